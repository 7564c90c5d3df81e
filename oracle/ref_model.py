"""Plain-PyTorch CPU restatement of the reference networks (test oracle only).

Follows /root/reference/Speech_enhancement_by_AAS/model.py:
  L1Loss_mask      model.py:19-31   (mask NOT applied; nElement = #valid (n,t))
  SequenceWise     model.py:34-49
  BatchRNN         model.py:66-86   (BN over T*N rows, bias-free RNN, dir-sum)
  BRNN             model.py:88-105
  stackedBRNN      model.py:203-252 (k=1 conv, 4x(BRNN+residual), k=1 conv)
  DeepSpeech       model.py:256-335 (conv/BN/LeakyReLU(slope=map) x2, GRU stack,
                                     BN+Linear, logits in train mode)
Module/parameter names are chosen so that ``state_dict()`` keys equal the
reference's, which lets golden weights load into either.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

RNN_TYPES = {"lstm": nn.LSTM, "gru": nn.GRU, "rnn": nn.RNN}


def l1loss_mask(inp, target, mask):
    """model.py:23-31: sum|a-b| over everything / number of un-masked (n,t) frames."""
    n_element = mask.numel() - int(mask.sum().item())
    loss = (inp - target).abs().sum() / n_element
    return loss, n_element


class _RowWise(nn.Module):
    """model.py:34-49 – apply `module` to [T*N, H] rows."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x):
        t, n = x.shape[0], x.shape[1]
        return self.module(x.reshape(t * n, -1)).reshape(t, n, -1)


class RefBRNN(nn.Module):
    def __init__(self, size_in, size_h, rnn_type=nn.LSTM, batch_norm=False):
        super().__init__()
        if batch_norm:
            self.batch_norm = _RowWise(nn.BatchNorm1d(size_in))
        self.has_bn = batch_norm
        self.rnn = rnn_type(input_size=size_in, hidden_size=size_h, bidirectional=True, bias=False)
        self.size_h = size_h

    def forward(self, x):
        if self.has_bn:
            x = self.batch_norm(x)
        y, _ = self.rnn(x)
        return y[..., : self.size_h] + y[..., self.size_h:]


class RefStackedBRNN(nn.Module):
    """E and D.  The reference ignores L (always 4 layers, model.py:211-214)."""

    def __init__(self, I, O, H, L=4, rnn_type=nn.LSTM):
        super().__init__()
        self.L = L
        for l in range(1, L + 1):
            setattr(self, "rnn%d" % l, RefBRNN(H, H, rnn_type))
        self.first_linear = nn.Conv1d(I, H, kernel_size=1)
        self.final_linear = nn.Conv1d(H, O, kernel_size=1)

    def forward(self, x):  # [N,I,T] -> [N,O,T]
        h = self.first_linear(x).permute(2, 0, 1)  # T,N,H
        for l in range(1, self.L + 1):
            h = getattr(self, "rnn%d" % l)(h) + h
        return self.final_linear(h.permute(1, 2, 0))

    def forward_paired(self, x, paired):
        return self.forward(torch.cat((x, paired), dim=1))


class RefDeepSpeech(nn.Module):
    def __init__(self, rnn_type=nn.GRU, labels="abc", rnn_hidden_size=512, rnn_layers=2,
                 kernel_sz=11, stride=2, map=256, cnn_layers=2, nFreq=40, nDownsample=1, include_first_BN=True):
        super().__init__()
        # include_first_BN: AM_training/model.py:341,364-367 (DeepSpeech_ken) - False leaves the first convolution without BatchNorm
        convs = [nn.Conv1d(nFreq, map, kernel_sz, stride=stride)] + ([nn.BatchNorm1d(map)] if include_first_BN else []) + [
                 nn.LeakyReLU(map)]  # negative_slope == map (model.py:291)
        s2 = 1 if nDownsample == 1 else stride
        for _ in range(cnn_layers - 1):
            convs += [nn.Conv1d(map, map, kernel_sz, stride=s2), nn.BatchNorm1d(map), nn.LeakyReLU(map)]
        self.conv = nn.Sequential(*convs)
        layers = [("0", RefBRNN(map, rnn_hidden_size, rnn_type, batch_norm=False))]
        for i in range(1, rnn_layers):
            layers.append((str(i), RefBRNN(rnn_hidden_size, rnn_hidden_size, rnn_type, batch_norm=True)))
        self.rnns = nn.Sequential(OrderedDict(layers))
        self.fc = nn.Sequential(_RowWise(nn.Sequential(
            nn.BatchNorm1d(rnn_hidden_size), nn.Linear(rnn_hidden_size, len(labels), bias=False))))

    def forward(self, x):  # [N,F,T] -> [N,T',C]  (logits in train mode, softmax in eval)
        h = self.conv(x).permute(2, 0, 1)
        h = self.fc(self.rnns(h)).transpose(0, 1)
        if not self.training:
            h = F.softmax(h, dim=-1)
        return h


def conv_out_len(T, kernel_sz=11, stride=2, cnn_layers=2, nDownsample=1):
    t = (T - kernel_sz) // stride + 1
    s2 = 1 if nDownsample == 1 else stride
    for _ in range(cnn_layers - 1):
        t = (t - kernel_sz) // s2 + 1
    return t
