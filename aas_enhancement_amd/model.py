"""Reference-compatible model API on the MI355X HIP path.

Same class names, constructor arguments, forward signatures and ``state_dict`` keys as the
reference's Speech_enhancement_by_AAS/model.py (L1Loss_mask :19-31, SequenceWise :34-49,
InferenceBatchSoftmax :58-64, BatchRNN :66-86, BRNN :88-105, stackedBRNN :203-252,
DeepSpeech :256-450), so reference checkpoints load and ``main.py --trainer AAS`` is a drop-in.
The modules below only own parameters; all arithmetic runs in libaas_hip.so via ``ops``.

Documented deviations: ``stackedBRNN`` honours ``L`` (the reference hard-codes 4
layers; identical at L=4) and ``O`` defaults to ``I`` (the reference trainers omit it and would
raise TypeError).
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops

supported_rnns = {"lstm": nn.LSTM, "rnn": nn.RNN, "gru": nn.GRU}
supported_rnns_inv = dict((v, k) for k, v in supported_rnns.items())


def _rnn_kind(rnn_type):
    if isinstance(rnn_type, str):
        kind = rnn_type.lower()
    else:
        kind = supported_rnns_inv.get(rnn_type, getattr(rnn_type, "__name__", str(rnn_type)).lower())
    if kind not in ("lstm", "gru", "rnn"):
        raise NotImplementedError("rnn_type %r: supported_rnns are lstm | gru | rnn (bias-free, bidirectional)" % (rnn_type,))
    return kind


class L1Loss_mask(nn.Module):
    """sum|input-target| over ALL elements / nElement, nElement = #unmasked (n,t) frames.
    The mask is NOT applied to the error (the reference drops the masked_fill result, model.py:29)."""

    def forward(self, input, target, mask):
        n_valid = getattr(mask, "n_valid", None)
        if n_valid is None:
            n_valid = int(mask.numel()) - int(mask.sum().item())
        loss = ops.l1_sum(input, target) / n_valid
        return loss, n_valid


class _RNNWeights(nn.Module):
    """Parameter container with nn.LSTM/nn.GRU names (weight_ih_l0, weight_hh_l0, *_reverse), bias-free."""

    def __init__(self, input_size, hidden_size, kind):
        super().__init__()
        g = {"lstm": 4, "gru": 3, "rnn": 1}[kind]
        self.kind, self.input_size, self.hidden_size = kind, input_size, hidden_size
        self.weight_ih_l0 = nn.Parameter(torch.empty(g * hidden_size, input_size))
        self.weight_hh_l0 = nn.Parameter(torch.empty(g * hidden_size, hidden_size))
        self.weight_ih_l0_reverse = nn.Parameter(torch.empty(g * hidden_size, input_size))
        self.weight_hh_l0_reverse = nn.Parameter(torch.empty(g * hidden_size, hidden_size))
        b = 1.0 / math.sqrt(hidden_size)
        for p in self.parameters():
            nn.init.uniform_(p, -b, b)
        self._aas_layer_id = ops.register_layer("%s %d->%d" % (kind, input_size, hidden_size))

    def flatten_parameters(self):
        pass

    def run(self, x, residual, rs=None):
        return ops.birnn_layer(x, self.weight_ih_l0, self.weight_hh_l0, self.weight_ih_l0_reverse,
                               self.weight_hh_l0_reverse, self.kind, residual, rs, self._aas_layer_id)


class _BNParams(nn.Module):
    """nn.BatchNorm1d-named parameter/buffer container; forward = train-mode batch statistics."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x, slope=1.0):
        if not self.training:   # model.eval(): running statistics (AM_training/train.py:357 validation; the AAS trainers
            return ops.batchnorm_eval(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, slope)  # never do)
        return ops.batchnorm_rows(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                  self.momentum, slope, self.num_batches_tracked)   # (the counter: += 1 inside the apply launch)


class SequenceWise(nn.Module):
    """Collapses T*N*H to (T*N)*H and applies `module` (model.py:34-49).  Our row-wise modules accept
    the un-collapsed tensor directly (statistics are over all leading dims), so this only delegates."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x):
        return self.module(x)


class InferenceBatchSoftmax(nn.Module):
    def forward(self, input_):
        if not self.training:
            return ops.softmax_rows(input_)
        return input_


class BatchRNN(nn.Module):
    def __init__(self, input_size, hidden_size, rnn_type=nn.LSTM, bidirectional=False, batch_norm=True):
        super().__init__()
        if not bidirectional:
            raise NotImplementedError("only bidirectional layers are on the hot path")
        self.input_size, self.hidden_size, self.bidirectional = input_size, hidden_size, bidirectional
        self.batch_norm = SequenceWise(_BNParams(input_size)) if batch_norm else None
        self.rnn = _RNNWeights(input_size, hidden_size, _rnn_kind(rnn_type))
        self.num_directions = 2

    def flatten_parameters(self):
        pass

    def forward(self, x):  # [T,N,I] -> [T,N,H] (directions summed)
        if self.batch_norm is not None:
            x = self.batch_norm(x)
        return self.rnn.run(x, residual=False)


class BRNN(nn.Module):
    def __init__(self, input_size, hidden_size, rnn_type=nn.LSTM, bidirectional=False):
        super().__init__()
        if not bidirectional:
            raise NotImplementedError("only bidirectional layers are on the hot path")
        self.input_size, self.hidden_size, self.bidirectional = input_size, hidden_size, bidirectional
        self.rnn = _RNNWeights(input_size, hidden_size, _rnn_kind(rnn_type))
        self.num_directions = 2

    def flatten_parameters(self):
        pass

    def forward(self, x, residual=False, wgrad_row_scale=None):
        return self.rnn.run(x, residual=residual, rs=wgrad_row_scale)


class _PointwiseConv(nn.Module):
    """nn.Conv1d(kernel_size=1)-named container: weight [out,in,1], bias [out]."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c_out, c_in, 1))
        self.bias = nn.Parameter(torch.empty(c_out))
        b = 1.0 / math.sqrt(c_in)
        nn.init.uniform_(self.weight, -b, b)
        nn.init.uniform_(self.bias, -b, b)


class stackedBRNN(nn.Module):
    """E and D: k=1 conv I->H, L x (BiRNN + residual), k=1 conv H->O (model.py:203-252)."""

    def __init__(self, I, O=None, H=None, L=4, rnn_type=nn.LSTM):
        super().__init__()
        if H is None:
            raise TypeError("stackedBRNN needs H")
        O = I if O is None else O
        self.I, self.O, self.H, self.L, self.rnn_type = I, O, H, L, rnn_type
        for l in range(1, L + 1):
            setattr(self, "rnn%d" % l, BRNN(input_size=H, hidden_size=H, rnn_type=rnn_type, bidirectional=True))
        self.first_linear = _PointwiseConv(I, H)
        self.final_linear = _PointwiseConv(H, O)

    def _trunk(self, input, rs=None, tnc=None):
        h = tnc if tnc is not None else ops.layout(input, "nct_tnc")      # [N,I,T] -> [T,N,I]
        h = ops.linear_rows(h, self.first_linear.weight, self.first_linear.bias, rs)
        for l in range(1, self.L + 1):
            h = getattr(self, "rnn%d" % l)(h, residual=True, wgrad_row_scale=rs)   # BRNN(h) + h
        return h

    def forward(self, input, wgrad_row_scale=None, tnc=None):
        """`wgrad_row_scale` [N] (extension, default None = reference behaviour): per-utterance weights applied to
        the PARAMETER gradients only (input gradients are unaffected) - lets D(enhanced) and D(clean) share one
        batched pass while the enhanced half's parameter gradients carry the BEGAN factor (-kt).
        `tnc` (extension): the input already laid down time-major [T, N, I] (`input` is then ignored) - a caller that
        assembles a batch from several tensors writes it in that layout directly (ops.layout_paired_cat)."""
        h = self._trunk(input, wgrad_row_scale, tnc)
        out = ops.linear_rows(h, self.final_linear.weight, self.final_linear.bias, wgrad_row_scale)
        return ops.layout(out, "tnc_nct")                                 # [T,N,O] -> [N,O,T]

    def forward_stages(self, input, wgrad_row_scale=None, pair=None):
        """forward() as a generator that yields after every layer (None) and finally the output: lets a trainer queue
        two independent networks layer by layer on two HIP streams (trainer_AAS: discriminator beside acoustic model).
        pair = (a, b): the input is [a ; b] along the batch axis, laid down time-major without a concatenated copy."""
        rs = wgrad_row_scale
        h = ops.layout_cat_nct_tnc(pair[0], pair[1]) if pair is not None else ops.layout(input, "nct_tnc")
        h = ops.linear_rows(h, self.first_linear.weight, self.first_linear.bias, rs)
        yield None
        for l in range(1, self.L + 1):
            h = getattr(self, "rnn%d" % l)(h, residual=True, wgrad_row_scale=rs)
            yield None
        out = ops.linear_rows(h, self.final_linear.weight, self.final_linear.bias, rs)
        yield ops.layout(out, "tnc_nct")

    def forward_paired(self, input, paired):
        return self.forward(torch.cat((input, paired), dim=1))

    def forward_with_intermediate_output(self, input):
        h = self._trunk(input)
        out = ops.linear_rows(h, self.final_linear.weight, self.final_linear.bias)
        return [ops.layout(out, "tnc_nct"), ops.layout(h, "tnc_nct")]


class _ConvK(nn.Module):
    """nn.Conv1d-named container (weight [out,in,k], bias [out]) for the DeepSpeech front-end."""

    def __init__(self, c_in, c_out, kernel_size, stride):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, stride
        self.weight = nn.Parameter(torch.empty(c_out, c_in, kernel_size))
        self.bias = nn.Parameter(torch.empty(c_out))
        b = 1.0 / math.sqrt(c_in * kernel_size)
        nn.init.uniform_(self.weight, -b, b)
        nn.init.uniform_(self.bias, -b, b)


class _LeakySlope(nn.Module):
    """Placeholder keeping the reference's Sequential indexing (conv.2 / conv.5); the LeakyReLU
    (negative_slope = map, model.py:291) is fused into the BatchNorm apply kernel."""

    def __init__(self, slope):
        super().__init__()
        self.negative_slope = slope


class _FCWeights(nn.Module):
    """nn.Linear(bias=False)-named container."""

    def __init__(self, n_in, n_out):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in))
        b = 1.0 / math.sqrt(n_in)
        nn.init.uniform_(self.weight, -b, b)


DeepSpeech_ken = None   # bound below: AM_training/model.py's name for the same class


class _BNLinear(nn.Sequential):
    """fc.0.module = Sequential(BatchNorm1d, Linear(bias=False)) (model.py:315-318)."""

    def forward(self, x):
        return ops.linear_rows(self[0](x), self[1].weight, None)


class DeepSpeech(nn.Module):
    """AAS/model.py:256-335 `DeepSpeech` = AM_training/model.py:337-470 `DeepSpeech_ken` (the class AM_training/train.py builds;
    same layers and state_dict keys).  `include_first_BN` is the AM_training constructor's extra argument (:341,364-367): False
    leaves the first convolution without its BatchNorm - conv.0 conv, conv.1 LeakyReLU, conv.2 conv, conv.3 BatchNorm, as the
    reference's Sequential then numbers them."""

    def __init__(self, rnn_type=nn.LSTM, labels="abc", rnn_hidden_size=512, rnn_layers=2, bidirectional=True,
                 kernel_sz=11, stride=2, map=256, cnn_layers=2, nFreq=40, nDownsample=1, audio_conf=None, include_first_BN=True):
        super().__init__()
        self.nFreq = nFreq
        self._version = "0.0.1"
        self._audio_conf = audio_conf
        self.rnn_size, self.rnn_layers, self.rnn_type, self.bidirectional = rnn_hidden_size, rnn_layers, rnn_type, bidirectional
        self.cnn_stride, self.cnn_map, self.cnn_kernel, self.nDownsample = stride, map, kernel_sz, nDownsample
        self.cnn_layers = cnn_layers
        self._labels = labels
        num_classes = len(labels)
        self.include_first_BN = bool(include_first_BN)
        conv_list = [_ConvK(nFreq, map, kernel_sz, stride)] + ([_BNParams(map)] if include_first_BN else []) + [_LeakySlope(map)]
        s2 = 1 if nDownsample == 1 else stride
        for _ in range(cnn_layers - 1):
            conv_list += [_ConvK(map, map, kernel_sz, s2), _BNParams(map), _LeakySlope(map)]
        self.conv = nn.Sequential(*conv_list)
        rnns = [("0", BatchRNN(map, rnn_hidden_size, rnn_type, bidirectional, batch_norm=False))]
        for x in range(rnn_layers - 1):
            rnns.append(("%d" % (x + 1), BatchRNN(rnn_hidden_size, rnn_hidden_size, rnn_type, bidirectional)))
        self.rnns = nn.Sequential(OrderedDict(rnns))
        self.fc = nn.Sequential(SequenceWise(_BNLinear(_BNParams(rnn_hidden_size), _FCWeights(rnn_hidden_size, num_classes))))
        self.inference_softmax = InferenceBatchSoftmax()

    def output_length(self, T):
        """Number of output frames T' for T input frames (two un-padded temporal convolutions, model.py:288-301)."""
        t = T
        for m in self.conv:
            if isinstance(m, _ConvK):
                t = (t - m.kernel_size) // m.stride + 1
        return t

    def forward(self, x):  # [N,nFreq,T] -> [N,T',C]
        out = None
        for out in self.forward_stages(x):
            pass
        return out

    def forward_stages(self, x):
        """forward() as a generator: yields None after the convolutional front-end and after every recurrent layer,
        finally the output [N,T',C] (see stackedBRNN.forward_stages)."""
        h = ops.layout(x, "nct_ntc")                                     # channels-last [N,T,F]
        mods, i = list(self.conv), 0
        while i < len(mods):
            cv = mods[i]
            h = ops.conv1d_cl(h, cv.weight, cv.bias, cv.stride)
            if isinstance(mods[i + 1], _BNParams):     # BatchNorm with the LeakyReLU fused into its apply launch
                h = mods[i + 1](h, slope=float(mods[i + 2].negative_slope))
                i += 3
            else:                                      # include_first_BN=False: the activation alone
                h = ops.leaky_relu(h, float(mods[i + 1].negative_slope))
                i += 2
        h = ops.layout(h, "swap01")                                      # [N,T',M] -> [T',N,M]
        yield None
        for layer in self.rnns:
            h = layer(h)
            yield None
        h = self.fc(h)                                                   # [T',N,C]
        h = h.transpose(0, 1)
        yield self.inference_softmax(h)

    # ---- (de)serialisation, same package format as model.py:337-410 ---------------------------
    @classmethod
    def load_model(cls, path, gpu=-1):
        package = torch.load(path, map_location=lambda storage, loc: storage)
        blacklist = ["rnns.0.batch_norm.module.weight", "rnns.0.batch_norm.module.bias",
                     "rnns.0.batch_norm.module.running_mean", "rnns.0.batch_norm.module.running_var"]
        for x in blacklist:
            package["state_dict"].pop(x, None)
        return cls.load_model_package(package, gpu)

    @classmethod
    def load_model_package(cls, package, gpu=-1):
        sd = package["state_dict"]
        n_freq = package.get("nFreq", sd["conv.0.weight"].shape[1])  # the reference always builds 40-in
        # (the reference's packages carry neither include_first_BN nor nDownsample, AM_training/model.py:457-470 rebuilds with the
        #  defaults; the first is visible in the keys - a BatchNorm at conv.1 or not - and is honoured so such a checkpoint loads)
        first_bn = package.get("include_first_BN", "conv.1.weight" in sd)
        model = cls(rnn_hidden_size=package["rnn_size"], rnn_layers=package["rnn_layers"],
                    rnn_type=supported_rnns[package["rnn_type"]], map=package["cnn_map"], stride=package["cnn_stride"],
                    kernel_sz=package["cnn_kernel"], cnn_layers=package["cnn_layers"], labels=package["labels"],
                    nFreq=n_freq, nDownsample=package.get("nDownsample", 1), include_first_BN=first_bn)
        model.load_state_dict(sd)
        if gpu >= 0:
            model = model.cuda()
        return model

    @staticmethod
    def serialize(model, optimizer=None, epoch=None, iteration=None, loss_results=None, cer_results=None,
                  wer_results=None, avg_loss=None, meta=None):
        package = {
            "version": model._version, "rnn_size": model.rnn_size, "rnn_layers": model.rnn_layers,
            "cnn_map": model.cnn_map, "cnn_kernel": model.cnn_kernel, "cnn_stride": model.cnn_stride,
            "cnn_layers": model.cnn_layers,
            "rnn_type": supported_rnns_inv.get(model.rnn_type, getattr(model.rnn_type, "__name__", "gru").lower()),
            "labels": model._labels, "state_dict": model.state_dict(),
        }
        if getattr(model, "nDownsample", 1) != 1:      # (extra keys only when they differ from what the reference's loader assumes)
            package["nDownsample"] = model.nDownsample
        if not getattr(model, "include_first_BN", True):
            package["include_first_BN"] = False
        if optimizer is not None:
            package["optim_dict"] = optimizer.state_dict()
        if avg_loss is not None:
            package["avg_loss"] = avg_loss
        if epoch is not None:
            package["epoch"] = epoch + 1
        if iteration is not None:
            package["iteration"] = iteration
        if loss_results is not None:
            package["loss_results"], package["cer_results"], package["wer_results"] = loss_results, cer_results, wer_results
        if meta is not None:
            package["meta"] = meta
        return package

    @staticmethod
    def get_labels(model):
        return model._labels

    @staticmethod
    def get_param_size(model):
        return sum(p.numel() for p in model.parameters())

    @staticmethod
    def get_audio_conf(model):
        return model._audio_conf

    @staticmethod
    def get_meta(model):
        return {"version": model._version, "rnn_size": model.rnn_size, "rnn_layers": model.rnn_layers,
                "cnn_map": model.cnn_map, "cnn_kernel": model.cnn_kernel, "cnn_stride": model.cnn_stride,
                "cnn_layers": model.cnn_layers, "rnn_type": supported_rnns_inv[model.rnn_type]}


DeepSpeech_ken = DeepSpeech   # AM_training/model.py:337, AM_training/train.py:152 (`--arch_ver ken`)
