"""warpctc_pytorch.CTCLoss-compatible op on the HIP CTC kernel.

Reference call sites: Speech_enhancement_by_AAS/trainer_AAS.py:10,62,168,349;
trainer_acoustic.py:132,280; AM_training/train.py:10,151,319.  Semantics of the third-party
op being replaced (SeanNaren/warp-ctc binding, defaults): ``CTCLoss()(acts[T,N,C] pre-softmax,
labels[sum L] int32 (no blanks), act_lens[N] int32, label_lens[N] int32) -> FloatTensor[1]`` = sum of
per-utterance costs, blank index 0, differentiable wrt ``acts`` only, gradients zero for t >= act_len.
"""
import torch.nn as nn

from . import ops


class CTCLoss(nn.Module):
    def __init__(self, size_average=False, length_average=False, blank=0):
        super().__init__()
        self.size_average, self.length_average, self.blank = size_average, length_average, blank

    def prepare(self, labels, act_lens, label_lens, device):
        """Optional: upload labels/lengths ahead of the forward pass (see ops.ctc_prepare); pass the result as
        ``prepared=`` to forward so the loss itself issues no host->device copy."""
        return ops.ctc_prepare(labels, act_lens, label_lens, device)

    def forward(self, acts, labels, act_lens, label_lens, prepared=None):
        if labels.dim() != 1 or act_lens.dim() != 1 or label_lens.dim() != 1:
            raise ValueError("CTCLoss: labels / act_lens / label_lens must be 1-D")
        if act_lens.numel() != acts.size(1) or label_lens.numel() != acts.size(1):
            raise ValueError("CTCLoss: lengths must have one entry per utterance")
        cost = ops.ctc_sum(acts, labels, act_lens, label_lens, self.blank, prepared)
        if self.size_average:
            cost = cost / acts.size(1)
        if self.length_average:
            cost = cost / float(act_lens.sum().item())
        return cost
