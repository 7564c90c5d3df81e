"""Acoustic-model (A) CTC pre-training step on the HIP path - the hot loop of the reference's
AM_training/train.py:293-349 (BASELINE config 5): A(x) -> CTC / N -> Adam(lr) (plain Adam, not amsgrad, :246-247).
Same batch tuple as the AAS trainer (`_collate_fn` order).  Data-parallel: the flat gradient buffer of A is
SUM-all-reduced over RCCL with the loss normalised by the global batch size."""
import torch

from . import ops
from .ctc import CTCLoss
from .dist import DPContext, FlatBuffers
from .optim import FlatAdam
from .utils import _get_variable_nograd


class AMTrainer(object):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), dp=None):
        self.model = model
        self.criterion = CTCLoss()
        self.flat = FlatBuffers(model)
        self.opt = FlatAdam(self.flat, lr=lr, betas=betas, amsgrad=False)
        self.dp = dp or DPContext.from_env()
        ops.DIRECT_WGRAD[0] = True

    def train_step(self, data_list):
        inputs, targets, input_percentages, target_sizes = data_list[0], data_list[1], data_list[2], data_list[3]
        inputs = _get_variable_nograd(inputs)
        N = inputs.size(0)
        t_out = self.model.output_length(inputs.size(2))
        sizes = input_percentages.clone().mul_(int(t_out)).int()
        meta = self.criterion.prepare(targets, sizes, target_sizes, inputs.device)
        N_glob = self.dp.global_counts([N])[0] if self.dp.active else N
        ops.sync_wgrad()
        self.flat.zero_grad()
        out = self.model(inputs).transpose(0, 1)
        loss = self.criterion(out, targets, sizes, target_sizes, prepared=meta) / N_glob
        loss.backward()
        ops.sync_wgrad()
        if self.dp.active:
            self.dp.allreduce_sum_(self.flat.flat_g)
        self.opt.step()
        v = self.dp.reduce_scalars(loss.detach().reshape(1).clone())
        loss_value = float(v)
        return dict(loss=loss_value, is_inf=loss_value in (float("inf"), float("-inf")), logits=out)
