#!/usr/bin/env python3
"""Errors of one 500-unit BiLSTM layer's gradients against an fp64 nn.LSTM in the three arithmetic modes (and mode 2 with the exact
BPTT kernel): rms and max of |got - want| / max|want| for dx and the four weight gradients.   PYTHONPATH=. python tools/r03_x6_bptt_error.py [N]"""
import os
import sys

import torch
import torch.nn as nn

from aas_enhancement_amd import _lib, ops
from aas_enhancement_amd.dist import FlatBuffers

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
T, H = 48, 500
torch.manual_seed(3)
ref = nn.LSTM(H, H, bidirectional=True, bias=False).double()
g = torch.Generator().manual_seed(4)
x = torch.randn(T, N, H, generator=g) * 0.5
gy = torch.randn(T, N, H, generator=g)
names = ("weight_ih_l0", "weight_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse")
xr = x.double().requires_grad_(True)
yr, _ = ref(xr)
(yr[..., :H] + yr[..., H:] + xr).backward(gy.double())
want = [xr.grad] + [getattr(ref, k).grad for k in names]


def run(mode, flags=0):
    holder = nn.ParameterList([nn.Parameter(getattr(ref, k).detach().float().clone()) for k in names]).cuda()
    FlatBuffers(holder)
    ops.set_precision(mode)
    _lib.lib().aas_set_debug_flags(flags)
    xg = x.clone().cuda().requires_grad_(True)
    ops.birnn_layer(xg, *list(holder), kind="lstm", residual=True).backward(gy.cuda())
    ops.sync_wgrad()
    torch.cuda.synchronize()
    _lib.lib().aas_set_debug_flags(0)
    got = [xg.grad] + [p_.grad for p_ in holder]
    out = []
    for a, b in zip(got, want):
        e = (a.detach().double().cpu() - b).abs() / b.abs().max()
        out.append((float(e.pow(2).mean().sqrt()), float(e.max())))
    return out


for rep in range(2):
    for label, m, f in (("fp32", 0, 0), ("fp32eq x6-bptt", 2, 0), ("fp32eq exact-bptt", 2, 536870912), ("bf16x3", 1, 0)):
        e = run(m, f)
        print("%-18s " % label + "  ".join("%s rms %.2e max %.2e" % (n, a, b) for n, (a, b) in zip(("dx", "dWih", "dWhh", "dWih_r", "dWhh_r"), e)), flush=True)
