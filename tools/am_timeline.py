#!/usr/bin/env python3
"""Gantt of the recurrent / GEMM / CTC launches of the AM step (config 5) from HIP events: python tools/am_timeline.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn as nn

from aas_enhancement_amd import ops, prng
from aas_enhancement_amd.am_train import AMTrainer
from aas_enhancement_amd.model import DeepSpeech

LABELS = "_'abcdefghijklmnopqrstuvwxyz "


def main():
    n, F, T, L = 30, 80, 200, 20
    A = DeepSpeech(nn.GRU, LABELS, 1000, 5, True, 11, 2, 128, 2, nFreq=F)
    sd = prng.fill_state_dict(A.state_dict(), 9203, conv_std=0.1)
    A.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    tr = AMTrainer(A.cuda(), lr=1e-4)
    b = (torch.from_numpy(prng.uniform(9210, (n, F, T), 0.0, 6.0)), torch.from_numpy(prng.randint(9220, (n * L,), 1, 28).astype(np.int32)),
         torch.ones(n), torch.full((n,), L, dtype=torch.int32))
    prev = None
    for it in range(10):
        cur = tr.train_step_async(b)
        if prev is not None:
            tr.read_loss(prev["handle"])
        prev = cur
    torch.cuda.synchronize()
    ops.Profiler.start(("rnn", "gemm", "ctc"))
    marks = []
    for it in range(5):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((e, len(ops.Profiler.records)))
        cur = tr.train_step_async(b)
        tr.read_loss(prev["handle"])
        prev = cur
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((e, len(ops.Profiler.records)))
    torch.cuda.synchronize()
    recs = list(ops.Profiler.records)
    ops.Profiler.enabled = False
    print("step times:", ["%.2f" % marks[i][0].elapsed_time(marks[i + 1][0]) for i in range(5)])
    i = 2
    for j in range(marks[i][1], marks[i + 1][1]):
        name, _, e0, e1, _T = recs[j]
        print("%8.3f %8.3f %7.3f  %s" % (marks[i][0].elapsed_time(e0), marks[i][0].elapsed_time(e1), e0.elapsed_time(e1), name))


if __name__ == "__main__":
    main()
