#!/usr/bin/env python3
"""How often do the sets of the XCD-aware persistent launches really land on one XCD inside the training step?  Reads the placement
statistics the recurrent kernels keep behind the phase stamps of their sync buffers (rnn_split_kernel.h: XSTAT_WORD).
    python tools/xcd_stats.py [--steps 10]"""
import argparse
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    from aas_enhancement_amd import ops
    from aas_enhancement_amd.trainer_AAS import Trainer
    dev = torch.device("cuda", 0)
    cfg = types.SimpleNamespace(lr=1e-5, beta1=0.5, beta2=0.999, optimizer="adam", batch_size=30, expnum=0, lambda_k=0.001, gamma=0.5,
                                gpu=0, load_path="", mode="train", write_log=False, w_adversarial=1.0, w_acoustic=1.0,
                                allow_ASR_update_iter=10 ** 9, schedule="fused")
    tr = Trainer(cfg, None, models=bench.build_models())
    ny, cl = bench.make_batches(0, dev)
    for it in range(5):
        tr.train_step_async(ny, cl, it)
    torch.cuda.synchronize()
    for k, b in ops._scratch.items():
        if k[0] == "sync":
            b.view(torch.int32)[1060:1076].zero_()
    for it in range(a.steps):
        tr.train_step_async(ny, cl, 10 + it)
    torch.cuda.synchronize()
    names = ("lstm_fwd", "lstm_bwd", "gru_fwd", "gru_bwd")
    tot = [[0, 0, 0, 0] for _ in names]
    for k, b in ops._scratch.items():
        if k[0] == "sync":
            w = b.view(torch.int32)[1060:1076].tolist()
            for c in range(4):
                for j in range(4):
                    tot[c][j] += w[4 * c + j]
    for n, (wg, full, peers, P) in zip(names, tot):
        if wg:
            print("%-9s %6d workgroups over %d steps: whole set on their XCD %5.1f %%, co-located peers %5.1f %%" % (n, wg, a.steps, 100.0 * full / wg, 100.0 * peers / max(P, 1)))


if __name__ == "__main__":
    main()
