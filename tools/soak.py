#!/usr/bin/env python3
"""Soak run of the device-resident step (GPU box): STEPS iterations of train_step_async at BASELINE config-2 sizes, switching
between the batched-D schedule (equal padded lengths), its two-row-class form (ragged pair, clean T = 184) and the two-lane schedule
(clean T = 120: below the length-ratio threshold) every 25 steps, trainable A from
step 100 on, scalars read back every 25 steps (where a raised exchange-timeout word or a non-finite loss raises).
Usage: python tools/soak.py [STEPS=400]"""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    from aas_enhancement_amd import ops
    from aas_enhancement_amd.trainer_AAS import Trainer
    dev = torch.device("cuda", 0)
    cfg = types.SimpleNamespace(lr=1e-5, beta1=0.5, beta2=0.999, optimizer="adam", batch_size=30, expnum=0, lambda_k=0.001, gamma=0.5,
                                gpu=0, load_path="", mode="train", write_log=False, w_adversarial=1.0, w_acoustic=1.0,
                                allow_ASR_update_iter=100, schedule="fused")
    tr = Trainer(cfg, None, models=bench.build_models())
    ny, cl = bench.make_batches(0, dev)
    pairs = [cl]
    for tc in (184, 120):
        c_ = (cl[0][:, :, :tc].contiguous(), None, None, None, torch.zeros(30, 1, tc, dtype=torch.uint8, device=dev))
        c_[4].n_valid = 30 * tc
        pairs.append(c_)
    t0 = time.time()
    hist = []
    for it in range(steps):
        tr.train_step_async(ny, pairs[(it // 25) % 3], it)
        if (it + 1) % 25 == 0:
            r = tr.read_scalars()          # raises on an exchange timeout / divergence
            hist.append((it + 1, tr._last_schedule, r["l_adv_ny_G"], r["l_adv_cl"], r["l_ctc"], r["kt"]))
            print("step %4d %-14s adv_ny %.4f adv_cl %.4f ctc %.4f kt %.4f" % hist[-1], flush=True)
    torch.cuda.synchronize()
    assert not ops.rnn_timeout_flag()
    print("soak ok: %d steps in %.1f s (%.2f ms / step incl. read-backs); CTC %.3f -> %.3f" % (steps, time.time() - t0, 1e3 * (time.time() - t0) / steps, hist[0][4], hist[-1][4]))


if __name__ == "__main__":
    main()
