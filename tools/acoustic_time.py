#!/usr/bin/env python3
"""Time the acoustic_supervision step (trainer_acoustic) at config-2 size: python tools/acoustic_time.py [--trainable]"""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    from aas_enhancement_amd.trainer_acoustic import Trainer
    dev = torch.device("cuda", 0)
    cfg = types.SimpleNamespace(lr=1e-5, beta1=0.5, beta2=0.999, optimizer="adam", batch_size=30, expnum=0, lambda_k=0.001, gamma=0.5, gpu=0,
                                load_path="", mode="train", write_log=False, w_adversarial=1.0, w_acoustic=1.0,
                                allow_ASR_update_iter=0 if "--trainable" in sys.argv else 10 ** 9, schedule="fused")
    g, _, a = bench.build_models()
    tr = Trainer(cfg, None, models=(g, a))
    ny, _ = bench.make_batches(0, dev)
    for it in range(8):
        tr.train_step_async(ny, it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(20):
        tr.train_step_async(ny, 10 + it)
    torch.cuda.synchronize()
    print("acoustic step %.2f ms (%s A)" % ((time.perf_counter() - t0) / 20 * 1e3, "trainable" if "--trainable" in sys.argv else "frozen"))
    tr.read_scalars()


if __name__ == "__main__":
    main()
