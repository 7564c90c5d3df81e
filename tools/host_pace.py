#!/usr/bin/env python3
"""Does the host run ahead of the GPU in the async step loop?  Host timestamps after each train_step_async call of a
free-running loop (no device spin in front): ~5 ms apart = the host queues ahead; ~step time apart = something blocks it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    import types
    from aas_enhancement_amd.trainer_AAS import Trainer
    dev = torch.device("cuda", 0)
    cfg = types.SimpleNamespace(lr=1e-5, beta1=0.5, beta2=0.999, optimizer="adam", batch_size=30, expnum=0, lambda_k=0.001, gamma=0.5,
                                gpu=0, load_path="", mode="train", write_log=False, w_adversarial=1.0, w_acoustic=1.0,
                                allow_ASR_update_iter=10 ** 9, schedule="fused")
    tr = Trainer(cfg, None, models=bench.build_models())
    ny, cl = bench.make_batches(0, dev)
    for it in range(8):
        tr.train_step_async(ny, cl, it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = []
    for rep in range(24):
        tr.train_step_async(ny, cl, 10 + rep)
        ts.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    tot = 1e3 * (time.perf_counter() - t0)
    print("host done queuing step i at (ms):", ["%.1f" % t for t in ts])
    print("all 24 steps finished on the device at %.1f ms (%.2f ms / step)" % (tot, tot / 24))


if __name__ == "__main__":
    main()
