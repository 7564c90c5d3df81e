#!/usr/bin/env python3
"""Host time to QUEUE one training step (Python + ctypes + HIP launch calls), measured while the GPU is kept busy by a
long spin kernel so that no launch ever waits for the device: the floor the eager step can reach when the GPU is faster
than the host.  Usage: python tools/host_time.py [--lanes 0|1]"""
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
    for it in range(5):
        tr.train_step_async(ny, cl, it)
    torch.cuda.synchronize()
    res = []
    for rep in range(5):
        torch.cuda._sleep(int(2.0e9 * 0.15))      # ~150 ms of device spin on the current stream: the step queues behind it
        t0 = time.perf_counter()
        tr.train_step_async(ny, cl, 10 + rep)
        res.append(1e3 * (time.perf_counter() - t0))
        torch.cuda.synchronize()
    print("host ms to queue one step:", ["%.2f" % r for r in res])


if __name__ == "__main__":
    main()
