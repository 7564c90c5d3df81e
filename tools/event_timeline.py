#!/usr/bin/env python3
"""Gantt of the recurrent + GEMM launches of the async step from HIP events (ops.Profiler), i.e. WITHOUT a tracing tool
slowing the host down: rocprofv3's kernel trace makes the host the bottleneck of parts of the step, so the gaps it shows
between launches are partly launch latency that an untraced run does not have.
    python tools/event_timeline.py [--steps 6] [--classes rnn,gemm]     (env switches as for bench.py)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--classes", default="rnn,gemm")
    a = ap.parse_args()
    import types
    from aas_enhancement_amd import ops
    from aas_enhancement_amd.trainer_AAS import Trainer
    dev = torch.device("cuda", 0)
    if os.environ.get("AAS_DP_FORCE") == "1":     # one-rank RCCL: the data-parallel code path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    cfg = types.SimpleNamespace(lr=1e-5, beta1=0.5, beta2=0.999, optimizer="adam", batch_size=30, expnum=0, lambda_k=0.001, gamma=0.5,
                                gpu=0, load_path="", mode="train", write_log=False, w_adversarial=1.0, w_acoustic=1.0,
                                allow_ASR_update_iter=10 ** 9, schedule="fused")
    tr = Trainer(cfg, None, models=bench.build_models())
    ny, cl = bench.make_batches(0, dev)
    for it in range(12):
        tr.train_step_async(ny, cl, it)
    torch.cuda.synchronize()
    marks = []
    ops.Profiler.start(tuple(a.classes.split(",")))
    for rep in range(a.steps):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        marks.append((e, len(ops.Profiler.records)))
        tr.train_step_async(ny, cl, 20 + rep)
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    marks.append((e, len(ops.Profiler.records)))
    torch.cuda.synchronize()
    recs = list(ops.Profiler.records)
    ops.Profiler.enabled = False
    per = [marks[i][0].elapsed_time(marks[i + 1][0]) for i in range(a.steps)]
    print("step times (ms, main-stream marks):", ["%.2f" % p for p in per])
    n0 = marks[1][1] - marks[0][1]
    ok = all(marks[i + 1][1] - marks[i][1] == n0 for i in range(a.steps))
    print("%d bracketed launches per step%s" % (n0, "" if ok else " (varies!)"))
    # average offsets over steps 1.. (skip the first: its neighbours' tails differ)
    use = range(1, a.steps)
    rows = []
    for j in range(n0):
        st = en = 0.0
        for i in use:
            name, _, e0, e1, T = recs[marks[i][1] + j]
            st += marks[i][0].elapsed_time(e0)
            en += marks[i][0].elapsed_time(e1)
        rows.append((st / len(use), en / len(use), recs[marks[1][1] + j][0]))
    rows.sort()
    print("start_ms   end_ms   dur_ms  launch (offsets from the main-stream mark at the head of the step's host code)")
    for s_, e_, n_ in rows:
        print("%8.3f %8.3f %8.3f  %s" % (s_, e_, e_ - s_, n_))


if __name__ == "__main__":
    main()
