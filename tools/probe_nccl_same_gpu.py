#!/usr/bin/env python3
"""Probe (GPU box): can two processes form an RCCL ("nccl" backend) group on ONE device?  Prints the outcome; used to decide
whether the data-parallel GPU tests can run over RCCL on the single-GPU test box or must stay on host-staged gloo."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp


def worker(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    import torch.distributed as dist
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", 0))
        t = torch.full((1 << 20,), float(rank + 1), device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        q.put((rank, "ok", float(t[0])))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put((rank, "error", repr(e)[:300]))


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = []
    try:
        for _ in range(2):
            out.append(q.get(timeout=90))
    except Exception as e:  # noqa: BLE001
        out.append(("?", "timeout", repr(e)))
    for p in ps:
        p.join(timeout=10)
        if p.is_alive():
            p.terminate()
    print("nccl two ranks on one GPU:", out)
