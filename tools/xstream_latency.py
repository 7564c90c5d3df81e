#!/usr/bin/env python3
"""Device-side cost of a cross-stream dependency (event record on one stream, wait on another) against in-stream order:
a ping-pong of tiny kernels between two streams, queued behind a device spin so that the host never paces it."""
import torch


def run(n, two):
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.zeros(64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(int(2.0e9 * 0.05))
        e0.record(a)
        for _ in range(n):
            x.add_(1.0)
            if two:
                ev = torch.cuda.Event(); ev.record(a); b.wait_event(ev)
                with torch.cuda.stream(b):
                    x.add_(1.0)
                    ev2 = torch.cuda.Event(); ev2.record(b)
                a.wait_event(ev2)
            else:
                x.add_(1.0)
        e1.record(a)
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (2 * n)


if __name__ == "__main__":
    for _ in range(2):
        print("in-stream  : %.1f us per kernel" % run(200, False))
        print("ping-pong  : %.1f us per kernel (one cross-stream dependency each)" % run(200, True))
