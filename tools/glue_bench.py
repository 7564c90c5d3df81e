"""Isolated timings of the HBM-bound glue launches that sit on the A chain between persistent launches:
train-mode BatchNorm forward / backward (16-byte partial-sum path vs the column-per-thread kernels with fp64 atomics, debug bit 32768).

    python tools/glue_bench.py            # on the GPU box
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aas_enhancement_amd import ops
from aas_enhancement_amd._lib import lib


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (R, C, slope) in [(2550, 1000, 1.0), (2850, 128, 128.0), (2550, 128, 128.0), (6000, 1000, 1.0)]:
        x = torch.randn(R, C, device=dev)
        dy = torch.randn(R, C, device=dev)
        gamma = torch.rand(C, device=dev) + 0.5
        beta = torch.randn(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        out = {}
        for name, flags in (("partials16", 0), ("atomics", 32768)):
            lib().aas_set_debug_flags(flags)
            xx = x.clone().requires_grad_(True)
            y = ops.batchnorm_rows(xx, gamma, beta, rm, rv, 1e-5, 0.1, slope)
            y.backward(dy)
            out[name] = (y.detach().clone(), xx.grad.clone())
            tf = timed(lambda: ops.batchnorm_rows(x, gamma, beta, rm, rv, 1e-5, 0.1, slope))

            def fb():
                xx.grad = None
                ops.batchnorm_rows(xx, gamma, beta, rm, rv, 1e-5, 0.1, slope).backward(dy)
            tfb = timed(fb)
            mb = R * C * 4 / 1e6
            print("bn R=%d C=%d slope=%g %-10s fwd %.1f us (%.2f TB/s of 3 passes)  fwd+bwd %.1f us" % (R, C, slope, name, tf, 3 * mb / tf, tfb))
        lib().aas_set_debug_flags(0)
        dyv = (out["partials16"][0] - out["atomics"][0]).abs().max().item()
        dxv = (out["partials16"][1] - out["atomics"][1]).abs().max().item()
        print("   max |y diff| %.2e  max |dx diff| %.2e" % (dyv, dxv))


if __name__ == "__main__":
    main()
