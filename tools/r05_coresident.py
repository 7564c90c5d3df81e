#!/usr/bin/env python3
"""Experiment: weight-gradient GEMMs CO-RESIDENT with E's forward recurrent launches (256-thread workgroups, half of a CU's registers).
Would D's weight gradients of step k fit under E's forward of step k+1?  E forward x 4 alone, n GEMMs alone, both at once."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aas_enhancement_amd import _lib, ops
from tools.rnn_bench import rnn_case


def main():
    L = _lib.lib()
    L.aas_set_precision(0)
    L.aas_set_gemm_max_steps(int(os.environ.get("CAP", "48")))
    dev = torch.device("cuda:0")
    f, _ = rnn_case("lstm", 200, 30, 500)
    # D weight gradient shape: dW[2000, 500] = dgates^T [12000 x 2000] x[12000 x 500]
    R = 12000
    dg = torch.randn(R, 2000, device=dev)
    x = torch.randn(R, 500, device=dev)
    dw = [torch.zeros(2000, 500, device=dev) for _ in range(4)]
    wg = ops.wgrad_stream(dev)
    main_s = torch.cuda.current_stream()
    ngemm = int(os.environ.get("NGEMM", "8"))

    def gemms():
        for i in range(ngemm):
            ops.gemm(ops.TN, 2000, 500, R, dg, 2000, x, 500, dw[i % 4], 500)

    def efwd():
        for _ in range(4):
            f()

    def timed(fn, n=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    ta = timed(efwd)
    tb = timed(gemms)
    print("E forward x4 alone %.3f ms; %d weight-gradient GEMMs (48 GFLOP... %.1f GFLOP each) alone %.3f ms (%.1f TFLOP/s)" % (
        ta, ngemm, 2e-9 * 2000 * 500 * R, tb, ngemm * 2e-12 * 2000 * 500 * R / (tb * 1e-3)), flush=True)

    def both(gemm_first):
        def run():
            wg.wait_stream(main_s)
            if gemm_first:
                with torch.cuda.stream(wg):
                    gemms()
                efwd()
            else:
                efwd_ev = None
                # queue the persistent launches first, the GEMMs right behind on the other stream
                efwd()
                with torch.cuda.stream(wg):
                    gemms()
            main_s.wait_stream(wg)
        return run
    for gf in (True, False):
        # per-stream end times
        run = both(gf)
        run(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        ends = []
        for _ in range(5):
            wg.wait_stream(main_s)
            s0 = torch.cuda.Event(enable_timing=True); s0.record()
            if gf:
                with torch.cuda.stream(wg):
                    gemms()
                    eg = torch.cuda.Event(enable_timing=True); eg.record()
                efwd()
                ee = torch.cuda.Event(enable_timing=True); ee.record()
            else:
                efwd()
                ee = torch.cuda.Event(enable_timing=True); ee.record()
                with torch.cuda.stream(wg):
                    gemms()
                    eg = torch.cuda.Event(enable_timing=True); eg.record()
            main_s.wait_stream(wg)
            torch.cuda.synchronize()
            ends.append((s0.elapsed_time(ee), s0.elapsed_time(eg)))
        print("both (%s queued first): E forward ends at %s ms, GEMMs end at %s ms  (sequential: %.3f)" % (
            "GEMMs" if gf else "E forward", " ".join("%.2f" % a for a, _ in ends), " ".join("%.2f" % b for _, b in ends), ta + tb), flush=True)
    assert not ops.rnn_timeout_flag()


if __name__ == "__main__":
    main()
