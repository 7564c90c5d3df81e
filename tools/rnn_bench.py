#!/usr/bin/env python3
"""Micro-benchmark / ablation of the persistent RNN kernels and GEMM shapes of the AAS step (GPU box only).
    python tools/rnn_bench.py [--flags 0,1,2,...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aas_enhancement_amd import _lib, ops


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def rnn_case(kind, T, N, H):
    G = 4 if kind == "lstm" else 3
    dev = "cuda"
    x = torch.randn(T, N, H, device=dev) * 0.5
    w = [torch.randn(G * H, H, device=dev) / H ** 0.5 for _ in range(4)]
    hout, gact, cst = ops._birnn_fwd(kind, x, *w)
    dy = torch.randn(T, N, H, device=dev)
    L = _lib.lib()
    sync = ops._sync_buf(x.device)
    xc = ops._xchg_buf(x.device, T, N, H, G)
    pre = torch.randn(T, N, 2, G * H, device=dev)
    dgx = torch.empty(T, N, 2, G * H, device=dev)
    dgh = torch.empty(T, N, 2, G * H, device=dev)
    s = _lib.stream()
    p = _lib.ptr
    fx = None
    if kind == "lstm":
        f = lambda: L.aas_lstm_fwd(s, T, N, H, p(pre), p(w[1]), p(w[3]), p(hout), p(gact), p(cst), p(sync), p(xc))
        # the same launch with the input projection inside (-> 3 when the shape / CU budget is not covered), and the GEMM it replaces
        fx = lambda: L.aas_lstm_fwd_x_ex(s, T, N, H, H, p(x), p(w[0]), p(w[2]), p(w[1]), p(w[3]), p(hout), p(gact), p(cst), p(sync), p(xc), None)
        wcat = torch.cat((w[0], w[2]), 0).contiguous()
        fx.gemm = lambda: ops.gemm(ops.NT, T * N, 2 * G * H, H, x.view(T * N, H), H, wcat, H, pre.view(T * N, 2 * G * H), 2 * G * H)
        b = lambda: L.aas_lstm_bwd(s, T, N, H, p(dy), p(w[1]), p(w[3]), p(gact), p(cst), p(dgx), p(sync), p(xc))
    else:
        f = lambda: L.aas_gru_fwd(s, T, N, H, p(pre), p(w[1]), p(w[3]), p(hout), p(gact), p(sync), p(xc))
        b = lambda: L.aas_gru_bwd(s, T, N, H, p(dy), p(w[1]), p(w[3]), p(hout), p(gact), p(dgx), p(dgh), p(sync), p(xc))
    f.fx = fx
    return f, b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", default="0")
    ap.add_argument("--gemm", action="store_true")
    ap.add_argument("--precision", type=int, default=1)
    ap.add_argument("--gflags", default="0")
    ap.add_argument("--skip-rnn", action="store_true")
    ap.add_argument("--cus", type=int, default=0, help="aas_set_rnn_cu_limit for the recurrent launches")
    ap.add_argument("--only", default="", help="comma list of case indices (0 lstm30, 1 gru30, 2 lstm60)")
    ap.add_argument("--tsweep", action="store_true", help="launch duration against T (4 ... 200): the intercept is what a persistent launch costs before its first step")
    a = ap.parse_args()
    L = _lib.lib()
    L.aas_set_precision(a.precision)
    L.aas_set_rnn_cu_limit(a.cus)
    cases = [("lstm", 200, 30, 500), ("gru", 85, 30, 1000), ("lstm", 200, 60, 500)]
    if a.only:
        cases = [cases[int(i)] for i in a.only.split(",")]
    if a.tsweep:
        for kind, _, N, H in cases:
            rows = []
            for T in (4, 8, 16, 50, 100, 200):
                f, b = rnn_case(kind, T, N, H)
                r = [T, timeit(f, 20), timeit(b, 20)]
                if f.fx is not None and a.precision == 0 and f.fx() == 0:
                    r.append(timeit(f.fx, 20))
                rows.append(r)
            for j, nm in enumerate(("fwd", "bwd", "fwdx")[:len(rows[0]) - 1]):
                (t0, y0), (t1, y1) = (rows[2][0], rows[2][j + 1]), (rows[-1][0], rows[-1][j + 1])
                slope = (y1 - y0) / (t1 - t0)
                print("%s N=%d H=%d %-4s: %s  -> %.2f us/step, intercept %.1f us" % (kind, N, H, nm, "  ".join("T=%d %.3f" % (r[0], r[j + 1]) for r in rows),
                                                                                     1e3 * slope, 1e3 * (y1 - slope * t1)), flush=True)
        return
    for kind, T, N, H in ([] if a.skip_rnn else cases):
        f, b = rnn_case(kind, T, N, H)
        for fl in [int(v) for v in a.flags.split(",")]:
            L.aas_set_debug_flags(fl)
            tf, tb = timeit(f), timeit(b)
            print("%s T=%d N=%d H=%d flags=%2d  fwd %.3f ms (%.2f us/step)  bwd %.3f ms (%.2f us/step)" % (kind, T, N, H, fl, tf, 1e3 * tf / T, tb, 1e3 * tb / T), flush=True)
            if f.fx is not None and a.precision == 0:
                if f.fx() == 0:
                    tx = timeit(f.fx)
                    L.aas_set_debug_flags(0)      # (the bits mean other things to the GEMM: 64 = "no operand loads" - its output would be
                    tg = timeit(f.fx.gemm)        #  garbage, and a NaN of the poison pattern in `pre` stalls the next plain launch)
                    L.aas_set_debug_flags(fl)
                    print("   input projection inside the launch: %.3f ms (%.2f us/step)  vs  GEMM %.3f + launch %.3f = %.3f ms" % (tx, 1e3 * tx / T, tg, tf, tg + tf), flush=True)
                    if fl & 64:
                        f.fx(); torch.cuda.synchronize()
                        st = ops._sync_buf(torch.device("cuda", 0)).view(torch.int64)[520:527].tolist()
                        print("   fwdx phases (us/step, WG0 wave0): wait %.2f  load+mfma %.2f  lds+barrier %.2f  gate+publish+xproj %.2f  total %.2f   | %d reloads in %d of %d steps" % (tuple(v * 0.01 / T for v in st[:5]) + (st[5], st[6], T)), flush=True)
                else:
                    print("   input projection inside the launch: not covered on this CU budget", flush=True)
            if fl & 64:
                for nm, fn in (("fwd", f), ("bwd", b)):
                    fn(); torch.cuda.synchronize()
                    st = ops._sync_buf(torch.device("cuda", 0)).view(torch.int64)[520:525].tolist()  # word 1040 = int64 index 520
                    print("   %s phases (us/step, WG0 wave0): wait %.2f  load+mfma %.2f  lds+barrier %.2f  gate+publish %.2f  total %.2f" % ((nm,) + tuple(v * 0.01 / T for v in st)), flush=True)
        L.aas_set_debug_flags(0)
    torch.cuda.synchronize()
    assert not ops.rnn_timeout_flag() or a.flags != "0", "timeout flag set"
    if a.gemm:
        shapes = [("NT pre", ops.NT, 6000, 2000, 500), ("NN dx", ops.NN, 6000, 500, 2000), ("TN dWih", ops.TN, 2000, 500, 6000),
                  ("NT gru pre", ops.NT, 2550, 3000, 1000), ("NN gru dx", ops.NN, 2550, 1000, 3000), ("TN gru dW", ops.TN, 3000, 1000, 2550),
                  ("NT first", ops.NT, 6000, 500, 80), ("NT final", ops.NT, 6000, 80, 500), ("NT big", ops.NT, 8192, 8192, 1024)]
        for name, mode, M, N, K in shapes:
            if mode == ops.NT:
                A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"); lda, ldb = K, K
            elif mode == ops.NN:
                A, B = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda"); lda, ldb = K, N
            else:
                A, B = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda"); lda, ldb = M, N
            C = torch.empty(M, N, device="cuda")
            res = []
            for fl in [int(v) for v in a.gflags.split(",")]:
                L.aas_set_debug_flags(fl)
                res.append("f%d %.3f" % (fl, timeit(lambda: ops.gemm(mode, M, N, K, A, lda, B, ldb, C, N), n=10)))
            L.aas_set_debug_flags(0)
            t = float(res[0].split()[1])
            print("gemm %-12s M=%5d N=%5d K=%5d  %.3f ms  %.1f TFLOP/s  | %s" % (name, M, N, K, t, 2.0 * M * N * K / t / 1e9, "  ".join(res)), flush=True)


if __name__ == "__main__":
    main()
