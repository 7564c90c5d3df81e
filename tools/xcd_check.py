#!/usr/bin/env python3
"""A/B of the XCD-aware recurrent launches: debug bit 262144 = the plain 3-D grid (write-through publish stores), 524288 = the
XCD-aware grid with write-through stores, 0 = the default (XCD-aware grid, plain stores once a set is verified to share an XCD).
Results must be bit-identical; timing printed.  python tools/xcd_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aas_enhancement_amd import _lib, ops
import rnn_bench


def main():
    L = _lib.lib()
    L.aas_set_rnn_cu_limit(128)
    for kind, T, N, H in (("lstm", 200, 30, 500), ("lstm", 200, 60, 500)):
        G = 4
        dev = "cuda"
        torch.manual_seed(1)
        x = torch.randn(T, N, H, device=dev) * 0.5
        w = [torch.randn(G * H, H, device=dev) / H ** 0.5 for _ in range(4)]
        hout, gact, cst = ops._birnn_fwd(kind, x, *w)
        dy = torch.randn(T, N, H, device=dev)
        sync = ops._sync_buf(x.device)
        xc = ops._xchg_buf(x.device, T, N, H, G)
        s, p = _lib.stream(), _lib.ptr
        outs = {}
        for fl in (262144, 524288, 0):
            L.aas_set_debug_flags(fl)
            dgx = torch.zeros(T, N, 2, G * H, device=dev)
            b = lambda: L.aas_lstm_bwd(s, T, N, H, p(dy), p(w[1]), p(w[3]), p(gact), p(cst), p(dgx), p(sync), p(xc))
            t = rnn_bench.timeit(b, n=10)
            torch.cuda.synchronize()
            outs[fl] = dgx.clone()
            print("%s N=%d flags=%7d  bwd %.3f ms (%.2f us/step)  timeout=%s  max|diff vs 3-D grid| %.3e" %
                  (kind, N, fl, t, 1e3 * t / T, ops.rnn_timeout_flag(), (outs[fl] - outs[262144]).abs().max().item()), flush=True)
        L.aas_set_debug_flags(0)
        for cus in (128, 0):
            L.aas_set_rnn_cu_limit(cus)
            pre = torch.randn(T, N, 2, G * H, device=dev)
            fo = {}
            for fl in (262144, 524288, 0):
                L.aas_set_debug_flags(fl)
                ho, ga, cs = torch.zeros_like(hout), torch.zeros_like(gact), torch.zeros_like(cst)
                f = lambda: L.aas_lstm_fwd(s, T, N, H, p(pre), p(w[1]), p(w[3]), p(ho), p(ga), p(cs), p(sync), p(xc))
                t = rnn_bench.timeit(f, n=10)
                torch.cuda.synchronize()
                fo[fl] = (ho.clone(), cs.clone())
                print("%s N=%d cus=%3d flags=%7d  fwd %.3f ms (%.2f us/step)  timeout=%s  max|diff vs 3-D grid| %.3e" %
                      (kind, N, cus, fl, t, 1e3 * t / T, ops.rnn_timeout_flag(),
                       max((fo[fl][0] - fo[262144][0]).abs().max().item(), (fo[fl][1] - fo[262144][1]).abs().max().item())), flush=True)
            L.aas_set_debug_flags(0)
        L.aas_set_rnn_cu_limit(128)


def gru():
    L = _lib.lib()
    L.aas_set_rnn_cu_limit(128)
    T, N, H, G, dev = 85, 30, 1000, 3, "cuda"
    torch.manual_seed(2)
    x = torch.randn(T, N, H, device=dev) * 0.5
    w = [torch.randn(G * H, H, device=dev) / H ** 0.5 for _ in range(4)]
    hout, gact, _ = ops._birnn_fwd("gru", x, *w)
    dy = torch.randn(T, N, H, device=dev)
    sync, xc = ops._sync_buf(x.device), ops._xchg_buf(x.device, T, N, H, G)
    s, p = _lib.stream(), _lib.ptr
    outs = {}
    for fl in (262144, 524288, 0):
        L.aas_set_debug_flags(fl)
        dgx, dgh = torch.zeros(T, N, 2, G * H, device=dev), torch.zeros(T, N, 2, G * H, device=dev)
        b = lambda: L.aas_gru_bwd(s, T, N, H, p(dy), p(w[1]), p(w[3]), p(hout), p(gact), p(dgx), p(dgh), p(sync), p(xc))
        t = rnn_bench.timeit(b, n=10)
        torch.cuda.synchronize()
        outs[fl] = (dgx.clone(), dgh.clone())
        print("gru N=%d flags=%7d  bwd %.3f ms (%.2f us/step)  timeout=%s  max|diff vs 3-D grid| %.3e" %
              (N, fl, t, 1e3 * t / T, ops.rnn_timeout_flag(),
               max((outs[fl][0] - outs[262144][0]).abs().max().item(), (outs[fl][1] - outs[262144][1]).abs().max().item())), flush=True)
    L.aas_set_debug_flags(0)


if __name__ == "__main__":
    gru()
    main()
