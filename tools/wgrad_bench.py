#!/usr/bin/env python3
"""Weight-gradient products of one recurrent layer: transposed-plane path (planes_t + split_rows_t x3 + multi-problem NT plane
GEMM) against the row-major TN plane GEMM (aas_gemm_planes_tn), isolated on an idle GPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aas_enhancement_amd import ops


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(T, Nb, H, I, classes):
    dev = torch.device("cuda")
    GH, R_ = 4 * H, T * Nb
    dg = torch.randn(R_, 2 * GH, device=dev)
    x = torch.randn(R_, I, device=dev)
    h = torch.randn(2 * R_, H, device=dev)
    pd, px, ph = ops.split_planes(dg, R_, 2 * GH), ops.split_planes(x, R_, I), ops.split_planes(h, 2 * R_, H)
    outs = [torch.zeros(GH, I, device=dev), torch.zeros(GH, H, device=dev), torch.zeros(GH, I, device=dev), torch.zeros(GH, H, device=dev)]
    row = lambda pl: pl.Kp * 4
    one = torch.ones(1, device=dev)

    def tn():
        probs = []
        for n0, ns in classes:
            common = dict(A=pd.buf.data_ptr(), lda=row(pd), acols=pd.Kp, n0=n0, alpha=one)
            probs += [dict(common, B=px.buf.data_ptr(), ldb=row(px), bcols=px.Kp, acol0=0, M=2 * GH, N=I, K=T * ns, C0=outs[0].data_ptr(),
                           C1=outs[2].data_ptr(), msplit=GH, ldc=I, ta=0, tb=0),
                      dict(common, B=ph.buf.data_ptr(), ldb=row(ph), bcols=ph.Kp, acol0=0, M=GH, N=H, K=(T - 1) * ns, C0=outs[1].data_ptr(),
                           C1=0, msplit=GH, ldc=H, ta=1, tb=0),
                      dict(common, B=ph.buf.data_ptr() + R_ * row(ph), ldb=row(ph), bcols=ph.Kp, acol0=GH, M=GH, N=H, K=(T - 1) * ns,
                           C0=outs[3].data_ptr(), C1=0, msplit=GH, ldc=H, ta=0, tb=1)]
        ops.gemm_planes_tn(probs, classes[0][1], Nb, dev, accumulate=True)

    rs = torch.ones(Nb, device=dev)
    hout = h.view(2, T, Nb, H)

    def old():
        nbp = (Nb + 31) // 32 * 32
        K = T * nbp
        Kp = ops._kp(K + nbp)
        bf = torch.bfloat16
        dgT = torch.empty((2 * GH, 2 * Kp), device=dev, dtype=bf)
        ops.check(ops.lib().aas_planes_transpose(ops.stream(), ops.ptr(pd.buf), pd.Kp, T, Nb, nbp, 2 * GH, Kp, ops.ptr(dgT), ops.ptr(rs)), "t")
        xT = torch.empty((I, 2 * Kp), device=dev, dtype=bf)
        ops.split_planes_t_into(xT, x, T, Nb, nbp, I, Kp, ld=I)
        hT = torch.empty((2 * H, 2 * Kp), device=dev, dtype=bf)
        ops.split_planes_t_into(hT, hout, T, Nb, nbp, H, Kp, ld=H)
        ops.split_planes_t_into(hT[H:], hout, T, Nb, nbp, H, Kp, ld=H, off=T * Nb * H)
        rowb = 4 * Kp
        a_f, a_r = dgT.data_ptr(), dgT.data_ptr() + GH * rowb
        shift = nbp * 4
        ih = [(a_f, xT.data_ptr(), outs[0].data_ptr()), (a_r, xT.data_ptr(), outs[2].data_ptr())]
        hh = [(a_f + shift, hT.data_ptr(), outs[1].data_ptr()), (a_r, hT.data_ptr() + H * rowb + shift, outs[3].data_ptr())]
        if I == H:
            ops.gemm_planes_multi(GH, H, K, ih + hh, Kp, Kp, H)
        else:
            ops.gemm_planes_multi(GH, I, K, ih, Kp, Kp, I)
            ops.gemm_planes_multi(GH, H, K, hh, Kp, Kp, H)

    flops = 2.0 * R_ * 2 * GH * (I + H)
    t_old, t_tn = timeit(old), timeit(tn)
    print("T=%d Nb=%d H=%d I=%d classes=%d: transposed-plane path %.3f ms, TN plane GEMM %.3f ms (%.0f TFLOP/s fp32-equivalent)"
          % (T, Nb, H, I, len(classes), t_old, t_tn, flops / t_tn / 1e9))


if __name__ == "__main__":
    case(200, 30, 500, 500, [(0, 30)])
    case(200, 60, 500, 500, [(0, 30), (30, 30)])
    case(200, 60, 500, 500, [(0, 60)])
    case(200, 30, 500, 80, [(0, 30)])
    case(85, 30, 750, 1000, [(0, 30)])
