#!/usr/bin/env python3
"""Error of the six-product bf16 GEMM (aas_set_precision(2): ops.gemm_planes6) against fp64, beside the fp32 MFMA GEMM's own and the
three-product fast mode's, at the K values of the AAS step (GPU box).  python tools/x6_error.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aas_enhancement_amd import ops


def errs(M, N, K, dist, seed=0):
    g = torch.Generator().manual_seed(seed)
    if dist == "randn":
        A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    elif dist == "pos":
        A, B = torch.rand(M, K, generator=g) + 0.5, torch.rand(N, K, generator=g) + 0.5
    else:   # wide dynamic range, random signs
        A = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-12, 12, (M, K), generator=g).float())
        B = torch.randn(N, K, generator=g) * torch.exp2(torch.randint(-12, 12, (N, K), generator=g).float())
    A, B = A.cuda(), B.cuda()
    ref = A.double() @ B.double().t()
    scale = (A.double().abs() @ B.double().abs().t())          # sum_k |a||b|: the natural error scale of a dot product
    out = {}
    ops.set_precision(0)
    C = torch.empty(M, N, device="cuda")
    ops.gemm(ops.NT, M, N, K, A, K, B, K, C, N)
    out["fp32"] = C.double()
    Kp = ops._kp(K)
    A3, B3 = ops.split_planes3(A, M, K), ops.split_planes3(B, N, K)
    assert torch.equal(A3.to_float()[:, :K], A), "three-term split is not exact"
    C6 = torch.empty(M, N, device="cuda")
    ops.gemm_planes6(M, N, Kp, A3, B3, C6, N)
    out["bf16x6"] = C6.double()
    A2, B2 = ops.split_planes(A, M, K), ops.split_planes(B, N, K)
    C3 = torch.empty(M, N, device="cuda")
    ops.gemm_planes(M, N, Kp, A2, B2, C3, N)
    out["bf16x3"] = C3.double()
    torch.cuda.synchronize()
    res = {}
    for k, v in out.items():
        e = (v - ref).abs() / scale
        res[k] = (float(e.max()), float(e.pow(2).mean().sqrt()))
    return res


def main():
    for dist in ("randn", "pos", "wide"):
        for K in (500, 1000, 6016):
            r = errs(384, 256, K, dist)
            print("%-6s K=%5d  " % (dist, K) + "  ".join("%s max %.2e rms %.2e" % (k, v[0], v[1]) for k, v in r.items()), flush=True)
    ops.set_precision(int(os.environ.get("AAS_PRECISION", "0")))


if __name__ == "__main__":
    main()
