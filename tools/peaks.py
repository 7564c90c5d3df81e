#!/usr/bin/env python3
"""Re-confirm the machine peaks the roofline objects are priced against, on the GPU box (SURVEY 8d asks for it):
  * HBM: streaming kernels of this library on buffers far larger than the 256 MB Infinity Cache (add3: 2 reads + 1 write,
    axpby: 2 reads + 1 write, a plain device copy) -> achieved GB/s vs the 8 TB/s datasheet figure;
  * MFMA: the library's own GEMMs on a compute-bound shape (8192 x 8192 x 2048): exact fp32-input MFMA and the split-bf16
    plane GEMM (3 bf16 MFMAs per fp32-class multiply-accumulate), in fp32-equivalent TFLOP/s.
Writes one JSON object (profiles/r02_peaks.json)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from aas_enhancement_amd import ops


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    out = {"device": torch.cuda.get_device_name(0), "cus": ops.device_cus()}
    n = 1 << 29                      # 2 GiB per fp32 buffer
    a, b = torch.rand(n, device="cuda"), torch.rand(n, device="cuda")
    c = torch.empty_like(a)
    t = timeit(lambda: ops.lib().aas_add3_f32(ops.stream(), c.data_ptr(), a.data_ptr(), b.data_ptr(), None, n))
    out["hbm_add3_GBs"] = 3 * 4 * n / t / 1e9
    t = timeit(lambda: ops.axpby_(c, a, 0.5, 0.25))
    out["hbm_axpby_GBs"] = 3 * 4 * n / t / 1e9
    t = timeit(lambda: c.copy_(a))
    out["hbm_copy_GBs"] = 2 * 4 * n / t / 1e9
    del a, b, c
    M, N, K = 8192, 8192, 2048
    A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    C = torch.empty(M, N, device="cuda")
    ops.set_precision(0)
    t = timeit(lambda: ops.gemm(ops.NT, M, N, K, A, K, B, K, C, N), 5)
    out["mfma_fp32_exact_TFLOPs"] = 2.0 * M * N * K / t / 1e12
    ops.set_precision(1)
    t = timeit(lambda: ops.gemm(ops.NT, M, N, K, A, K, B, K, C, N), 5)
    out["mfma_split_bf16_inloop_TFLOPs_fp32_equiv"] = 2.0 * M * N * K / t / 1e12
    pa, pb = ops.split_planes(A, M, K), ops.split_planes(B, N, K)
    t = timeit(lambda: ops.gemm_planes(M, N, pa.Kp, pa, pb, C, N), 5)
    out["mfma_split_bf16_planes_TFLOPs_fp32_equiv"] = 2.0 * M * N * K / t / 1e12
    out["mfma_split_bf16_planes_TFLOPs_bf16_issued"] = 3 * out["mfma_split_bf16_planes_TFLOPs_fp32_equiv"]
    out["datasheet"] = {"hbm_GBs": 8000, "mfma_fp32_TFLOPs": 157.3, "mfma_bf16_dense_TFLOPs": 2500}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
