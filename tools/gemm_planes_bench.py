"""Correctness + throughput of the plane GEMM (aas_gemm_planes) against fp64 and the in-loop split kernel."""
import sys

import torch

sys.path.insert(0, ".")
from aas_enhancement_amd import _lib, ops  # noqa: E402


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    torch.manual_seed(0)
    L = _lib.lib()
    flags = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
    shapes = [("pre E", 6000, 4000, 500), ("pre D", 12000, 4000, 500), ("dx E", 6000, 500, 4000), ("dx D", 12000, 500, 4000),
              ("wgrad E", 2000, 500, 6016), ("wgrad D", 2000, 500, 12032), ("gru pre", 3000, 6000, 1000), ("gru dx", 3000, 1000, 6000),
              ("first", 6000, 4000, 80), ("odd", 333, 77, 100), ("big", 8192, 8192, 1024)]
    for name, M, N, K in shapes:
        A = torch.randn(M, K, device="cuda")
        B = torch.randn(N, K, device="cuda")
        bias = torch.randn(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        pa, pb = ops.split_planes(A, M, K), ops.split_planes(B, N, K)
        ops.gemm_planes(M, N, pa.Kp, pa, pb, C, N, bias=bias)
        if M * N * K < 4e10:
            ref = (A.double() @ B.double().t() + bias.double())
            err = ((C.double() - ref).norm() / ref.norm()).item()
        else:
            err = float("nan")
        C2 = torch.empty(M, N, device="cuda")
        res = []
        for fl in flags:
            L.aas_set_debug_flags(fl)
            res.append("f%d %.3f" % (fl, timeit(lambda: ops.gemm_planes(M, N, pa.Kp, pa, pb, C, N))))
        L.aas_set_debug_flags(0)
        t_old = timeit(lambda: ops.gemm(ops.NT, M, N, K, A, K, B, K, C2, N))
        t_sa = timeit(lambda: ops.split_planes(A, M, K))
        t = float(res[0].split()[1])
        print("%-8s M=%5d N=%5d K=%5d  planes %.3f ms %6.1f TF | in-loop split %.3f ms %6.1f TF | split(A) %.3f ms | rel err %.2e | %s"
              % (name, M, N, K, t, 2.0 * M * N * K / t / 1e9, t_old, 2.0 * M * N * K / t_old / 1e9, t_sa, err, "  ".join(res)), flush=True)
    # transposed split check
    T, nb, Cc = 7, 30, 200
    x = torch.randn(T * nb, Cc, device="cuda")
    rs = torch.rand(nb, device="cuda") + 0.5
    pl, nbp = ops.split_planes_t(x, T, nb, Cc, row_scale=rs, extra=32)
    full = pl.to_float()
    rec = full[:, :T * nbp].view(Cc, T, nbp)
    ref = (x.view(T, nb, Cc) * rs.view(1, nb, 1)).permute(2, 0, 1)
    print("split_t err %.2e  pad max %.1e  tail max %.1e" % ((rec[:, :, :nb] - ref).abs().max().item() / ref.abs().max().item(),
                                                            rec[:, :, nb:].abs().max().item(), full[:, T * nbp:].abs().max().item()))


if __name__ == "__main__":
    main()
