import torch, os
from aas_enhancement_amd import ops
K=6016; M,N=384,256
g = torch.Generator().manual_seed(K)
A, B = torch.rand(M, K, generator=g) + 0.5, torch.rand(N, K, generator=g) + 0.5
A,B=A.cuda(),B.cuda()
ref = A.double() @ B.double().t(); scale = A.double().abs() @ B.double().abs().t()
ops.set_precision(0)
for i in range(8):
    C32 = torch.empty(M, N, device="cuda"); ops.gemm(ops.NT, M, N, K, A, K, B, K, C32, N)
    A3, B3 = ops.split_planes3(A, M, K), ops.split_planes3(B, N, K)
    C6 = torch.empty(M, N, device="cuda"); ops.gemm_planes6(M, N, A3.Kp, A3, B3, C6, N)
    torch.cuda.synchronize()
    e32, e6 = [((c.double() - ref).abs() / scale) for c in (C32, C6)]
    print("max %.3e %.3e  rms %.3e %.3e" % (e6.max(), e32.max(), e6.pow(2).mean().sqrt(), e32.pow(2).mean().sqrt()))
