"""Does the plain LSTM forward launch misbehave right after a launch with the input projection inside (same unmanaged exchange buffer)?
Per-launch durations, the sticky timeout word, outputs against the first plain launch."""
import sys

import torch

sys.path.insert(0, ".")
from aas_enhancement_amd import _lib, ops  # noqa: E402

L = _lib.lib()
L.aas_set_precision(0)
T, N, H, G = 200, 30, 500, 4
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(T, N, H, device=dev) * 0.5
w = [torch.randn(G * H, H, device=dev) / H ** 0.5 for _ in range(4)]
pre = torch.empty(T, N, 2, G * H, device=dev)
wcat = torch.cat((w[0], w[2]), 0).contiguous()
ops.gemm(ops.NT, T * N, 2 * G * H, H, x.view(T * N, H), H, wcat, H, pre.view(T * N, 2 * G * H), 2 * G * H)
sync = ops._sync_buf(x.device)
xc = ops._xchg_buf(x.device, T, N, H, G)
s, p = _lib.stream(), _lib.ptr


def outs():
    return torch.empty(2, T, N, H, device=dev), torch.empty(2, T, N, 4 * H, device=dev), torch.empty(2, T, N, H, device=dev)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); rc = fn(); e1.record(); torch.cuda.synchronize()
    return rc, e0.elapsed_time(e1)


ref = outs()
f = lambda o: L.aas_lstm_fwd(s, T, N, H, p(pre), p(w[1]), p(w[3]), p(o[0]), p(o[1]), p(o[2]), p(sync), p(xc))
fx = lambda o: L.aas_lstm_fwd_x_ex(s, T, N, H, H, p(x), p(w[0]), p(w[2]), p(w[1]), p(w[3]), p(o[0]), p(o[1]), p(o[2]), p(sync), p(xc), None)
dy = torch.randn(T, N, H, device=dev)
dgx = torch.empty(T, N, 2, G * H, device=dev)
gact0, cst0 = torch.empty(2, T, N, 4 * H, device=dev), torch.empty(2, T, N, H, device=dev)
b = lambda o: L.aas_lstm_bwd(s, T, N, H, p(dy), p(w[1]), p(w[3]), p(ref[1]), p(ref[2]), p(dgx), p(sync), p(xc))
gm = lambda o: (ops.gemm(ops.NT, T * N, 2 * G * H, H, x.view(T * N, H), H, wcat, H, pre.view(T * N, 2 * G * H), 2 * G * H), 0)[1]
print("plain first:", timed(lambda: f(ref)))
for FL in (0, 262144, 524288, 64):
    L.aas_set_debug_flags(FL)
    ops.clear_rnn_timeout()
    for name, fn in (("plain", f), ("gemm", gm), ("fwdx", fx), ("plain", f), ("plain", f), ("gemm", gm), ("fwdx", fx), ("plain", f), ("bptt", b), ("gemm", gm), ("plain", f), ("gemm", gm), ("plain", f)):
        o = outs()
        rc, ms = timed(lambda: fn(o))
        err = float((o[0] - ref[0]).abs().max()) if name in ("plain", "fwdx") else 0.0
        print("flags %7d  %-5s rc %d  %8.3f ms  max|h - h_ref| %.2e  timeout word %s" % (FL, name, rc, ms, err, ops.rnn_timeout_flag()), flush=True)
L.aas_set_debug_flags(0)
