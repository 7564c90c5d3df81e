#!/usr/bin/env python3
"""Find which launches break hipGraph capture: each case runs in its own subprocess (a failure segfaults)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["gemm", "gemm_splitk", "memset_big", "lstm_fwd", "lstm_fwdbwd", "lstm_direct", "gru_fwdbwd", "bn", "conv", "ctc", "l1", "adam_dev", "twostream", "stacked"]

def run_case(name):
    sys.path.insert(0, ROOT)
    import torch, torch.nn as nn
    from aas_enhancement_amd import ops, _lib
    from aas_enhancement_amd.model import stackedBRNN, DeepSpeech
    dev = "cuda"
    T, N, H = 20, 6, 64
    x = torch.randn(T, N, H, device=dev, requires_grad=True)
    w = [torch.randn(4 * H, H, device=dev, requires_grad=True) for _ in range(4)]
    w3 = [torch.randn(3 * H, H, device=dev, requires_grad=True) for _ in range(4)]
    def body():
        if name == "gemm":
            a, b = torch.randn(300, 200, device=dev), torch.randn(100, 200, device=dev); c = torch.empty(300, 100, device=dev)
            ops.gemm(ops.NT, 300, 100, 200, a, 200, b, 200, c, 100); return c
        if name == "gemm_splitk":
            a, b = torch.randn(6000, 128, device=dev), torch.randn(6000, 64, device=dev); c = torch.empty(128, 64, device=dev)
            ops.gemm(ops.TN, 128, 64, 6000, a, 128, b, 64, c, 64); return c
        if name == "memset_big":
            b = torch.empty(64 << 20, dtype=torch.uint8, device=dev); b.fill_(255); return b
        if name == "lstm_fwd":
            with torch.no_grad():
                return ops.birnn_layer(x, *w, kind="lstm", residual=True)
        if name in ("lstm_fwdbwd", "lstm_direct"):
            if name == "lstm_direct":
                ops.DIRECT_WGRAD[0] = True
                for p in w:
                    if p.grad is None: p.grad = torch.zeros_like(p)
            y = ops.birnn_layer(x, *w, kind="lstm", residual=True); y.sum().backward()
            if name == "lstm_direct": ops.sync_wgrad()
            return y
        if name == "gru_fwdbwd":
            y = ops.birnn_layer(x, *w3, kind="gru", residual=False); y.sum().backward(); return y
        if name == "bn":
            g, b = torch.ones(H, device=dev, requires_grad=True), torch.zeros(H, device=dev, requires_grad=True)
            y = ops.batchnorm_rows(x, g, b, torch.zeros(H, device=dev), torch.ones(H, device=dev), 1e-5, 0.1, 128.0); y.sum().backward(); return y
        if name == "conv":
            xi = torch.randn(N, 40, 16, device=dev, requires_grad=True); W = torch.randn(8, 16, 11, device=dev, requires_grad=True); b = torch.zeros(8, device=dev, requires_grad=True)
            y = ops.conv1d_cl(xi, W, b, 2); y.sum().backward(); return y
        if name == "ctc":
            acts = torch.randn(15, 3, 29, device=dev, requires_grad=True)
            y = ops.ctc_sum(acts, None, None, None, 0, CTCMETA); y.backward(); return y
        if name == "l1":
            a, b = torch.randn(3, 8, 20, device=dev, requires_grad=True), torch.randn(3, 8, 20, device=dev)
            y = ops.l1_sum(a, b) * 0.5; y.backward(); return y
        if name == "adam_dev":
            from aas_enhancement_amd.dist import FlatBuffers
            from aas_enhancement_amd.optim import FlatAdam
            OPT.step_dev(); return OPT.flat.flat_p
        if name == "twostream":
            main = torch.cuda.current_stream(); SIDE.wait_stream(main)
            with torch.cuda.stream(SIDE):
                a = torch.randn(100, device=dev) * 2
            main.wait_stream(SIDE); return a + 1
        if name == "stacked":
            y = NET(torch.randn(3, 8, 40, device=dev)); y.sum().backward(); return y
    glb = globals()
    if name == "ctc":
        m = ops.ctc_prepare(torch.tensor([1, 2, 3, 4, 5, 6], dtype=torch.int32), torch.tensor([15, 12, 9], dtype=torch.int32), torch.tensor([3, 2, 1], dtype=torch.int32), dev)
        glb["CTCMETA"] = m
    if name == "adam_dev":
        from aas_enhancement_amd.dist import FlatBuffers
        from aas_enhancement_amd.optim import FlatAdam
        lin = nn.Linear(32, 32).cuda(); fb = FlatBuffers(lin); fb.flat_g.fill_(0.1)
        glb["OPT"] = FlatAdam(fb, lr=1e-3, betas=(0.5, 0.999), amsgrad=True)
    if name == "twostream":
        glb["SIDE"] = torch.cuda.Stream()
    if name == "stacked":
        glb["NET"] = stackedBRNN(I=8, H=16, L=2).cuda()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(); body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body()
    g.replay(); g.replay(); torch.cuda.synchronize()
    print("CASE", name, "OK", float(out.float().abs().sum()), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_case(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=120)
            ok = [l for l in r.stdout.splitlines() if l.startswith("CASE")]
            print(ok[0] if ok else "CASE %s FAILED rc=%d %s" % (c, r.returncode, (r.stderr.strip().splitlines() or [""])[-1][:200]), flush=True)
