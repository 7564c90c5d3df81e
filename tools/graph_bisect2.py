#!/usr/bin/env python3
"""Trainer-level capture bisect (each level in its own subprocess)."""
import os, subprocess, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEVELS = ["tsg", "tsg_trainA"]

def run(level):
    sys.path.insert(0, ROOT)
    import torch, torch.nn as nn, numpy as np
    from aas_enhancement_amd import ops, prng
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    from aas_enhancement_amd.trainer_AAS import Trainer
    LABELS = "_'abcdefghijklmnopqrstuvwxyz "
    F, H, HA, M, N, T, L = 8, 16, 12, 8, 3, 60, 3
    G, D = stackedBRNN(I=F, H=H, L=4), stackedBRNN(I=F, H=H, L=4)
    A = DeepSpeech(nn.GRU, LABELS, HA, 3, True, 11, 2, M, 2, nFreq=F)
    cfg = types.SimpleNamespace(lr=1e-3, beta1=0.5, beta2=0.999, optimizer="adam", batch_size=N, expnum=0, lambda_k=0.001, gamma=0.5, gpu=0,
                                load_path="", mode="train", write_log=False, w_adversarial=1.0, w_acoustic=1.0, allow_ASR_update_iter=(0 if level.endswith('trainA') else 10**9), schedule="fused")
    tr = Trainer(cfg, None, models=(G, D, A)); tr.make_optimizers()
    dev = "cuda"
    x = torch.rand(N, F, T, device=dev) * 6; cl = torch.rand(N, F, T, device=dev) * 6
    meta = ops.ctc_prepare(torch.randint(1, 28, (N * L,), dtype=torch.int32), torch.full((N,), 15, dtype=torch.int32), torch.full((N,), L, dtype=torch.int32), dev)
    if not level.startswith('tsg'):
        tr._kt_dev = torch.zeros(1, device=dev, dtype=torch.float64); tr._g_out = torch.zeros(4, device=dev, dtype=torch.float64)
    og, oa, od = tr._opts
    def body(capturing=False):
        for f in tr._flat.values(): f.flat_g.zero_()
        enhanced = tr.G(x)
        leaf = enhanced.detach().requires_grad_(True)
        gsum = None
        acoustic = None
        if level in ("E_D_Aside", "opt", "kt", "full"):
            acoustic = tr._acoustic_branch(enhanced, None, None, None, N, meta)
        if level != "E":
            rs = torch.empty(2 * N, device=dev); rs[:N].copy_((-tr._kt_dev).to(torch.float32).expand(N)); rs[N:] = 1.0
            ae = tr.D(torch.cat([leaf, cl], 0), wgrad_row_scale=rs)
            l1 = ops.l1_sum(ae[:N], leaf) * 0.01; l2 = ops.l1_sum(ae[N:], cl) * 0.01
            (l1 + l2).backward(); gsum = leaf.grad
        else:
            gsum = torch.ones_like(enhanced)
        if level == "E_D_Amain":
            la = enhanced.detach().requires_grad_(True)
            prob = tr.ASR(la).transpose(0, 1)
            lc = ops.ctc_sum(prob, None, None, None, 0, meta) / N; lc.backward(); gsum = ops.add3(gsum, la.grad)
        if acoustic is not None:
            prob, l_CTC, leaf_a = acoustic
            torch.cuda.current_stream().wait_stream(tr._side)
            leaf_a.grad.record_stream(torch.cuda.current_stream())
            gsum = ops.add3(gsum, leaf_a.grad)
        enhanced.backward(gsum)
        ops.sync_wgrad()
        if level in ("opt", "kt", "full"):
            og.step_dev(); od.step_dev()
        if level in ("kt", "full"):
            packed = torch.stack([l1.detach().reshape(()), l2.detach().reshape(()), l_CTC.detach().reshape(())]).double()
            bal = 0.5 * packed[1] - packed[0]
            tr._kt_dev.copy_(torch.clamp(tr._kt_dev + 0.001 * bal, 0.0, 1.0))
            tr._g_out[:3].copy_(packed); tr._g_out[3:4].copy_(tr._kt_dev)
        return enhanced
    if level.startswith("tsg"):
        ny = (x, torch.randint(1, 28, (N * L,), dtype=torch.int32), torch.ones(N), torch.full((N,), L, dtype=torch.int32), torch.zeros(N, 1, T, dtype=torch.uint8))
        c2 = (cl, None, None, None, torch.zeros(N, 1, T, dtype=torch.uint8))
        for it in range(4):
            r = tr.train_step_graph(ny, c2, it)
        print("LEVEL", level, "OK", r["l_ctc"], flush=True)
        return
    if "full" in level:
        body = lambda capturing=False: tr._device_core(x, cl, N * T, N * T, meta, capturing)[0]
        if level.startswith("eager"):
            ny = (x, torch.randint(1, 28, (N * L,), dtype=torch.int32), torch.ones(N), torch.full((N,), L, dtype=torch.int32), torch.zeros(N, 1, T, dtype=torch.uint8))
            c2 = (cl, None, None, None, torch.zeros(N, 1, T, dtype=torch.uint8))
            tr.train_step(ny, c2, 0, log_norms=False)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(); body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body(True)
    g.replay(); g.replay(); torch.cuda.synchronize()
    print("LEVEL", level, "OK", float(out.float().abs().sum()), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for c in LEVELS:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=200)
            ok = [l for l in r.stdout.splitlines() if l.startswith("LEVEL")]
            err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "File \"/root/repo" in l]
            print(ok[0] if ok else "LEVEL %s FAILED rc=%d | %s" % (c, r.returncode, " ; ".join(err[-4:])[:600]), flush=True)
