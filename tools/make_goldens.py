#!/usr/bin/env python3
"""Generate golden fixtures by IMPORTING the reference's model.py (never copied).

Run only in the build container, where /root/reference exists:
    python tools/make_goldens.py [--skip-big]
It instantiates the reference's own stackedBRNN / DeepSpeech / L1Loss_mask
(/root/reference/Speech_enhancement_by_AAS/model.py), loads deterministic weights from the
portable PRNG (aas_enhancement_amd/prng.py), and runs the training steps restated from
trainer_AAS.py:131-194, trainer_DCE.py:116-127, trainer_FSEGAN.py:128-182 and
AM_training/train.py:297-349 with the substitutions listed in SURVEY.md 8(c)
(.data[0]->.item(), no .cuda(), bool mask, O=nFeat, warp-ctc -> F.ctc_loss(log_softmax),
A left in train mode, DeepSpeech(nFreq=F) built directly).
Outputs: tests/golden/*.npz (data only: inputs, expected outputs, seeds).
"""
import argparse
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/Speech_enhancement_by_AAS")
import model as REF  # noqa: E402  (the reference's model.py)

from aas_enhancement_amd import prng  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
LABELS = "_'abcdefghijklmnopqrstuvwxyz "  # 29 symbols, blank '_' at 0 (Common/labels.json)
torch.set_num_threads(8)


def load_weights(mod, seed, conv_std=None):
    sd = mod.state_dict()
    w = prng.fill_state_dict(sd, seed, conv_std=conv_std)
    for k, v in w.items():
        sd[k].copy_(torch.from_numpy(v))
    return w


def make_batch(N, Fdim, lens, seed, label_lens=None, lab_seed=None):
    """_collate_fn layout (loader_functions.py:47-73): sorted desc by T, zero padded."""
    T = max(lens)
    x = np.zeros((N, Fdim, T), np.float32)
    mask = np.zeros((N, 1, T), np.uint8)
    pct = np.zeros(N, np.float32)
    for n in range(N):
        x[n, :, :lens[n]] = prng.uniform(seed + 17 * n, (Fdim, lens[n]), 0.0, 6.0)
        mask[n, :, lens[n]:] = 1
        pct[n] = lens[n] / float(T)
    out = dict(inputs=x, mask=mask, pct=pct)
    if label_lens is not None:
        tg = []
        for n in range(N):
            tg.extend(prng.randint(lab_seed + n, (label_lens[n],), 1, 28).tolist())
        out["targets"] = np.asarray(tg, np.int32)
        out["target_sizes"] = np.asarray(label_lens, np.int32)
    return out


def t(x):
    return torch.from_numpy(np.asarray(x))


def ctc_sum(prob, targets, sizes, target_sizes):
    return F.ctc_loss(F.log_softmax(prob, 2), targets.long(), sizes.long(), target_sizes.long(),
                      blank=0, reduction="sum")


def gnorm(model):  # trainer_AAS.py:353-361
    g = 0
    for p in model.parameters():
        g = g + torch.pow(p.grad, 2).sum()
    return float(torch.pow(g, 0.5))


def aas_iteration(G, D, A, og, od, oa, ny, cl, cfg, kt, it, diff, enh_grads=None):
    """trainer_AAS.py:131-194, line by line, on reference modules.  `enh_grads` (a list) collects, through a tensor hook that
    changes nothing, the gradient arriving at `enhanced` in each backward pass: [adversarial (:148), CTC (:170)]."""
    G.zero_grad(); D.zero_grad(); A.zero_grad()
    inputs, targets, pct, target_sizes = t(ny["inputs"]), t(ny["targets"]), t(ny["pct"]).clone(), t(ny["target_sizes"])
    mask = t(ny["mask"]).bool()
    N = inputs.size(0)
    enhanced = G(inputs)
    if enh_grads is not None:
        enhanced.register_hook(lambda g: enh_grads.append(g.detach().clone().numpy()))
    enhanced_D = enhanced.detach()
    ae_ny_G = D(enhanced)
    l_adv_ny_G, _ = diff(ae_ny_G, enhanced, mask)
    l_adv_ny_G = l_adv_ny_G * cfg["w_adversarial"]
    l_adv_ny_G_data = l_adv_ny_G.item()
    l_adv_ny_G.backward(retain_graph=True)
    g_adv = gnorm(G)
    D.zero_grad()
    ae_ny_D = D(enhanced_D)
    l_adv_ny_D, _ = diff(ae_ny_D, enhanced_D, mask)
    l_adv_ny_D = l_adv_ny_D * (-kt) * cfg["w_adversarial"]
    l_adv_ny_D.backward()
    prob = A(enhanced)
    prob = prob.transpose(0, 1)
    T = prob.size(0)
    sizes = pct.mul_(int(T)).int()
    l_CTC = cfg["w_acoustic"] * ctc_sum(prob, targets, sizes, target_sizes) / N
    l_ctc_data = l_CTC.item()
    l_CTC.backward()
    g_ctc_adv = gnorm(G)
    cinputs, cmask = t(cl["inputs"]), t(cl["mask"]).bool()
    ae_cl = D(cinputs)
    l_adv_cl, _ = diff(ae_cl, cinputs, cmask)
    l_adv_cl = cfg["w_adversarial"] * l_adv_cl
    l_adv_cl.backward()
    l_adv_cl_data = l_adv_cl.item()
    grads = {}
    for nm, m in (("G", G), ("D", D), ("A", A)):
        for k, p in m.named_parameters():
            grads[nm + "." + k] = p.grad.detach().clone().numpy()
    og.step(); od.step()
    if it > cfg["allow_ASR_update_iter"]:
        oa.step()
    bal = cfg["gamma"] * l_adv_cl_data - l_adv_ny_G_data
    kt = kt + cfg["lambda_k"] * bal
    kt = max(min(1, kt), 0)
    conv = l_adv_cl_data + abs(bal)
    sc = dict(l_adv_ny_G=l_adv_ny_G_data, l_adv_cl=l_adv_cl_data, l_ctc=l_ctc_data, g_adv=g_adv,
              g_ctc_adv=g_ctc_adv, kt=kt, conv_measure=conv)
    return kt, sc, grads, enhanced.detach().numpy(), ae_ny_G.detach().numpy(), prob.detach().numpy(), sizes.numpy()


def build_aas(Fdim, H, HA, M, nA, seed):
    G = REF.stackedBRNN(I=Fdim, O=Fdim, H=H, L=4)
    D = REF.stackedBRNN(I=Fdim, O=Fdim, H=H, L=4)
    A = REF.DeepSpeech(rnn_type=nn.GRU, labels=LABELS, rnn_hidden_size=HA, rnn_layers=nA,
                       kernel_sz=11, stride=2, map=M, cnn_layers=2, nFreq=Fdim)
    load_weights(G, seed + 1)
    load_weights(D, seed + 2)
    load_weights(A, seed + 3, conv_std=0.1)
    return G, D, A


def adam(m, lr, amsgrad=True):
    return torch.optim.Adam(m.parameters(), lr=lr, betas=(0.5, 0.999), amsgrad=amsgrad)


def f1_tiny():
    """F1: full tensors at tiny size; ragged lengths; 3 iterations."""
    Fdim, H, HA, M = 8, 16, 12, 8
    G, D, A = build_aas(Fdim, H, HA, M, 5, seed=1000)
    init = {}
    for nm, m in (("G", G), ("D", D), ("A", A)):
        for k, v in m.state_dict().items():
            init[nm + "." + k] = v.clone().numpy()
    cfg = dict(w_adversarial=1.0, w_acoustic=1.0, gamma=0.5, lambda_k=0.001, allow_ASR_update_iter=0)
    lr = 1e-3  # larger than the 1e-5 default so 3 Adam steps move the params measurably
    og, od, oa = adam(G, lr), adam(D, lr), adam(A, lr)
    diff = REF.L1Loss_mask()
    out = dict(cfg_lr=lr, **{"cfg_" + k: v for k, v in cfg.items()})
    out.update({"init." + k: v for k, v in init.items()})
    kt = 0.3  # non-zero so the D-step contributes on iteration 0
    out["kt0"] = kt
    for it in range(3):
        ny = make_batch(3, Fdim, [60, 50, 38], 2000 + 100 * it, [4, 3, 2], 3000 + 10 * it)
        cl = make_batch(3, Fdim, [60, 47, 41], 4000 + 100 * it)
        kt, sc, grads, enh, ae, prob, sizes = aas_iteration(G, D, A, og, od, oa, ny, cl, cfg, kt, it, diff)
        for k, v in ny.items():
            out["it%d.ny.%s" % (it, k)] = v
        for k, v in cl.items():
            out["it%d.cl.%s" % (it, k)] = v
        for k, v in sc.items():
            out["it%d.%s" % (it, k)] = np.float64(v)
        out["it%d.enhanced" % it] = enh
        out["it%d.ae_ny_G" % it] = ae
        out["it%d.logits_tnc" % it] = prob
        out["it%d.sizes" % it] = sizes
        if it == 0:
            out.update({"it0.grad." + k: v for k, v in grads.items()})
    for nm, m in (("G", G), ("D", D), ("A", A)):
        for k, v in m.state_dict().items():
            out["final." + nm + "." + k] = v.clone().numpy()
    np.savez_compressed(os.path.join(OUT, "f1_aas_tiny.npz"), **out)
    print("F1", {k: out[k] for k in out if k.startswith("it2.") and np.ndim(out[k]) == 0})


def sample_idx(seed, shape, n):
    tot = int(np.prod(shape))
    return prng.randint(seed, (n,), 0, tot - 1)


def f2_dce():
    """F2: config 1 (N=4,F=80,T=200,H=128, reference-equivalent 4 layers), 5 DCE steps."""
    Fdim, H, N, T = 80, 128, 4, 200
    G = REF.stackedBRNN(I=Fdim, O=Fdim, H=H, L=4)
    load_weights(G, 5001)
    og = adam(G, 1e-4)
    diff = REF.L1Loss_mask()
    out = dict(weight_seed=5001, lr=1e-4, N=N, F=Fdim, T=T, H=H)
    losses, gns, sums, samples = [], [], [], []
    idx = sample_idx(77, (N, Fdim, T), 64)
    for it in range(5):
        x = prng.uniform(6000 + it, (N, Fdim, T), 0.0, 6.0)
        c = prng.uniform(7000 + it, (N, Fdim, T), 0.0, 6.0)
        mask = torch.zeros(N, 1, T, dtype=torch.bool)
        o = G(t(x))
        loss, nel = diff(o, t(c), mask)
        G.zero_grad()
        loss.backward()
        gns.append(gnorm(G))
        og.step()
        losses.append(loss.item())
        sums.append(float(o.detach().double().sum()))
        samples.append(o.detach().numpy().reshape(-1)[idx])
    out.update(losses=np.asarray(losses), g_norms=np.asarray(gns), out_sums=np.asarray(sums),
               sample_idx=idx, out_samples=np.stack(samples), nElement=int(nel),
               input_seed0=6000, clean_seed0=7000)
    np.savez_compressed(os.path.join(OUT, "f2_dce_config1.npz"), **out)
    print("F2 losses", losses)


def f3_config2():
    """F3: config 2 (N=30,T=200,F=80,E/D 4x500, A 2conv+5x1000 GRU): scalars + samples, 2 iterations."""
    Fdim, H, HA, M, N, T, L = 80, 500, 1000, 128, 30, 200, 20
    G, D, A = build_aas(Fdim, H, HA, M, 5, seed=9000)
    cfg = dict(w_adversarial=1.0, w_acoustic=1.0, gamma=0.5, lambda_k=0.001, allow_ASR_update_iter=0)
    og, od, oa = adam(G, 1e-5), adam(D, 1e-5), adam(A, 1e-5)
    diff = REF.L1Loss_mask()
    out = dict(weight_seed=9000, lr=1e-5, N=N, F=Fdim, T=T, H=H, HA=HA, M=M, L=L, kt0=0.0,
               noisy_seed=123, clean_seed=124, label_seed=125)
    kt = 0.0
    for it in range(2):
        ny = dict(inputs=prng.uniform(123 + 1000 * it, (N, Fdim, T), 0.0, 6.0), mask=np.zeros((N, 1, T), np.uint8),
                  pct=np.ones(N, np.float32),
                  targets=prng.randint(125 + 1000 * it, (N * L,), 1, 28).astype(np.int32),
                  target_sizes=np.full(N, L, np.int32))
        cl = dict(inputs=prng.uniform(124 + 1000 * it, (N, Fdim, T), 0.0, 6.0), mask=np.zeros((N, 1, T), np.uint8))
        kt, sc, grads, enh, ae, prob, sizes = aas_iteration(G, D, A, og, od, oa, ny, cl, cfg, kt, it, diff)
        for k, v in sc.items():
            out["it%d.%s" % (it, k)] = np.float64(v)
        ie = sample_idx(31 + it, enh.shape, 256)
        il = sample_idx(41 + it, prob.shape, 256)
        out["it%d.enh_idx" % it] = ie
        out["it%d.enh_samples" % it] = enh.reshape(-1)[ie]
        out["it%d.logit_idx" % it] = il
        out["it%d.logit_samples" % it] = prob.reshape(-1)[il]
        out["it%d.enh_sum" % it] = float(np.float64(enh.astype(np.float64).sum()))
        out["it%d.logit_abs_sum" % it] = float(np.abs(prob.astype(np.float64)).sum())
        if it == 0:
            for k in ("G.rnn1.rnn.weight_hh_l0", "G.first_linear.weight", "D.rnn4.rnn.weight_ih_l0_reverse",
                      "A.conv.0.weight", "A.rnns.2.rnn.weight_hh_l0", "A.fc.0.module.1.weight"):
                g = grads[k]
                ig = sample_idx(51, g.shape, 64)
                out["it0.gradsample_idx." + k] = ig
                out["it0.gradsample." + k] = g.reshape(-1)[ig]
                out["it0.gradnorm." + k] = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        print("F3 it", it, sc, flush=True)
    np.savez_compressed(os.path.join(OUT, "f3_aas_config2.npz"), **out)


def f3b_config2_kt():
    """F3b: iteration 0 of config 2 with kt0 = 0.3, so that the D-step term (trainer_AAS.py:156-160) is NOT identically zero at
    size: every parameter gradient of E, D and A as norm + 64 samples, the three networks' total gradient norms, and the two
    gradients that arrive at `enhanced` (adversarial, CTC).  Same weights / batches as F3 iteration 0."""
    Fdim, H, HA, M, N, T, L = 80, 500, 1000, 128, 30, 200, 20
    G, D, A = build_aas(Fdim, H, HA, M, 5, seed=9000)
    cfg = dict(w_adversarial=1.0, w_acoustic=1.0, gamma=0.5, lambda_k=0.001, allow_ASR_update_iter=0)
    og, od, oa = adam(G, 1e-5), adam(D, 1e-5), adam(A, 1e-5)
    diff = REF.L1Loss_mask()
    kt0 = 0.3
    out = dict(weight_seed=9000, lr=1e-5, N=N, F=Fdim, T=T, H=H, HA=HA, M=M, L=L, kt0=kt0, noisy_seed=123, clean_seed=124, label_seed=125)
    ny = dict(inputs=prng.uniform(123, (N, Fdim, T), 0.0, 6.0), mask=np.zeros((N, 1, T), np.uint8), pct=np.ones(N, np.float32),
              targets=prng.randint(125, (N * L,), 1, 28).astype(np.int32), target_sizes=np.full(N, L, np.int32))
    cl = dict(inputs=prng.uniform(124, (N, Fdim, T), 0.0, 6.0), mask=np.zeros((N, 1, T), np.uint8))
    eg = []
    kt, sc, grads, enh, ae, prob, sizes = aas_iteration(G, D, A, og, od, oa, ny, cl, cfg, kt0, 0, diff, enh_grads=eg)
    assert len(eg) == 2
    for k, v in sc.items():
        out["it0." + k] = np.float64(v)
    ie, il = sample_idx(31, enh.shape, 256), sample_idx(41, prob.shape, 256)
    out["it0.enh_idx"], out["it0.enh_samples"] = ie, enh.reshape(-1)[ie]
    out["it0.logit_idx"], out["it0.logit_samples"] = il, prob.reshape(-1)[il]
    tot = dict(G=0.0, D=0.0, A=0.0)
    for k, g in grads.items():
        ig = sample_idx(51, g.shape, min(64, g.size))
        out["it0.gradsample_idx." + k] = ig
        out["it0.gradsample." + k] = g.reshape(-1)[ig]
        sq = float((g.astype(np.float64) ** 2).sum())
        out["it0.gradnorm." + k] = float(np.sqrt(sq))
        tot[k[0]] += sq
    for nm in tot:
        out["it0.gradnorm_total." + nm] = float(np.sqrt(tot[nm]))
    for nm, g in (("adv", eg[0]), ("ctc", eg[1])):
        ig = sample_idx(53, g.shape, 256)
        out["it0.enh_grad_idx." + nm], out["it0.enh_grad_samples." + nm] = ig, g.reshape(-1)[ig]
        out["it0.enh_grad_norm." + nm] = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
    print("F3b", sc, {k: out[k] for k in out if "total" in k or "enh_grad_norm" in k}, flush=True)
    np.savez_compressed(os.path.join(OUT, "f3b_aas_config2_kt.npz"), **out)


def f1r_tiny_ragged_pair():
    """F1r: F1's tiny models, 3 iterations, with the noisy and the clean batch of DIFFERENT padded length - what the reference's
    two loaders deliver on every real iteration (trainer_AAS.py:136-138,175-177; each batch zero-padded to its own max T by
    loader_functions.py:47-73).  Iteration 0 / 2: noisy padded 60 vs clean padded 47; iteration 1: the CLEAN batch is the longer
    one (noisy 47 vs clean 60).  Full tensors; every parameter gradient of E / D / A at every iteration."""
    Fdim, H, HA, M = 8, 16, 12, 8
    G, D, A = build_aas(Fdim, H, HA, M, 5, seed=1000)
    init = {}
    for nm, m in (("G", G), ("D", D), ("A", A)):
        for k, v in m.state_dict().items():
            init[nm + "." + k] = v.clone().numpy()
    cfg = dict(w_adversarial=1.0, w_acoustic=1.0, gamma=0.5, lambda_k=0.001, allow_ASR_update_iter=0)
    lr = 1e-3
    og, od, oa = adam(G, lr), adam(D, lr), adam(A, lr)
    diff = REF.L1Loss_mask()
    out = dict(cfg_lr=lr, **{"cfg_" + k: v for k, v in cfg.items()})
    out.update({"init." + k: v for k, v in init.items()})
    kt = 0.3
    out["kt0"] = kt
    lens = [([60, 50, 38], [47, 41, 33]), ([47, 41, 33], [60, 50, 38]), ([60, 52, 45], [47, 47, 30])]
    for it in range(3):
        ny = make_batch(3, Fdim, lens[it][0], 2500 + 100 * it, [4, 3, 2], 3500 + 10 * it)
        cl = make_batch(3, Fdim, lens[it][1], 4500 + 100 * it)
        eg = []
        kt, sc, grads, enh, ae, prob, sizes = aas_iteration(G, D, A, og, od, oa, ny, cl, cfg, kt, it, diff, enh_grads=eg)
        for k, v in ny.items():
            out["it%d.ny.%s" % (it, k)] = v
        for k, v in cl.items():
            out["it%d.cl.%s" % (it, k)] = v
        for k, v in sc.items():
            out["it%d.%s" % (it, k)] = np.float64(v)
        out["it%d.enhanced" % it] = enh
        out["it%d.ae_ny_G" % it] = ae
        out["it%d.logits_tnc" % it] = prob
        out["it%d.sizes" % it] = sizes
        out["it%d.enh_grad.adv" % it], out["it%d.enh_grad.ctc" % it] = eg[0], eg[1]
        out.update({"it%d.grad.%s" % (it, k): v for k, v in grads.items()})
    for nm, m in (("G", G), ("D", D), ("A", A)):
        for k, v in m.state_dict().items():
            out["final." + nm + "." + k] = v.clone().numpy()
    np.savez_compressed(os.path.join(OUT, "f1r_aas_tiny_ragged_pair.npz"), **out)
    print("F1r", {k: out[k] for k in out if k.startswith("it2.") and np.ndim(out[k]) == 0})


def f3r_config2_ragged_pair():
    """F3r: iteration 0 of config 2 with kt0 = 0.3 and a noisy / clean pair of different padded length: noisy T = 200, clean
    T = 184 (F3b's clean batch cut to its first 184 frames: the pair bench.py times as `ragged_pair*`).  Same contents as F3b: all
    scalars, 256 enhanced / logit samples, every E / D / A gradient as norm + 64 samples, totals, the two gradients arriving at
    `enhanced`."""
    Fdim, H, HA, M, N, T, L, Tc = 80, 500, 1000, 128, 30, 200, 20, 184
    G, D, A = build_aas(Fdim, H, HA, M, 5, seed=9000)
    cfg = dict(w_adversarial=1.0, w_acoustic=1.0, gamma=0.5, lambda_k=0.001, allow_ASR_update_iter=0)
    og, od, oa = adam(G, 1e-5), adam(D, 1e-5), adam(A, 1e-5)
    diff = REF.L1Loss_mask()
    kt0 = 0.3
    out = dict(weight_seed=9000, lr=1e-5, N=N, F=Fdim, T=T, T_clean=Tc, H=H, HA=HA, M=M, L=L, kt0=kt0, noisy_seed=123, clean_seed=124, label_seed=125)
    ny = dict(inputs=prng.uniform(123, (N, Fdim, T), 0.0, 6.0), mask=np.zeros((N, 1, T), np.uint8), pct=np.ones(N, np.float32),
              targets=prng.randint(125, (N * L,), 1, 28).astype(np.int32), target_sizes=np.full(N, L, np.int32))
    cl = dict(inputs=np.ascontiguousarray(prng.uniform(124, (N, Fdim, T), 0.0, 6.0)[:, :, :Tc]), mask=np.zeros((N, 1, Tc), np.uint8))
    eg = []
    kt, sc, grads, enh, ae, prob, sizes = aas_iteration(G, D, A, og, od, oa, ny, cl, cfg, kt0, 0, diff, enh_grads=eg)
    assert len(eg) == 2
    for k, v in sc.items():
        out["it0." + k] = np.float64(v)
    ie, il = sample_idx(31, enh.shape, 256), sample_idx(41, prob.shape, 256)
    out["it0.enh_idx"], out["it0.enh_samples"] = ie, enh.reshape(-1)[ie]
    out["it0.logit_idx"], out["it0.logit_samples"] = il, prob.reshape(-1)[il]
    tot = dict(G=0.0, D=0.0, A=0.0)
    for k, g in grads.items():
        ig = sample_idx(51, g.shape, min(64, g.size))
        out["it0.gradsample_idx." + k] = ig
        out["it0.gradsample." + k] = g.reshape(-1)[ig]
        sq = float((g.astype(np.float64) ** 2).sum())
        out["it0.gradnorm." + k] = float(np.sqrt(sq))
        tot[k[0]] += sq
    for nm in tot:
        out["it0.gradnorm_total." + nm] = float(np.sqrt(tot[nm]))
    for nm, g in (("adv", eg[0]), ("ctc", eg[1])):
        ig = sample_idx(53, g.shape, 256)
        out["it0.enh_grad_idx." + nm], out["it0.enh_grad_samples." + nm] = ig, g.reshape(-1)[ig]
        out["it0.enh_grad_norm." + nm] = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
    print("F3r", sc, {k: out[k] for k in out if "total" in k or "enh_grad_norm" in k}, flush=True)
    np.savez_compressed(os.path.join(OUT, "f3r_aas_config2_ragged_pair.npz"), **out)


def f3c_thread_spread():
    """F3c: how far the reference's OWN fp32 result moves when only the CPU thread count changes (summation orders inside the
    BLAS / oneDNN kernels): iteration 0 of F3b (kt0 = 0.3) run with 8, 3 and 1 threads; per parameter the 64 gradient samples of F3b
    as min / max over the runs.  The yardstick for "is a rounding-level difference of this build's kernels visible at the gradient
    level" (tests/test_gpu_round5.py: the 22-bit partial-sum exchange of the reduce-scatter BPTT)."""
    Fdim, H, HA, M, N, T, L = 80, 500, 1000, 128, 30, 200, 20
    cfg = dict(w_adversarial=1.0, w_acoustic=1.0, gamma=0.5, lambda_k=0.001, allow_ASR_update_iter=0)
    ny = dict(inputs=prng.uniform(123, (N, Fdim, T), 0.0, 6.0), mask=np.zeros((N, 1, T), np.uint8), pct=np.ones(N, np.float32),
              targets=prng.randint(125, (N * L,), 1, 28).astype(np.int32), target_sizes=np.full(N, L, np.int32))
    cl = dict(inputs=prng.uniform(124, (N, Fdim, T), 0.0, 6.0), mask=np.zeros((N, 1, T), np.uint8))
    runs = []
    threads = (8, 3, 1)
    for nt in threads:
        torch.set_num_threads(nt)
        G, D, A = build_aas(Fdim, H, HA, M, 5, seed=9000)
        og, od, oa = adam(G, 1e-5), adam(D, 1e-5), adam(A, 1e-5)
        _, sc, grads, *_ = aas_iteration(G, D, A, og, od, oa, ny, cl, cfg, 0.3, 0, REF.L1Loss_mask())
        runs.append({k: g.reshape(-1)[sample_idx(51, g.shape, min(64, g.size))].copy() for k, g in grads.items()})
        print("F3c threads", nt, sc["l_ctc"], flush=True)
    torch.set_num_threads(8)
    out = dict(threads=np.asarray(threads), kt0=0.3)
    for k in runs[0]:
        st = np.stack([r[k] for r in runs])
        out["lo." + k], out["hi." + k] = st.min(0), st.max(0)
    np.savez_compressed(os.path.join(OUT, "f3c_thread_spread.npz"), **out)
    worst = max(float(np.abs(out["hi." + k] - out["lo." + k]).max() / (np.abs(out["hi." + k]).max() + 1e-30)) for k in runs[0])
    print("F3c worst spread / max|sample|:", worst)


def f4_ops():
    """F4: per-op vectors from the reference's own BRNN / BatchRNN / DeepSpeech.conv / L1Loss_mask."""
    out = {}
    torch.manual_seed(0)
    # LSTM BRNN fwd+bwd (model.py:88-105)
    for tag, (T, N, H) in dict(s=(7, 2, 5), m=(23, 5, 24)).items():
        for kind, cls in (("lstm", nn.LSTM), ("gru", nn.GRU)):
            m = REF.BRNN(H, H, rnn_type=cls, bidirectional=True)
            load_weights(m, 1 + T + (0 if kind == "lstm" else 50))
            x = t(prng.normal(11 + T, (T, N, H))).requires_grad_(True)
            y = m(x)
            gy = t(prng.normal(12 + T, (T, N, H)))
            y.backward(gy)
            p = "brnn_%s_%s." % (kind, tag)
            out[p + "x"], out[p + "y"], out[p + "gy"], out[p + "gx"] = x.detach().numpy(), y.detach().numpy(), gy.numpy(), x.grad.numpy()
            for k, v in m.named_parameters():
                out[p + "w." + k] = v.detach().numpy()
                out[p + "gw." + k] = v.grad.numpy()
    # BatchRNN with SequenceWise BN, GRU, in != hidden (model.py:66-86)
    m = REF.BatchRNN(6, 9, rnn_type=nn.GRU, bidirectional=True, batch_norm=True)
    load_weights(m, 301)
    x = t(prng.normal(302, (11, 3, 6), 0.5, 2.0)).requires_grad_(True)
    y = m(x)
    gy = t(prng.normal(303, (11, 3, 9)))
    y.backward(gy)
    p = "batchrnn."
    out[p + "x"], out[p + "y"], out[p + "gy"], out[p + "gx"] = x.detach().numpy(), y.detach().numpy(), gy.numpy(), x.grad.numpy()
    for k, v in m.state_dict().items():
        out[p + "sd." + k] = v.numpy().copy()
    for k, v in m.named_parameters():
        out[p + "gw." + k] = v.grad.numpy()
    # DeepSpeech conv front-end: conv(k11,s2)+BN+LeakyReLU(slope=map), conv(k11,s1)+BN+LReLU (model.py:288-301)
    A = REF.DeepSpeech(rnn_type=nn.GRU, labels=LABELS, rnn_hidden_size=6, rnn_layers=2, map=8, nFreq=10)
    load_weights(A, 401, conv_std=0.1)
    x = t(prng.uniform(402, (3, 10, 50), 0.0, 6.0)).requires_grad_(True)
    y = A.conv(x)
    gy = t(prng.normal(403, tuple(y.shape)))
    y.backward(gy)
    p = "dsconv."
    out[p + "x"], out[p + "y"], out[p + "gy"], out[p + "gx"] = x.detach().numpy(), y.detach().numpy(), gy.numpy(), x.grad.numpy()
    for k, v in A.conv.state_dict().items():
        out[p + "sd." + k] = v.numpy().copy()
    for k, v in A.conv.named_parameters():
        out[p + "gw." + k] = v.grad.numpy()
    # L1Loss_mask with padding: shows the mask is NOT applied (model.py:23-31)
    a = t(prng.normal(501, (3, 4, 9))).requires_grad_(True)
    b = t(prng.normal(502, (3, 4, 9))).requires_grad_(True)
    mask = torch.zeros(3, 1, 9, dtype=torch.bool)
    mask[1, :, 6:] = True
    mask[2, :, 4:] = True
    loss, nel = REF.L1Loss_mask()(a, b, mask)
    loss.backward()
    out["l1.a"], out["l1.b"], out["l1.mask"] = a.detach().numpy(), b.detach().numpy(), mask.numpy().astype(np.uint8)
    out["l1.loss"], out["l1.nElement"], out["l1.ga"], out["l1.gb"] = loss.item(), int(nel), a.grad.numpy(), b.grad.numpy()
    # stackedBRNN.forward_paired (model.py:233-238)
    Dp = REF.stackedBRNN(I=12, O=6, H=10, L=4)
    load_weights(Dp, 601)
    xa, xb = t(prng.normal(602, (2, 6, 13))), t(prng.normal(603, (2, 6, 13)))
    out["paired.a"], out["paired.b"], out["paired.y"] = xa.numpy(), xb.numpy(), Dp.forward_paired(xa, xb).detach().numpy()
    for k, v in Dp.state_dict().items():
        out["paired.sd." + k] = v.numpy().copy()
    # stackedBRNN.forward_with_intermediate_output (model.py:240-252): [output, h4 as N x H x T]
    Gi = REF.stackedBRNN(I=6, O=6, H=10, L=4)
    load_weights(Gi, 611)
    xi = t(prng.normal(612, (3, 6, 17)))
    yo, yh = Gi.forward_with_intermediate_output(xi)
    out["inter.x"], out["inter.y"], out["inter.h"] = xi.numpy(), yo.detach().numpy(), yh.detach().numpy()
    for k, v in Gi.state_dict().items():
        out["inter.sd." + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "f4_ops.npz"), **out)
    print("F4 keys", len(out))


def f5_fsegan_am():
    """F5: FSEGAN (intended + as-written) and AM steps at tiny size."""
    out = {}
    Fdim, H = 8, 12
    for variant in ("intended", "as_written"):
        G = REF.stackedBRNN(I=Fdim, O=Fdim, H=H, L=4)
        D = REF.stackedBRNN(I=2 * Fdim, O=Fdim, H=H, L=4)
        load_weights(G, 7001)
        load_weights(D, 7002)
        og, od = adam(G, 1e-3), adam(D, 1e-3)
        diff = REF.L1Loss_mask()
        kt, w = 0.2, 0.01
        for it in range(2):
            b = make_batch(3, Fdim, [30, 26, 19], 7100 + it)
            mixture, mask = t(b["inputs"]), t(b["mask"]).bool()
            cl = make_batch(3, Fdim, [30, 26, 19], 7200 + it)["inputs"]
            cleans = t(cl)
            G.zero_grad(); D.zero_grad()
            enhanced = G(mixture)
            enhanced_D = enhanced.detach()
            ae = D.forward_paired(enhanced, mixture)
            l_g, _ = diff(ae, enhanced, mask)
            l_g = l_g * w
            l_g_data = l_g.item()
            l_g.backward(retain_graph=True)
            D.zero_grad()
            ae_d = D.forward_paired(enhanced_D, mixture)
            l_d, _ = diff(ae_d, enhanced_D, mask)
            (l_d * (-kt) * w).backward()
            dce, _ = diff(enhanced, cleans, mask)
            if variant == "intended":
                dce.backward()
            ae_cl = D.forward_paired(cleans, mixture)
            l_cl, _ = diff(ae_cl, cleans, mask)
            l_cl = w * l_cl
            l_cl.backward()
            l_cl_data = l_cl.item()
            gn = gnorm(G)
            og.step(); od.step()
            bal = 0.5 * l_cl_data - l_g_data
            kt = max(min(1, kt + 0.001 * bal), 0)
            p = "fsegan_%s.it%d." % (variant, it)
            out[p + "l_adv_ny_G"], out[p + "l_adv_cl"], out[p + "dce"], out[p + "kt"], out[p + "g_norm"] = l_g_data, l_cl_data, dce.item(), kt, gn
        for nm, m in (("G", G), ("D", D)):
            for k, v in m.state_dict().items():
                out["fsegan_%s.final.%s.%s" % (variant, nm, k)] = v.numpy().copy()
    # AM step (AM_training/train.py:297-349), plain Adam
    A = REF.DeepSpeech(rnn_type=nn.GRU, labels=LABELS, rnn_hidden_size=12, rnn_layers=3, map=8, nFreq=Fdim)
    load_weights(A, 8001, conv_std=0.1)
    opt = torch.optim.Adam(A.parameters(), lr=1e-3)
    for it in range(2):
        b = make_batch(3, Fdim, [60, 50, 38], 8100 + it, [4, 3, 2], 8200 + it)
        o = A(t(b["inputs"])).transpose(0, 1)
        sizes = t(b["pct"]).clone().mul_(int(o.size(0))).int()
        loss = ctc_sum(o, t(b["targets"]), sizes, t(b["target_sizes"])) / 3
        opt.zero_grad()
        loss.backward()
        opt.step()
        out["am.it%d.loss" % it] = loss.item()
        out["am.it%d.logits" % it] = o.detach().numpy()
    for k, v in A.state_dict().items():
        out["am.final." + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "f5_fsegan_am.npz"), **out)
    print("F5", {k: v for k, v in out.items() if np.ndim(v) == 0})


def f6_fsegan_config4():
    """F6: BASELINE config 4 - FSEGAN at size (N=30,T=200,F=80; E 4x500, D I=160 O=80 4x500, w_adversarial 0.01),
    intended step (DCE back-propagated, trainer_FSEGAN.py:128-182 with the fix list of SURVEY 0.13), 2 iterations:
    scalars + 256 sampled enhanced elements + sampled gradients."""
    Fdim, H, N, T = 80, 500, 30, 200
    G = REF.stackedBRNN(I=Fdim, O=Fdim, H=H, L=4)
    D = REF.stackedBRNN(I=2 * Fdim, O=Fdim, H=H, L=4)
    load_weights(G, 9101)
    load_weights(D, 9102)
    lr, w, kt = 1e-5, 0.01, 0.1
    og, od = adam(G, lr), adam(D, lr)
    diff = REF.L1Loss_mask()
    out = dict(weight_seed_G=9101, weight_seed_D=9102, lr=lr, w_adversarial=w, kt0=kt, N=N, F=Fdim, T=T, H=H,
               mixture_seed0=9110, clean_seed0=9120)
    for it in range(2):
        mixture = t(prng.uniform(9110 + it, (N, Fdim, T), 0.0, 6.0))
        cleans = t(prng.uniform(9120 + it, (N, Fdim, T), 0.0, 6.0))
        mask = torch.zeros(N, 1, T, dtype=torch.bool)
        G.zero_grad(); D.zero_grad()
        enhanced = G(mixture)
        enhanced_D = enhanced.detach()
        ae = D.forward_paired(enhanced, mixture)
        l_g, _ = diff(ae, enhanced, mask)
        l_g = l_g * w
        l_g_data = l_g.item()
        l_g.backward(retain_graph=True)
        D.zero_grad()
        ae_d = D.forward_paired(enhanced_D, mixture)
        l_d, _ = diff(ae_d, enhanced_D, mask)
        (l_d * (-kt) * w).backward()
        dce, _ = diff(enhanced, cleans, mask)
        dce.backward()
        ae_cl = D.forward_paired(cleans, mixture)
        l_cl, _ = diff(ae_cl, cleans, mask)
        l_cl = w * l_cl
        l_cl.backward()
        l_cl_data = l_cl.item()
        gn = gnorm(G)
        gd = gnorm(D)
        if it == 0:
            for nm, m, keys in (("G", G, ("rnn2.rnn.weight_hh_l0_reverse", "first_linear.weight")),
                                ("D", D, ("first_linear.weight", "rnn1.rnn.weight_ih_l0", "final_linear.bias"))):
                gr = dict(m.named_parameters())
                for k in keys:
                    g = gr[k].grad.detach().numpy()
                    ig = sample_idx(61, g.shape, min(64, g.size))
                    out["it0.gradsample_idx.%s.%s" % (nm, k)] = ig
                    out["it0.gradsample.%s.%s" % (nm, k)] = g.reshape(-1)[ig]
                    out["it0.gradnorm.%s.%s" % (nm, k)] = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        og.step(); od.step()
        bal = 0.5 * l_cl_data - l_g_data
        kt = max(min(1, kt + 0.001 * bal), 0)
        p = "it%d." % it
        enh = enhanced.detach().numpy()
        ie = sample_idx(71 + it, enh.shape, 256)
        out.update({p + "l_adv_ny_G": l_g_data, p + "l_adv_cl": l_cl_data, p + "dce": dce.item(), p + "kt": kt,
                    p + "g_norm": gn, p + "d_norm": gd, p + "enh_idx": ie, p + "enh_samples": enh.reshape(-1)[ie],
                    p + "enh_sum": float(enh.astype(np.float64).sum())})
        print("F6 it", it, {k: out[p + k] for k in ("l_adv_ny_G", "l_adv_cl", "dce", "kt", "g_norm", "d_norm")}, flush=True)
    np.savez_compressed(os.path.join(OUT, "f6_fsegan_config4.npz"), **out)


def f7_am_config5():
    """F7: BASELINE config 5 per-GPU step - AM_training/train.py:297-349 at size (N=30,T=200,F=80; A = 2xconv1d(128,k11)
    + 5x1000 BiGRU + fc, plain Adam lr 1e-4), 2 iterations: loss, 256 logit samples, gradient norms + samples."""
    Fdim, HA, M, N, T, L = 80, 1000, 128, 30, 200, 20
    A = REF.DeepSpeech(rnn_type=nn.GRU, labels=LABELS, rnn_hidden_size=HA, rnn_layers=5, kernel_sz=11, stride=2,
                       map=M, cnn_layers=2, nFreq=Fdim)
    load_weights(A, 9203, conv_std=0.1)
    lr = 1e-4
    opt = torch.optim.Adam(A.parameters(), lr=lr)
    out = dict(weight_seed=9203, lr=lr, N=N, F=Fdim, T=T, HA=HA, M=M, L=L, input_seed0=9210, label_seed0=9220)
    for it in range(2):
        x = t(prng.uniform(9210 + it, (N, Fdim, T), 0.0, 6.0))
        targets = t(prng.randint(9220 + it, (N * L,), 1, 28).astype(np.int32))
        target_sizes = torch.full((N,), L, dtype=torch.int32)
        o = A(x).transpose(0, 1)
        sizes = torch.ones(N).mul_(int(o.size(0))).int()
        loss = ctc_sum(o, targets, sizes, target_sizes) / N
        opt.zero_grad()
        loss.backward()
        p = "it%d." % it
        if it == 0:
            gr = dict(A.named_parameters())
            tot = 0.0
            for k, v in gr.items():
                tot += float((v.grad.double() ** 2).sum())
            out["it0.gradnorm_total"] = float(np.sqrt(tot))
            for k in ("conv.0.weight", "conv.3.weight", "rnns.0.rnn.weight_ih_l0", "rnns.2.rnn.weight_hh_l0_reverse",
                      "rnns.4.batch_norm.module.weight", "fc.0.module.1.weight"):
                g = gr[k].grad.detach().numpy()
                ig = sample_idx(81, g.shape, 64)
                out["it0.gradsample_idx." + k] = ig
                out["it0.gradsample." + k] = g.reshape(-1)[ig]
                out["it0.gradnorm." + k] = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        opt.step()
        lg = o.detach().numpy()
        il = sample_idx(91 + it, lg.shape, 256)
        out.update({p + "loss": loss.item(), p + "logit_idx": il, p + "logit_samples": lg.reshape(-1)[il],
                    p + "logit_abs_sum": float(np.abs(lg.astype(np.float64)).sum())})
        print("F7 it", it, loss.item(), flush=True)
    np.savez_compressed(os.path.join(OUT, "f7_am_config5.npz"), **out)


def f8_host_side():
    """F8: host-side contract vectors from the reference's OWN loader_functions._collate_fn / _collate_fn_paired /
    FeatDataset.parse_transcript (loader_functions.py:37-105) and AM_training/decoder.py GreedyDecoder
    (convert_to_strings / process_string / wer / cer, :45-74,146-201).  decoder.py imports the third-party
    `Levenshtein` C extension, absent from this image: a plain dynamic-programming edit distance is injected under that
    module name for the import (the quantity is the textbook Levenshtein distance; nothing else of the package is used)."""
    import tempfile
    import types as _types
    import loader_functions as LF   # the reference's
    lev = _types.ModuleType("Levenshtein")

    def distance(a, b):
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            cur = [i]
            for j, cb in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
            prev = cur
        return prev[-1]
    lev.distance = distance
    sys.modules["Levenshtein"] = lev
    sys.path.insert(0, "/root/reference/AM_training")
    import decoder as RDEC          # the reference's AM_training/decoder.py
    out = {}
    # ---- collate: ragged batch given in NON-sorted order, with ties
    lens, Fdim = [17, 31, 31, 9, 24], 6
    feats = [prng.uniform(1300 + i, (Fdim, n), 0.0, 6.0) for i, n in enumerate(lens)]
    paired = [prng.uniform(1400 + i, (Fdim, n), 0.0, 6.0) for i, n in enumerate(lens)]
    labs = [prng.randint(1500 + i, (k,), 1, 28).tolist() for i, k in enumerate([3, 5, 0, 2, 4])]
    for i in range(len(lens)):
        out["collate.feat%d" % i] = feats[i]
        out["collate.paired%d" % i] = paired[i]
        out["collate.label%d" % i] = np.asarray(labs[i], np.int32)
    r = LF._collate_fn([(t(feats[i]), labs[i]) for i in range(len(lens))])
    for k, v in zip(("inputs", "targets", "pct", "target_sizes", "mask"), r):
        out["collate.out." + k] = v.numpy()
    r = LF._collate_fn_paired([(t(feats[i]), labs[i], t(paired[i])) for i in range(len(lens))])
    for k, v in zip(("inputs", "outputs", "mask", "targets", "pct", "target_sizes"), r):
        out["collate_paired.out." + k] = v.numpy()
    # ---- parse_transcript: unknown characters AND index-0 characters are dropped (filter(None, ...))
    texts = ["hello world", "it's  a_test\nwith newline", "UPPER lower 123", "", "_'_ z"]
    with tempfile.TemporaryDirectory() as td:
        man = os.path.join(td, "m.csv")
        rows = []
        for i, s in enumerate(texts):
            tp = os.path.join(td, "t%d.txt" % i)
            open(tp, "w", encoding="utf8").write(s)
            rows.append("x.pt7,%s" % tp)
        open(man, "w").write("\n".join(rows) + "\n")
        ds = LF.FeatDataset(manifest=man, labels=LABELS)
        for i in range(len(texts)):
            out["transcript.text%d" % i] = np.frombuffer(texts[i].encode("utf8"), np.uint8).copy()
            out["transcript.ids%d" % i] = np.asarray(ds.parse_transcript(rows[i].split(",")[1]), np.int32)
    out["transcript.n"] = len(texts)
    # ---- greedy decoding: argmax paths -> strings (collapse repeats, drop blanks), WER / CER edit distances
    dec = RDEC.GreedyDecoder(LABELS)
    Tn, Nn = 40, 6
    paths = prng.randint(1600, (Nn, Tn), 0, 28).astype(np.int64)
    paths[:, ::3] = 0                      # blanks
    paths[1, 5:12] = 9                     # a long repeat
    paths[2, :] = 0                        # all blank -> empty string
    paths[3, 10:20] = 28                   # repeated spaces
    sizes = np.asarray([40, 33, 40, 25, 1, 0], np.int32)
    strings = dec.convert_to_strings([p.tolist() for p in paths], sizes.tolist(), remove_repetitions=True)
    out["decode.paths"], out["decode.sizes"] = paths, sizes
    for i, s in enumerate(strings):
        out["decode.str%d" % i] = np.frombuffer(s[0].encode("utf8"), np.uint8).copy()
    tgt = [prng.randint(1700 + i, (k,), 1, 28).tolist() for i, k in enumerate([9, 7, 0, 12, 3, 5])]
    tstr = dec.convert_to_strings(tgt)
    pairs = [(strings[i][0], tstr[i][0]) for i in range(Nn)] + [("the cat sat", "the cat sat on the mat"), ("a b c", "c b a"), ("", "x y")]
    out["decode.npairs"] = len(pairs)
    for i, (a, b) in enumerate(pairs):
        out["decode.pair%d.a" % i] = np.frombuffer(a.encode("utf8"), np.uint8).copy()
        out["decode.pair%d.b" % i] = np.frombuffer(b.encode("utf8"), np.uint8).copy()
        out["decode.pair%d.wer" % i] = dec.wer(a, b)
        out["decode.pair%d.cer" % i] = dec.cer(a, b)
    for i, g in enumerate(tgt):
        out["decode.target%d" % i] = np.asarray(g, np.int32)
        out["decode.target_str%d" % i] = np.frombuffer(tstr[i][0].encode("utf8"), np.uint8).copy()
    np.savez_compressed(os.path.join(OUT, "f8_host_side.npz"), **out)
    print("F8 keys", len(out), [s[0] for s in strings])


def f9_rnn_kind():
    """F9: the vanilla `rnn` kind of supported_rnns (model.py:12-17): the reference's BRNN with nn.RNN (tanh, bias-free,
    directions summed; model.py:88-105) at three sizes - the last at the enhancer's layer shape - and a whole
    stackedBRNN(rnn_type=nn.RNN) forward / backward (model.py:203-231)."""
    out = {}
    torch.manual_seed(0)
    for tag, (T, N, H) in dict(s=(7, 2, 5), m=(23, 5, 24), l=(60, 30, 500)).items():
        m = REF.BRNN(H, H, rnn_type=nn.RNN, bidirectional=True)
        load_weights(m, 700 + T)
        x = t(prng.normal(711 + T, (T, N, H), 0.0, 0.5)).requires_grad_(True)
        y = m(x)
        gy = t(prng.normal(712 + T, (T, N, H)))
        y.backward(gy)
        p = "brnn_rnn_%s." % tag
        if tag == "l":     # at size: samples and norms only
            idx = prng.randint(713, (256,), 0, T * N * H - 1).astype(np.int64)
            out[p + "dims"] = np.asarray([T, N, H])
            out[p + "seeds"] = np.asarray([700 + T, 711 + T, 712 + T])
            out[p + "idx"] = idx
            out[p + "y_samples"], out[p + "gx_samples"] = y.detach().reshape(-1)[idx].numpy(), x.grad.reshape(-1)[idx].numpy()
            out[p + "y_norm"], out[p + "gx_norm"] = float(y.detach().double().norm()), float(x.grad.double().norm())
            for k, v in m.named_parameters():
                out[p + "gw_norm." + k] = float(v.grad.double().norm())
                out[p + "gw_samples." + k] = v.grad.reshape(-1)[idx % v.numel()].numpy()
        else:
            out[p + "x"], out[p + "y"], out[p + "gy"], out[p + "gx"] = x.detach().numpy(), y.detach().numpy(), gy.numpy(), x.grad.numpy()
            for k, v in m.named_parameters():
                out[p + "w." + k] = v.detach().numpy()
                out[p + "gw." + k] = v.grad.numpy()
    G = REF.stackedBRNN(I=6, O=6, H=10, L=4, rnn_type=nn.RNN)
    load_weights(G, 750)
    x = t(prng.uniform(751, (3, 6, 17), 0.0, 6.0)).requires_grad_(True)
    y = G(x)
    gy = t(prng.normal(752, tuple(y.shape)))
    y.backward(gy)
    p = "stacked_rnn."
    out[p + "x"], out[p + "y"], out[p + "gy"], out[p + "gx"] = x.detach().numpy(), y.detach().numpy(), gy.numpy(), x.grad.numpy()
    for k, v in G.state_dict().items():
        out[p + "sd." + k] = v.numpy().copy()
    for k, v in G.named_parameters():
        out[p + "gw." + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "f9_rnn_kind.npz"), **out)
    print("F9 keys", len(out))


def acoustic_iteration(G, A, og, oa, ny, it, allow):
    """trainer_acoustic.py:120-142, line by line, on reference modules (loss = CTC / N: no w_acoustic, no discriminator)."""
    inputs, targets, pct, target_sizes = t(ny["inputs"]), t(ny["targets"]), t(ny["pct"]).clone(), t(ny["target_sizes"])
    N = inputs.size(0)
    enhanced = G(inputs)
    prob = A(enhanced)
    prob = prob.transpose(0, 1)
    T = prob.size(0)
    sizes = pct.mul_(int(T)).int()
    loss = ctc_sum(prob, targets, sizes, target_sizes)
    loss = loss / N
    G.zero_grad(); A.zero_grad()
    loss.backward()
    grads = {}
    for nm, m in (("G", G), ("A", A)):
        for k, p in m.named_parameters():
            grads[nm + "." + k] = p.grad.detach().clone().numpy()
    og.step()
    if it > allow:
        oa.step()
    return loss.item(), grads, enhanced.detach().numpy(), prob.detach().numpy()


def f10_acoustic():
    """F10: the acoustic_supervision trainer (trainer_acoustic.py:120-142): a tiny case with every tensor (3 iterations, ragged
    lengths, A trainable from iteration 1) and config-2 size (N=30, T=200, E 4x500, A 2conv+5x1000 GRU), 2 iterations, sampled."""
    out = {}
    # ---- tiny
    Fdim, H, HA, M, N, T, L = 8, 16, 12, 8, 4, 60, 3
    G, _, A = build_aas(Fdim, H, HA, M, 3, seed=7100)
    og, oa = adam(G, 1e-3), adam(A, 1e-3)
    for k, v in G.state_dict().items():
        out["tiny.G0." + k] = v.numpy().copy()
    for k, v in A.state_dict().items():
        out["tiny.A0." + k] = v.numpy().copy()
    for it in range(3):
        ny = make_batch(N, Fdim, [60, 52, 47, 33], seed=7200 + it, label_lens=[3, 2, 3, 1], lab_seed=7300 + it)
        loss, grads, enh, prob = acoustic_iteration(G, A, og, oa, ny, it, allow=0)
        p = "tiny.it%d." % it
        for k, v in ny.items():
            out[p + "ny." + k] = v
        out[p + "loss"], out[p + "enhanced"], out[p + "logits"] = np.float64(loss), enh, prob
        for k in ("G.rnn1.rnn.weight_hh_l0", "G.final_linear.weight", "A.conv.0.weight", "A.rnns.1.rnn.weight_ih_l0", "A.fc.0.module.1.weight"):
            out[p + "grad." + k] = grads[k]
    for k, v in G.state_dict().items():
        out["tiny.G3." + k] = v.numpy().copy()
    for k, v in A.state_dict().items():
        out["tiny.A3." + k] = v.numpy().copy()
    # ---- config-2 size (weights = F3's: seed 9000)
    Fdim, H, HA, M, N, T, L = 80, 500, 1000, 128, 30, 200, 20
    G, _, A = build_aas(Fdim, H, HA, M, 5, seed=9000)
    og, oa = adam(G, 1e-5), adam(A, 1e-5)
    out.update({"big.weight_seed": 9000, "big.lr": 1e-5, "big.N": N, "big.F": Fdim, "big.T": T, "big.L": L})
    for it in range(2):
        ny = dict(inputs=prng.uniform(123 + 1000 * it, (N, Fdim, T), 0.0, 6.0), mask=np.zeros((N, 1, T), np.uint8), pct=np.ones(N, np.float32),
                  targets=prng.randint(125 + 1000 * it, (N * L,), 1, 28).astype(np.int32), target_sizes=np.full(N, L, np.int32))
        loss, grads, enh, prob = acoustic_iteration(G, A, og, oa, ny, it, allow=0)
        p = "big.it%d." % it
        out[p + "loss"] = np.float64(loss)
        ie, il = sample_idx(61 + it, enh.shape, 256), sample_idx(71 + it, prob.shape, 256)
        out[p + "enh_idx"], out[p + "enh_samples"] = ie, enh.reshape(-1)[ie]
        out[p + "logit_idx"], out[p + "logit_samples"] = il, prob.reshape(-1)[il]
        for k in ("G.rnn1.rnn.weight_hh_l0", "G.rnn4.rnn.weight_ih_l0_reverse", "A.rnns.2.rnn.weight_hh_l0"):
            g = grads[k]
            ig = sample_idx(81, g.shape, 64)
            out[p + "gradsample_idx." + k], out[p + "gradsample." + k] = ig, g.reshape(-1)[ig]
            out[p + "gradnorm." + k] = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        print("F10 big it", it, loss, flush=True)
    np.savez_compressed(os.path.join(OUT, "f10_acoustic.npz"), **out)
    print("F10 keys", len(out))


def f11_am_model_ken():
    """F11: AM_training/model.py's own DeepSpeech_ken (:337-470), the class AM_training/train.py builds: include_first_BN
    True / False, nDownsample 1 / 2 - forward logits, the gradient of a fixed cotangent wrt input and every parameter, and the
    BatchNorm running statistics after the pass (tiny sizes, every tensor)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("am_ref_model", "/root/reference/AM_training/model.py")
    AMR = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(AMR)
    out = {}
    for tag, kw in dict(bn=dict(include_first_BN=True, nDownsample=1), nobn=dict(include_first_BN=False, nDownsample=1),
                        nobn_ds2=dict(include_first_BN=False, nDownsample=2), lstm_nobn=dict(include_first_BN=False, nDownsample=1)).items():
        rt = nn.LSTM if tag.startswith("lstm") else nn.GRU
        A = AMR.DeepSpeech_ken(rnn_type=rt, labels=LABELS, rnn_hidden_size=12, rnn_layers=3, kernel_sz=11, stride=2, map=8, cnn_layers=2,
                               nFreq=10, **kw)
        load_weights(A, 8300 + len(tag), conv_std=0.1)
        p = "ken_%s." % tag
        for k, v in A.state_dict().items():
            out[p + "sd0." + k] = v.numpy().copy()
        x = t(prng.uniform(8310, (3, 10, 90), 0.0, 6.0)).requires_grad_(True)
        y = A(x)
        gy = t(prng.normal(8311, tuple(y.shape)))
        y.backward(gy)
        out[p + "x"], out[p + "y"], out[p + "gy"], out[p + "gx"] = x.detach().numpy(), y.detach().numpy(), gy.numpy(), x.grad.numpy()
        for k, v in A.named_parameters():
            out[p + "gw." + k] = v.grad.numpy()
        for k, v in A.state_dict().items():
            if "running" in k:
                out[p + "sd1." + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "f11_am_model_ken.npz"), **out)
    print("F11 keys", len(out))


def f12_cli_defaults():
    """F12: the command-line contracts as data - every flag and default of AM_training/train.py's parser (:24-110) and of
    Speech_enhancement_by_AAS/config.py's.  train.py cannot be imported here (warpctc_pytorch, data.data_loader are absent), so
    its `parser.add_argument(...)` calls are read from its syntax tree and replayed on a fresh argparse parser; config.py imports."""
    import ast
    import json
    src = open("/root/reference/AM_training/train.py").read()
    tree = ast.parse(src)
    ns = {"str2bool": (lambda v: v.lower() in ("true", "1")), "int": int, "float": float, "str": str}
    par = argparse.ArgumentParser()
    for node in ast.walk(tree):
        if (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument"
                and isinstance(node.func.value, ast.Name) and node.func.value.id == "parser"):
            args = [ast.literal_eval(a_) for a_ in node.args]
            kw = {}
            for k in node.keywords:
                kw[k.arg] = ns[k.value.id] if (isinstance(k.value, ast.Name) and k.value.id in ns) else ast.literal_eval(k.value)
            par.add_argument(*args, **kw)
    am = vars(par.parse_args([]))
    import config as REFCFG     # the reference's Speech_enhancement_by_AAS/config.py
    argv, sys.argv = sys.argv, ["main.py"]
    try:
        c, _ = REFCFG.get_config()
    finally:
        sys.argv = argv
    aas = {k: v for k, v in vars(c).items()}
    with open(os.path.join(OUT, "f12_cli_defaults.json"), "w") as f:
        json.dump({"AM_training/train.py": am, "Speech_enhancement_by_AAS/config.py": aas}, f, indent=1, sort_keys=True)
    print("F12", len(am), "AM flags,", len(aas), "AAS flags")


def _ref_decoder():
    """The reference's AM_training/decoder.py GreedyDecoder (its `Levenshtein` import satisfied by a textbook edit distance, as in
    F8).  `decode()` itself indexes a dict with tensor elements and fails under torch 2.x, so the callers below pass the
    argmax path to `convert_to_strings(..., remove_repetitions=True)` as lists - the call decode() makes (:186-201)."""
    import types as _types
    if "Levenshtein" not in sys.modules:
        lev = _types.ModuleType("Levenshtein")

        def distance(a, b):
            prev = list(range(len(b) + 1))
            for i, ca in enumerate(a, 1):
                cur = [i]
                for j, cb in enumerate(b, 1):
                    cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
                prev = cur
            return prev[-1]
        lev.distance = distance
        sys.modules["Levenshtein"] = lev
    if "/root/reference/AM_training" not in sys.path:
        sys.path.insert(0, "/root/reference/AM_training")
    import decoder as RDEC
    return RDEC.GreedyDecoder(LABELS)


def _ref_greedy(dec, G, A, inputs, targets, pct, target_sizes):
    """Step 1 of the three validation functions (trainer_DCE.py:209-250, trainer_FSEGAN.py:277-306, trainer_AAS.py:301-340), line
    by line: unflatten targets, enhanced = G(inputs), prob = ASR(enhanced) time-major, sizes = pct.mul_(T').int(), argmax decode,
    per-utterance wer / cer sums; wer = we / total_word, cer = ce / total_word (sic)."""
    split_targets, offset = [], 0
    for size in target_sizes.tolist():
        split_targets.append(targets[offset:offset + size].tolist())
        offset += size
    enhanced = G(inputs)
    prob = A(enhanced)
    prob = prob.transpose(0, 1)
    Tn = prob.size(0)
    sizes = pct.clone().mul_(int(Tn)).int()
    _, max_probs = torch.max(prob.detach().transpose(0, 1), 2)
    decoded = dec.convert_to_strings(max_probs.tolist(), sizes.tolist(), remove_repetitions=True)
    target_strings = dec.convert_to_strings(split_targets)
    we = ce = total_word = total_char = 0
    for x in range(len(target_strings)):
        decoding, reference = decoded[x][0], target_strings[x][0]
        we += dec.wer(decoding, reference)
        ce += dec.cer(decoding, reference)
        total_word += len(reference.split())
        total_char += len(reference)
    return enhanced, prob, sizes, we / total_word, ce / total_word, total_word, total_char, [d[0] for d in decoded], [t_[0] for t_ in target_strings]


class _Meter(object):       # utils.py:35-51 AverageMeter
    def __init__(self):
        self.sum = self.count = self.avg = 0

    def update(self, val, n=1):
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def f13_validation():
    """F13: the validation passes of the three trainers on the reference's own modules - `greedy_decoding` + DCE
    (trainer_DCE.py:130-190,209-250), `greedy_decoding_and_FSEGAN` (trainer_FSEGAN.py:199-243,277-317, its
    `assert(nElement == nElement_)` included; D through forward_paired) and `greedy_decoding_and_AAS` (trainer_AAS.py:215-263,
    301-351) - over a two-batch validation set in the paired collate layout (loader_functions.py:76-105), with the AverageMeter
    weighting of the loops.  G in eval mode, A left in train mode (the reference never calls ASR.eval()).  Tiny models, every
    tensor; labels are word-like (spaces) so that WER is not degenerate."""
    dec = _ref_decoder()
    Fdim, H, HA, M = 8, 16, 12, 8
    G = REF.stackedBRNN(I=Fdim, O=Fdim, H=H, L=4)
    Dp = REF.stackedBRNN(I=2 * Fdim, O=Fdim, H=H, L=4)      # FSEGAN's discriminator (forward_paired)
    Da = REF.stackedBRNN(I=Fdim, O=Fdim, H=H, L=4)          # AAS's
    A = REF.DeepSpeech(rnn_type=nn.GRU, labels=LABELS, rnn_hidden_size=HA, rnn_layers=3, kernel_sz=11, stride=2, map=M, cnn_layers=2, nFreq=Fdim)
    load_weights(G, 8401); load_weights(Dp, 8402); load_weights(Da, 8403); load_weights(A, 8404, conv_std=0.1)
    with torch.no_grad():       # a wider output layer, so that the argmax path is not one symbol throughout
        A.fc[0].module[1].weight.mul_(6.0)
    out = {}
    for nm, m in (("G", G), ("Dp", Dp), ("Da", Da), ("A", A)):
        for k, v in m.state_dict().items():
            out["init.%s.%s" % (nm, k)] = v.clone().numpy()
    G.eval()
    w_adv, w_ac = 0.01, 1.0
    out["w_adversarial"], out["w_acoustic"] = w_adv, w_ac
    batches = []
    for b, (lens, lab_lens) in enumerate((([70, 64, 51], [7, 5, 6]), ([66, 40], [4, 8]))):
        bt = make_batch(len(lens), Fdim, lens, 8500 + 100 * b, lab_lens, 8600 + 10 * b)
        cl = make_batch(len(lens), Fdim, lens, 8700 + 100 * b)["inputs"]
        # transcripts derived from what the models decode (the decode does not depend on them), perturbed: utterance 0 exact,
        # utterance 1 with one character replaced, the others with their first word dropped - so WER / CER are neither 0 nor 1
        with torch.no_grad():
            _, _, _, _, _, _, _, dstr, _ = _ref_greedy(dec, G, A, t(bt["inputs"]), t(bt["targets"]), t(bt["pct"]), t(bt["target_sizes"]))
        c2i = {c: i for i, c in enumerate(LABELS)}
        tg, tl = [], []
        for i, s_ in enumerate(dstr):
            s_ = s_.strip() or "a"
            if i == 1:
                k_ = len(s_) // 2
                s_ = s_[:k_] + ("q" if s_[k_] != "q" else "r") + s_[k_ + 1:]
            elif i >= 2 and " " in s_:
                s_ = s_.split(" ", 1)[1]
            ids = [c2i[c] for c in s_]
            tg.extend(ids); tl.append(len(ids))
        bt["targets"], bt["target_sizes"] = np.asarray(tg, np.int32), np.asarray(tl, np.int32)
        batches.append((bt, cl))
        for k, v in bt.items():
            out["b%d.%s" % (b, k)] = v
        out["b%d.cleans" % b] = cl
    diff = REF.L1Loss_mask()
    m_dce, m_wer, m_cer = _Meter(), _Meter(), _Meter()
    f_dce, f_adv, f_wer, f_cer = _Meter(), _Meter(), _Meter(), _Meter()
    a_ctc, a_adv, a_wer, a_cer = _Meter(), _Meter(), _Meter(), _Meter()
    with torch.no_grad():
        for b, (bt, cl) in enumerate(batches):
            inputs, cleans, mask = t(bt["inputs"]), t(cl), t(bt["mask"]).bool()
            targets, pct, tsz = t(bt["targets"]), t(bt["pct"]), t(bt["target_sizes"])
            p = "b%d." % b
            # ---- minimize_DCE (trainer_DCE.py:137-153): DCE on G(inputs), then greedy_decoding
            outputs = G(inputs)
            dce, nEl = diff(outputs, cleans, mask)
            m_dce.update(dce.item(), int(nEl))
            enh, prob, sizes, wer, cer, nW, nC, dstr, tstr = _ref_greedy(dec, G, A, inputs, targets, pct, tsz)
            m_wer.update(wer, nW); m_cer.update(cer, nC)
            out[p + "dce.dce"], out[p + "dce.nElement"] = dce.item(), int(nEl)
            out[p + "dce.wer"], out[p + "dce.cer"], out[p + "dce.nWord"], out[p + "dce.nChar"] = wer, cer, nW, nC
            out[p + "enhanced"], out[p + "logits_tnc"], out[p + "sizes"] = enh.numpy(), prob.numpy(), sizes.numpy()
            for i, (a_, b_) in enumerate(zip(dstr, tstr)):
                out[p + "decoded%d" % i] = np.frombuffer(a_.encode("utf8"), np.uint8).copy()
                out[p + "reference%d" % i] = np.frombuffer(b_.encode("utf8"), np.uint8).copy()
            # ---- FSEGAN (trainer_FSEGAN.py:277-317)
            enh, prob, sizes, wer, cer, nW, nC, _, _ = _ref_greedy(dec, G, A, inputs, targets, pct, tsz)
            ae_ny = Dp.forward_paired(enh, inputs)
            l_adv_ny, nElement = diff(ae_ny, enh, mask)
            l_adv_ny = l_adv_ny * w_adv
            dce2, nElement_ = diff(enh, cleans, mask)
            assert (nElement == nElement_)
            f_dce.update(dce2.item(), int(nElement)); f_adv.update(l_adv_ny.item(), int(nElement)); f_wer.update(wer, nW); f_cer.update(cer, nC)
            for k, v in zip(("dce", "l_adv_ny", "nElement", "wer", "cer", "total_word", "total_char"),
                            (dce2.item(), l_adv_ny.item(), int(nElement), wer, cer, nW, nC)):
                out[p + "fsegan." + k] = v
            # ---- AAS (trainer_AAS.py:301-351)
            enh, prob, sizes, wer, cer, nW, nC, _, _ = _ref_greedy(dec, G, A, inputs, targets, pct, tsz)
            ae_ny = Da(enh)
            l_adv, nElement = diff(ae_ny, enh, mask)
            l_adv = l_adv * w_adv
            N = inputs.size(0)
            l_ctc = w_ac * ctc_sum(prob, targets, sizes, tsz) / N
            a_ctc.update(l_ctc.item(), N); a_adv.update(l_adv.item(), int(nElement)); a_wer.update(wer, nW); a_cer.update(cer, nC)
            for k, v in zip(("l_CTC", "l_adv_ny", "nElement", "wer", "cer", "total_word", "total_char"),
                            (l_ctc.item(), l_adv.item(), int(nElement), wer, cer, nW, nC)):
                out[p + "aas." + k] = v
    out.update({"avg.dce.dce": m_dce.avg, "avg.dce.wer": m_wer.avg, "avg.dce.cer": m_cer.avg,
                "avg.fsegan.dce": f_dce.avg, "avg.fsegan.adv_ny": f_adv.avg, "avg.fsegan.wer": f_wer.avg, "avg.fsegan.cer": f_cer.avg,
                "avg.aas.ctc": a_ctc.avg, "avg.aas.adv_ny": a_adv.avg, "avg.aas.wer": a_wer.avg, "avg.aas.cer": a_cer.avg})
    np.savez_compressed(os.path.join(OUT, "f13_validation.npz"), **out)
    print("F13", {k: v for k, v in out.items() if k.startswith("avg.")},
          [bytes(out["b0.decoded%d" % i]).decode() for i in range(3)], [bytes(out["b0.reference%d" % i]).decode() for i in range(3)])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-big", action="store_true", help="skip F3 (config 2, minutes of CPU)")
    ap.add_argument("--only", default="", help="comma list of fixtures to (re)generate: f1,...,f10")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    if a.only:
        table = dict(f1=f1_tiny, f2=f2_dce, f3=f3_config2, f4=f4_ops, f5=f5_fsegan_am, f6=f6_fsegan_config4,
                     f7=f7_am_config5, f8=f8_host_side, f9=f9_rnn_kind, f10=f10_acoustic, f3b=f3b_config2_kt, f11=f11_am_model_ken, f12=f12_cli_defaults, f3c=f3c_thread_spread,
                     f1r=f1r_tiny_ragged_pair, f3r=f3r_config2_ragged_pair, f13=f13_validation)
        for k in a.only.split(","):
            table[k]()
        sys.exit(0)
    f1_tiny()
    f1r_tiny_ragged_pair()
    f4_ops()
    f5_fsegan_am()
    f2_dce()
    f8_host_side()
    f9_rnn_kind()
    f11_am_model_ken()
    f12_cli_defaults()
    f13_validation()
    if not a.skip_big:
        f3_config2()
        f3b_config2_kt()
        f3r_config2_ragged_pair()
        f3c_thread_spread()
        f6_fsegan_config4()
        f7_am_config5()
        f10_acoustic()
