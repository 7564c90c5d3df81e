"""GPU parity tests (-m gpu) added in round 4: the acoustic_supervision trainer (no discriminator) against F10 (the reference's
stackedBRNN / DeepSpeech run through the restated trainer_acoustic.py:120-142 loop by tools/make_goldens.py), the LDS-DMA fp32 GEMM
and its multi-problem weight-gradient launch against fp64, the hybrid forward schedule against the batched one.
north_star tolerances: logits / enhanced within 1e-3 relative, losses within 1e-2 relative."""
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from tests.helpers import LABELS, batch_from, grad_close, load, load_sd, rel_err, sub

pytestmark = pytest.mark.gpu

REL_OUT, REL_LOSS = 1e-3, 1e-2


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def cfg(**kw):
    c = types.SimpleNamespace(lr=1e-3, beta1=0.5, beta2=0.999, optimizer="adam", batch_size=4, expnum=0, lambda_k=0.001, gamma=0.5, gpu=0,
                              load_path="", mode="train", write_log=False, w_adversarial=1.0, w_acoustic=1.0, allow_ASR_update_iter=0,
                              schedule="fused", nFeat=8, rnn_size=16, rnn_layers=4, rnn_type="lstm")
    c.__dict__.update(kw)
    return c


def _tiny_models(z):
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    G = stackedBRNN(I=8, H=16, L=4)
    A = DeepSpeech(nn.GRU, LABELS, 12, 3, True, 11, 2, 8, 2, nFreq=8)
    load_sd(G, sub(z, "tiny.G0."))
    load_sd(A, sub(z, "tiny.A0."))
    return G, A


@pytest.mark.parametrize("form", ["sync", "async"])
def test_acoustic_supervision_tiny_golden(gpu, precision, form):
    """trainer_acoustic.Trainer (E + A, loss = CTC / N, no discriminator anywhere) on F10's tiny ragged batches: iteration 0 at
    north_star's tolerances incl. sampled gradients; the iterations behind Adam steps at the level its sign flips of
    noise-level gradient elements allow."""
    from aas_enhancement_amd.trainer_acoustic import Trainer
    z = load("f10_acoustic.npz")
    tr = Trainer(cfg(), None, models=_tiny_models(z))
    assert not any(True for _ in tr.D.parameters())
    for it in range(3):
        ny = batch_from(z, "tiny.it%d.ny." % it)
        if form == "sync":
            r = tr.train_step(ny, it)
            loss = r["l_ctc"]
        else:
            r = tr.train_step_async(ny, it)
            loss = tr.read_scalars()["l_ctc"]
        p = "tiny.it%d." % it
        assert loss == pytest.approx(float(z[p + "loss"]), rel=REL_LOSS), it
        tol = REL_OUT if it == 0 else 1e-2
        assert rel_err(r["enhanced"], z[p + "enhanced"]) < tol and rel_err(r["prob"], z[p + "logits"]) < tol, it
        if it == 0:
            for nm, m in (("G", tr.G), ("A", tr.ASR)):
                for k, v in m.named_parameters():
                    if p + "grad.%s.%s" % (nm, k) in z.files:
                        assert grad_close(v.grad, z[p + "grad.%s.%s" % (nm, k)], rtol=2e-3), (nm, k)
    from aas_enhancement_amd import ops
    assert not ops.rnn_timeout_flag()


def test_acoustic_supervision_config2_golden(gpu, precision2):
    """F10 at config-2 size (N=30, T=200, E 4x500 BiLSTM, A 2xconv + 5x1000 BiGRU, weights = F3's) through the device-resident
    step: losses, sampled enhanced frames / logits, sampled weight gradients and their norms."""
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    from aas_enhancement_amd.trainer_acoustic import Trainer
    z = load("f10_acoustic.npz")
    N, F, T, L, seed = int(z["big.N"]), int(z["big.F"]), int(z["big.T"]), int(z["big.L"]), int(z["big.weight_seed"])
    G = stackedBRNN(I=F, H=500, L=4)
    A = DeepSpeech(nn.GRU, LABELS, 1000, 5, True, 11, 2, 128, 2, nFreq=F)
    for m, s, cs in ((G, seed + 1, None), (A, seed + 3, 0.1)):
        load_sd(m, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(m.state_dict(), s, conv_std=cs).items()}, strict=False)
    tr = Trainer(cfg(lr=float(z["big.lr"]), nFeat=F, rnn_size=500, batch_size=N), None, models=(G, A))
    for it in range(2):
        ny = (torch.from_numpy(prng.uniform(123 + 1000 * it, (N, F, T), 0.0, 6.0)),
              torch.from_numpy(prng.randint(125 + 1000 * it, (N * L,), 1, 28).astype(np.int32)),
              torch.ones(N), torch.full((N,), L, dtype=torch.int32), torch.zeros(N, 1, T, dtype=torch.uint8))
        r = tr.train_step_async(ny, it)
        loss = tr.read_scalars()["l_ctc"]
        p = "big.it%d." % it
        assert loss == pytest.approx(float(z[p + "loss"]), rel=REL_LOSS), it
        enh, prob = r["enhanced"].detach().reshape(-1), r["prob"].detach().reshape(-1)
        e_ref, p_ref = z[p + "enh_samples"], z[p + "logit_samples"]
        e_got = enh[torch.from_numpy(z[p + "enh_idx"]).cuda()].cpu().numpy()
        p_got = prob[torch.from_numpy(z[p + "logit_idx"]).cuda()].cpu().numpy()
        assert np.abs(e_got - e_ref).max() < REL_OUT * np.abs(e_ref).max(), it
        assert np.abs(p_got - p_ref).max() < REL_OUT * np.abs(p_ref).max(), it
        if it == 0:    # (A starts stepping at iteration 1: its gradients are formed from iteration 0 on, as in the reference)
            for key in [k[len(p + "gradnorm."):] for k in z.files if k.startswith(p + "gradnorm.")]:
                nm, k = key.split(".", 1)
                g = dict((tr.G if nm == "G" else tr.ASR).named_parameters())[k].grad
                got = g.reshape(-1)[torch.from_numpy(z[p + "gradsample_idx." + key]).cuda()].cpu().numpy()
                want = z[p + "gradsample." + key]
                gtol = 2e-3 if precision2 == 0 else 1.5e-2     # (the fast mode's 2^-17 products through nine recurrent layers)
                assert np.abs(got - want).max() < gtol * np.abs(want).max() + 1e-9, key
                assert float(g.double().pow(2).sum().sqrt()) == pytest.approx(float(z[p + "gradnorm." + key]), rel=gtol), key
    from aas_enhancement_amd import ops
    assert not ops.rnn_timeout_flag()


@pytest.mark.parametrize("M,N,K", [(6000, 2000, 500), (2000, 500, 6000), (700, 512, 260), (130, 68, 100)])
def test_lds_dma_gemm_against_register_staged_kernel_and_fp64(gpu, M, N, K):
    """aas_gemm_f32 in fp32 arithmetic: the LDS-DMA kernel (aas_set_gemm_variant(0), the default) and the register-staged one
    (variant 1) in all three modes with bias / addend / accumulate, both against fp64 at the fp32 chain's error level."""
    from aas_enhancement_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(7)
    a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    bias, add = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = a.double() @ b.double().t()
    scale = float((a.abs().double() @ b.abs().double().t()).max())
    A, B, Bt, At = a.cuda(), b.cuda(), b.t().contiguous().cuda(), a.t().contiguous().cuda()
    try:
        for variant in (0, 1):
            L.aas_set_gemm_variant(variant)
            c = torch.empty(M, N, device="cuda")
            ops.gemm(ops.NT, M, N, K, A, K, B, K, c, N, bias=bias.cuda(), addend=add.cuda(), ldd=N)
            assert float((c.double().cpu() - (ref + bias.double() + add.double())).abs().max()) < 4e-7 * scale, (variant, "nt")
            ops.gemm(ops.NN, M, N, K, A, K, Bt, N, c, N)
            assert float((c.double().cpu() - ref).abs().max()) < 4e-7 * scale, (variant, "nn")
            c.fill_(1.0)
            ops.gemm(ops.TN, M, N, K, At, M, Bt, N, c, N, accumulate=True)
            assert float((c.double().cpu() - (ref + 1.0)).abs().max()) < 4e-7 * scale, (variant, "tn")
    finally:
        L.aas_set_gemm_variant(0)


def test_multi_problem_weight_gradient_launch_equals_four_products(gpu):
    """aas_gemm_f32_multi: the four weight-gradient products of a bidirectional layer (per-problem reduction extents, shared
    d(gates)) in one launch == four aas_gemm_f32 launches, with and without the workgroup-lifetime cap (extra split-K slabs)."""
    from aas_enhancement_amd import _lib, ops
    L = _lib.lib()
    T, Nb, H = 50, 30, 500
    GH, R = 4 * H, T * Nb
    g = torch.Generator().manual_seed(3)
    dg = torch.randn(R, 2 * GH, generator=g).cuda()
    x, h = torch.randn(R, H, generator=g).cuda(), torch.randn(2 * R, H, generator=g).cuda()
    Rm = (T - 1) * Nb
    want = [dg[:, :GH].double().t() @ x.double(), dg[:, GH:].double().t() @ x.double(),
            dg[Nb:, :GH].double().t() @ h[:Rm].double(), dg[:Rm, GH:].double().t() @ h[R + Nb:].double()]
    a0, b0, bh = dg.data_ptr(), x.data_ptr(), h.data_ptr()
    As = [a0, a0 + 4 * GH, a0 + 4 * Nb * 2 * GH, a0 + 4 * GH]
    Bs = [b0, b0, bh, bh + 4 * (R + Nb) * H]
    try:
        for cap in (0, 16):
            L.aas_set_gemm_max_steps(cap)
            outs = [torch.ones(GH, H, device="cuda") for _ in range(4)]
            ops.gemm_multi(ops.TN, GH, H, [R, R, Rm, Rm], As, 2 * GH, Bs, H, [o.data_ptr() for o in outs], H, accumulate=True)
            for o, w in zip(outs, want):
                assert rel_err(o.double().cpu() - 1.0, w.cpu()) < 2e-5, cap
    finally:
        L.aas_set_gemm_max_steps(48)


@pytest.mark.parametrize("T,Nb,H", [(50, 60, 500), (7, 6, 16)])
def test_weight_gradients_per_utterance_class_with_alpha_equal_scaled_products(gpu, T, Nb, H):
    """The batched [enhanced; clean] discriminator pass: weight gradients as one multi-problem launch per utterance CLASS - two-level
    reduction rows (the class's rows of every time step), the class weight as a device-scalar alpha - against fp64 products of
    explicitly scaled operands; at the discriminator's size (LDS-DMA kernel, kdiv = 30 < one k-step) and at a tiny size (general
    kernel, alpha on its reduction rows)."""
    from aas_enhancement_amd import ops
    GH, R, ns = 4 * H, T * Nb, Nb // 2
    g = torch.Generator().manual_seed(11)
    dg = torch.randn(T, Nb, 2 * GH, generator=g).cuda()
    x, h = torch.randn(T, Nb, H, generator=g).cuda(), torch.randn(2, T, Nb, H, generator=g).cuda()
    kt = torch.tensor([-0.37], device="cuda")
    w = torch.ones(Nb, dtype=torch.float64)
    w[:ns] = -0.37
    dgd, xd, hd = dg.double().cpu(), x.double().cpu() * w[None, :, None], h.double().cpu() * w[None, None, :, None]
    want = [torch.einsum("tng,tni->gi", dgd[..., :GH], xd), torch.einsum("tng,tni->gi", dgd[..., GH:], xd),
            torch.einsum("tng,tni->gi", dgd[1:, :, :GH], hd[0, :-1]), torch.einsum("tng,tni->gi", dgd[:-1, :, GH:], hd[1, 1:])]
    outs = [torch.zeros(GH, H, device="cuda") for _ in range(4)]
    a0, b0, bh = dg.data_ptr(), x.data_ptr(), h.data_ptr()
    for n0, alpha in ((0, kt), (ns, None)):
        oa, ox = 4 * n0 * 2 * GH, 4 * n0 * H
        As = [a0 + oa, a0 + oa + 4 * GH, a0 + oa + 4 * Nb * 2 * GH, a0 + oa + 4 * GH]
        Bs = [b0 + ox, b0 + ox, bh + ox, bh + ox + 4 * (T * Nb * H + Nb * H)]
        ops.gemm_multi(ops.TN, GH, H, [T * ns, T * ns, (T - 1) * ns, (T - 1) * ns], As, 2 * GH, Bs, H, [o.data_ptr() for o in outs], H,
                       accumulate=True, kdiv=ns, kouterA=Nb * 2 * GH, kouterB=Nb * H, alpha=alpha)
    for o, wv in zip(outs, want):
        assert rel_err(o, wv) < 2e-5


# ------------------------------------------------------------------------------------------------ acoustic trainer, data parallel
def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _acoustic_models(seed=40):
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    G = stackedBRNN(I=8, H=16, L=2)
    A = DeepSpeech(nn.GRU, LABELS, 12, 2, True, 11, 2, 8, 2, nFreq=8)
    for m, s, cs in ((G, seed + 1, None), (A, seed + 3, 0.1)):
        load_sd(m, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(m.state_dict(), s, conv_std=cs).items()}, strict=False)
    return G, A


def _acoustic_batch(it):
    from tests.tools_shim import make_batch
    b = make_batch(4, 8, [60, 60, 60, 60], 510 + it, [3, 3, 2, 2], 520 + it)
    return (torch.from_numpy(b["inputs"]), torch.from_numpy(b["targets"]), torch.from_numpy(b["pct"]), torch.from_numpy(b["target_sizes"]),
            torch.from_numpy(b["mask"]))


def _acoustic_worker(rank, world, port, q, form):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=180))
    try:
        torch.cuda.set_device(0)
        from aas_enhancement_amd.trainer_acoustic import Trainer
        tr = Trainer(cfg(sync_bn=True), None, models=_acoustic_models())
        tr.make_optimizers()
        assert tr.dp.active
        out = []
        for it in range(3):
            b = tr.dp.shard_collated(_acoustic_batch(it))
            if form == "sync":
                out.append(tr.train_step(b, it)["l_ctc"])
            else:
                tr.train_step_async(b, it)
                out.append(tr.read_scalars()["l_ctc"])
        q.put((rank, np.asarray(out), tr._flat["G"].flat_p.detach().cpu().numpy(), tr._flat["A"].flat_p.detach().cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("form", ["sync", "async"])
def test_acoustic_trainer_dp_two_ranks_equal_single(gpu, form, precision2):
    """trainer_acoustic data parallel: 2 ranks x 2 utterances (global N as the loss normaliser, bucketed all-reduce of E's and A's flat
    gradient buffers, SyncBN in A) == 1 rank x 4 utterances: losses, and E's / A's parameters after three Adam steps."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_acoustic_worker, args=(r, 2, port, q, form)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from aas_enhancement_amd.trainer_acoustic import Trainer
    tr = Trainer(cfg(), None, models=_acoustic_models())
    ref = np.asarray([tr.train_step(_acoustic_batch(it), it)["l_ctc"] for it in range(3)])
    for rank, out, gp, ap in res:
        assert np.allclose(out, ref, rtol=3e-4), (rank, out, ref)
        for got, want in ((gp, tr._flat["G"].flat_p), (ap, tr._flat["A"].flat_p)):
            d_ = np.abs(got - want.detach().cpu().numpy())
            assert float((d_ > 2e-4).mean()) < 1e-2 and float(d_.max()) < 6.1e-3, rank      # (Adam sign flips of noise-level gradients)
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])


@pytest.mark.parametrize("R_,C,slope", [(2550, 1000, 1.0), (2850, 128, 128.0), (37, 8, 0.5), (5000, 260, 1.0)])
def test_batchnorm_partial_sum_path_vs_fp64_and_atomic_path(gpu, R_, C, slope):
    """Train-mode BatchNorm (model.py:72,82,290,298,316) on the 16-byte partial-sum kernels (default where C % 4 == 0) against torch
    in fp64, against the column-per-thread kernels with fp64 atomics (debug bit 32768), through the two-launch SyncBN entries, and
    bit-identical between two runs (no atomics in the statistics)."""
    import torch.nn.functional as F
    from aas_enhancement_amd import ops
    from aas_enhancement_amd._lib import lib
    g0 = torch.Generator().manual_seed(5)
    x = (torch.randn(R_, C, generator=g0) * 3 + 1.5).to(gpu)
    dy = torch.randn(R_, C, generator=g0).to(gpu)
    gamma = (torch.rand(C, generator=g0) + 0.5).to(gpu)
    beta = torch.randn(C, generator=g0).to(gpu)

    def run():
        rm, rv = torch.zeros(C, device=gpu), torch.ones(C, device=gpu)
        xx, gg, bb = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        y = ops.batchnorm_rows(xx, gg, bb, rm, rv, 1e-5, 0.1, slope)
        y.backward(dy)
        return [t.detach().clone() for t in (y, xx.grad, gg.grad, bb.grad, rm, rv)]

    new = run()
    again = run()
    for a, b in zip(new, again):
        assert torch.equal(a, b)
    lib().aas_set_debug_flags(32768)
    try:
        old = run()
    finally:
        lib().aas_set_debug_flags(0)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rmd, rvd = torch.zeros(C, device=gpu, dtype=torch.float64), torch.ones(C, device=gpu, dtype=torch.float64)
    yd = F.batch_norm(xd, rmd, rvd, gd, bd, True, 0.1, 1e-5)
    if slope != 1.0:
        yd = F.leaky_relu(yd, slope)
    yd.backward(dy.double())
    want = [yd, xd.grad, gd.grad, bd.grad, rmd, rvd]
    for name, got_new, got_old, w in zip(("y", "dx", "dgamma", "dbeta", "running_mean", "running_var"), new, old, want):
        e_new, e_old = rel_err(got_new, w), rel_err(got_old, w)
        assert e_new < 2e-5, (name, e_new)
        assert e_new < 2 * e_old + 1e-6, (name, e_new, e_old)
    # the SyncBN form: statistics, (all-reduce by the caller), apply - totals handed over as ONE partial row
    from aas_enhancement_amd.ops import check, ptr, stream
    red = torch.empty(2 * C, device=gpu, dtype=torch.float64)
    check(lib().aas_bn_stats(stream(), ptr(x), R_, C, ptr(red)), "aas_bn_stats")
    assert rel_err(red[:C], x.double().sum(0)) < 1e-6 and rel_err(red[C:], (x.double() ** 2).sum(0)) < 1e-6
    y2, st = torch.empty_like(x), torch.empty((4, C), device=gpu)
    rm, rv = torch.zeros(C, device=gpu), torch.ones(C, device=gpu)
    nbt = torch.full((1,), 41, device=gpu, dtype=torch.int64)     # nn.BatchNorm1d's num_batches_tracked: += 1 inside the apply launch
    check(lib().aas_bn_apply(stream(), ptr(x), ptr(y2), R_, C, ptr(gamma), ptr(beta), 1e-5, float(slope), ptr(st), ptr(rm), ptr(rv), 0.1,
                             ptr(red), None, ptr(nbt)), "aas_bn_apply")
    assert rel_err(y2, new[0]) < 1e-6 and rel_err(rm, new[4]) < 1e-6 and rel_err(rv, new[5]) < 1e-6
    assert int(nbt.item()) == 42
    for fl in (0, 32768):
        lib().aas_set_debug_flags(fl)
        try:
            ops.batchnorm_rows(x, gamma, beta, rm, rv, 1e-5, 0.1, slope, nbt)
        finally:
            lib().aas_set_debug_flags(0)
    assert int(nbt.item()) == 44


@pytest.mark.parametrize("T,N,H,cus", [(40, 30, 500, 128), (40, 60, 500, 128), (25, 26, 500, 128), (25, 60, 500, 0), (30, 13, 320, 64),
                                      (3, 30, 500, 128), (1, 30, 500, 128), (12, 70, 500, 128), (16, 30, 264, 128)])
def test_lstm_bptt_fp32_vs_fp64_recurrence_on_capped_grids(gpu, T, N, H, cus):
    """fp32 LSTM BPTT (nn.LSTM backward under model.py:94-95,101-105) against the recurrence in fp64 on the host, on the grids the
    training step uses (CU budgets of 128 / 64 and the whole chip): row groups of 4 / 8 rows on the 4 x 4 x 1 block kernels (tile
    reads running ahead of the MFMAs), 16 rows on the 16 x 16 x 4 tile kernels (debug bit 268435456 forces them everywhere), ragged
    last groups, row chunks over several launches (N = 70), T = 1 and T = 3; run-to-run identical; no timeout."""
    import os
    from aas_enhancement_amd import _lib, ops
    L = _lib.lib()
    dev = "cuda"
    g = torch.Generator().manual_seed(11)
    R = lambda *shape, scale=1.0: (torch.randn(*shape, generator=g) * scale).to(dev)
    ops.set_precision(0)
    L.aas_set_rnn_cu_limit(cus)
    try:
        w = [R(4 * H, H, scale=1.0 / H ** 0.5) for _ in range(2)]
        pre, dy = R(T, N, 2, 4 * H), R(T, N, H)
        sync, xc = ops._sync_buf(torch.device(dev, 0)), ops._xchg_buf(torch.device(dev, 0), T, N, H, 4)
        s, p = _lib.stream(), _lib.ptr
        hout, gact = torch.zeros(2, T, N, H, device=dev), torch.zeros(2, T, N, H, 4, device=dev)
        cst = torch.zeros(2, T, N, H, device=dev)
        ops.check(L.aas_lstm_fwd(s, T, N, H, p(pre), p(w[0]), p(w[1]), p(hout), p(gact), p(cst), p(sync), p(xc)), "fwd")
        res = {}
        for fl in (268435456, 0, 0):
            L.aas_set_debug_flags(fl)
            dgx = torch.zeros(T, N, 2, 4 * H, device=dev)
            ops.check(L.aas_lstm_bwd(s, T, N, H, p(dy), p(w[0]), p(w[1]), p(gact), p(cst), p(dgx), p(sync), p(xc)), "bwd")
            torch.cuda.synchronize()
            assert not ops.rnn_timeout_flag()
            res.setdefault(fl, []).append(dgx.clone())
        assert torch.equal(res[0][0], res[0][1])                       # run to run
        a, b = res[0][0], res[268435456][0]
        assert torch.isfinite(a).all() and a.abs().max() > 0
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7
        gd, cd = gact.double().cpu(), cst.double().cpu()
        want = torch.zeros(T, N, 2, 4 * H, dtype=torch.float64)
        for d_ in range(2):
            W = w[d_].double().cpu()
            dh_rec = torch.zeros(N, H, dtype=torch.float64)
            dc_next = torch.zeros(N, H, dtype=torch.float64)
            for t in (range(T - 1, -1, -1) if d_ == 0 else range(T)):
                tq = t - 1 if d_ == 0 else t + 1
                ig, fg, gg, og = gd[d_, t, :, :, 0], gd[d_, t, :, :, 1], gd[d_, t, :, :, 2], gd[d_, t, :, :, 3]
                c_t = cd[d_, t]
                cp = cd[d_, tq] if 0 <= tq < T else torch.zeros_like(c_t)
                dh = dy[t].double().cpu() + dh_rec
                tc = torch.tanh(c_t)
                dc = dh * og * (1 - tc * tc) + dc_next
                dc_next = dc * fg
                dg = torch.cat([dc * gg * ig * (1 - ig), dc * cp * fg * (1 - fg), dc * ig * (1 - gg * gg), dh * tc * og * (1 - og)], dim=1)
                want[t, :, d_] = dg
                dh_rec = dg @ W
        assert rel_err(a, want) < 2e-5 and rel_err(b, want) < 2e-5
    finally:
        L.aas_set_debug_flags(0)
        L.aas_set_rnn_cu_limit(0)
        ops.set_precision(int(os.environ.get("AAS_PRECISION", "0")))
