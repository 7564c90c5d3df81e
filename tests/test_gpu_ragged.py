"""Row classes of different sequence length inside one persistent recurrent launch (aas_set_rnn_row_classes) and the batched
discriminator pass over a ragged noisy / clean pair built on them (trainer_AAS._batched_D_core; the reference runs D twice,
trainer_AAS.py:94-107).  The layer-level tests compare with what the library itself computes with one launch per class; the
step-level tests at the end compare with the REFERENCE on pairs of different padded length (fixtures F1r / F3r)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _layer(kind, x, w, rs=None, residual=False):
    from aas_enhancement_amd import ops
    return ops.birnn_layer(x, w[0], w[1], w[2], w[3], kind=kind, residual=residual, rs=rs, lid=0)


@pytest.mark.parametrize("kind,H,I,Na,Nb,Ta,Tb", [
    ("lstm", 500, 500, 30, 30, 200, 184),     # D's layers at config 2: the 16-unit / 32-unit fp32 forward kernels
    ("lstm", 500, 500, 30, 30, 150, 200),     # the FIRST class shorter
    ("gru", 320, 96, 6, 10, 61, 40),          # GRU (z = 0, n = 1 encoding of the dead steps), unequal class sizes
    ("lstm", 48, 24, 3, 5, 37, 23),           # small hidden size: the split-kernel family
    ("gru", 40, 40, 4, 4, 19, 33),
])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_row_classes_match_one_launch_per_class(gpu, kind, H, I, Na, Nb, Ta, Tb, mode):
    from aas_enhancement_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1234 + H + Ta)
    G = {"lstm": 4, "gru": 3}[kind]
    flat = ((torch.rand(2 * G * H * (I + H), generator=g) - 0.5) * (2.0 / H ** 0.5)).to(dev)
    o, w = 0, []
    for shape in ((G * H, I), (G * H, H), (G * H, I), (G * H, H)):      # W_ih, W_hh, W_ih_rev, W_hh_rev in one flat buffer
        n = shape[0] * shape[1]
        w.append(flat[o:o + n].view(shape))
        o += n
    w = [w[0], w[1], w[2], w[3]]
    xa = torch.randn(Ta, Na, I, generator=g).to(dev)
    xb = torch.randn(Tb, Nb, I, generator=g).to(dev)
    ga = torch.randn(Ta, Na, H, generator=g).to(dev)
    gb = torch.randn(Tb, Nb, H, generator=g).to(dev)
    T, N = max(Ta, Tb), Na + Nb
    with ops.precision(mode):
        # one launch per class
        ref = {}
        ws = [t.detach().clone().requires_grad_(True) for t in w]
        xas, xbs = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True)
        ya = _layer(kind, xas, ws)
        yb = _layer(kind, xbs, ws)
        ((ya * ga).sum() + (yb * gb).sum()).backward()
        torch.cuda.synchronize()
        ref = dict(ya=ya.detach(), yb=yb.detach(), dxa=xas.grad, dxb=xbs.grad, dw=[t.grad for t in ws])
        # one launch, two row classes; junk beyond a class's last frame in x AND in dy
        ws2 = [t.detach().clone().requires_grad_(True) for t in w]
        x = torch.randn(T, N, I, generator=g).to(dev)
        x[:Ta, :Na] = xa
        x[:Tb, Na:] = xb
        x = x.requires_grad_(True)
        rs = ops.RowWeights(torch.ones(N, device=dev), row_len=(Na, Ta, Tb))
        y = _layer(kind, x, ws2, rs=rs)
        dy = torch.randn(T, N, H, generator=g).to(dev)
        dy[:Ta, :Na] = ga
        dy[:Tb, Na:] = gb
        y.backward(dy)
        torch.cuda.synchronize()
    tol = {0: 2e-5, 1: 2e-3, 2: 2e-5}[mode]

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))
    yd = y.detach()
    assert rel(yd[:Ta, :Na], ref["ya"]) < tol and rel(yd[:Tb, Na:], ref["yb"]) < tol
    assert float(yd[Ta:, :Na].abs().max()) == 0.0 if Ta < T else True
    assert float(yd[Tb:, Na:].abs().max()) == 0.0 if Tb < T else True
    assert rel(x.grad[:Ta, :Na], ref["dxa"]) < tol and rel(x.grad[:Tb, Na:], ref["dxb"]) < tol
    dead = x.grad[Ta:, :Na] if Ta < T else x.grad[Tb:, Na:]
    assert float(dead.abs().max()) == 0.0          # no gradient reaches the padding, whatever dy held there
    for a, b in zip([t.grad for t in ws2], ref["dw"]):
        assert rel(a, b) < tol


@pytest.mark.parametrize("kind,how", [("lstm", "chunks"), ("gru", "chunks"), ("lstm", "counter"), ("gru", "counter")])
def test_row_classes_on_row_chunked_launches_and_the_counter_kernels(gpu, kind, how):
    """The class boundary is a GLOBAL row index: a batch that a CU-capped launch processes in several row chunks (one kernel launch
    per chunk), and the round-1 counter-based kernels (debug bit 134217728) that remain the fallback family."""
    from aas_enhancement_amd import _lib, ops
    dev = torch.device("cuda:0")
    H, I, Na, Nb, Ta, Tb = 96, 32, 70, 90, 21, 30
    G = {"lstm": 4, "gru": 3}[kind]
    g = torch.Generator().manual_seed(99)
    w = [((torch.rand(s, generator=g) - 0.5) * 0.2).to(dev) for s in ((G * H, I), (G * H, H), (G * H, I), (G * H, H))]
    xa, xb = torch.randn(Ta, Na, I, generator=g).to(dev), torch.randn(Tb, Nb, I, generator=g).to(dev)
    T, N = max(Ta, Tb), Na + Nb
    x = torch.randn(T, N, I, generator=g).to(dev)
    x[:Ta, :Na], x[:Tb, Na:] = xa, xb
    dy = torch.randn(T, N, H, generator=g).to(dev)
    rs = ops.RowWeights(torch.ones(N, device=dev), row_len=(Na, Ta, Tb))
    lib = _lib.lib()
    try:
        if how == "chunks":
            ops.set_rnn_cu_limit(24)     # 6 unit slices x 2 directions x 2 row groups: the 160 rows take several launches
        else:
            lib.aas_set_debug_flags(134217728)
        outs = []
        for split in (True, False):
            ws = [t.detach().clone().requires_grad_(True) for t in w]
            if split:
                a_, b_ = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True)
                ya, yb = _layer(kind, a_, ws), _layer(kind, b_, ws)
                ((ya * dy[:Ta, :Na]).sum() + (yb * dy[:Tb, Na:]).sum()).backward()
                outs.append((ya.detach(), yb.detach(), a_.grad, b_.grad, [t.grad for t in ws]))
            else:
                x_ = x.clone().requires_grad_(True)
                y = _layer(kind, x_, ws, rs=rs)
                y.backward(dy)
                outs.append((y.detach()[:Ta, :Na], y.detach()[:Tb, Na:], x_.grad[:Ta, :Na], x_.grad[:Tb, Na:], [t.grad for t in ws]))
                assert float(y.detach()[Ta:, :Na].abs().max()) == 0.0 and float(x_.grad[Ta:, :Na].abs().max()) == 0.0
        torch.cuda.synchronize()
    finally:
        ops.set_rnn_cu_limit(0)
        lib.aas_set_debug_flags(0)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))
    for a, b in zip(outs[1][:4], outs[0][:4]):
        assert rel(a, b) < 2e-5
    for a, b in zip(outs[1][4], outs[0][4]):
        assert rel(a, b) < 2e-5


def test_row_classes_are_one_shot_and_validated(gpu):
    from aas_enhancement_amd import _lib, ops
    lib = _lib.lib()
    assert lib.aas_set_rnn_row_classes(-1, 4, 4) != 0
    assert lib.aas_set_rnn_row_classes(2, 0, 4) != 0
    dev = torch.device("cuda:0")
    H, I, T, N = 32, 16, 12, 4
    g = torch.Generator().manual_seed(7)
    w = [((torch.rand(s, generator=g) - 0.5) * 0.3).to(dev) for s in ((4 * H, I), (4 * H, H), (4 * H, I), (4 * H, H))]
    x = torch.randn(T, N, I, generator=g).to(dev)
    rs = ops.RowWeights(torch.ones(N, device=dev), row_len=(2, T, 5))
    y1 = _layer("lstm", x, w, rs=rs)
    y2 = _layer("lstm", x, w)              # the setting was consumed by the launch above
    y3 = _layer("lstm", x, w)
    torch.cuda.synchronize()
    assert float(y1[5:, 2:].abs().max()) == 0.0 and float(y2[5:, 2:].abs().max()) > 0.0
    assert torch.equal(y2, y3)
    assert torch.equal(y1[:, :2], y2[:, :2])
    rs.row_len = (2, T + 1, 5)        # longer than the launch
    with pytest.raises(RuntimeError, match="row classes"):
        _layer("lstm", x, w, rs=rs)
    rs.row_len = (2, 5, 6)            # neither class spans the launch
    with pytest.raises(RuntimeError, match="row classes"):
        _layer("lstm", x, w, rs=rs)
    y4 = _layer("lstm", x, w)              # a refused setting does not linger
    torch.cuda.synchronize()
    assert torch.equal(y2, y4)
    with pytest.raises(NotImplementedError):
        rs.row_len = (2, T, 5)
        ops.birnn_layer(x, w[0][:H], w[1][:H], w[2][:H], w[3][:H], kind="rnn", rs=rs)


def test_row_classes_random_shapes_and_degenerate_classes(gpu):
    """Seeded sweep over layer kinds, hidden sizes (all three forward kernel families), class sizes and lengths - including a class
    of one frame, an empty first class and an empty second class - against one launch per class, fp32."""
    import random
    from aas_enhancement_amd import ops
    dev = torch.device("cuda:0")
    rnd = random.Random(20261003)
    cases = [("lstm", 64, 3, 0, 9, 9), ("gru", 64, 0, 4, 7, 7), ("lstm", 500, 5, 7, 1, 12), ("gru", 256, 6, 2, 14, 1)]
    for _ in range(14):
        cases.append((rnd.choice(["lstm", "gru"]), rnd.choice([24, 40, 72, 128, 256, 320, 500]), rnd.randint(1, 24), rnd.randint(1, 24),
                      rnd.randint(1, 40), rnd.randint(1, 40)))
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))
    for ci, (kind, H, Na, Nb, Ta, Tb) in enumerate(cases):
        if Ta == Tb and Na and Nb:
            Tb += 1
        I = 32
        G = {"lstm": 4, "gru": 3}[kind]
        g = torch.Generator().manual_seed(500 + ci)
        w = [((torch.rand(s, generator=g) - 0.5) * (1.5 / H ** 0.5)).to(dev) for s in ((G * H, I), (G * H, H), (G * H, I), (G * H, H))]
        T, N = max(Ta if Na else 0, Tb if Nb else 0), Na + Nb
        x = torch.randn(T, N, I, generator=g).to(dev)
        dy = torch.randn(T, N, H, generator=g).to(dev)
        rs = ops.RowWeights(torch.ones(N, device=dev), row_len=(Na, Ta if Na else T, Tb if Nb else T))
        ws = [t.clone().requires_grad_(True) for t in w]
        xj = x.clone().requires_grad_(True)
        y = _layer(kind, xj, ws, rs=rs)
        y.backward(dy)
        ref_w = [torch.zeros_like(t) for t in w]
        for lo, n, Tc in ((0, Na, Ta), (Na, Nb, Tb)):
            if n == 0:
                continue
            ws2 = [t.clone().requires_grad_(True) for t in w]
            xs = x[:Tc, lo:lo + n].clone().requires_grad_(True)
            ys = _layer(kind, xs, ws2)
            ys.backward(dy[:Tc, lo:lo + n].contiguous())
            torch.cuda.synchronize()
            tag = (ci, kind, H, Na, Nb, Ta, Tb, lo)
            assert rel(y.detach()[:Tc, lo:lo + n], ys.detach()) < 2e-5, tag
            assert rel(xj.grad[:Tc, lo:lo + n], xs.grad) < 2e-5, tag
            if Tc < T:
                assert float(y.detach()[Tc:, lo:lo + n].abs().max()) == 0.0 and float(xj.grad[Tc:, lo:lo + n].abs().max()) == 0.0, tag
            for a, b in zip(ref_w, ws2):
                a += b.grad
        for a, b in zip(ws, ref_w):
            assert rel(a.grad, b) < 3e-5, (ci, kind, H, Na, Nb, Ta, Tb)
    assert not ops.rnn_timeout_flag()


# ---- the ragged-pair STEP against the reference itself (F1r / F3r: T_noisy != T_clean) -------------------------------------------
# Everything above compares the library with itself.  These compare with vectors generated from the reference's own modules through
# the restated loop of trainer_AAS.py:131-194 on a noisy / clean pair of DIFFERENT padded length (tools/make_goldens.py: f1r, f3r):
# the reference draws the two batches from two loaders (:136-138, :175-177), each padded to its own max T (loader_functions.py:47-73).
REL_OUT, REL_LOSS = 1e-3, 1e-2


def _check_all_grads(z, it, nets, tol):
    """Every parameter gradient of `nets` (read from .grad = the flat buffers) against F1r's full tensors of iteration `it`."""
    from tests.helpers import NOISE_PARAMS, rel_err
    n = 0
    for nm, m in nets:
        for k, p in m.named_parameters():
            if nm == "A" and k in NOISE_PARAMS:
                continue
            ref = z["it%d.grad.%s.%s" % (it, nm, k)]
            assert p.grad is not None, (nm, k)
            scale = max(float(np.abs(ref).max()), float(np.sqrt((ref.astype(np.float64) ** 2).sum() / ref.size)), 1e-30)
            err = float((p.grad.detach().cpu().double() - torch.from_numpy(ref).double().reshape(p.shape)).abs().max()) / scale
            assert err < tol, (it, nm, k, err)
            n += 1
    return n


@pytest.mark.parametrize("schedule", ["fused", "as_executed"])
def test_f1r_synchronous_step_on_ragged_pairs_vs_reference(gpu, precision, schedule):
    """F1r through Trainer.train_step (what logging iterations run): 3 iterations, noisy padded 60 / clean padded 47, then the
    clean batch the longer one, then 60 / 47 again - scalars, both gradient norms, enhanced, logits, every parameter gradient at
    every iteration, the final parameters."""
    from tests.helpers import NOISE_PARAMS, batch_from, load, rel_err
    from tests.test_gpu_step import build_tiny, cfg
    from aas_enhancement_amd.trainer_AAS import Trainer
    z = load("f1r_aas_tiny_ragged_pair.npz")
    tr = Trainer(cfg(lr=float(z["cfg_lr"]), schedule=schedule), None, models=build_tiny(z))
    tr.kt = float(z["kt0"])
    gt = 2e-3 if precision != 1 else 3e-2
    for it in range(3):
        ny, cl = batch_from(z, "it%d.ny." % it), batch_from(z, "it%d.cl." % it)
        assert ny[0].shape[2] != cl[0].shape[2]
        r = tr.train_step(ny, cl, it, log_norms=True)
        for k in ("l_adv_ny_G", "l_adv_cl", "l_ctc", "kt", "conv_measure"):
            assert r[k] == pytest.approx(float(z["it%d.%s" % (it, k)]), rel=REL_LOSS), (it, k)
        assert float(r["g_adv"]) == pytest.approx(float(z["it%d.g_adv" % it]), rel=REL_LOSS)
        assert float(r["g_ctc_adv"]) == pytest.approx(float(z["it%d.g_ctc_adv" % it]), rel=REL_LOSS)
        assert rel_err(r["enhanced"], z["it%d.enhanced" % it]) < REL_OUT
        assert rel_err(r["prob"], z["it%d.logits_tnc" % it]) < REL_OUT
        assert _check_all_grads(z, it, [("G", tr.G), ("D", tr.D), ("A", tr.ASR)], gt) > 60
    for nm, m in (("G", tr.G), ("D", tr.D), ("A", tr.ASR)):
        for k, v in m.state_dict().items():
            if nm == "A" and k in NOISE_PARAMS:
                continue
            assert rel_err(v, z["final.%s.%s" % (nm, k)]) < 2e-3, (nm, k)


@pytest.mark.parametrize("lanes,expect", [("auto", "batched-ragged"), ("1", "lanes")], ids=["batched_ragged", "twolanes"])
@pytest.mark.parametrize("frozen", [True, False], ids=["frozenA", "trainableA"])
def test_f1r_device_resident_step_on_ragged_pairs_vs_reference(gpu, precision, frozen, lanes, expect):
    """F1r through train_step_async on BOTH device-resident schedules - the batched discriminator pass with two row classes of
    different length in its recurrent launches (the default for ragged pairs) and the two-lane schedule - frozen / trainable A,
    three arithmetic modes: scalars, enhanced, logits, the two gradients arriving at `enhanced` and every parameter gradient at
    every iteration; with a trainable A the whole 3-iteration trajectory and the final parameters (a frozen A leaves the
    reference's trajectory after iteration 1, where the reference's optimizer_asr first steps)."""
    from tests.helpers import NOISE_PARAMS, batch_from, load, rel_err
    from tests.test_gpu_step import build_tiny, cfg
    from aas_enhancement_amd import knobs, ops
    from aas_enhancement_amd.trainer_AAS import Trainer
    z = load("f1r_aas_tiny_ragged_pair.npz")
    gt = 2e-3 if precision != 1 else 3e-2
    with knobs.override(TWO_LANES=lanes):
        tr = Trainer(cfg(lr=float(z["cfg_lr"]), allow_ASR_update_iter=10 ** 9 if frozen else 0), None, models=build_tiny(z))
        tr.kt = float(z["kt0"])
        tr.keep_enh_grads = True
        for it in range(2 if frozen else 3):
            ny, cl = batch_from(z, "it%d.ny." % it), batch_from(z, "it%d.cl." % it)
            r = tr.train_step_async(ny, cl, it)
            assert tr._last_schedule == expect, tr._last_schedule
            sc = tr.read_scalars()
            for k in ("l_adv_ny_G", "l_adv_cl", "l_ctc", "kt", "conv_measure"):
                assert sc[k] == pytest.approx(float(z["it%d.%s" % (it, k)]), rel=REL_LOSS), (it, k)
            assert rel_err(r["enhanced"], z["it%d.enhanced" % it]) < REL_OUT
            assert rel_err(r["prob"], z["it%d.logits_tnc" % it]) < REL_OUT
            nets = [("G", tr.G), ("D", tr.D)] + ([] if frozen else [("A", tr.ASR)])
            assert _check_all_grads(z, it, nets, gt) >= 40
            for nm, g in zip(("adv", "ctc"), tr._enh_grads):
                assert rel_err(g, z["it%d.enh_grad.%s" % (it, nm)]) < gt, (it, nm)
    assert not ops.rnn_timeout_flag()
    if not frozen:
        for nm, m in (("G", tr.G), ("D", tr.D), ("A", tr.ASR)):
            for k, v in m.state_dict().items():
                if nm == "A" and k in NOISE_PARAMS:
                    continue
                assert rel_err(v, z["final.%s.%s" % (nm, k)]) < 2e-3, (nm, k)


def _f3r_batches():
    from tests.test_gpu_round2 import _config2_batches
    ny, cl = _config2_batches(0)
    Tc = 184
    return ny, (cl[0][:, :, :Tc].contiguous(), None, None, None, torch.zeros(cl[0].size(0), 1, Tc, dtype=torch.uint8))


@pytest.mark.parametrize("lanes,expect", [("auto", "batched-ragged"), ("1", "lanes")], ids=["batched_ragged", "twolanes"])
@pytest.mark.parametrize("frozen", [True, False], ids=["frozenA", "trainableA"])
def test_f3r_config2_ragged_pair_timed_path_vs_reference(gpu, precision, frozen, lanes, expect):
    """F3r: iteration 0 of config 2 with kt0 = 0.3, noisy T = 200 / clean T = 184 (the pair bench.py times as `ragged_pair*`),
    through train_step_async on both schedules x frozen / trainable A x three modes: scalars, 256 enhanced / logit samples, EVERY
    parameter gradient of E / D (/ A) as norm + 64 samples, the networks' total norms and the two gradients arriving at
    `enhanced` - against the reference's own modules (trainer_AAS.py:136-181 with two batches of different padded length)."""
    from tests.helpers import NOISE_PARAMS, load
    from tests.test_gpu_round2 import _config2_models
    from tests.test_gpu_round5 import _check_param_grads, _gtol, _sqnorm
    from tests.test_gpu_step import cfg
    from aas_enhancement_amd import knobs, ops
    from aas_enhancement_amd.trainer_AAS import Trainer
    z = load("f3r_aas_config2_ragged_pair.npz")
    assert int(z["T"]) == 200 and int(z["T_clean"]) == 184
    with knobs.override(TWO_LANES=lanes):
        tr = Trainer(cfg(lr=float(z["lr"]), nFeat=80, rnn_size=500, allow_ASR_update_iter=10 ** 9 if frozen else 0), None, models=_config2_models())
        tr.kt = float(z["kt0"])
        tr.keep_enh_grads = True
        ny, cl = _f3r_batches()
        r = tr.train_step_async(ny, cl, 0)
        assert tr._last_schedule == expect, tr._last_schedule
        sc = tr.read_scalars()
    torch.cuda.synchronize()
    assert not ops.rnn_timeout_flag()
    for k in ("l_adv_ny_G", "l_adv_cl", "l_ctc", "kt", "conv_measure"):
        assert sc[k] == pytest.approx(float(z["it0." + k]), rel=REL_LOSS), k
    enh, prob = r["enhanced"].detach().reshape(-1), r["prob"].detach().reshape(-1)
    e_got = enh[torch.from_numpy(z["it0.enh_idx"]).cuda()].cpu().numpy()
    p_got = prob[torch.from_numpy(z["it0.logit_idx"]).cuda()].cpu().numpy()
    assert np.abs(e_got - z["it0.enh_samples"]).max() < REL_OUT * np.abs(z["it0.enh_samples"]).max()
    assert np.abs(p_got - z["it0.logit_samples"]).max() < REL_OUT * np.abs(z["it0.logit_samples"]).max()
    nets = [("G", tr.G), ("D", tr.D)] + ([] if frozen else [("A", tr.ASR)])
    n, worst = _check_param_grads(z, nets, "it0.", precision, skip=NOISE_PARAMS)
    assert n >= 40 + (0 if frozen else 30), n
    st, nt = _gtol(precision)
    for nm, m in nets:
        assert float(tr.get_gradient_norm(m).item()) == pytest.approx(float(z["it0.gradnorm_total." + nm]), rel=nt), nm
    for nm, g in zip(("adv", "ctc"), tr._enh_grads):
        ref_s = z["it0.enh_grad_samples." + nm]
        got_s = g.detach().reshape(-1)[torch.from_numpy(z["it0.enh_grad_idx." + nm].astype(np.int64)).cuda()].cpu().numpy()
        assert _sqnorm(g) == pytest.approx(float(z["it0.enh_grad_norm." + nm]), rel=nt), nm
        assert np.abs(got_s - ref_s).max() < st * np.abs(ref_s).max(), nm
    print("F3r worst parameter-gradient error", worst)
