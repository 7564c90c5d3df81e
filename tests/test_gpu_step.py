"""GPU parity tests (-m gpu): whole modules and whole training steps (the code main.py runs) against
golden vectors generated from the reference's model.py and against the CPU oracle.
north_star tolerances: logits / enhanced within 1e-3 relative, losses within 1e-2 relative."""
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from tests.helpers import LABELS, NOISE_PARAMS, batch_from, grad_close, load, load_sd, rel_err, sub
from tests.tools_shim import make_batch

pytestmark = pytest.mark.gpu

REL_OUT, REL_LOSS = 1e-3, 1e-2


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def cfg(**kw):
    c = types.SimpleNamespace(lr=1e-5, beta1=0.5, beta2=0.999, optimizer="adam", batch_size=30, expnum=0, lambda_k=0.001,
                              gamma=0.5, gpu=0, load_path="", mode="train", write_log=False, w_adversarial=1.0,
                              w_acoustic=1.0, allow_ASR_update_iter=0, schedule="fused", nFeat=8, rnn_size=16,
                              rnn_layers=4, rnn_type="lstm")
    c.__dict__.update(kw)
    return c


def build_tiny(z):
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    G = stackedBRNN(I=8, H=16, L=4)
    D = stackedBRNN(I=8, H=16, L=4)
    A = DeepSpeech(nn.GRU, LABELS, 12, 5, True, 11, 2, 8, 2, nFreq=8)
    for nm, m in (("G", G), ("D", D), ("A", A)):
        load_sd(m, sub(z, "init.%s." % nm))
    return G, D, A



@pytest.mark.parametrize("schedule", ["fused", "as_executed"])
def test_aas_step_tiny_golden(gpu, precision, schedule):
    """F1: 3 full AAS iterations, ragged batch, every tensor checked."""
    from aas_enhancement_amd.trainer_AAS import Trainer
    z = load("f1_aas_tiny.npz")
    tr = Trainer(cfg(lr=float(z["cfg_lr"]), schedule=schedule), None, models=build_tiny(z))
    tr.kt = float(z["kt0"])
    for it in range(3):
        ny, cl = batch_from(z, "it%d.ny." % it), batch_from(z, "it%d.cl." % it)
        r = tr.train_step(ny, cl, it, log_norms=True)
        for k in ("l_adv_ny_G", "l_adv_cl", "l_ctc", "kt", "conv_measure"):
            assert r[k] == pytest.approx(float(z["it%d.%s" % (it, k)]), rel=REL_LOSS), (it, k)
        assert float(r["g_adv"]) == pytest.approx(float(z["it%d.g_adv" % it]), rel=REL_LOSS)
        assert float(r["g_ctc_adv"]) == pytest.approx(float(z["it%d.g_ctc_adv" % it]), rel=REL_LOSS)
        assert rel_err(r["enhanced"], z["it%d.enhanced" % it]) < REL_OUT
        assert rel_err(r["prob"], z["it%d.logits_tnc" % it]) < REL_OUT
    for nm, m in (("G", tr.G), ("D", tr.D), ("A", tr.ASR)):
        for k, v in m.state_dict().items():
            if nm == "A" and k in NOISE_PARAMS:
                continue
            assert rel_err(v, z["final.%s.%s" % (nm, k)]) < 2e-3, (nm, k)


def test_chain_schedules_agree_config2(gpu, monkeypatch, precision2):
    """The three ways of queueing the discriminator and acoustic passes - one after the other on one stream, on two
    streams one chain after the other, and layer by layer in alternation with one combined backward - are the same
    computation: identical scalars, enhanced output and parameters over 2 steps at BASELINE config-2 sizes."""
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    from aas_enhancement_amd.trainer_AAS import Trainer
    N, F, T, H, HA, M, L = 30, 80, 200, 500, 1000, 128, 20
    res = {}
    from aas_enhancement_amd import knobs
    for mode, kn in (("serial", dict(OVERLAP_ASR=False)), ("two_streams", dict(OVERLAP_ASR=True, INTERLEAVE=False)),
                     ("alternating", dict(OVERLAP_ASR=True, INTERLEAVE=True))):
        for k, v in kn.items():
            monkeypatch.setitem(knobs._values, k, v)
        G, D = stackedBRNN(I=F, H=H, L=4), stackedBRNN(I=F, H=H, L=4)
        A = DeepSpeech(nn.GRU, LABELS, HA, 5, True, 11, 2, M, 2, nFreq=F)
        for m, s, cs in ((G, 9001, None), (D, 9002, None), (A, 9003, 0.1)):
            load_sd(m, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(m.state_dict(), s, conv_std=cs).items()}, strict=False)
        tr = Trainer(cfg(nFeat=F, rnn_size=H, allow_ASR_update_iter=10 ** 9), None, models=(G, D, A))
        out = []
        for it in range(2):
            ny = (torch.from_numpy(prng.uniform(123 + it, (N, F, T), 0.0, 6.0)), torch.from_numpy(prng.randint(125 + it, (N * L,), 1, 28).astype(np.int32)),
                  torch.ones(N), torch.full((N,), L, dtype=torch.int32), torch.zeros(N, 1, T, dtype=torch.uint8))
            cl = (torch.from_numpy(prng.uniform(124 + it, (N, F, T), 0.0, 6.0)), None, None, None, torch.zeros(N, 1, T, dtype=torch.uint8))
            r = tr.train_step(ny, cl, it, log_norms=False)
            out.append([r[k] for k in ("l_adv_ny_G", "l_adv_cl", "l_ctc", "kt")] + [float(r["enhanced"].detach().double().abs().sum()), float(r["prob"].detach().double().abs().sum())])
        torch.cuda.synchronize()
        from aas_enhancement_amd import ops
        assert not ops.rnn_timeout_flag()
        out.append([float(sum(p.double().abs().sum() for p in m.parameters())) for m in (G, D)] + [0.0] * 4)
        res[mode] = np.asarray(out)
    assert np.allclose(res["serial"], res["two_streams"], rtol=2e-5), (res["serial"], res["two_streams"])
    assert np.allclose(res["serial"], res["alternating"], rtol=2e-5), (res["serial"], res["alternating"])


def test_async_steps_match_synchronous_steps(gpu, precision2):
    """Trainer.train_step_async (kt / losses device-resident, no read-back) == Trainer.train_step over 3 steps (tiny shapes)."""
    from aas_enhancement_amd.trainer_AAS import Trainer
    z = load("f1_aas_tiny.npz")
    res = {}
    for mode in ("sync", "async"):
        tr = Trainer(cfg(lr=float(z["cfg_lr"]), schedule="fused", allow_ASR_update_iter=10 ** 9), None, models=build_tiny(z))
        tr.kt = float(z["kt0"])
        out = []
        for it in range(3):
            ny, cl = batch_from(z, "it%d.ny." % it), batch_from(z, "it%d.cl." % it)
            if mode == "sync":
                r = tr.train_step(ny, cl, it, log_norms=False)
            else:
                tr.train_step_async(ny, cl, it)
                r = tr.read_scalars()
            out.append([r[k] for k in ("l_adv_ny_G", "l_adv_cl", "l_ctc", "kt")])
        res[mode] = np.asarray(out)
    assert np.allclose(res["sync"], res["async"], rtol=1e-5, atol=1e-9), (res["sync"], res["async"])


def test_forward_stages_equal_forward(gpu):
    """stackedBRNN / DeepSpeech.forward_stages (the layer-by-layer generators the trainer alternates) give forward()."""
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    torch.manual_seed(0)
    D = stackedBRNN(I=12, H=24, L=3).cuda()
    A = DeepSpeech(nn.GRU, LABELS, 16, 2, True, 11, 2, 8, 2, nFreq=12).cuda()
    x = torch.rand(3, 12, 50, device="cuda")
    for m in (D, A):
        *_, last = m.forward_stages(x)
        assert torch.equal(last, m(x))


def test_aas_grads_tiny_golden(gpu, precision2):
    """F1 iteration 0: every parameter gradient of E, D and A at optimiser-step time (as-executed schedule)."""
    from aas_enhancement_amd.ctc import CTCLoss
    from aas_enhancement_amd.model import L1Loss_mask
    z = load("f1_aas_tiny.npz")
    G, D, A = [m.cuda() for m in build_tiny(z)]
    ny, cl = batch_from(z, "it0.ny.", "cuda"), batch_from(z, "it0.cl.", "cuda")
    kt = float(z["kt0"])
    diff = L1Loss_mask()
    enhanced = G(ny[0])
    l, _ = diff(D(enhanced), enhanced, ny[4])
    l.backward(retain_graph=True)
    D.zero_grad()
    ed = enhanced.detach()
    l, _ = diff(D(ed), ed, ny[4])
    (l * (-kt)).backward()
    prob = A(enhanced).transpose(0, 1)
    sizes = ny[2].clone().mul_(int(prob.size(0))).int()
    assert torch.equal(sizes, torch.from_numpy(z["it0.sizes"]))
    (CTCLoss()(prob, ny[1], sizes, ny[3]) / 3).backward()
    l, _ = diff(D(cl[0]), cl[0], cl[4])
    l.backward()
    for nm, m in (("G", G), ("D", D), ("A", A)):
        for k, p in m.named_parameters():
            assert grad_close(p.grad, z["it0.grad.%s.%s" % (nm, k)], rtol=2e-3, atol=1e-5), (nm, k)


def test_dce_config1_golden(gpu, precision2):
    """F2: BASELINE config 1 (N=4,F=80,T=200,H=128, 4 layers), 5 DCE steps."""
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.model import stackedBRNN
    from aas_enhancement_amd.trainer_DCE import Trainer
    z = load("f2_dce_config1.npz")
    N, F, T, H = int(z["N"]), int(z["F"]), int(z["T"]), int(z["H"])
    G = stackedBRNN(I=F, H=H, L=4)
    load_sd(G, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(G.state_dict(), int(z["weight_seed"])).items()})
    tr = Trainer(cfg(lr=float(z["lr"]), nFeat=F, rnn_size=H), None, models=(G,))
    for it in range(5):
        x = torch.from_numpy(prng.uniform(int(z["input_seed0"]) + it, (N, F, T), 0.0, 6.0))
        c = torch.from_numpy(prng.uniform(int(z["clean_seed0"]) + it, (N, F, T), 0.0, 6.0))
        r = tr.train_step((x, c, torch.zeros(N, 1, T, dtype=torch.uint8)), it)
        assert float(r["dce"]) == pytest.approx(float(z["losses"][it]), rel=1e-4)
        got = r["outputs"].detach().reshape(-1)[torch.from_numpy(z["sample_idx"]).cuda()]
        assert rel_err(got, z["out_samples"][it]) < REL_OUT
        assert float(r["outputs"].detach().double().sum()) == pytest.approx(float(z["out_sums"][it]), rel=1e-3)


def test_aas_config2_golden(gpu, precision):
    """F3: BASELINE config 2 (N=30,T=200,F=80; E/D 4x500 BiLSTM; A 2xconv+5x1000 BiGRU+CTC), 2 iterations."""
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    from aas_enhancement_amd.trainer_AAS import Trainer
    z = load("f3_aas_config2.npz")
    N, F, T, H, HA, M, L = [int(z[k]) for k in ("N", "F", "T", "H", "HA", "M", "L")]
    seed = int(z["weight_seed"])
    G, D = stackedBRNN(I=F, H=H, L=4), stackedBRNN(I=F, H=H, L=4)
    A = DeepSpeech(nn.GRU, LABELS, HA, 5, True, 11, 2, M, 2, nFreq=F)
    for m, s, cs in ((G, seed + 1, None), (D, seed + 2, None), (A, seed + 3, 0.1)):
        load_sd(m, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(m.state_dict(), s, conv_std=cs).items()}, strict=False)
    tr = Trainer(cfg(lr=float(z["lr"]), nFeat=F, rnn_size=H), None, models=(G, D, A))
    tr.kt = float(z["kt0"])
    for it in range(2):
        ny = (torch.from_numpy(prng.uniform(123 + 1000 * it, (N, F, T), 0.0, 6.0)),
              torch.from_numpy(prng.randint(125 + 1000 * it, (N * L,), 1, 28).astype(np.int32)),
              torch.ones(N), torch.full((N,), L, dtype=torch.int32), torch.zeros(N, 1, T, dtype=torch.uint8))
        cl = (torch.from_numpy(prng.uniform(124 + 1000 * it, (N, F, T), 0.0, 6.0)), None, None, None, torch.zeros(N, 1, T, dtype=torch.uint8))
        r = tr.train_step(ny, cl, it, log_norms=True)
        for k in ("l_adv_ny_G", "l_adv_cl", "l_ctc", "kt", "conv_measure"):
            assert r[k] == pytest.approx(float(z["it%d.%s" % (it, k)]), rel=REL_LOSS), (it, k)
        assert float(r["g_adv"]) == pytest.approx(float(z["it%d.g_adv" % it]), rel=REL_LOSS)
        assert float(r["g_ctc_adv"]) == pytest.approx(float(z["it%d.g_ctc_adv" % it]), rel=REL_LOSS)
        enh, prob = r["enhanced"].detach().reshape(-1), r["prob"].detach().reshape(-1)
        e_ref, p_ref = z["it%d.enh_samples" % it], z["it%d.logit_samples" % it]
        e_got = enh[torch.from_numpy(z["it%d.enh_idx" % it]).cuda()].cpu().numpy()
        p_got = prob[torch.from_numpy(z["it%d.logit_idx" % it]).cuda()].cpu().numpy()
        assert np.abs(e_got - e_ref).max() < REL_OUT * np.abs(e_ref).max(), it
        assert np.abs(p_got - p_ref).max() < REL_OUT * np.abs(p_ref).max(), it
        assert float(enh.double().sum()) == pytest.approx(float(z["it%d.enh_sum" % it]), rel=1e-3)
    from aas_enhancement_amd import ops
    assert not ops.rnn_timeout_flag()


def test_fsegan_golden(gpu, precision2):
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.model import stackedBRNN
    from aas_enhancement_amd.trainer_FSEGAN import Trainer
    z = load("f5_fsegan_am.npz")
    for variant in ("intended", "as_written"):
        G, D = stackedBRNN(I=8, O=8, H=12, L=4), stackedBRNN(I=16, O=8, H=12, L=4)
        load_sd(G, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(G.state_dict(), 7001).items()})
        load_sd(D, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(D.state_dict(), 7002).items()})
        tr = Trainer(cfg(lr=1e-3, w_adversarial=0.01, fsegan_as_written=(variant == "as_written"), rnn_size=12), None, models=(G, D))
        tr.kt = 0.2
        for it in range(2):
            b = make_batch(3, 8, [30, 26, 19], 7100 + it)
            cl = make_batch(3, 8, [30, 26, 19], 7200 + it)["inputs"]
            r = tr.train_step((torch.from_numpy(b["inputs"]), torch.from_numpy(cl), torch.from_numpy(b["mask"])), it)
            p = "fsegan_%s.it%d." % (variant, it)
            for k in ("l_adv_ny_G", "l_adv_cl", "dce", "kt", "g_norm"):
                assert r[k] == pytest.approx(float(z[p + k]), rel=REL_LOSS), (variant, it, k)
        for nm, m in (("G", tr.G), ("D", tr.D)):
            for k, v in m.state_dict().items():
                assert rel_err(v, z["fsegan_%s.final.%s.%s" % (variant, nm, k)]) < 2e-3, (variant, nm, k)


def test_am_step_golden(gpu, precision2):
    """AM_training/train.py:297-349 (config 5's per-GPU step): A(x) -> CTC/N -> plain Adam."""
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.ctc import CTCLoss
    from aas_enhancement_amd.model import DeepSpeech
    from aas_enhancement_amd.optim import Adam
    z = load("f5_fsegan_am.npz")
    A = DeepSpeech(nn.GRU, LABELS, 12, 3, True, 11, 2, 8, 2, nFreq=8)
    load_sd(A, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(A.state_dict(), 8001, conv_std=0.1).items()}, strict=False)
    A.cuda()
    opt = Adam(A.parameters(), lr=1e-3)
    crit = CTCLoss()
    for it in range(2):
        b = make_batch(3, 8, [60, 50, 38], 8100 + it, [4, 3, 2], 8200 + it)
        out = A(torch.from_numpy(b["inputs"]).cuda()).transpose(0, 1)
        sizes = torch.from_numpy(b["pct"]).clone().mul_(int(out.size(0))).int()
        loss = crit(out, torch.from_numpy(b["targets"]), sizes, torch.from_numpy(b["target_sizes"])) / 3
        opt.zero_grad()
        loss.backward()
        opt.step()
        assert float(loss) == pytest.approx(float(z["am.it%d.loss" % it]), rel=REL_LOSS)
        assert rel_err(out, z["am.it%d.logits" % it]) < REL_OUT
    for k, v in A.state_dict().items():
        if k in NOISE_PARAMS:
            continue
        assert rel_err(v.double(), z["am.final." + k]) < 2e-3, k


def test_am_trainer_class_golden(gpu, precision2):
    """aas_enhancement_amd.am_train.AMTrainer (flat buffers, fused Adam, side-stream wgrads) on the F5 AM vectors."""
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.am_train import AMTrainer
    from aas_enhancement_amd.model import DeepSpeech
    z = load("f5_fsegan_am.npz")
    A = DeepSpeech(nn.GRU, LABELS, 12, 3, True, 11, 2, 8, 2, nFreq=8)
    load_sd(A, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(A.state_dict(), 8001, conv_std=0.1).items()}, strict=False)
    tr = AMTrainer(A.cuda(), lr=1e-3)
    for it in range(2):
        b = make_batch(3, 8, [60, 50, 38], 8100 + it, [4, 3, 2], 8200 + it)
        r = tr.train_step((torch.from_numpy(b["inputs"]), torch.from_numpy(b["targets"]), torch.from_numpy(b["pct"]), torch.from_numpy(b["target_sizes"])))
        assert r["loss"] == pytest.approx(float(z["am.it%d.loss" % it]), rel=REL_LOSS)
        assert rel_err(r["logits"], z["am.it%d.logits" % it]) < REL_OUT
    for k, v in A.state_dict().items():
        if k in NOISE_PARAMS:
            continue
        assert rel_err(v.double(), z["am.final." + k]) < 2e-3, k


def test_am_async_steps_equal_synchronous_steps_and_goldens(gpu, precision2):
    """AMTrainer.train_step_async (no host read-back: device step counter, loss through a pinned ring, read one step late) on
    the F5 AM vectors - same losses, logits and final weights as the reference; interleaved with synchronous steps the Adam
    counters stay in step."""
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.am_train import AMTrainer
    from aas_enhancement_amd.model import DeepSpeech
    z = load("f5_fsegan_am.npz")
    A = DeepSpeech(nn.GRU, LABELS, 12, 3, True, 11, 2, 8, 2, nFreq=8)
    load_sd(A, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(A.state_dict(), 8001, conv_std=0.1).items()}, strict=False)
    tr = AMTrainer(A.cuda(), lr=1e-3)
    recs = []
    for it in range(2):
        b = make_batch(3, 8, [60, 50, 38], 8100 + it, [4, 3, 2], 8200 + it)
        recs.append(tr.train_step_async((torch.from_numpy(b["inputs"]), torch.from_numpy(b["targets"]), torch.from_numpy(b["pct"]),
                                         torch.from_numpy(b["target_sizes"]))))
    for it, r in enumerate(recs):      # read late: nothing of the update depended on the host seeing the loss
        loss, is_inf = tr.read_loss(r["handle"])
        assert not is_inf and loss == pytest.approx(float(z["am.it%d.loss" % it]), rel=REL_LOSS)
        assert rel_err(r["logits"], z["am.it%d.logits" % it]) < REL_OUT
    for k, v in A.state_dict().items():
        if k in NOISE_PARAMS:
            continue
        assert rel_err(v.double(), z["am.final." + k]) < 2e-3, k
    # alternate the two forms on a second model: equal to synchronous steps only
    outs = []
    for mode in ("sync", "mixed"):
        A2 = DeepSpeech(nn.GRU, LABELS, 12, 3, True, 11, 2, 8, 2, nFreq=8)
        load_sd(A2, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(A2.state_dict(), 8001, conv_std=0.1).items()}, strict=False)
        t2 = AMTrainer(A2.cuda(), lr=1e-3)
        for it in range(4):
            b = make_batch(3, 8, [60, 50, 38], 8100 + it, [4, 3, 2], 8200 + it)
            data = (torch.from_numpy(b["inputs"]), torch.from_numpy(b["targets"]), torch.from_numpy(b["pct"]), torch.from_numpy(b["target_sizes"]))
            if mode == "mixed" and it % 2 == 1:
                t2.read_loss(t2.train_step_async(data)["handle"])
            else:
                t2.train_step(data)
        outs.append({k: v.detach().clone() for k, v in A2.state_dict().items()})
    for k in outs[0]:
        if k not in NOISE_PARAMS:
            assert rel_err(outs[1][k].double(), outs[0][k].double()) < 1e-4, k      # (bias corrections: device fp32 vs host fp64)


def test_round_trip_properties_full_size(gpu):
    """Size-independent properties at BASELINE config-2 sizes: (i) linearity of backward in the upstream
    gradient, (ii) fused and as-executed schedules agree, (iii) the step is deterministic run to run
    up to atomics-order rounding."""
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.model import stackedBRNN
    N, F, T, H = 30, 80, 200, 500
    G = stackedBRNN(I=F, H=H, L=4)
    load_sd(G, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(G.state_dict(), 31).items()})
    G.cuda()
    x = torch.from_numpy(prng.uniform(5, (N, F, T), 0.0, 6.0)).cuda().requires_grad_(True)
    g1 = torch.from_numpy(prng.normal(6, (N, F, T))).cuda()
    y = G(x)
    (gx1,) = torch.autograd.grad(y, x, g1, retain_graph=True)
    (gx2,) = torch.autograd.grad(y, x, 2.5 * g1, retain_graph=True)
    assert rel_err(gx2, 2.5 * gx1) < 1e-4  # split-bf16 products are not exactly linear in fp32 rounding
    y2 = G(x)
    assert rel_err(y2, y) < 1e-6
