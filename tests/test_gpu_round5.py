"""GPU parity tests (-m gpu) added in round 5: gradients of the TIMED path at config-2 size against the reference (F3 samples with
kt0 = 0, F3b every parameter with kt0 = 0.3), the two module methods that had no direct test (forward_paired,
forward_with_intermediate_output), AM_training's DeepSpeech_ken variants (F11), per-trainer launch state, the warp-ctc-ABI
entry points, and the multi-problem GEMM with per-problem reduction extents."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn as nn

from tests.helpers import LABELS, NOISE_PARAMS, load, load_sd, rel_err, sub
from tests.test_gpu_round2 import _config2_batches, _config2_models
from tests.test_gpu_step import cfg

pytestmark = pytest.mark.gpu

REL_OUT, REL_LOSS = 1e-3, 1e-2


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _gtol(precision):
    """(samples, norms) relative tolerance of a parameter gradient at config-2 size vs the reference's fp32-CPU one.  Measured
    (gpurun_out r05): fp32 and fp32-equivalent modes: samples <= 4.5e-4 of the tensor's scale, norms <= 1.5e-5; the split-bf16 fast
    mode (a labelled extra, narrower than fp32 by construction): samples up to 2e-2 on E's first layer - behind A's slope-128
    LeakyReLUs and nine recurrent layers - norms <= 9e-4."""
    return (1e-3, 1e-4) if precision != 1 else (3e-2, 3e-3)


def _sqnorm(t):
    return float(t.detach().double().pow(2).sum().sqrt())


def _check_param_grads(z, nets, prefix, precision, skip=()):
    st, nt = _gtol(precision)
    n = 0
    errs = []
    for nm, m in nets:
        for k, p in m.named_parameters():
            key = "%s.%s" % (nm, k)
            if (prefix + "gradnorm." + key) not in z.files or k in skip:
                continue
            assert p.grad is not None, key
            ref_n = float(z[prefix + "gradnorm." + key])
            got_n = _sqnorm(p.grad)
            ref_s = z[prefix + "gradsample." + key]
            idx = torch.from_numpy(z[prefix + "gradsample_idx." + key].astype(np.int64)).cuda()
            got_s = p.grad.detach().reshape(-1)[idx].cpu().numpy()
            # samples are held to the tensor's own scale: max |sample| can be tiny for a sparse draw, the norm says how big entries are
            scale = max(float(np.abs(ref_s).max()), ref_n / np.sqrt(p.numel()))
            e_s = float(np.abs(got_s - ref_s).max()) / scale
            e_n = abs(got_n - ref_n) / max(ref_n, 1e-30)
            errs.append((e_s, e_n, key))
            n += 1
    errs.sort(reverse=True)
    print("largest parameter-gradient errors (sample, norm, key):", errs[:4], "| worst norm:", max((e[1], e[2]) for e in errs))
    for e_s, e_n, key in errs:
        assert e_n < nt, (key, e_n)
        assert e_s < st, (key, e_s)
    return n, errs[0]


@pytest.mark.parametrize("lanes", ["auto", "1"], ids=["batchedD", "twolanes"])
@pytest.mark.parametrize("frozen", [True, False], ids=["frozenA", "trainableA"])
def test_timed_async_path_config2_gradients_with_live_D_step(gpu, precision, frozen, lanes):
    """F3b: iteration 0 of config 2 with kt0 = 0.3 through the path bench.py times (train_step_async -> _device_core), all
    four schedule variants, three arithmetic modes: EVERY parameter gradient of E and D (A's too when it is trainable) - read
    from the flat gradient buffers after the step, before the next step zeroes them - against the reference's
    (trainer_AAS.py:146-181: G-step, D-step with (-kt), CTC, clean), as norm + 64 samples; the networks' total norms; the two
    gradients that arrive at `enhanced`.  With kt0 != 0 the D-step identity (class-wise (-kt)-weighted weight-gradient
    launches) is checked against the reference at size, not only against an fp64 product."""
    from aas_enhancement_amd import knobs, ops
    from aas_enhancement_amd.trainer_AAS import Trainer
    z = load("f3b_aas_config2_kt.npz")
    with knobs.override(TWO_LANES=lanes):
        tr = Trainer(cfg(lr=float(z["lr"]), nFeat=80, rnn_size=500, allow_ASR_update_iter=10 ** 9 if frozen else 0), None, models=_config2_models())
        tr.kt = float(z["kt0"])
        tr.keep_enh_grads = True
        ny, cl = _config2_batches(0)
        r = tr.train_step_async(ny, cl, 0)
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
        tot = float(tr.get_gradient_norm(m).item())
        assert tot == pytest.approx(float(z["it0.gradnorm_total." + nm]), rel=nt), nm
    for nm, g in zip(("adv", "ctc"), tr._enh_grads):
        ref_s = z["it0.enh_grad_samples." + nm]
        got_s = g.detach().reshape(-1)[torch.from_numpy(z["it0.enh_grad_idx." + nm].astype(np.int64)).cuda()].cpu().numpy()
        assert _sqnorm(g) == pytest.approx(float(z["it0.enh_grad_norm." + nm]), rel=nt), nm
        assert np.abs(got_s - ref_s).max() < st * np.abs(ref_s).max(), nm
    print("worst parameter-gradient error", worst)


def test_timed_async_path_config2_f3_gradient_samples(gpu, precision):
    """The six gradient samples F3 has carried since round 2 (kt0 = 0: D's gradients are the clean pass's alone), trainable A,
    both device-resident schedules."""
    from aas_enhancement_amd import knobs
    from aas_enhancement_amd.trainer_AAS import Trainer
    z = load("f3_aas_config2.npz")
    for lanes in ("auto", "1"):
        with knobs.override(TWO_LANES=lanes):
            tr = Trainer(cfg(lr=float(z["lr"]), nFeat=80, rnn_size=500, allow_ASR_update_iter=0), None, models=_config2_models())
            tr.kt = float(z["kt0"])
            ny, cl = _config2_batches(0)
            tr.train_step_async(ny, cl, 0)
            tr.read_scalars()
        n, _ = _check_param_grads(z, [("G", tr.G), ("D", tr.D), ("A", tr.ASR)], "it0.", precision)
        assert n == 6


# ---- the two stackedBRNN methods without a direct test (model.py:233-252) ----------------------------------------
def test_forward_paired_golden(gpu, precision):
    """F4 `paired.*`: the reference's own forward_paired (cat on the feature axis, I = 12, O = 6)."""
    from aas_enhancement_amd.model import stackedBRNN
    z = load("f4_ops.npz")
    D = stackedBRNN(I=12, O=6, H=10, L=4)
    load_sd(D, sub(z, "paired.sd."))
    D.cuda()
    y = D.forward_paired(torch.from_numpy(z["paired.a"]).cuda(), torch.from_numpy(z["paired.b"]).cuda())
    assert tuple(y.shape) == tuple(z["paired.y"].shape)
    assert rel_err(y, z["paired.y"]) < (1e-5 if precision != 1 else 1e-4)


def test_forward_with_intermediate_output_golden(gpu, precision):
    """F4 `inter.*`: [output N x O x T, last recurrent layer's output as N x H x T] (model.py:240-252)."""
    from aas_enhancement_amd.model import stackedBRNN
    z = load("f4_ops.npz")
    G = stackedBRNN(I=6, O=6, H=10, L=4)
    load_sd(G, sub(z, "inter.sd."))
    G.cuda()
    out = G.forward_with_intermediate_output(torch.from_numpy(z["inter.x"]).cuda())
    assert isinstance(out, list) and len(out) == 2
    tol = 1e-5 if precision != 1 else 1e-4
    assert tuple(out[0].shape) == tuple(z["inter.y"].shape) and tuple(out[1].shape) == tuple(z["inter.h"].shape)
    assert rel_err(out[0], z["inter.y"]) < tol
    assert rel_err(out[1], z["inter.h"]) < tol
    assert rel_err(G(torch.from_numpy(z["inter.x"]).cuda()), z["inter.y"]) < tol
