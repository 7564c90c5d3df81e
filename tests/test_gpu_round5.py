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


# ---- AM_training's model class (DeepSpeech_ken) ---------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["bn", "nobn", "nobn_ds2", "lstm_nobn"])
def test_deepspeech_ken_variants_golden(gpu, precision, tag):
    """F11: AM_training/model.py:337-470 DeepSpeech_ken with / without the first BatchNorm (`--include_first_BN`), `--nDownsample`
    1 / 2, GRU / LSTM: same state_dict keys as the reference's Sequential numbering, logits, input gradient, every parameter
    gradient, BatchNorm running statistics after the pass."""
    from aas_enhancement_amd.model import DeepSpeech_ken
    z = load("f11_am_model_ken.npz")
    p = "ken_%s." % tag
    A = DeepSpeech_ken(nn.LSTM if tag.startswith("lstm") else nn.GRU, LABELS, 12, 3, True, 11, 2, 8, 2, nFreq=10,
                       nDownsample=2 if tag.endswith("ds2") else 1, include_first_BN=(tag == "bn"))
    sd0 = sub(z, p + "sd0.")
    assert set(A.state_dict().keys()) == set(sd0.keys())
    load_sd(A, sd0)
    A.cuda()
    x = torch.from_numpy(z[p + "x"]).cuda().requires_grad_(True)
    y = A(x)
    assert tuple(y.shape) == tuple(z[p + "y"].shape) and y.shape[1] == A.output_length(x.shape[2])
    y.backward(torch.from_numpy(z[p + "gy"]).cuda())
    ft, gt = (1e-5, 2e-4) if precision != 1 else (1e-4, 1e-3)
    assert rel_err(y, z[p + "y"]) < ft
    assert rel_err(x.grad, z[p + "gx"]) < gt
    for k, v in A.named_parameters():
        g = z[p + "gw." + k]
        if k.endswith(".bias") and np.abs(g).max() < 1e-5:    # conv bias in front of a train-mode BatchNorm: exact gradient 0
            continue
        assert float((v.grad.cpu() - torch.from_numpy(g)).abs().max()) <= gt * float(np.abs(g).max()) + 1e-6, k
    for k, v in A.state_dict().items():
        if "running" in k and not (k.startswith("conv") and "running_mean" in k):
            assert rel_err(v, z[p + "sd1." + k]) < 1e-4, k
    # the package round trip keeps the variant (the reference's packages do not record it: the keys do)
    B = DeepSpeech_ken.load_model_package(DeepSpeech_ken.serialize(A))
    assert set(B.state_dict().keys()) == set(sd0.keys()) and B.output_length(90) == A.output_length(90)


# ---- per-trainer launch state --------------------------------------------------------------------------------------
def test_two_trainers_with_different_launch_state_interleave(gpu):
    """Two trainers in one process with different library settings - arithmetic mode, GEMM workgroup-lifetime cap, kernel-selection
    bits - stepping in alternation produce, bit for bit, what each produces alone; and neither leaves its settings behind."""
    from aas_enhancement_amd import ops
    from aas_enhancement_amd._lib import lib
    from aas_enhancement_amd.trainer_AAS import Trainer
    from tests.test_gpu_step import build_tiny
    from tests.helpers import batch_from
    z = load("f1_aas_tiny.npz")
    settings = [dict(precision=0, launch=ops.LaunchState(gemm_max_steps=48, debug_flags=0)),
                dict(precision=1, launch=ops.LaunchState(gemm_max_steps=0, debug_flags=512))]

    def make(i):
        tr = Trainer(cfg(lr=float(z["cfg_lr"])), None, models=build_tiny(z))
        tr.kt = float(z["kt0"])
        tr.set_precision(settings[i]["precision"])
        tr.launch = settings[i]["launch"]
        return tr

    def run(trs, order):
        out = {id(t): [] for t in trs}
        its = {id(t): 0 for t in trs}
        for i in order:
            t = trs[i]
            it = its[id(t)]
            r = t.train_step(batch_from(z, "it%d.ny." % it), batch_from(z, "it%d.cl." % it), it, log_norms=False)
            out[id(t)].append([r[k] for k in ("l_adv_ny_G", "l_adv_cl", "l_ctc", "kt")] + [float(r["enhanced"].double().sum())])
            its[id(t)] += 1
        return [np.asarray(out[id(t)]) for t in trs], [{k: v.detach().clone() for k, v in t.G.state_dict().items()} for t in trs]
    before = (ops.get_precision(), int(lib().aas_get_gemm_max_steps()), int(lib().aas_get_debug_flags()))
    alone = [run([make(i)], [0, 0, 0]) for i in (0, 1)]
    both = run([make(0), make(1)], [0, 1, 1, 0, 0, 1])
    assert (ops.get_precision(), int(lib().aas_get_gemm_max_steps()), int(lib().aas_get_debug_flags())) == before
    for i in (0, 1):
        assert np.array_equal(alone[i][0][0], both[0][i]), i
        for k, v in alone[i][1][0].items():
            assert torch.equal(v, both[1][i][k]), (i, k)
    assert not np.array_equal(both[0][0], both[0][1])       # (the two modes do differ in the last bits)


# ---- warp-ctc's own C ABI -------------------------------------------------------------------------------------------
class _CtcOptUnion(ctypes.Union):
    _fields_ = [("num_threads", ctypes.c_uint), ("stream", ctypes.c_void_p)]


class _CtcOptions(ctypes.Structure):
    _anonymous_ = ("u",)
    _fields_ = [("loc", ctypes.c_int), ("u", _CtcOptUnion), ("blank_label", ctypes.c_int)]


def test_warpctc_abi_entry_points(gpu):
    """include/aas_warpctc.h: get_workspace_size / compute_ctc_loss with warp-ctc's exact signatures (ctcOptions by value, costs on
    the host) against the numpy fp64 oracle - what warpctc_pytorch's binding calls (trainer_AAS.py:168 through CTCLoss)."""
    from aas_enhancement_amd import _lib
    from oracle import ctc_np
    L = ctypes.CDLL(_lib.LIB_PATH)
    L.get_workspace_size.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _CtcOptions, ctypes.POINTER(ctypes.c_size_t)]
    L.compute_ctc_loss.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_void_p, _CtcOptions]
    L.ctcGetStatusString.restype = ctypes.c_char_p
    assert L.get_warpctc_version() >= 2 and L.ctcGetStatusString(0) == b"no error"
    rng = np.random.RandomState(5)
    T, N, C = 85, 6, 29
    acts = torch.from_numpy((rng.randn(T, N, C) * 2).astype(np.float32)).cuda()
    ll = np.array([20, 7, 1, 0, 12, 20], np.int32)
    al = np.array([85, 60, 85, 2, 40, 85], np.int32)
    labels = np.concatenate([rng.randint(1, C, size=k) for k in ll]).astype(np.int32)
    st = torch.cuda.Stream()
    opt = _CtcOptions()
    opt.loc, opt.stream, opt.blank_label = 1, st.cuda_stream, 0
    sz = ctypes.c_size_t(0)
    assert L.get_workspace_size(ll.ctypes.data, al.ctypes.data, C, N, opt, ctypes.byref(sz)) == 0 and sz.value > 0
    ws = torch.empty(sz.value, dtype=torch.uint8, device="cuda")
    grads = torch.full_like(acts, 7.0)
    costs = np.zeros(N, np.float32)
    torch.cuda.synchronize()
    assert L.compute_ctc_loss(acts.data_ptr(), grads.data_ptr(), labels.ctypes.data, ll.ctypes.data, al.ctypes.data, C, N, costs.ctypes.data,
                              ws.data_ptr(), opt) == 0
    rc, rg = ctc_np.ctc_batch(acts.cpu().numpy(), labels, al, ll)
    assert np.allclose(costs, rc, rtol=1e-5)
    assert np.abs(grads.cpu().numpy() - rg).max() < 2e-5
    # gradients may be NULL (costs only); CPU location and a bad blank label are refused with warp-ctc's status codes
    costs2 = np.zeros(N, np.float32)
    assert L.compute_ctc_loss(acts.data_ptr(), None, labels.ctypes.data, ll.ctypes.data, al.ctypes.data, C, N, costs2.ctypes.data, ws.data_ptr(), opt) == 0
    assert np.allclose(costs2, rc, rtol=1e-5)
    opt.loc = 0
    assert L.compute_ctc_loss(acts.data_ptr(), None, labels.ctypes.data, ll.ctypes.data, al.ctypes.data, C, N, costs2.ctypes.data, ws.data_ptr(), opt) == 3
    opt.loc, opt.blank_label = 1, C
    assert L.compute_ctc_loss(acts.data_ptr(), None, labels.ctypes.data, ll.ctypes.data, al.ctypes.data, C, N, costs2.ctypes.data, ws.data_ptr(), opt) == 2


# ---- ADVICE r4 ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
def test_multi_problem_gemm_with_unequal_reduction_extents(gpu, mode):
    """aas_gemm_f32_multi with per-problem K that are NOT multiples of 4 ({100, 98}) in all three operand forms: a k-contiguous
    operand is fetched in 4-k chunks, so such problems must take the general kernel (the chunk guard looked at the
    launch's largest K only and added k = 98..99 of the second problem's operands into its result)."""
    from aas_enhancement_amd import ops
    M, N, Ks = 128, 192, [100, 98]
    g = torch.Generator().manual_seed(17)
    Kmax = max(Ks)
    outs, want, As, Bs = [], [], [], []
    keep = []
    for K in Ks:
        a, b = torch.randn(M, Kmax, generator=g), torch.randn(N, Kmax, generator=g)      # columns K..Kmax hold data that must NOT be summed
        want.append(a[:, :K].double() @ b[:, :K].double().t())
        if mode == "nt":
            A, B = a.cuda(), b.cuda()
        elif mode == "nn":
            A, B = a.cuda(), b.t().contiguous().cuda()
        else:
            A, B = a.t().contiguous().cuda(), b.t().contiguous().cuda()
        keep += [A, B]
        As.append(A.data_ptr()); Bs.append(B.data_ptr())
        outs.append(torch.zeros(M, N, device="cuda"))
    lda = Kmax if mode != "tn" else M
    ldb = Kmax if mode == "nt" else N
    ops.gemm_multi({"nt": ops.NT, "nn": ops.NN, "tn": ops.TN}[mode], M, N, Ks, As, lda, Bs, ldb, [o.data_ptr() for o in outs], N)
    for o, w in zip(outs, want):
        assert rel_err(o, w) < 1e-5, mode


def test_split_k_workspace_growth_keeps_captured_graphs_valid(gpu):
    """A hipGraph captured while the split-K slab block of its stream was small must still replay correctly after a deeper product
    on the same stream made the block grow (the old block used to be freed; it is retired now)."""
    from aas_enhancement_amd import _lib, ops
    L = _lib.lib()
    prev = int(L.aas_get_gemm_max_steps())
    L.aas_set_gemm_max_steps(16)
    try:
        st = torch.cuda.Stream()
        g = torch.Generator().manual_seed(23)
        M, N, K = 256, 256, 16384                    # deep and narrow: split-K with slabs
        a, b = torch.randn(K, M, generator=g).cuda(), torch.randn(K, N, generator=g).cuda()
        c = torch.zeros(M, N, device="cuda")
        want = a.double().t() @ b.double()
        with torch.cuda.stream(st):
            ops.gemm(ops.TN, M, N, K, a, M, b, N, c, N)      # sizes the block for this stream
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            ops.gemm(ops.TN, M, N, K, a, M, b, N, c, N)
        c.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert rel_err(c, want) < 2e-5
        # a much larger split product on the same stream: the block has to grow past the 64 MB floor
        M2, N2, K2 = 2048, 2048, 8192
        a2, b2 = torch.randn(K2, M2, generator=g).cuda(), torch.randn(K2, N2, generator=g).cuda()
        c2 = torch.zeros(M2, N2, device="cuda")
        with torch.cuda.stream(st):
            ops.gemm(ops.TN, M2, N2, K2, a2, M2, b2, N2, c2, N2)
        torch.cuda.synchronize()
        assert rel_err(c2, a2.double().t() @ b2.double()) < 2e-5
        junk = [torch.full((32 << 20,), float("nan"), device="cuda") for _ in range(4)]     # would land on a freed block
        c.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert rel_err(c, want) < 2e-5
        del junk
    finally:
        L.aas_set_gemm_max_steps(prev)


def test_22bit_bptt_exchange_is_below_the_references_own_thread_count_spread(gpu):
    """fp32 headline: the reduce-scatter BPTT exchanges partial sums whose two low mantissa bits carry the step tag (a 22-bit
    exchange, <= 2 ulp per partial).  Is that visible at the gradient level?  The same F3b step runs twice: with the shipped
    kernels, and with the BPTT on the counter-based kernels of round 1 (kernel-selection bit 256: whole fp32 words exchanged, and a
    different summation order on top).  The difference between the two - an UPPER bound of the 22-bit effect - is compared, per
    parameter, with how far the reference's own fp32-CPU gradient samples move when only the CPU thread count changes (F3c: 8 / 3 /
    1 threads).  Both are measured against the same 64 samples per parameter, in units of the largest sample."""
    from aas_enhancement_amd import ops
    from aas_enhancement_amd.trainer_AAS import Trainer
    z, zc = load("f3b_aas_config2_kt.npz"), load("f3c_thread_spread.npz")
    grads = []
    for flags in (0, 256):
        tr = Trainer(cfg(lr=float(z["lr"]), nFeat=80, rnn_size=500, allow_ASR_update_iter=0), None, models=_config2_models())
        tr.kt = float(z["kt0"])
        tr.launch = ops.LaunchState(debug_flags=flags)
        ny, cl = _config2_batches(0)
        tr.train_step_async(ny, cl, 0)
        tr.read_scalars()
        g = {}
        for nm, m in (("G", tr.G), ("D", tr.D), ("A", tr.ASR)):
            for k, p in m.named_parameters():
                key = "%s.%s" % (nm, k)
                if k in NOISE_PARAMS or ("it0.gradsample_idx." + key) not in z.files:
                    continue
                idx = torch.from_numpy(z["it0.gradsample_idx." + key].astype(np.int64)).cuda()
                g[key] = p.grad.detach().reshape(-1)[idx].cpu().numpy()
        grads.append(g)
    assert not ops.rnn_timeout_flag()
    dev22, spread, vs_ref = {}, {}, {}
    for key in grads[0]:
        ref = z["it0.gradsample." + key]
        scale = float(np.abs(ref).max()) + 1e-30
        dev22[key] = float(np.abs(grads[0][key] - grads[1][key]).max()) / scale
        spread[key] = float(np.abs(zc["hi." + key] - zc["lo." + key]).max()) / scale
        vs_ref[key] = float(np.abs(grads[0][key] - ref).max()) / scale
    w22, wsp = max(dev22.values()), max(spread.values())
    m22, msp = float(np.median(list(dev22.values()))), float(np.median(list(spread.values())))
    print("shipped vs whole-word-exchange BPTT: worst %.2e median %.2e | reference across thread counts: worst %.2e median %.2e | shipped vs reference: worst %.2e"
          % (w22, m22, wsp, msp, max(vs_ref.values())))
    assert len(dev22) >= 70
    # no parameter's samples move more than the reference's own worst thread-count movement, and the typical movement is smaller too
    assert w22 <= wsp, (w22, wsp, max(dev22, key=dev22.get))
    assert m22 <= msp, (m22, msp)


# ---- managed exchange buffers (no poison memset launch in front of a persistent launch) ----------------------------------------------
def _layer_run(ops, kind, T, N, H, seed):
    g = torch.Generator().manual_seed(seed)
    G = 4 if kind == "lstm" else 3
    b = 1.0 / np.sqrt(H)
    # (one flat buffer, as the trainers' FlatBuffers lay the weights out: whether the two directions' products run as one launch
    #  depends on the distance between the weight tensors, which separate allocations would leave to the allocator)
    flat = ((torch.rand(4, G * H, H, generator=g) * 2 - 1) * b).cuda()
    w = [flat[i].detach().requires_grad_(True) for i in range(4)]
    x = (torch.randn(T, N, H, generator=g) * 0.5).cuda().requires_grad_(True)
    gy = torch.randn(T, N, H, generator=g).cuda()
    y = ops.birnn_layer(x, *w, kind=kind, residual=True)
    y.backward(gy)
    return [y.detach()] + [x.grad] + [wi.grad for wi in w]


def test_managed_exchange_buffers_equal_the_poison_fill_path_bit_for_bit(gpu):
    """A chain of persistent launches of changing kind / T / N / H queued back to back on one stream - the exchange buffer alternates
    between its halves, every kernel re-poisons what its predecessor dirtied, the fp32 forward kernels exchange h_t through the
    self-cleaning ring of four time slots - gives, bit for bit, what the same chain gives with a poison memset in front of every
    launch (knobs.MANAGED_XCHG off).  Includes T < 4 (no ring: falls back and is prepared again), T = 4 / 5 (the ring's wrap),
    ragged row groups (N = 7, 33), the 1000-unit GRU, and a kernel-selection bit that takes the whole buffer in between."""
    from aas_enhancement_amd import knobs, ops
    from aas_enhancement_amd._lib import lib
    shapes = [("lstm", 200, 30, 500), ("gru", 85, 30, 1000), ("lstm", 200, 60, 500), ("lstm", 5, 7, 500), ("gru", 4, 33, 512),
              ("lstm", 3, 30, 500), ("lstm", 64, 30, 500), ("gru", 85, 30, 1000), ("lstm", 9, 3, 16), ("lstm", 200, 30, 500)]
    res = {}
    for managed in (True, False):
        with knobs.override(MANAGED_XCHG=managed):
            out = []
            for rep in range(2):                      # twice: the second pass starts on buffers the first one left behind
                for i, (kind, T, N, H) in enumerate(shapes):
                    if managed and rep == 1 and i == 4:
                        lib().aas_set_debug_flags(256)        # all-gather / counter-based BPTT: takes the whole buffer (legacy fill)
                    out.append(_layer_run(ops, kind, T, N, H, 100 + i))
                    lib().aas_set_debug_flags(0)
            torch.cuda.synchronize()
            assert not ops.rnn_timeout_flag()
            res[managed] = out
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        if i == len(shapes) + 4:      # (the launch under the kernel-selection bit ran another BPTT kernel: same values to rounding)
            for ta, tb in zip(a, b):
                assert rel_err(ta, tb) < 1e-5
            continue
        kind, T, N, H = shapes[i % len(shapes)]
        for j, (ta, tb) in enumerate(zip(a, b)):
            if j == 1 and T * N < 64:    # dx of a tiny batch: the general GEMM's split-K epilogue adds with fp32 atomics (order varies run to run)
                assert rel_err(ta, tb) < 1e-5
                continue
            assert torch.equal(ta, tb), (i, j, shapes[i % len(shapes)])
    # the buffers of this stream are managed again after the fall-backs
    st = torch.cuda.current_stream().cuda_stream
    keys = [k for k in ops._scratch if k[0] in ("xchg_fwd", "xchg_bwd") and k[2] == st]
    assert len(keys) == 2 and all(lib().aas_rnn_xchg_is_managed(ops._scratch[k].data_ptr()) for k in keys)


# ---- the fused step-glue launches, op by op ------------------------------------------------------------------------------------------
def test_step_prologue_zeroes_exactly_its_buffers_and_writes_the_weights(gpu):
    from aas_enhancement_amd import ops
    big = torch.full((3 * 1024 * 1024 + 256,), 7.0, device="cuda")
    a = big[64:64 + 1000003]                      # 16-byte aligned start (64 floats in), odd length: bytewise tail
    b = big[2 * 1024 * 1024:2 * 1024 * 1024 + 4096]
    acc = torch.full((2,), 3.0, device="cuda", dtype=torch.float64)
    rs = torch.full((70,), 9.0, device="cuda")
    kt = torch.tensor([0.37], device="cuda", dtype=torch.float64)
    ops.step_prologue([a, b, acc], rs[:60], 30, 30, kt)
    torch.cuda.synchronize()
    assert float(a.abs().sum()) == 0.0 and float(b.abs().sum()) == 0.0 and float(acc.abs().sum()) == 0.0
    assert float(big[:64].min()) == 7.0 and float(big[64 + 1000003:2 * 1024 * 1024].min()) == 7.0     # neighbours untouched
    assert float(big[2 * 1024 * 1024 + 4096:].min()) == 7.0
    assert torch.equal(rs[:30], torch.full((30,), -float(np.float32(0.37)), device="cuda")) and torch.equal(rs[30:60], torch.ones(30, device="cuda"))
    assert float(rs[60:].min()) == 9.0
    ops.step_prologue([], rs[:8], 8, 0, kt)       # weights only
    ops.step_prologue([b])                        # buffers only
    torch.cuda.synchronize()
    assert torch.equal(rs[:8], torch.full((8,), -float(np.float32(0.37)), device="cuda"))


def test_fused_loss_roots_equal_the_autograd_composition(gpu):
    """ops.layout_cat_nct_tnc + ops.l1_pair + ops.ctc_scaled + ops.began_step_raw against what they replace (torch.cat + ops.layout,
    ops.l1_sum * scale on slices, ops.ctc_sum * scale, ops.began_step): same values and gradients."""
    from aas_enhancement_amd import ops
    g = torch.Generator().manual_seed(5)
    N, F, T = 5, 12, 23
    leaf0 = (torch.rand(N, F, T, generator=g) * 6).cuda()
    clean = (torch.rand(N, F, T, generator=g) * 6).cuda()
    W = (torch.randn(F, F, generator=g) * 0.3).cuda()
    s_ny, s_cl, s_ctc = 1.0 / 97.0, 1.0 / 89.0, 1.0 / N

    def net(x_tnc):                               # a stand-in for D: [T, 2N, F] -> [2N, F, T]
        return ops.layout(torch.tanh(x_tnc @ W), "tnc_nct")
    # fused
    leaf = leaf0.clone().requires_grad_(True)
    ae = net(ops.layout_cat_nct_tnc(leaf, clean))
    acc = torch.zeros(2, device="cuda", dtype=torch.float64)
    tg = []
    root = ops.l1_pair(ae, leaf, clean, s_ny, s_cl, acc, tg)
    torch.autograd.backward([root], [ops.unit_root(root)])
    g_fused = leaf.grad + tg[0]
    # unfused
    leaf2 = leaf0.clone().requires_grad_(True)
    ae2 = net(ops.layout(torch.cat([leaf2, clean], 0), "nct_tnc"))
    l_ny = ops.l1_sum(ae2[:N], leaf2) * s_ny
    l_cl = ops.l1_sum(ae2[N:], clean) * s_cl
    (l_ny + l_cl).backward()
    assert rel_err(ae, ae2) < 1e-6
    assert float(acc[0]) * s_ny == pytest.approx(float(l_ny), rel=1e-6) and float(acc[1]) * s_cl == pytest.approx(float(l_cl), rel=1e-6)
    assert rel_err(g_fused, leaf2.grad) < 1e-5
    # CTC
    Tp, C, L = 19, 29, 4
    acts0 = torch.randn(Tp, N, C, generator=g).cuda()
    labels = torch.randint(1, C, (N * L,), generator=g).int()
    meta = ops.ctc_prepare(labels, torch.full((N,), Tp, dtype=torch.int32), torch.full((N,), L, dtype=torch.int32), torch.device("cuda"))
    a1 = acts0.clone().requires_grad_(True)
    costs = ops.ctc_scaled(a1, 0, meta, s_ctc)
    torch.autograd.backward([costs], [ops.unit_root(costs)])
    a2 = acts0.clone().requires_grad_(True)
    l_ctc = ops.ctc_sum(a2, None, None, None, 0, meta) * s_ctc
    l_ctc.backward()
    assert float(costs.sum()) * s_ctc == pytest.approx(float(l_ctc), rel=1e-6)
    assert rel_err(a1.grad, a2.grad) < 1e-6
    # controller
    kt1, kt2 = torch.tensor([0.3], device="cuda", dtype=torch.float64), torch.tensor([0.3], device="cuda", dtype=torch.float64)
    o1, o2 = torch.zeros(6, device="cuda", dtype=torch.float64), torch.zeros(6, device="cuda", dtype=torch.float64)
    ops.began_step_raw(acc, s_ny, s_cl, costs.detach(), s_ctc, kt1, o1, 0.5, 0.001, float(N))
    ops.began_step(l_ny, l_cl, l_ctc, kt2, o2, 0.5, 0.001, float(N))
    assert torch.allclose(o1, o2, rtol=1e-6) and float(kt1) == pytest.approx(float(kt2), rel=1e-9)
