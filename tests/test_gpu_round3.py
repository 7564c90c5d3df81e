"""GPU parity tests (-m gpu) added in round 3: the device-resident FSEGAN step against the reference goldens and the
synchronous step, SGD-with-Nesterov-momentum (AM_training/train.py:172-174) against torch.optim.SGD, data parallelism for the
DCE / FSEGAN trainers, and `Trainer.train()` data parallel with a sharding loader and --sync_bn across a save_iter."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn

from tests.helpers import LABELS, load, load_sd
from tests.test_gpu_step import cfg

pytestmark = pytest.mark.gpu
REL_OUT, REL_LOSS = 1e-3, 1e-2


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _fill(m, seed, conv_std=None):
    from aas_enhancement_amd import prng
    load_sd(m, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(m.state_dict(), seed, conv_std=conv_std).items()}, strict=False)
    return m


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


# ------------------------------------------------------------------------------------------------ FSEGAN, device-resident
def test_fsegan_async_config4_golden(gpu, precision2):
    """F6 (config 4 at size) through train_step_async + read_scalars: scalars, kt trajectory and enhanced samples of the
    reference-module step, two iterations (the second one sees the kt the device advanced)."""
    from aas_enhancement_amd import ops, prng
    from aas_enhancement_amd.model import stackedBRNN
    from aas_enhancement_amd.trainer_FSEGAN import Trainer
    z = load("f6_fsegan_config4.npz")
    N, F, T, H = [int(z[k]) for k in ("N", "F", "T", "H")]
    G, D = _fill(stackedBRNN(I=F, O=F, H=H, L=4), int(z["weight_seed_G"])), _fill(stackedBRNN(I=2 * F, O=F, H=H, L=4), int(z["weight_seed_D"]))
    tr = Trainer(cfg(lr=float(z["lr"]), w_adversarial=float(z["w_adversarial"]), nFeat=F, rnn_size=H), None, models=(G, D))
    tr.kt = float(z["kt0"])
    outs = []
    for it in range(2):       # both queued before anything is read back
        mix = torch.from_numpy(prng.uniform(int(z["mixture_seed0"]) + it, (N, F, T), 0.0, 6.0))
        cln = torch.from_numpy(prng.uniform(int(z["clean_seed0"]) + it, (N, F, T), 0.0, 6.0))
        mask = torch.zeros(N, 1, T, dtype=torch.uint8)
        r = tr.train_step_async((mix, cln, mask), it)
        outs.append((r["enhanced"], r["scalars"].clone()))
    sc = tr.read_scalars()
    for it, (enh, s) in enumerate(outs):
        got = dict(zip(("l_adv_ny_G", "l_adv_cl", "dce", "kt"), s[:4].tolist()))
        for k, v in got.items():
            assert v == pytest.approx(float(z["it%d.%s" % (it, k)]), rel=REL_LOSS), (it, k)
        e = enh.detach().reshape(-1)
        g = e[torch.from_numpy(z["it%d.enh_idx" % it]).cuda()].cpu().numpy()
        ref = z["it%d.enh_samples" % it]
        assert np.abs(g - ref).max() < REL_OUT * np.abs(ref).max(), it
    assert sc["kt"] == pytest.approx(float(z["it1.kt"]), rel=REL_LOSS) and tr.kt == sc["kt"]
    assert not ops.rnn_timeout_flag()


def _tiny_fsegan(as_written=False):
    from aas_enhancement_amd.model import stackedBRNN
    from aas_enhancement_amd.trainer_FSEGAN import Trainer
    G, D = _fill(stackedBRNN(I=8, O=8, H=16, L=2), 31), _fill(stackedBRNN(I=16, O=8, H=16, L=2), 32)
    tr = Trainer(cfg(lr=1e-3, w_adversarial=0.3, nFeat=8, rnn_size=16, fsegan_as_written=as_written), None, models=(G, D))
    tr.kt = 0.2
    return tr


def _tiny_paired(seed, n=4, T=40, lens=None):
    from aas_enhancement_amd import prng
    lens = lens or [T] * n
    x = torch.from_numpy(prng.uniform(seed, (n, 8, T), 0.0, 6.0))
    y = torch.from_numpy(prng.uniform(seed + 1, (n, 8, T), 0.0, 6.0))
    mask = torch.zeros(n, 1, T, dtype=torch.uint8)
    for i, L in enumerate(lens):
        x[i, :, L:] = 0; y[i, :, L:] = 0; mask[i, 0, L:] = 1
    return x, y, mask


def test_fsegan_async_and_sync_steps_interchange(gpu, precision2):
    """async, async, sync, async == sync x 4 (host / device kt and the Adam step counters stay in step), ragged lengths."""
    ref, mix = _tiny_fsegan(), _tiny_fsegan()
    want, got = [], []
    for it in range(4):
        b = _tiny_paired(700 + 2 * it, lens=[40, 33, 25, 25])
        r = ref.train_step(b, it)
        want.append([r[k] for k in ("l_adv_ny_G", "l_adv_cl", "dce", "kt")])
        if it == 2:
            r2 = mix.train_step(b, it)
        else:
            mix.train_step_async(b, it)
            r2 = mix.read_scalars() if it in (1, 3) else None
        if r2 is not None:
            got.append((it, [r2[k] for k in ("l_adv_ny_G", "l_adv_cl", "dce", "kt")]))
    for it, g in got:
        assert np.allclose(g, want[it], rtol=2e-4), (it, g, want[it])
    for k in ("G", "D"):
        a, b_ = ref._flat[k].flat_p, mix._flat[k].flat_p
        assert float((a - b_).abs().max()) < 5e-5


# ------------------------------------------------------------------------------------------------ SGD + Nesterov momentum
def test_flat_sgd_nesterov_vs_torch(gpu):
    from aas_enhancement_amd.dist import FlatBuffers
    from aas_enhancement_amd.optim import FlatSGD
    torch.manual_seed(0)
    mod = nn.ParameterList([nn.Parameter(torch.randn(37, 5)), nn.Parameter(torch.randn(11))]).cuda()
    ref = [p.detach().clone().requires_grad_(True) for p in mod]
    opt_ref = torch.optim.SGD(ref, lr=0.05, momentum=0.9, nesterov=True)
    flat = FlatBuffers(mod)
    opt = FlatSGD(flat, lr=0.05, momentum=0.9)
    for step in range(4):
        gs = [torch.randn_like(p) for p in ref]
        for p, r, g in zip(mod, ref, gs):
            p.grad.copy_(g)
            r.grad = g.clone()
        opt.step() if step % 2 == 0 else opt.step_dev()
        opt_ref.step()
        for p, r in zip(mod, ref):
            assert float((p.detach() - r.detach()).abs().max()) < 1e-6, step
    sd = opt.state_dict()
    assert sd["param_groups"][0]["nesterov"] and sd["param_groups"][0]["momentum"] == 0.9
    assert torch.allclose(sd["state"][0]["momentum_buffer"], opt_ref.state_dict()["state"][0]["momentum_buffer"], atol=1e-6)
    opt2 = FlatSGD(flat, lr=0.01, momentum=0.5)
    opt2.load_state_dict(sd)
    assert opt2.lr == 0.05 and opt2.momentum == 0.9 and torch.equal(opt2.buf, opt.buf)


def test_am_trainer_sgd_matches_oracle_step(gpu, precision2):
    """AMTrainer(optim='sgd') (train.py:172-174: SGD(momentum, nesterov=True)) against the CPU oracle model under torch.optim.SGD."""
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.am_train import AMTrainer
    from aas_enhancement_amd.model import DeepSpeech
    from oracle import ref_model as RM
    from oracle import ref_step as RS
    A = _fill(DeepSpeech(nn.GRU, LABELS, 12, 3, True, 11, 2, 8, 2, nFreq=8), 51, 0.1)
    R = RM.RefDeepSpeech(nn.GRU, LABELS, 12, 3, 11, 2, 8, 2, nFreq=8)
    load_sd(R, A.state_dict(), strict=False)
    tr = AMTrainer(A.cuda(), lr=1e-2, optim="sgd", momentum=0.9)
    opt = torch.optim.SGD(R.parameters(), lr=1e-2, momentum=0.9, nesterov=True)
    N, T, L = 3, 60, 3
    for it in range(3):
        b = (torch.from_numpy(prng.uniform(60 + it, (N, 8, T), 0.0, 6.0)), torch.from_numpy(prng.randint(70 + it, (N * L,), 1, 28).astype(np.int32)),
             torch.ones(N), torch.full((N,), L, dtype=torch.int32))
        r = tr.train_step(b) if it != 1 else None
        if it == 1:
            h = tr.train_step_async(b)
            r = dict(loss=tr.read_loss(h["handle"])[0])
        ref = RS.am_step(R, opt, b)
        assert r["loss"] == pytest.approx(float(ref["loss"]), rel=1e-3), it
    sd_r = R.state_dict()
    for k, v in A.state_dict().items():
        if "running" in k or "num_batches" in k or k in ("conv.0.bias", "conv.3.bias"):
            continue
        assert float((v.cpu() - sd_r[k]).abs().max()) < 2e-3 * float(sd_r[k].abs().max()) + 1e-5, k


# ------------------------------------------------------------------------------------------------ DP: DCE / FSEGAN trainers
def _dp_env(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=180))
    torch.cuda.set_device(0)
    return dist


def _paired_shard(dp, b):
    idx = torch.tensor(dp.shard_rows(b[0].size(0)))
    return tuple(t.index_select(0, idx) for t in b)


def _pair_worker(rank, world, port, q, which):
    dist = _dp_env(rank, world, port)
    try:
        from aas_enhancement_amd.dist import DPContext
        out = []
        if which == "dce":
            from aas_enhancement_amd.model import stackedBRNN
            from aas_enhancement_amd.trainer_DCE import Trainer
            tr = Trainer(cfg(lr=1e-3, nFeat=8, rnn_size=16), None, models=(_fill(stackedBRNN(I=8, H=16, L=2), 31),))
            tr.make_optimizers()
            for it in range(2):
                r = tr.train_step(_paired_shard(tr.dp, _tiny_paired(800 + 2 * it, lens=[40, 36, 30, 22])), it)
                out.append([float(r["dce"])])
            flats = [tr._flat.flat_p]
        else:
            tr = _tiny_fsegan()
            tr.make_optimizers()
            for it in range(3):
                b = _paired_shard(tr.dp, _tiny_paired(800 + 2 * it, lens=[40, 36, 30, 22]))
                if which == "fsegan_sync" or it == 1:
                    r = tr.train_step(b, it)
                else:
                    tr.train_step_async(b, it)
                    r = tr.read_scalars()
                out.append([r[k] for k in ("l_adv_ny_G", "l_adv_cl", "dce", "kt")])
            flats = [tr._flat["G"].flat_p, tr._flat["D"].flat_p]
        assert isinstance(tr.dp, DPContext) and tr.dp.active
        q.put((rank, np.asarray(out), [f.detach().cpu().numpy() for f in flats]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which", ["dce", "fsegan_sync", "fsegan_async"])
def test_dce_and_fsegan_trainers_dp_two_ranks_equal_single(gpu, which, precision2):
    """2 ranks x 2 utterances (global nElement, SUM all-reduce of the flat gradient buffers, all-reduced kt inputs) == 1 rank x 4."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pair_worker, args=(r, 2, port, q, which)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = []
    if which == "dce":
        from aas_enhancement_amd.model import stackedBRNN
        from aas_enhancement_amd.trainer_DCE import Trainer
        tr = Trainer(cfg(lr=1e-3, nFeat=8, rnn_size=16), None, models=(_fill(stackedBRNN(I=8, H=16, L=2), 31),))
        for it in range(2):
            ref.append([float(tr.train_step(_tiny_paired(800 + 2 * it, lens=[40, 36, 30, 22]), it)["dce"])])
        flats = [tr._flat.flat_p]
    else:
        tr = _tiny_fsegan()
        for it in range(3):
            r = tr.train_step(_tiny_paired(800 + 2 * it, lens=[40, 36, 30, 22]), it)
            ref.append([r[k] for k in ("l_adv_ny_G", "l_adv_cl", "dce", "kt")])
        flats = [tr._flat["G"].flat_p, tr._flat["D"].flat_p]
    ref = np.asarray(ref)
    for rank, out, pars in res:
        assert np.allclose(out, ref, rtol=3e-4), (rank, out, ref)
        for a, b in zip(pars, flats):
            d_ = np.abs(a - b.detach().cpu().numpy())
            assert float((d_ > 2e-4).mean()) < 2e-3 and float(d_.max()) < 6.1e-3, rank    # (Adam sign flips of noise-level gradients)
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a, b)


# ------------------------------------------------------------------------------------------------ Trainer.train() data parallel
def _manifests(tmp):
    from aas_enhancement_amd import prng
    lens = [36, 40, 44, 47, 50, 52, 55, 60]                   # ascending (make_manifest_librispeech.py:58-59)
    rows = []
    for i, T in enumerate(lens):
        torch.save(torch.from_numpy(prng.uniform(900 + i, (8, T), 0.0, 6.0)), os.path.join(tmp, "f%d.pt7" % i))
        open(os.path.join(tmp, "t%d.txt" % i), "w").write("ab cd"[: 2 + i % 3])
        rows.append("%s,%s" % (os.path.join(tmp, "f%d.pt7" % i), os.path.join(tmp, "t%d.txt" % i)))
    for name in ("ny.csv", "cl.csv", "val.csv"):
        open(os.path.join(tmp, name), "w").write("\n".join(rows if name != "val.csv" else rows[:3]) + "\n")
    return tmp


def _aas_models():
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    return (_fill(stackedBRNN(I=8, H=16, L=2), 11), _fill(stackedBRNN(I=8, H=16, L=2), 12),
            _fill(DeepSpeech(nn.GRU, LABELS, 12, 2, True, 11, 2, 8, 2, nFreq=8), 13, 0.1))


def _train_cfg(tmp, expnum):
    return cfg(lr=1e-3, nFeat=8, rnn_size=16, batch_size=4, allow_ASR_update_iter=0, sync_bn=True, start_iter=0, max_iter=4, log_iter=2, save_iter=2,
               expnum=expnum, write_log=False, w_adversarial=1.0, w_acoustic=1.0)


def _run_train(tmp, expnum, dp):
    from aas_enhancement_amd.data_loader import DataLoader
    from aas_enhancement_amd.trainer_AAS import Trainer
    np.random.seed(3)
    dl = DataLoader(batch_size=4, tr_ny_manifest=os.path.join(tmp, "ny.csv"), tr_cl_manifest=os.path.join(tmp, "cl.csv"),
                    trsub_manifest=os.path.join(tmp, "val.csv"), val_manifest=os.path.join(tmp, "val.csv"), labels=LABELS, num_workers=0,
                    pin_memory=True, dp=dp)
    tr = Trainer(_train_cfg(tmp, expnum), dl, models=_aas_models())
    tr.model_dir = os.path.join(tmp, "logs%d" % expnum)
    tr.train()
    torch.cuda.synchronize()
    return tr


def _train_worker(rank, world, port, tmp, q):
    dist = _dp_env(rank, world, port)
    try:
        from aas_enhancement_amd.dist import DPContext
        tr = _run_train(tmp, 70, DPContext.from_env())
        bufs = np.concatenate([b.detach().double().cpu().numpy().reshape(-1) for b in tr.ASR.buffers()])
        q.put((rank, tr._flat["G"].flat_p.detach().cpu().numpy(), tr._flat["A"].flat_p.detach().cpu().numpy(), float(tr.kt),
               sorted(os.listdir(tr.model_dir)) if os.path.isdir(tr.model_dir) else [], bufs))
    finally:
        dist.destroy_process_group()


def test_train_loop_dp_with_sharding_loader_and_syncbn_across_save_iter(gpu, tmp_path, precision2):
    """rank 0 alone validates at a save_iter while A stays in train mode; with --sync_bn armed its BatchNorm
    would issue all-reduces no other rank matches.  Two ranks, a loader that shards before loading, sync_bn on, two save_iters
    inside four iterations: the run finishes, only rank 0 writes checkpoints, every rank ends with identical parameters, and
    they equal the single-process run on the same global batches."""
    tmp = _manifests(str(tmp_path))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, tmp, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=420) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2]) and res[0][3] == res[1][3]
    # replicas INCLUDING buffers: rank 0's validation moved A's BatchNorm running statistics, the broadcast behind it levels them
    assert np.array_equal(res[0][5], res[1][5])
    assert any(f.startswith("G_3") for f in res[0][4]) and any(f.startswith("ASR_3") for f in res[0][4])
    from aas_enhancement_amd import ops
    single = _run_train(tmp, 71, None)
    assert single.launch.sync_bn is None and ops.state().sync_bn is None
    for got, want in ((res[0][1], single._flat["G"].flat_p), (res[0][2], single._flat["A"].flat_p)):
        d_ = np.abs(got - want.detach().cpu().numpy())
        assert float((d_ > 2e-4).mean()) < 5e-3 and float(d_.max()) < 8.1e-3
    assert res[0][3] == pytest.approx(float(single.kt), rel=1e-3, abs=1e-6)


# ------------------------------------------------------------------------------------------------ exact-fp32 recurrent kernels
@pytest.mark.parametrize("half_chip", [True, False])
@pytest.mark.parametrize("kind,T,N,H", [("lstm", 60, 30, 500), ("lstm", 60, 60, 500), ("gru", 40, 30, 1000), ("lstm", 25, 30, 64), ("gru", 9, 5, 24),
                                        ("lstm", 20, 4, 200), ("gru", 20, 7, 256), ("gru", 12, 3, 500), ("lstm", 15, 13, 384)])
def test_exact_fp32_flag_kernels_vs_counter_kernels_and_xcd_variants(gpu, kind, T, N, H, half_chip):
    """aas_set_precision(0) runs the data-is-the-flag kernels (all-gather forward, reduce-scatter BPTT) with fp32-input MFMA on fp32
    exchange words.  Against the counter-based fp32 kernels of round 1 (debug bit 134217728: same products, another summation
    order) they agree to fp32 rounding; their XCD-aware launch variants (262144 plain grid, 524288 write-through stores) are bit
    identical; no timeout.  Config-2 layer shapes on half-chip and whole-chip grids, plus shapes only the fallback covers.  Row groups of
    <= 8 rows run the 4 x 4 x 1 block form of the product (forward: one / two row blocks, LSTM and GRU; LSTM BPTT): it agrees with the
    16 x 16 x 4 tile form (debug bit 268435456) to fp32 rounding - same products, k summed in four interleaved chains."""
    from aas_enhancement_amd import _lib, ops
    L = _lib.lib()
    G = 4 if kind == "lstm" else 3
    dev = "cuda"
    g = torch.Generator().manual_seed(5)
    R = lambda *shape, scale=1.0: (torch.randn(*shape, generator=g) * scale).to(dev)
    ops.set_precision(0)
    L.aas_set_rnn_cu_limit(ops.device_cus() // 2 if half_chip else 0)
    try:
        w = [R(G * H, H, scale=1.0 / H ** 0.5) for _ in range(4)]
        pre = R(T, N, 2, G * H)
        dy = R(T, N, H)
        sync, xc = ops._sync_buf(torch.device(dev, 0)), ops._xchg_buf(torch.device(dev, 0), T, N, H, G)
        s, p = _lib.stream(), _lib.ptr
        res = {}
        for fl in (134217728, 268435456, 262144, 524288, 0):
            L.aas_set_debug_flags(fl)
            hout = torch.zeros(2, T, N, H, device=dev)
            gact = torch.zeros(2, T, N, H, 4, device=dev)
            cst = torch.zeros(2, T, N, H, device=dev)
            dgx = torch.zeros(T, N, 2, G * H, device=dev)
            dgh = torch.zeros(T, N, 2, G * H, device=dev)
            if kind == "lstm":
                ops.check(L.aas_lstm_fwd(s, T, N, H, p(pre), p(w[1]), p(w[3]), p(hout), p(gact), p(cst), p(sync), p(xc)), "fwd")
                ops.check(L.aas_lstm_bwd(s, T, N, H, p(dy), p(w[1]), p(w[3]), p(gact), p(cst), p(dgx), p(sync), p(xc)), "bwd")
            else:
                ops.check(L.aas_gru_fwd(s, T, N, H, p(pre), p(w[1]), p(w[3]), p(hout), p(gact), p(sync), p(xc)), "fwd")
                ops.check(L.aas_gru_bwd(s, T, N, H, p(dy), p(w[1]), p(w[3]), p(hout), p(gact), p(dgx), p(dgh), p(sync), p(xc)), "bwd")
            torch.cuda.synchronize()
            assert not ops.rnn_timeout_flag()
            res[fl] = [t.clone() for t in (hout, gact, cst, dgx, dgh)]
        for fl in (524288, 0):
            for a, b in zip(res[fl], res[262144]):
                assert torch.equal(a, b), (kind, fl)
        for other in (134217728, 268435456):         # summation order differs (K split over waves / producers / chains): fp32 rounding only
            for a, b in zip(res[0], res[other]):
                assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7, (kind, other)
        assert torch.isfinite(res[0][3]).all() and res[0][3].abs().max() > 0
    finally:
        L.aas_set_debug_flags(0)
        L.aas_set_rnn_cu_limit(0)
        ops.set_precision(int(os.environ.get("AAS_PRECISION", "0")))


# ------------------------------------------------------------------------------------------------ vanilla `rnn` kind

@pytest.mark.parametrize("tag", ["s", "m", "l"])
def test_brnn_vanilla_rnn_golden(gpu, precision, tag):
    """F9: the reference's BRNN with nn.RNN (supported_rnns['rnn'], model.py:12-17,88-105): forward, input gradient and all four
    weight gradients; `l` is the enhancer's layer shape (T=60, N=30, H=500) checked on samples and norms."""
    from aas_enhancement_amd import ops, prng
    from aas_enhancement_amd.model import BRNN
    from tests.helpers import rel_err, sub
    z = load("f9_rnn_kind.npz")
    p = "brnn_rnn_%s." % tag
    ft, gt = (1e-5, 1e-4) if precision != 1 else (1e-4, 5e-4)
    if tag != "l":
        H = z[p + "x"].shape[2]
        m = BRNN(H, H, nn.RNN, bidirectional=True)
        load_sd(m, sub(z, p + "w."))
        m.cuda()
        x = torch.from_numpy(z[p + "x"]).cuda().requires_grad_(True)
        y = m(x)
        y.backward(torch.from_numpy(z[p + "gy"]).cuda())
        assert rel_err(y, z[p + "y"]) < ft and rel_err(x.grad, z[p + "gx"]) < gt
        for k, v in m.named_parameters():
            assert rel_err(v.grad, z[p + "gw." + k]) < gt, k
    else:
        T, N, H = [int(v) for v in z[p + "dims"]]
        s_w, s_x, s_g = [int(v) for v in z[p + "seeds"]]
        m = _fill(BRNN(H, H, nn.RNN, bidirectional=True), s_w).cuda()
        x = torch.from_numpy(prng.normal(s_x, (T, N, H), 0.0, 0.5)).cuda().requires_grad_(True)
        y = m(x)
        y.backward(torch.from_numpy(prng.normal(s_g, (T, N, H))).cuda())
        idx = torch.from_numpy(z[p + "idx"]).cuda()
        for got, key, tol in ((y.detach(), "y", ft), (x.grad, "gx", gt)):
            ref = z[p + key + "_samples"]
            assert np.abs(got.reshape(-1)[idx].cpu().numpy() - ref).max() < 4 * tol * np.abs(ref).max(), key
            assert float(got.double().norm()) == pytest.approx(float(z[p + key + "_norm"]), rel=4 * tol), key
        for k, v in m.named_parameters():
            assert float(v.grad.double().norm()) == pytest.approx(float(z[p + "gw_norm." + k]), rel=4 * gt), k
            ref = z[p + "gw_samples." + k]
            got = v.grad.reshape(-1)[idx % v.numel()].cpu().numpy()
            assert np.abs(got - ref).max() < 4 * gt * np.abs(ref).max() + 1e-7, k
    assert not ops.rnn_timeout_flag()


def test_stacked_brnn_vanilla_rnn_golden_and_dce_step(gpu, precision):
    """stackedBRNN(rnn_type=nn.RNN) (model.py:203-231) forward / backward against the reference module, then a DCE step with
    `--rnn_type rnn` through the trainer (flat buffers, side-stream weight gradients)."""
    from aas_enhancement_amd.model import stackedBRNN, supported_rnns
    from aas_enhancement_amd.trainer_DCE import Trainer
    from tests.helpers import rel_err, sub
    z = load("f9_rnn_kind.npz")
    p = "stacked_rnn."
    G = stackedBRNN(I=6, O=6, H=10, L=4, rnn_type=nn.RNN)
    load_sd(G, sub(z, p + "sd."))
    G.cuda()
    x = torch.from_numpy(z[p + "x"]).cuda().requires_grad_(True)
    y = G(x)
    y.backward(torch.from_numpy(z[p + "gy"]).cuda())
    ft, gt = (2e-5, 2e-4) if precision != 1 else (2e-4, 1e-3)
    assert rel_err(y, z[p + "y"]) < ft and rel_err(x.grad, z[p + "gx"]) < gt
    for k, v in G.named_parameters():
        assert rel_err(v.grad, z[p + "gw." + k]) < gt, k
    tr = Trainer(cfg(lr=1e-3, nFeat=8, rnn_size=16, rnn_type="rnn", rnn_layers=2), None, models=(_fill(stackedBRNN(I=8, H=16, L=2, rnn_type=supported_rnns["rnn"]), 77),))
    losses = [float(tr.train_step(_tiny_paired(880, lens=[40, 36, 30, 22]), it)["dce"]) for it in range(6)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_gemm_tn_rowscaled_fp32(gpu):
    """aas_gemm_tn_rowscaled_f32: C (+)= sum_r s[r % nb] A[r]^T B[r] - the fp32 weight-gradient product with the per-utterance BEGAN
    weights applied while the reduction rows are staged (no scaling pass over d(gates)); strided A, row offsets, accumulate."""
    from aas_enhancement_amd import _lib, ops
    L = _lib.lib()
    ops.set_precision(0)
    try:
        g = torch.Generator().manual_seed(9)
        T, nb, M, N = 37, 12, 200, 96
        K = T * nb
        A = torch.randn(K, 2 * M, generator=g).cuda()          # lda = 2M: the product takes the second column half
        B = torch.randn(K, N, generator=g).cuda()
        sc = torch.randn(nb, generator=g).cuda()
        C0 = torch.randn(M, N, generator=g).cuda()
        C = C0.clone()
        ops.check(L.aas_gemm_tn_rowscaled_f32(_lib.stream(), M, N, K - nb, A.data_ptr() + 4 * (nb * 2 * M + M), 2 * M, B.data_ptr(), N,
                                              C.data_ptr(), N, 1, sc.data_ptr(), nb), "tn_rowscaled")
        w = sc.double().repeat(T - 1)[:, None]
        ref = C0.double() + ((A[nb:, M:].double() * w).t() @ B[:K - nb].double())
        assert float((C.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
        ops.set_precision(1)
        assert L.aas_gemm_tn_rowscaled_f32(_lib.stream(), M, N, K, A.data_ptr(), 2 * M, B.data_ptr(), N, C.data_ptr(), N, 0, sc.data_ptr(), nb) != 0
    finally:
        ops.set_precision(int(os.environ.get("AAS_PRECISION", "0")))


# ------------------------------------------------------------------------------------------------ fp32-equivalent six-product GEMM
@pytest.mark.parametrize("dist", ["randn", "positive", "wide"])
@pytest.mark.parametrize("K", [500, 1000, 6016])
def test_six_product_gemm_error_not_above_the_fp32_gemms(gpu, K, dist):
    """aas_set_precision(2): three-term operands x = h + m + l (exact), six bf16 products: the three-product plane kernel run over
    the (m | h) and (h | l) plane sets laid side by side as one k extent.  Against fp64 its error - scaled by sum_k |a||b|, maximum and rms - is NOT ABOVE
    the fp32-input MFMA GEMM's own error on the same operands, at the K of the enhancer's layers (500), the acoustic model's (1000)
    and a weight-gradient reduction (6016 rows), on N(0,1) data, on all-positive data (no cancellation: accumulation rounding
    dominates) and on data spread over 2^+-12 with random signs.  The three-product fast mode is shown to be 2-15x worse on the same
    data - the reason it is not the headline."""
    from aas_enhancement_amd import ops
    g = torch.Generator().manual_seed(K)
    M, N = 384, 256
    if dist == "randn":
        A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    elif dist == "positive":
        A, B = torch.rand(M, K, generator=g) + 0.5, torch.rand(N, K, generator=g) + 0.5
    else:
        A = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-12, 12, (M, K), generator=g).float())
        B = torch.randn(N, K, generator=g) * torch.exp2(torch.randint(-12, 12, (N, K), generator=g).float())
    A, B = A.cuda(), B.cuda()
    ref = A.double() @ B.double().t()
    scale = A.double().abs() @ B.double().abs().t()
    try:
        ops.set_precision(0)
        C32 = torch.empty(M, N, device="cuda")
        ops.gemm(ops.NT, M, N, K, A, K, B, K, C32, N)
        A3, B3 = ops.split_planes3(A, M, K), ops.split_planes3(B, N, K)
        assert torch.equal(A3.to_float()[:, :K], A) and torch.equal(B3.to_float()[:, :K], B)      # h + m + l == x, bit for bit
        assert float(A3.to_float()[:, K:].abs().sum()) == 0.0
        C6 = torch.empty(M, N, device="cuda")
        ops.gemm_planes6(M, N, A3.Kp, A3, B3, C6, N)
        A2, B2 = ops.split_planes(A, M, K), ops.split_planes(B, N, K)
        C3 = torch.empty(M, N, device="cuda")
        ops.gemm_planes(M, N, A2.Kp, A2, B2, C3, N)
        torch.cuda.synchronize()
    finally:
        ops.set_precision(int(os.environ.get("AAS_PRECISION", "0")))
    e32, e6, e3 = [((c.double() - ref).abs() / scale) for c in (C32, C6, C3)]
    # (both kernels split K over workgroups and add the partial sums with atomics in arrival order, so on accumulation-dominated
    # data their errors move by ~10 % from run to run - tools/r03_x6pos.py: max 3.9-4.6e-7 vs 3.9-4.9e-7; hence the margins)
    assert float(e6.max()) <= 1.25 * float(e32.max()), (float(e6.max()), float(e32.max()))
    assert float(e6.pow(2).mean().sqrt()) <= 1.03 * float(e32.pow(2).mean().sqrt())
    assert float(e6.max()) < 2e-6                                     # a few units of fp32's 2^-24 at most
    if dist != "positive":
        assert float(e3.pow(2).mean().sqrt()) > 2.0 * float(e32.pow(2).mean().sqrt())    # the fast mode IS narrower


@pytest.mark.parametrize("kind,T,N,H,classes", [("lstm", 200, 30, 500, 1), ("lstm", 64, 60, 500, 2), ("gru", 85, 30, 1000, 1), ("lstm", 9, 3, 16, 1)])
def test_birnn_layer_fp32_equivalent_mode_vs_cpu(gpu, kind, T, N, H, classes):
    """A recurrent layer in the fp32-equivalent mode (fp32 recurrent kernels + six-product projections / input gradients / row-major
    weight gradients incl. two utterance classes) against torch's CPU nn.LSTM / nn.GRU, held to the fp32 mode's tolerances."""
    from aas_enhancement_amd import ops
    from aas_enhancement_amd.dist import FlatBuffers
    from tests.helpers import rel_err
    torch.manual_seed(0)
    ref = (nn.LSTM if kind == "lstm" else nn.GRU)(H, H, bidirectional=True, bias=False)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(T, N, H, generator=g) * 0.5
    gy = torch.randn(T, N, H, generator=g)
    names = ("weight_ih_l0", "weight_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse")
    holder = nn.ParameterList([nn.Parameter(getattr(ref, k).detach().clone()) for k in names]).cuda()
    flat = FlatBuffers(holder)        # flat-buffer gradients: the side-stream accumulate path the trainers use
    w = list(holder)
    rs = None
    scale = torch.ones(N)
    if classes == 2:
        w_ = torch.empty(N, device="cuda")
        w_[:N // 2] = -0.37
        w_[N // 2:] = 1.0
        rs = ops.RowWeights(w_, classes=[(0, N // 2, w_[0:1]), (N // 2, N - N // 2, None)])
        scale[:N // 2] = -0.37
    try:
        ops.set_precision(2)
        xg = x.clone().cuda().requires_grad_(True)
        yg = ops.birnn_layer(xg, *w, kind=kind, residual=True, rs=rs)
        yg.backward(gy.cuda())
        ops.sync_wgrad()
        torch.cuda.synchronize()
    finally:
        ops.set_precision(int(os.environ.get("AAS_PRECISION", "0")))
    assert not ops.rnn_timeout_flag()
    xr = x.clone().requires_grad_(True)
    yr, _ = ref(xr)
    yr = yr[..., :H] + yr[..., H:] + xr
    yr.backward(gy)
    assert rel_err(yg, yr) < 2e-5 and rel_err(xg.grad, xr.grad) < 2e-4
    if classes == 2:     # parameter gradients with per-utterance weights: a second reference backward with the weighted output gradient
        for p_ in ref.parameters():
            p_.grad = None
        xr2 = x.clone().requires_grad_(True)
        y2, _ = ref(xr2)
        (y2[..., :H] + y2[..., H:] + xr2).backward(gy)       # = same graph; weights applied per utterance below via a hook-free trick:
        # d(params) is linear in the per-utterance output gradients, so run one backward per class and combine
        tot = {k: torch.zeros_like(getattr(ref, k)) for k in names}
        for lo, hi, wgt in ((0, N // 2, -0.37), (N // 2, N, 1.0)):
            for p_ in ref.parameters():
                p_.grad = None
            xs = x.clone().requires_grad_(True)
            ys, _ = ref(xs)
            gm = torch.zeros_like(gy)
            gm[:, lo:hi] = gy[:, lo:hi]
            # the residual path does not touch the parameters; the recurrent gradient of an utterance depends on its own rows only
            (ys[..., :H] + ys[..., H:]).backward(gm)
            for k in names:
                tot[k] += wgt * getattr(ref, k).grad
        for wg, k in zip(w, names):
            assert rel_err(wg.grad, tot[k]) < 3e-4, k
    else:
        for wg, k in zip(w, names):
            assert rel_err(wg.grad, getattr(ref, k).grad) < 2e-4, k


@pytest.mark.parametrize("N", [30, 60])
def test_six_product_bptt_layer_error_vs_fp64_at_the_fp32_modes_level(gpu, N):
    """aas_set_precision(2) runs the 500-unit LSTM's BPTT with SIX bf16 products of three-term operands (rnn_bwd_rs_kernel<.., X6>:
    d(gates) = h + m + l and W = h' + m' + l' exactly, dropped cross terms <= 2^-25) instead of fp32-input MFMA.  Against an fp64
    nn.LSTM (errors scaled by the largest element, rms and maximum; tools/r03_x6_bptt_error.py prints the table):
      * the six-product BPTT changes nothing measurable: every gradient's error equals the one with the exact BPTT kernel (debug
        bit 536870912) within the run-to-run spread of the atomics' arrival order;
      * the input gradient of mode 2 is not above the fp32 mode's (1.5e-8 vs 2.1e-8 rms);
      * the weight gradients of mode 2 (row-major six-product plane GEMM) stay at fp32's rounding level - 1.2-1.5x the fp32 mode's rms
        (4.8-6.7e-8 vs 4.0-4.3e-8: its reduction chains over T N rows are longer than the fp32 GEMM's split-K ones), 8-10x below the
        three-product fast mode's (5.4e-7)."""
    from aas_enhancement_amd import _lib, ops
    from aas_enhancement_amd.dist import FlatBuffers
    T, H = 48, 500
    torch.manual_seed(3)
    ref = nn.LSTM(H, H, bidirectional=True, bias=False).double()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(T, N, H, generator=g) * 0.5
    gy = torch.randn(T, N, H, generator=g)
    names = ("weight_ih_l0", "weight_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse")
    xr = x.double().requires_grad_(True)
    yr, _ = ref(xr)
    (yr[..., :H] + yr[..., H:] + xr).backward(gy.double())
    want = [xr.grad] + [getattr(ref, k).grad for k in names]

    def run(mode, flags=0):
        holder = nn.ParameterList([nn.Parameter(getattr(ref, k).detach().float().clone()) for k in names]).cuda()
        FlatBuffers(holder)
        try:
            ops.set_precision(mode)
            _lib.lib().aas_set_debug_flags(flags)
            xg = x.clone().cuda().requires_grad_(True)
            ops.birnn_layer(xg, *list(holder), kind="lstm", residual=True).backward(gy.cuda())
            ops.sync_wgrad()
            torch.cuda.synchronize()
        finally:
            _lib.lib().aas_set_debug_flags(0)
            ops.set_precision(int(os.environ.get("AAS_PRECISION", "0")))
        assert not ops.rnn_timeout_flag()
        got = [xg.grad] + [p_.grad for p_ in holder]
        out = []
        for a, b in zip(got, want):
            e = (a.detach().double().cpu() - b).abs() / b.abs().max()
            out.append((float(e.pow(2).mean().sqrt()), float(e.max())))
        return out
    e0, e2, e2x, e1 = run(0), run(2), run(2, 536870912), run(1)
    for i in range(5):
        # X6 BPTT == exact BPTT (the maxima are single elements and move by up to 2x between two runs of the SAME kernels: atomics order)
        # (rms ratio: 1.33 seen once in 16 full-suite runs for the input gradient, 1.0-1.15 otherwise)
        assert e2[i][0] <= 1.6 * e2x[i][0] + 1e-9 and e2[i][1] <= 2.5 * e2x[i][1] + 1e-8, (i, e2[i], e2x[i])
        assert e1[i][0] > 5.0 * e2[i][0], (i, e1[i], e2[i])                                                       # the fast mode IS narrower
    assert e2[0][0] <= 1.1 * e0[0][0] and e2[0][1] <= 2.0 * e0[0][1], (e2[0], e0[0])                            # input gradient
    for i in range(1, 5):
        assert e2[i][0] <= 2.0 * e0[i][0] and e2[i][0] < 1e-7 and e2[i][1] < 3e-6, (i, e2[i], e0[i])             # weight gradients: fp32's rounding level


def test_six_product_bptt_plane_sets_equal_the_split_of_its_fp32_output(gpu):
    """aas_lstm_bwd_planes3 (precision 2) writes d(gates) as the two three-term plane sets of the layer's GEMMs: bit for bit what
    aas_split_planes3 makes of aas_lstm_bwd's fp32 output (pad columns zero), so the layer's gradients do not depend on which of
    the two forms the BPTT launch delivered (ops.PLANES_EMIT)."""
    from aas_enhancement_amd import _lib, ops
    L = _lib.lib()
    T, N, H, G = 40, 60, 500, 4
    g = torch.Generator().manual_seed(11)
    R_ = lambda *shape, scale=1.0: (torch.randn(*shape, generator=g) * scale).cuda()
    ops.set_precision(2)
    try:
        w = [R_(G * H, H, scale=1.0 / H ** 0.5) for _ in range(2)]
        pre, dy = R_(T, N, 2, G * H), R_(T, N, H)
        dev = torch.device("cuda", 0)
        sync, xc = ops._sync_buf(dev), ops._xchg_buf(dev, T, N, H, G)
        s, p = _lib.stream(), _lib.ptr
        hout, gact, cst = torch.zeros(2, T, N, H, device="cuda"), torch.zeros(2, T, N, H, 4, device="cuda"), torch.zeros(2, T, N, H, device="cuda")
        ops.check(L.aas_lstm_fwd(s, T, N, H, p(pre), p(w[0]), p(w[1]), p(hout), p(gact), p(cst), p(sync), p(xc)), "fwd")
        dgx = torch.zeros(T, N, 2, G * H, device="cuda")
        ops.check(L.aas_lstm_bwd(s, T, N, H, p(dy), p(w[0]), p(w[1]), p(gact), p(cst), p(dgx), p(sync), p(xc)), "bwd")
        want = ops.split_planes3(dgx.view(T * N, 2 * G * H), T * N, 2 * G * H)
        got = ops._new_planes3(T * N, 2 * G * H, dev)
        got.buf.fill_(7)     # (stale contents must not survive, pad columns included)
        ops.check(L.aas_lstm_bwd_planes3(s, T, N, H, p(dy), p(w[0]), p(w[1]), p(gact), p(cst), got.buf.data_ptr(), got.Kp, p(sync), p(xc)), "bwd planes3")
        torch.cuda.synchronize()
        assert not ops.rnn_timeout_flag()
        assert got.Kp == want.Kp and torch.equal(got.buf.view(torch.int16), want.buf.view(torch.int16))
        assert torch.equal(got.to_float()[:, :2 * G * H], dgx.view(T * N, 2 * G * H))       # h + m + l == the fp32 value, exactly
        # a shape without a six-product kernel: refused with code 3, nothing written
        H2 = 64
        w2 = [R_(G * H2, H2) for _ in range(2)]
        dgs = ops._new_planes3(T * N, 2 * G * H2, dev)
        rc = L.aas_lstm_bwd_planes3(s, T, N, H2, p(dy[..., :H2].contiguous()), p(w2[0]), p(w2[1]), p(gact[..., :H2, :].contiguous()),
                                    p(cst[..., :H2].contiguous()), dgs.buf.data_ptr(), dgs.Kp, p(sync), p(ops._xchg_buf(dev, T, N, H2, G)))
        assert rc == 3
    finally:
        ops.set_precision(int(os.environ.get("AAS_PRECISION", "0")))
