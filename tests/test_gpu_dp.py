"""GPU (-m gpu): the trainer's data-parallel code path, two ranks sharing the one GPU of the test box over gloo
(gradient buffers are staged through the host by DPContext for gloo; on a real node the same code runs over RCCL).
Property: 2 ranks x 15 utterances with global normalisers == 1 rank x 30 utterances (E/D have no batch statistics;
A is frozen and its BatchNorm uses local-batch statistics, so only the E/D-side quantities are compared exactly)."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _models():
    import torch.nn as nn
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    from tests.helpers import LABELS, load_sd
    G, D = stackedBRNN(I=8, H=16, L=2), stackedBRNN(I=8, H=16, L=2)
    A = DeepSpeech(nn.GRU, LABELS, 12, 2, True, 11, 2, 8, 2, nFreq=8)
    for m, s, cs in ((G, 11, None), (D, 12, None), (A, 13, 0.1)):
        load_sd(m, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(m.state_dict(), s, conv_std=cs).items()}, strict=False)
    return G, D, A


def _cfg(**kw):
    c = types.SimpleNamespace(lr=1e-3, beta1=0.5, beta2=0.999, optimizer="adam", batch_size=4, expnum=0, lambda_k=0.001, gamma=0.5, gpu=0,
                              load_path="", mode="train", write_log=False, w_adversarial=1.0, w_acoustic=0.0, allow_ASR_update_iter=10 ** 9,
                              schedule="fused")
    c.__dict__.update(kw)
    return c


def _batches():
    from tests.tools_shim import make_batch
    b = make_batch(4, 8, [60, 60, 60, 60], 501, [3, 3, 2, 2], 502)
    c = make_batch(4, 8, [60, 60, 60, 60], 601)
    ny = (torch.from_numpy(b["inputs"]), torch.from_numpy(b["targets"]), torch.from_numpy(b["pct"]), torch.from_numpy(b["target_sizes"]), torch.from_numpy(b["mask"]))
    cl = (torch.from_numpy(c["inputs"]), None, torch.from_numpy(c["pct"]), None, torch.from_numpy(c["mask"]))
    return ny, cl


def _worker(rank, world, port, q, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank); os.environ["WORLD_SIZE"] = str(world)
    import torch.distributed as dist
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # loopback only: no hostname / interface discovery
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=120))
    try:
        torch.cuda.set_device(0)
        from aas_enhancement_amd.dist import DPContext
        from aas_enhancement_amd.trainer_AAS import Trainer
        tr = Trainer(_cfg(**_mode_cfg(mode)), None, models=_models())
        tr.dp = DPContext.from_env()
        ny, cl = _batches()
        ny_s = tr.dp.shard_collated(ny)
        cl_s = tr.dp.shard_collated((cl[0], torch.zeros(0, dtype=torch.int32), cl[2], torch.zeros(4, dtype=torch.int32), cl[4]))
        out, prob = [], None
        for it in range(2):
            if mode == "sync":
                r = tr.train_step(ny_s, cl_s, it, log_norms=False)
            else:
                r0 = tr.train_step_async(ny_s, cl_s, it)
                r = tr.read_scalars()
                prob = r0["prob"].detach().cpu().numpy() if it == 0 else prob
            out.append([r["l_adv_ny_G"], r["l_adv_cl"], r["l_ctc"], r["kt"]])
        q.put((rank, np.asarray(out), tr._flat["G"].flat_p.detach().cpu().numpy(), tr._flat["D"].flat_p.detach().cpu().numpy(), prob))
    finally:
        dist.destroy_process_group()


def _mode_cfg(mode):
    if mode == "syncbn":   # the acoustic branch matters only here: global-batch BatchNorm statistics in A
        return dict(w_acoustic=1.0, sync_bn=True)
    return dict()


@pytest.mark.parametrize("mode", ["sync", "async", "syncbn"])
def test_trainer_dp_two_ranks_equals_single(mode, precision2):
    """sync: Trainer.train_step; async: the device-resident step bench.py times (global normalisers and kt inputs all-reduced
    on the device, bucketed gradient all-reduce); syncbn: + A's BatchNorm statistics all-reduced forward and backward, so
    even the acoustic branch (logits, CTC loss, the gradient it sends into E) equals the single-process global batch."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from aas_enhancement_amd.trainer_AAS import Trainer
    tr = Trainer(_cfg(**dict(_mode_cfg(mode), sync_bn=False)), None, models=_models())
    ny, cl = _batches()
    ref, prob_ref = [], None
    for it in range(2):
        r = tr.train_step(ny, cl, it, log_norms=False)
        if it == 0:
            prob_ref = r["prob"].detach().cpu().numpy()      # [T', N, C]
        ref.append([r["l_adv_ny_G"], r["l_adv_cl"], r["l_ctc"], r["kt"]])
    ref = np.asarray(ref)
    cols = [0, 1, 3] if mode != "syncbn" else [0, 1, 2, 3]      # (local-batch BN: the CTC value differs by design)

    def same(a_, b_):
        """Parameters after two Adam steps agree, except for isolated elements whose gradient is rounding noise around zero:
        Adam normalises those to +-lr steps whose SIGN depends on the summation order (atomics, rank order)."""
        d_ = np.abs(a_ - b_)
        return float((d_ > 2e-4).mean()) < 2e-3 and float(d_.max()) < 4.1e-3
    for rank, out, gp, dpar, prob in res:
        assert np.allclose(out[:, cols], ref[:, cols], rtol=2e-4), (rank, out, ref)
        # sync / async: w_acoustic = 0, so E and D are untouched by A's local-batch statistics; syncbn: the acoustic gradient
        # flows into E too and still matches, because A's BatchNorm runs on the global batch
        assert same(gp, tr._flat["G"].flat_p.detach().cpu().numpy()), rank
        assert same(dpar, tr._flat["D"].flat_p.detach().cpu().numpy()), rank
        if mode == "syncbn":
            assert np.abs(prob - prob_ref[:, rank::2]).max() < 1e-3 * np.abs(prob_ref).max()
    assert np.array_equal(res[0][2], res[1][2])  # identical parameters on every rank


def _am_model():
    import torch.nn as nn
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.model import DeepSpeech
    from tests.helpers import LABELS, load_sd
    A = DeepSpeech(nn.GRU, LABELS, 12, 3, True, 11, 2, 8, 2, nFreq=8)
    load_sd(A, {k: torch.from_numpy(v) for k, v in prng.fill_state_dict(A.state_dict(), 21, conv_std=0.1).items()}, strict=False)
    return A.cuda()


def _am_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank); os.environ["WORLD_SIZE"] = str(world)
    import torch.distributed as dist
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # loopback only: no hostname / interface discovery
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=120))
    try:
        torch.cuda.set_device(0)
        from aas_enhancement_amd.am_train import AMTrainer
        from aas_enhancement_amd.dist import DPContext
        tr = AMTrainer(_am_model(), lr=1e-3, dp=DPContext.from_env(), sync_bn=True)
        ny, _ = _batches()
        shard = tr.dp.shard_collated(ny)
        out = [tr.train_step(shard)["loss"] for _ in range(2)]
        q.put((rank, out, tr.flat.flat_p.detach().cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_am_trainer_dp_with_syncbn_equals_single(precision2):
    """AM pre-training step (config 5's per-GPU step) data parallel: global batch size as a device scalar, bucketed all-reduce of
    A's gradients, SyncBN - 2 ranks x 2 utterances == 1 rank x 4 utterances (loss and parameters after two Adam steps)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_am_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from aas_enhancement_amd.am_train import AMTrainer
    tr = AMTrainer(_am_model(), lr=1e-3)
    ny, _ = _batches()
    ref = [tr.train_step(ny)["loss"] for _ in range(2)]
    want = tr.flat.flat_p.detach().cpu().numpy()
    for rank, out, par in res:
        assert np.allclose(out, ref, rtol=2e-4), (rank, out, ref)
        d_ = np.abs(par - want)
        assert float((d_ > 2e-4).mean()) < 5e-3 and float(d_.max()) < 4.1e-3, rank   # (Adam sign flips of noise-level gradients)
    assert np.array_equal(res[0][2], res[1][2])
