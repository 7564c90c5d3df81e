"""CPU (-m "not gpu"): the data-parallel path with world_size 2 over gloo.
Checks DPContext (global normalisers, strided sharding, flat-buffer SUM all-reduce, scalar reduce) and the
property the AAS trainer relies on: shard losses normalised by GLOBAL counts + SUM all-reduce of the
gradients == single-process gradients of the global batch (E/D have no batch statistics)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.tools_shim import make_batch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _full_batch():
    b = make_batch(5, 6, [21, 19, 16, 12, 9], 77, [3, 2, 2, 1, 1], 78)
    return (torch.from_numpy(b["inputs"]), torch.from_numpy(b["targets"]), torch.from_numpy(b["pct"]),
            torch.from_numpy(b["target_sizes"]), torch.from_numpy(b["mask"]))


def _net():
    from aas_enhancement_amd import prng
    from oracle.ref_model import RefStackedBRNN
    torch.manual_seed(0)
    D = RefStackedBRNN(6, 6, 8, 2)
    sd = D.state_dict()
    for k, v in prng.fill_state_dict(sd, 5).items():
        sd[k].copy_(torch.from_numpy(v))
    return D


def _loss_and_backward(D, batch, n_el_global, n_global, lin):
    from oracle.ref_step import ctc_sum
    x, targets, pct, tsz, mask = batch
    out = D(x)
    l1 = (out - x).abs().sum() / n_el_global
    acts = lin(out.permute(2, 0, 1))  # [T,N,C]
    sizes = (pct.clone() * acts.size(0)).int()
    ctc = ctc_sum(acts, targets, sizes, tsz) / n_global
    (l1 + ctc).backward()
    return torch.stack([l1.detach(), ctc.detach()])


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # loopback only: no hostname / interface discovery
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=120))
    try:
        from aas_enhancement_amd.dist import DPContext, FlatBuffers
        dp = DPContext.from_env()
        assert dp.active and dp.world == world and dp.rank == rank
        full = _full_batch()
        shard = dp.shard_collated(full)
        rows = list(range(rank, 5, world))
        assert shard[0].shape[0] == len(rows) and shard[0].shape[2] == full[0].shape[2]  # global padding kept
        assert torch.equal(shard[0], full[0][rows]) and shard[3].tolist() == full[3][rows].tolist()
        n_valid_local = shard[4].n_valid
        N_glob, nv_glob = dp.global_counts([shard[0].shape[0], n_valid_local])
        assert N_glob == 5 and nv_glob == 21 + 19 + 16 + 12 + 9
        D = _net()
        torch.manual_seed(1)
        lin = torch.nn.Linear(6, 29)
        mods = torch.nn.ModuleList([D, lin])
        flat = FlatBuffers(mods)
        flat.zero_grad()
        sc = _loss_and_backward(D, shard, nv_glob, N_glob, lin)
        h = dp.allreduce_sum_(flat.flat_g, async_op=True)
        h.wait()
        sc = dp.reduce_scalars(sc)
        q.put((rank, flat.flat_g.clone().numpy(), sc.numpy()))
    finally:
        dist.destroy_process_group()


def test_dp_two_ranks_equals_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the global batch
    from aas_enhancement_amd.dist import FlatBuffers
    D = _net()
    torch.manual_seed(1)
    lin = torch.nn.Linear(6, 29)
    flat = FlatBuffers(torch.nn.ModuleList([D, lin]))
    full = _full_batch()
    sc = _loss_and_backward(D, full, 77, 5, lin)
    ref = flat.flat_g.numpy()
    for rank, g, s in res:
        assert np.abs(g - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), rank
        assert np.allclose(s, sc.numpy(), rtol=1e-5), rank
    assert np.array_equal(res[0][1], res[1][1])  # every rank holds identical gradients -> identical updates / kt


def test_flat_buffers_keep_module_api():
    from aas_enhancement_amd.dist import FlatBuffers
    D = _net()
    before = {k: v.clone() for k, v in D.state_dict().items()}
    fb = FlatBuffers(D)
    for k, v in D.state_dict().items():
        assert torch.equal(v, before[k])
    x = torch.randn(2, 6, 7)
    D(x).sum().backward()
    assert fb.flat_g.abs().sum() > 0 and all(p.grad.data_ptr() >= fb.flat_g.data_ptr() for p in D.parameters())
    fb.zero_grad()
    assert float(fb.flat_g.abs().sum()) == 0.0
    with torch.no_grad():
        fb.flat_p.add_(1.0)
    for k, v in D.state_dict().items():
        assert torch.allclose(v, before[k] + 1.0)


def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # loopback only: no hostname / interface discovery
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=120))
    try:
        from aas_enhancement_amd.dist import BucketReducer, DPContext, FlatBuffers
        dp = DPContext.from_env()
        torch.manual_seed(3)
        # three "layers" of four tensors each (the shape of a recurrent layer's parameters) plus two small tensors
        mods = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(8, 5)) for _ in range(12)] + [torch.nn.Parameter(torch.randn(3)) for _ in range(2)])
        flat = FlatBuffers(mods)
        red = BucketReducer(dp, [flat])
        red.MIN_ELEMS = 100          # 4 x 40 = 160 elements per layer bucket
        g = torch.Generator().manual_seed(10 + rank)
        flat.flat_g.copy_(torch.randn(flat.flat_g.numel(), generator=g))
        local = flat.flat_g.clone()
        red.begin()
        params = list(mods)
        red.on_wgrad([p.grad for p in params[8:12]])   # backward order: last layer first
        red.on_wgrad([p.grad for p in params[0:4]])
        red.on_wgrad([params[12].grad])                 # too small for a bucket: left to flush()
        fired = sorted(red.done[id(flat)])
        red.flush(flat)                                 # layer 2 (never announced) + the small tensors
        red.wait()
        q.put((rank, local.numpy(), flat.flat_g.clone().numpy(), fired))
    finally:
        dist.destroy_process_group()


def test_bucket_reducer_covers_every_element_exactly_once():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = res[0][1] + res[1][1]
    for rank, _, reduced, fired in res:
        assert np.allclose(reduced, total, rtol=0, atol=1e-6), rank      # summed once - not twice, not never
        assert fired == [(0, 160), (320, 480)]


def _main_worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # loopback only: no hostname / interface discovery
    torch.set_num_threads(1)
    import types
    from aas_enhancement_amd import main as M
    seen = {}

    class FakeTrainer(object):
        def __init__(self, config, data_loader):
            from aas_enhancement_amd.dist import DPContext
            self.dp = DPContext.from_env()
            seen["world"], seen["rank"] = self.dp.world, self.dp.rank
            assert data_loader.dp is not None and data_loader.dp.world == self.dp.world     # main.py hands the loader its DP context
            seen["batch"] = data_loader.next(cl_ny="ny", type="train")                     # already this rank's shard

        def train(self):
            pass

    import aas_enhancement_amd.trainer_AAS as TA
    TA.Trainer = FakeTrainer
    torch.cuda.set_device = lambda d: None          # CPU test: device selection is a no-op
    torch.cuda.manual_seed = lambda s: None
    cfg, _ = M.get_config(["--trainer", "AAS", "--dist_backend", "gloo", "--batch_size", "4", "--labels_path", os.path.join(tmp, "labels.json"),
                           "--DB_name", "none", "--tr_ny_manifest", os.path.join(tmp, "ny.csv"), "--tr_cl_manifest", os.path.join(tmp, "cl.csv"),
                           "--expnum", str(40 + rank), "--gpu", "-1"])
    os.chdir(tmp)
    M.main(cfg)
    q.put((rank, seen["world"], seen["batch"][0].numpy(), seen["batch"][2].numpy(), seen["batch"][3].tolist()))


def test_main_initialises_process_group_and_shards(tmp_path):
    """`torchrun -m aas_enhancement_amd.main --trainer AAS` wiring: RANK/WORLD_SIZE from the environment -> process group ->
    a data-parallel aware loader: every rank walks the SAME global bins and loads only its strided shard, padded to the global
    longest utterance."""
    import json
    tmp = str(tmp_path)
    json.dump(list("_'abcdefghijklmnopqrstuvwxyz "), open(os.path.join(tmp, "labels.json"), "w"))
    rows = []
    for i, T in enumerate([30, 28, 27, 25, 22, 20, 19, 15]):
        torch.save(torch.full((6, T), float(i)), os.path.join(tmp, "f%d.pt7" % i))
        open(os.path.join(tmp, "t%d.txt" % i), "w").write("ab c"[: 1 + i % 4])
        rows.append("%s,%s" % (os.path.join(tmp, "f%d.pt7" % i), os.path.join(tmp, "t%d.txt" % i)))
    for name in ("ny.csv", "cl.csv"):
        open(os.path.join(tmp, name), "w").write("\n".join(rows) + "\n")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_main_worker, args=(r, world, port, tmp, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=90) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == 2 and res[1][1] == 2
    x0, x1 = res[0][2], res[1][2]
    assert x0.shape == x1.shape and x0.shape[0] == 2                  # batch_size 4 over 2 ranks, same (global) padded length
    ids0, ids1 = sorted(int(v) for v in x0[:, 0, 0]), sorted(int(v) for v in x1[:, 0, 0])   # (utterance i is filled with the value i)
    assert not set(ids0) & set(ids1)
    both = sorted(ids0 + ids1)
    assert both in ([0, 1, 2, 3], [4, 5, 6, 7])                       # together: one global bin of the length-sorted manifest
    lens = [30, 28, 27, 25, 22, 20, 19, 15]
    t_glob = max(lens[i] for i in both)
    assert x0.shape[2] == t_glob
    for x, pct in ((x0, res[0][3]), (x1, res[1][3])):
        for row, pc in zip(x, pct):
            assert int(round(float(pc) * t_glob)) == lens[int(row[0, 0])]


def _buffers_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=120))
    try:
        import torch.nn as nn
        from aas_enhancement_amd.dist import DPContext
        from aas_enhancement_amd.model import DeepSpeech
        from tests.helpers import LABELS
        dp = DPContext.from_env()
        A = DeepSpeech(nn.GRU, LABELS, 12, 3, True, 11, 2, 8, 2, nFreq=8)      # parameter / buffer containers only: no kernel runs here
        # what a rank-0-only validation pass leaves behind: rank 0's BatchNorm running statistics (and batch counters) have moved
        g = torch.Generator().manual_seed(5)
        for b in A.buffers():
            if rank == 0:
                b.copy_((torch.rand(b.shape, generator=g) * 3).to(b.dtype) if b.dtype.is_floating_point else torch.full_like(b, 7))
        before = [b.clone() for b in A.buffers()]
        dp.broadcast_buffers(A, src=0)
        q.put((rank, [b.numpy().copy() for b in A.buffers()], [b.numpy().copy() for b in before]))
    finally:
        dist.destroy_process_group()


def test_broadcast_buffers_makes_ranks_replicas_again_after_rank0_only_validation():
    """A stays in train mode while rank 0 alone validates, so its BatchNorm running statistics move on rank 0
    only.  DPContext.broadcast_buffers (called by the trainers behind the post-validation barrier) hands every rank rank 0's
    buffers, floating point and integer ones, bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_buffers_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, after0, before0), (_, after1, before1) = res
    assert len(after0) >= 9
    assert any(not np.array_equal(a, b) for a, b in zip(before0, before1))      # the ranks really had diverged
    for a0, a1, b0 in zip(after0, after1, before0):
        assert np.array_equal(a0, b0) and np.array_equal(a1, b0) and a0.dtype == b0.dtype


def test_precision_context_restores_the_previous_mode():
    """ops.precision(mode): the mode inside the block, the previous one back on exit - also when the block raises."""
    from aas_enhancement_amd import ops
    base = ops.get_precision()
    try:
        with ops.precision(2):
            assert ops.get_precision() == 2
            with ops.precision(1):
                assert ops.get_precision() == 1
            assert ops.get_precision() == 2
            with ops.precision(None):
                assert ops.get_precision() == 2
        assert ops.get_precision() == base
        with pytest.raises(ValueError):
            with ops.precision(1):
                raise ValueError("x")
        assert ops.get_precision() == base
    finally:
        ops.set_precision(base)


# ---- bench.py --gpus 8: the worker-side bookkeeping on eight gloo ranks (CPU) ----------------------------------------------------
def _bench8_worker(rank, world, port, q):
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=120))
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        from aas_enhancement_amd.dist import DPContext
        dp = DPContext.from_env()
        assert dp.active and dp.world == world and dp.rank == rank
        dev = torch.device("cpu")
        # every rank draws its own shard of the synthetic global batch: same shapes, different members of the PRNG families
        ny, cl = bench.make_batches(rank, dev, n=3)
        sig = torch.tensor([float(ny[0].double().sum()), float(cl[0].double().sum()), float(ny[1].double().sum())], dtype=torch.float64)
        gathered = [torch.zeros_like(sig) for _ in range(world)]
        dist.all_gather(gathered, sig)
        # global normalisers as the trainer forms them (N, nElement noisy / clean summed over the ranks)
        counts = dp.global_counts([ny[0].shape[0], ny[4].n_valid, cl[4].n_valid])
        # the timed region: rank r "measured" (1 + r/10) s for 4 steps - the contract takes the MAX over ranks
        per_rank = bench.per_rank_times(1.0 + rank / 10.0, world, dev)
        dt = bench.max_over_ranks(1.0 + rank / 10.0, world, dev)
        value, ms = bench.whole_job(world, bench.N_PER * bench.T, dt, 4)
        q.put((rank, [g.tolist() for g in gathered], list(counts), dt, value, ms, bench.rank_seeds(rank), per_rank))
    finally:
        dist.destroy_process_group()


def test_bench_worker_bookkeeping_world8():
    """What `bench.py --gpus 8` does on each of its eight ranks besides stepping the trainer, on eight gloo ranks here: per-rank
    shards of the synthetic batch are distinct and reproducible, the global normalisers are the sums over the ranks, the step time
    is the MAX over ranks on every rank, `value` is the whole job's frames / s (8 x 30 x 200 frames per step), and the launcher
    path hands `--gpus 8` to eight workers (tests/test_abi.py covers the command line)."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys_path = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path.insert(0, sys_path)
    import bench
    sigs = res[0][1]
    assert len({tuple(s) for s in sigs}) == world                          # eight different shards ...
    assert all(r[1] == sigs for r in res)                                  # ... and every rank saw the same eight
    assert [r[6] for r in res] == [(123 + r, 125 + r, 124 + r) for r in range(world)]
    for r in res:
        assert r[2] == [3 * world, 3 * bench.T * world, 3 * bench.T * world]   # global N, nElement(noisy), nElement(clean)
        assert r[3] == pytest.approx(1.7)                                  # MAX over ranks (rank 7's 1.7 s), on every rank
        assert r[4] == pytest.approx(world * bench.N_PER * bench.T / (1.7 / 4)) and r[5] == pytest.approx(425.0)
        assert r[7] == pytest.approx([1.0 + k / 10.0 for k in range(world)])     # every rank's own time, on every rank (config.per_rank_ms_per_step)
