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
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
