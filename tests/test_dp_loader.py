"""CPU (-m "not gpu"): data-parallel loading shards BEFORE it loads (SURVEY 8f3 "DP-aware sharding";
Speech_enhancement_by_AAS/loader_functions.py:118-137, data_loader.py:42-83).  World 2 over gloo:

* every rank walks the same global bins with the same random draws and opens exactly ITS files - rank r the positions
  r, r+W, ... of the bin in longest-first order - never another rank's;
* each rank's batch is padded to the GLOBAL longest utterance and `input_percentages` refer to it, so the shards put back
  together are the single-process `_collate_fn` batch element for element (hence the step is the single-process step:
  tests/test_gpu_dp.py runs it on the GPU);
* a tail bin with fewer utterances than ranks is dropped on every rank alike; the sampler's length says so;
* paired manifests (DCE / FSEGAN loaders) shard the clean targets with the inputs.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

LABELS = "_'abcdefghijklmnopqrstuvwxyz "
LENS = [9, 12, 15, 16, 19, 21, 22, 26, 30]         # ascending, as make_manifest_librispeech.py:58-59 writes them; 9 = 4 + 4 + 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _write(tmp, paired):
    rows = []
    for i, T in enumerate(LENS):
        g = torch.Generator().manual_seed(100 + i)
        torch.save(torch.rand(6, T, generator=g), os.path.join(tmp, "f%d.pt7" % i))
        open(os.path.join(tmp, "t%d.txt" % i), "w").write("hello world"[: 2 + i % 7])
        row = "%s,%s" % (os.path.join(tmp, "f%d.pt7" % i), os.path.join(tmp, "t%d.txt" % i))
        if paired:
            torch.save(torch.rand(6, T, generator=g), os.path.join(tmp, "c%d.pt7" % i))
            row += "," + os.path.join(tmp, "c%d.pt7" % i)
        rows.append(row)
    path = os.path.join(tmp, "paired.csv" if paired else "plain.csv")
    open(path, "w").write("\n".join(rows) + "\n")
    return path


def _worker(rank, world, port, man, paired, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=120))
    try:
        from aas_enhancement_amd import loader_functions as LF
        from aas_enhancement_amd.data_loader import DataLoader
        from aas_enhancement_amd.dist import DPContext
        opened = []
        inner = LF.FeatDataset.__getitem__

        def spy(self, index):
            opened.append(int(index))
            return inner(self, index)
        LF.FeatDataset.__getitem__ = spy
        np.random.seed(7)                       # main.py seeds numpy identically on every rank
        dl = DataLoader(batch_size=4, paired=paired, tr_ny_manifest=man, labels=LABELS, num_workers=0, pin_memory=False, dp=DPContext.from_env())
        sp = dl._sp["ny/train"]
        sp.log = []
        out = []
        for _ in range(2 * len(sp)):            # two epochs: the wrap-around reshuffles identically on both ranks
            mark = len(opened)
            b = dl.next("ny", "train")
            out.append(([t.numpy() if torch.is_tensor(t) else t for t in b], opened[mark:]))
        q.put((rank, len(sp), out, sp.log))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("paired", [False, True], ids=["plain", "paired"])
def test_each_rank_opens_only_its_shard_and_the_union_is_the_single_process_batch(tmp_path, paired):
    from aas_enhancement_amd import loader_functions as LF
    man = _write(str(tmp_path), paired)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, man, paired, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ds = LF.FeatDataset(man, LABELS)
    collate = LF._collate_fn_paired if paired else LF._collate_fn
    (_, n0, out0, log0), (_, n1, out1, log1) = res
    assert n0 == n1 == 2                                     # 9 utterances / 4: the tail bin of ONE utterance is dropped on both ranks
    assert [g for g, _ in log0] == [g for g, _ in log1]      # same global bins, same order, same permutation on both ranks
    assert len(out0) == len(out1) == 4
    seen = []
    for k in range(4):
        glob = log0[k][0]
        want = sorted(glob, reverse=True)                    # longest first (manifest order = length order)
        (b0, open0), (b1, open1) = out0[k], out1[k]
        assert sorted(open0) == sorted(want[0::2]) and sorted(open1) == sorted(want[1::2])      # exactly its files, no others
        assert not set(open0) & set(open1)
        seen.append(tuple(sorted(glob)))
        ref = collate([ds[i] for i in glob])                 # the single-process batch of this bin
        big = [i for i, t in enumerate(ref) if torch.is_tensor(t) and t.dim() == 3]
        pct_i, tsz_i, tg_i = (4, 5, 3) if paired else (2, 3, 1)
        t_glob = ref[big[0]].size(2)
        for i in big:                                        # rows interleave back into the global longest-first order
            assert b0[i].shape[2] == b1[i].shape[2] == t_glob
            assert np.array_equal(b0[i], ref[i][0::2].numpy()) and np.array_equal(b1[i], ref[i][1::2].numpy())
        assert np.array_equal(b0[pct_i], ref[pct_i][0::2].numpy()) and np.array_equal(b1[pct_i], ref[pct_i][1::2].numpy())
        assert np.array_equal(b0[tsz_i], ref[tsz_i][0::2].numpy()) and np.array_equal(b1[tsz_i], ref[tsz_i][1::2].numpy())
        offs = np.concatenate([[0], np.cumsum(ref[tsz_i].numpy())])
        for b, r in ((b0, 0), (b1, 1)):
            tg = np.concatenate([ref[tg_i].numpy()[offs[j]:offs[j + 1]] for j in range(r, len(glob), 2)])
            assert np.array_equal(b[tg_i], tg)
    assert set(seen[:2]) == set(seen[2:]) == {(0, 1, 2, 3), (4, 5, 6, 7)}       # each epoch covers both full bins


def test_shard_collated_refuses_a_batch_smaller_than_the_world_and_counts_frames_on_the_host():
    from aas_enhancement_amd.dist import DPContext
    dp = DPContext(world=4, rank=1)
    x = torch.zeros(3, 2, 5)
    batch = (x, torch.zeros(3, dtype=torch.int32), torch.ones(3), torch.ones(3, dtype=torch.int32), torch.zeros(3, 1, 5, dtype=torch.uint8))
    with pytest.raises(ValueError, match="cannot be sharded over 4 ranks"):
        dp.shard_collated(batch)
    dp2 = DPContext(world=2, rank=1)
    mask = torch.zeros(3, 1, 5, dtype=torch.uint8)
    mask[1, 0, 3:] = 1
    pct = torch.tensor([1.0, 0.6, 1.0])
    sh = dp2.shard_collated((x, torch.tensor([1, 2, 3], dtype=torch.int32), pct, torch.ones(3, dtype=torch.int32), mask))
    assert sh[0].size(0) == 1 and sh[4].n_valid == 3 and sh[1].tolist() == [2]
