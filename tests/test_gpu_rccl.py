"""GPU (-m gpu): the data-parallel code path on RCCL itself.  The test box has ONE GPU and RCCL refuses two ranks on a
device, so a child process forms a one-rank `nccl` group (AAS_DP_FORCE=1 arms the data-parallel path on it) and runs the
device-resident steps at CONFIG-2 SIZE - the size at which every recurrent layer's gradient slice is >= 1 Mi elements, so
dist.BucketReducer's per-layer buckets really fire from the weight-gradient stream (the tiny models of test_gpu_dp.py only
ever reach flush()).  Checked: the F3 goldens of the reference (iterations 0 and 1, A trainable from iteration 1), every
element of every flat gradient buffer all-reduced EXACTLY once per step (the collectives are recorded by wrapping
DPContext.allreduce_sum_; with one rank SUM is the identity, so only the record can show a double or a missed reduction),
layer buckets among them, and the sticky exchange-timeout word clear (RCCL's kernels share the chip with the persistent
grids).  bench.py's own launcher is exercised too: `--gpus 2` on this one-GPU box must fail fast with a clear message."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _record_allreduces(dp, flats):
    """Wrap dp.allreduce_sum_: per call append (flat name, lo, hi) of the reduced range."""
    log = []
    inner = dp.allreduce_sum_

    def wrapped(tensor, async_op=False):
        for name, f in flats.items():
            off = (tensor.data_ptr() - f.flat_g.data_ptr()) // 4
            if 0 <= off < f.flat_g.numel():
                log.append((name, int(off), int(off + tensor.numel())))
                break
        else:
            log.append(("?", 0, int(tensor.numel())))
        return inner(tensor, async_op=async_op)
    dp.allreduce_sum_ = wrapped
    return log


def _coverage(log, flats, names):
    """-> {name: (covered exactly once?, number of ranges >= 1 Mi elements that are not the whole buffer)}"""
    out = {}
    for name in names:
        n = flats[name].flat_g.numel()
        cnt = np.zeros(n, dtype=np.int32)
        big = 0
        for nm, lo, hi in log:
            if nm == name:
                cnt[lo:hi] += 1
                big += int(hi - lo >= (1 << 20) and hi - lo < n)
        out[name] = (bool((cnt == 1).all()), big, int(cnt.min()), int(cnt.max()))
    return out


def _aas_worker(port, q, precision="1"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", AAS_DP_FORCE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", AAS_PRECISION=precision)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from aas_enhancement_amd import ops
        from aas_enhancement_amd.trainer_AAS import Trainer
        from tests.helpers import load
        from tests.test_gpu_round2 import _config2_batches, _config2_models
        from tests.test_gpu_step import cfg
        z = load("f3_aas_config2.npz")
        tr = Trainer(cfg(lr=float(z["lr"]), nFeat=80, rnn_size=500, allow_ASR_update_iter=0), None, models=_config2_models())
        tr.kt = float(z["kt0"])
        tr.make_optimizers()
        assert tr.dp.active and tr._reducer is not None and dist.get_backend() == "nccl"
        log = _record_allreduces(tr.dp, tr._flat)
        res = []
        for it in range(2):
            del log[:]
            ny, cl = _config2_batches(it)
            r = tr.train_step_async(ny, cl, it)
            sc = tr.read_scalars()
            names = ["G", "D"] + (["A"] if it > 0 else [])
            cov = _coverage(log, tr._flat, names)
            enh, prob = r["enhanced"].detach().reshape(-1), r["prob"].detach().reshape(-1)
            e_got = enh[torch.from_numpy(z["it%d.enh_idx" % it]).cuda()].cpu().numpy()
            p_got = prob[torch.from_numpy(z["it%d.logit_idx" % it]).cuda()].cpu().numpy()
            res.append(dict(sc=sc, cov=cov, e=e_got, p=p_got, unknown=[e for e in log if e[0] == "?"]))
        q.put(dict(ok=True, res=res, timeout=ops.rnn_timeout_flag()))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put(dict(ok=False, err=traceback.format_exc() + repr(e)))
    finally:
        dist.destroy_process_group()


def _run_child(target, *extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=target, args=(_free_port(), q) + extra)
    p.start()
    out = q.get(timeout=900)
    p.join(timeout=120)
    assert out["ok"], out.get("err")
    assert p.exitcode == 0
    return out


@pytest.mark.parametrize("precision", ["0", "1", "2"], ids=["fp32", "splitbf16", "fp32eq"])
def test_aas_async_steps_on_one_rank_rccl_config2_buckets_and_goldens(precision):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tests.helpers import load
    z = load("f3_aas_config2.npz")
    out = _run_child(_aas_worker, precision)
    assert not out["timeout"], "a persistent recurrent launch timed out beside the RCCL kernels"
    for it, r in enumerate(out["res"]):
        for k in ("l_adv_ny_G", "l_adv_cl", "l_ctc", "kt", "conv_measure"):
            assert r["sc"][k] == pytest.approx(float(z["it%d.%s" % (it, k)]), rel=1e-2), (it, k)
        e_ref, p_ref = z["it%d.enh_samples" % it], z["it%d.logit_samples" % it]
        assert np.abs(r["e"] - e_ref).max() < 1e-3 * np.abs(e_ref).max(), it
        assert np.abs(r["p"] - p_ref).max() < 1e-3 * np.abs(p_ref).max(), it
        assert not r["unknown"], r["unknown"]
        for name, (once, big, lo, hi) in r["cov"].items():
            assert once, "iteration %d: flat gradient buffer %s reduced between %d and %d times per element" % (it, name, lo, hi)
            # 4 recurrent layers in E / D (one of E's is left to the flush by design), 5 in A: the per-layer buckets fired
            assert big >= (3 if name != "A" else 4), (it, name, big)


def _am_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", AAS_DP_FORCE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import torch.nn as nn
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from aas_enhancement_amd import ops, prng
        from aas_enhancement_amd.am_train import AMTrainer
        from aas_enhancement_amd.model import DeepSpeech
        from tests.helpers import LABELS, load
        from tests.test_gpu_round2 import _fill
        z = load("f7_am_config5.npz")
        N, F, T, HA, M, L = [int(z[k]) for k in ("N", "F", "T", "HA", "M", "L")]
        A = _fill(DeepSpeech(nn.GRU, LABELS, HA, 5, True, 11, 2, M, 2, nFreq=F), int(z["weight_seed"]), 0.1).cuda()
        tr = AMTrainer(A, lr=float(z["lr"]))
        assert tr.dp.active and tr._reducer is not None
        flats = {"A": tr.flat}
        log = _record_allreduces(tr.dp, flats)
        res = []
        for it in range(2):
            del log[:]
            x = torch.from_numpy(prng.uniform(int(z["input_seed0"]) + it, (N, F, T), 0.0, 6.0))
            tg = torch.from_numpy(prng.randint(int(z["label_seed0"]) + it, (N * L,), 1, 28).astype(np.int32))
            r = tr.train_step_async((x, tg, torch.ones(N), torch.full((N,), L, dtype=torch.int32)))
            loss, is_inf = tr.read_loss(r["handle"])
            lg = r["logits"].detach().reshape(-1)
            got = lg[torch.from_numpy(z["it%d.logit_idx" % it]).cuda()].cpu().numpy()
            res.append(dict(loss=loss, cov=_coverage(log, flats, ["A"]), p=got))
        q.put(dict(ok=True, res=res, timeout=ops.rnn_timeout_flag()))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put(dict(ok=False, err=traceback.format_exc() + repr(e)))
    finally:
        dist.destroy_process_group()


def test_am_async_steps_on_one_rank_rccl_config5_buckets_and_goldens():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tests.helpers import load
    z = load("f7_am_config5.npz")
    out = _run_child(_am_worker)
    assert not out["timeout"]
    for it, r in enumerate(out["res"]):
        assert r["loss"] == pytest.approx(float(z["it%d.loss" % it]), rel=1e-2)
        ref = z["it%d.logit_samples" % it]
        assert np.abs(r["p"] - ref).max() < 1e-3 * np.abs(ref).max(), it
        once, big, lo, hi = r["cov"]["A"]
        assert once, (it, lo, hi)
        assert big >= 4, (it, big)


def _readiness_worker(port, q):
    """One-rank RCCL, config 2, frozen A (the headline's configuration), steady state: (1) the data-parallel step issues NO host
    synchronisation (torch's sync debug mode raises on one); (2) HIP events on the issuing streams say when each gradient bucket's
    all-reduce becomes eligible relative to the launch that first consumes the reduced gradients (optimizer_g.step_dev)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", AAS_DP_FORCE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", AAS_PRECISION="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from aas_enhancement_amd import ops
        from aas_enhancement_amd.trainer_AAS import Trainer
        from tests.test_gpu_round2 import _config2_batches, _config2_models
        from tests.test_gpu_step import cfg
        tr = Trainer(cfg(lr=1e-5, nFeat=80, rnn_size=500, allow_ASR_update_iter=10 ** 9), None, models=_config2_models())
        ny, cl = _config2_batches(0)
        dev = torch.device("cuda", 0)
        ny = tuple(t_.to(dev) if (torch.is_tensor(t_) and t_.dim() == 3) else t_ for t_ in ny)
        cl = tuple(t_.to(dev) if (torch.is_tensor(t_) and t_.dim() == 3) else t_ for t_ in cl)
        ny[4].n_valid = cl[4].n_valid = 30 * 200
        for it in range(10):                       # allocator pools, scratch, streams, the pinned staging ring: all settled
            tr.train_step_async(ny, cl, it)
        torch.cuda.synchronize()
        assert tr.dp.active and tr._last_schedule == "batched"
        sync_error = None
        torch.cuda.set_sync_debug_mode("error")
        try:
            for it in range(10, 14):
                tr.train_step_async(ny, cl, it)
        except Exception as e:  # noqa: BLE001
            sync_error = repr(e)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
        marks = []
        ops.Profiler.start(("coll", "mark"))
        for it in range(14, 18):
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            marks.append((e, len(ops.Profiler.records)))
            tr.train_step_async(ny, cl, it)
        marks.append((None, len(ops.Profiler.records)))
        torch.cuda.synchronize()
        recs = list(ops.Profiler.records)
        ops.Profiler.enabled = False
        steps = []
        for i in range(4):
            rows = []
            for name, _, e0, e1, _t in recs[marks[i][1]:marks[i + 1][1]]:
                rows.append((name, marks[i][0].elapsed_time(e0)))
            steps.append(rows)
        tr.read_scalars()
        q.put(dict(ok=True, sync_error=sync_error, steps=steps, timeout=ops.rnn_timeout_flag()))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put(dict(ok=False, err=traceback.format_exc() + repr(e)))
    finally:
        dist.destroy_process_group()


def test_dp_step_issues_no_host_sync_and_buckets_run_ahead_of_the_optimiser():
    """Multi-GPU readiness on the one GPU there is (no 8-GPU node has been available in any round): the device-resident
    data-parallel step queues without a host synchronisation, and every gradient bucket except the one of the layer that is
    back-propagated last (E's first, left to the final flush by design) becomes eligible >= 0.5 ms before optimizer_g.step_dev is
    queued - an 8-rank ring all-reduce of a 15.3 MB bucket takes 0.18 ms on one xGMI link direction, so only that last
    bucket's duration is exposed."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = _run_child(_readiness_worker)
    assert not out["timeout"]
    assert out["sync_error"] is None, out["sync_error"]
    for rows in out["steps"][1:]:
        opt = [t for n, t in rows if n.startswith("optimizer_g")]
        colls = sorted((t, n) for n, t in rows if n.startswith("allreduce"))
        assert len(opt) == 1 and len(colls) >= 8, rows
        big = [(t, n) for t, n in colls if float(n.split("[")[1].split(" MB")[0]) >= 4.0]
        assert len(big) >= 7, colls                          # 4 layer buckets of D, 3 of E (+ the final flush holding E's first layer)
        early = [t for t, n in big[:-1]]
        assert all(t <= opt[0] - 0.5 for t in early), (opt, big)
        assert all(t <= opt[0] + 0.05 for t, n in colls), (opt, colls)       # nothing is issued behind the optimiser


def test_bench_gpus_2_on_a_one_gpu_box_fails_fast_with_a_clear_message():
    """`python bench.py --gpus N` starts its own N ranks (no launcher needed); with fewer devices than ranks it must say so
    and exit non-zero before anything touches the GPU - not hang, not exit 0."""
    have = torch.cuda.device_count()
    if have >= 2:
        pytest.skip("box has %d GPUs" % have)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert cp.returncode != 0
    assert "only %d device" % have in cp.stderr
    assert cp.stdout.strip() == ""


def test_bench_launcher_path_relays_rank0_line_and_exit_code():
    """The self-launching path of bench.py on the one GPU of the test box (AAS_BENCH_FORCE_SPAWN=1 takes it for N=1): the parent never
    touches the GPU, starts `python -m torch.distributed.run --nproc-per-node 1 bench.py ...` as a child, relays exactly one JSON
    line (rank 0's) on stdout and returns the workers' exit code; a failing worker (unknown flag) gives a non-zero code and no line."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "AAS_PRECISION")}
    env["AAS_BENCH_FORCE_SPAWN"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-extras", "--no-cpu-baseline",
           "--profile-steps", "0"]
    cp = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert cp.returncode == 0, cp.stderr[-2000:]
    lines = [ln for ln in cp.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["dtype"] == "f32" and j["parity_gate"]["status"] == "ok" and j["value"] > 0
    bad = subprocess.run(cmd + ["--no-such-flag"], capture_output=True, text=True, timeout=600, env=env)
    assert bad.returncode != 0 and bad.stdout.strip() == ""
