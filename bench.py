#!/usr/bin/env python3
"""Headline benchmark: AAS train-step frames/sec (80-dim LMFB, batch 30 per GPU) on MI355X.

    python bench.py [--gpus N --steps K --warmup W]            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full iteration of trainer_AAS.py:131-194 (the exact code main.py --trainer AAS runs,
Trainer.train_step) on BASELINE.json configs[1]: E = D = 4x500 BiLSTM, frozen A = 2xconv1d + 5x1000
BiGRU + CTC, noisy + clean synthetic LMFB batches [30,80,200] already resident in HBM, fp32.
frames/sec = (noisy frames of all ranks) / step time, MAX over ranks.
Prints ONE JSON line on rank 0 with the `roofline` (dominant recurrent kernel, HIP-event timed on
its launch stream inside the timed region) and `cpu_baseline` (oracle step on the host cores) objects.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn as nn

LABELS = "_'abcdefghijklmnopqrstuvwxyz "
F, H, HA, M, N_PER, T, L = 80, 500, 1000, 128, 30, 200, 20
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA dense peak
PEAK_HBM_GBS = 8000.0


def build_models(rank_seed=9000):
    from aas_enhancement_amd import prng
    from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
    G, D = stackedBRNN(I=F, H=H, L=4), stackedBRNN(I=F, H=H, L=4)
    A = DeepSpeech(nn.GRU, LABELS, HA, 5, True, 11, 2, M, 2, nFreq=F)
    for m, s, cs in ((G, rank_seed + 1, None), (D, rank_seed + 2, None), (A, rank_seed + 3, 0.1)):
        sd = m.state_dict()
        for k, v in prng.fill_state_dict(sd, s, conv_std=cs).items():
            sd[k].copy_(torch.from_numpy(v))
    return G, D, A


def make_batches(rank, dev):
    from aas_enhancement_amd import prng
    ny = (torch.from_numpy(prng.uniform(123 + rank, (N_PER, F, T), 0.0, 6.0)).to(dev),
          torch.from_numpy(prng.randint(125 + rank, (N_PER * L,), 1, 28).astype(np.int32)),
          torch.ones(N_PER), torch.full((N_PER,), L, dtype=torch.int32),
          torch.zeros(N_PER, 1, T, dtype=torch.uint8, device=dev))
    ny[4].n_valid = N_PER * T
    cl = (torch.from_numpy(prng.uniform(124 + rank, (N_PER, F, T), 0.0, 6.0)).to(dev), None, None, None,
          torch.zeros(N_PER, 1, T, dtype=torch.uint8, device=dev))
    cl[4].n_valid = N_PER * T
    return ny, cl


def usable_cores():
    """CPUs this process may actually use: min(os.cpu_count, affinity mask, cgroup cpu.max quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def cpu_baseline():
    """Oracle AAS step (as-executed schedule of trainer_AAS.py:131-194, stock torch CPU kernels) on the same
    synthetic config-2 batch: a bounded sample of whole steps (~10-30 s of CPU work, no warm-up)."""
    from aas_enhancement_amd import prng
    from oracle import ref_model as RM
    from oracle import ref_step as RS
    cores = usable_cores()
    torch.set_num_threads(cores)
    G, D = RM.RefStackedBRNN(F, F, H, 4), RM.RefStackedBRNN(F, F, H, 4)
    A = RM.RefDeepSpeech(nn.GRU, LABELS, HA, 5, 11, 2, M, 2, nFreq=F)
    for m, s, cs in ((G, 9001, None), (D, 9002, None), (A, 9003, 0.1)):
        sd = m.state_dict()
        for k, v in prng.fill_state_dict(sd, s, conv_std=cs).items():
            sd[k].copy_(torch.from_numpy(v))
    cfg = RS.StepConfig(allow_ASR_update_iter=10 ** 9)
    og, od, oa = RS.make_optim(G, cfg), RS.make_optim(D, cfg), RS.make_optim(A, cfg)
    ny = (torch.from_numpy(prng.uniform(123, (N_PER, F, T), 0.0, 6.0)), torch.from_numpy(prng.randint(125, (N_PER * L,), 1, 28).astype(np.int32)),
          torch.ones(N_PER), torch.full((N_PER,), L, dtype=torch.int32), torch.zeros(N_PER, 1, T, dtype=torch.uint8))
    cl = (torch.from_numpy(prng.uniform(124, (N_PER, F, T), 0.0, 6.0)), None, None, None, torch.zeros(N_PER, 1, T, dtype=torch.uint8))
    # bounded sample: whole steps until ~12 s of CPU work have been spent (at least 1, at most 4 steps), no warm-up step
    t0 = time.time()
    steps, kt = 0, 0.0
    while steps < 4 and (steps == 0 or time.time() - t0 < 12.0):
        kt, _ = RS.aas_step(G, D, A, og, od, oa, ny, cl, cfg, kt, steps)
        steps += 1
    dt = time.time() - t0
    return {"value": steps * N_PER * T / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d full AAS steps (config 2: N=30,T=200,F=80; E/D 4x500 BiLSTM, frozen A 5x1000 BiGRU+CTC), "
                      "oracle/ref_step.aas_step on torch-CPU fp32, %d threads, %.1f s" % (steps, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--schedule", default="fused")
    ap.add_argument("--profile-gemm", action="store_true", help="also bracket every GEMM launch with HIP events")
    ap.add_argument("--graph", action="store_true", help="replay a captured hipGraph of the step instead of eager launches (single GPU)")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)  # the default; kept for older command lines
    ap.add_argument("--sync-steps", action="store_true", help="read the loss scalars back to the host every step (Trainer.train_step) "
                                                              "instead of leaving them on the device (Trainer.train_step_async)")
    ap.add_argument("--profile-steps", type=int, default=3, help="eager steps bracketed with HIP events for the roofline object")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline()))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world != a.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (a.gpus, world), file=sys.stderr)
        if world == 1 and a.gpus > 1:
            sys.exit(2)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from aas_enhancement_amd import ops
    from aas_enhancement_amd.trainer_AAS import Trainer
    cfg = types.SimpleNamespace(lr=1e-5, beta1=0.5, beta2=0.999, optimizer="adam", batch_size=N_PER, expnum=0,
                                lambda_k=0.001, gamma=0.5, gpu=local_rank, load_path="", mode="train", write_log=False,
                                w_adversarial=1.0, w_acoustic=1.0, allow_ASR_update_iter=10 ** 9, schedule=a.schedule,
                                world_size=world, rank=rank)
    tr = Trainer(cfg, None, models=build_models())
    ny, cl = make_batches(rank, dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Default: eager launches.  The step queues two independent chains of persistent launches layer by layer on two
    # streams (Trainer._interleaved_DA); a hipGraph replay enqueues its nodes one after the other in creation order
    # (~35 us apiece, 12.6 ms per replay with all kernels stubbed out vs 9.8 ms of host time for the eager step) and ran
    # the two branches one after the other, so the replay is slower (21.7 vs 20.6 ms) - `--graph` still selects it.
    use_graph = a.graph and world == 1
    use_async = world == 1 and not use_graph and not a.sync_steps
    if use_graph:
        step = lambda it: tr.train_step_graph(ny, cl, it)
    elif use_async:
        # no host read-back inside the step (kt and the losses stay on the device; the reference reads them only on logging
        # iterations): the host queues step i+1 while the GPU runs step i.  The timed region still ends with a barrier +
        # device synchronisation, and the scalars of the last step are read after it.
        step = lambda it: tr.train_step_async(ny, cl, it)
    else:
        step = lambda it: tr.train_step(ny, cl, it, log_norms=False)
    for it in range(a.warmup):
        step(it)
    barrier()
    t0 = time.perf_counter()
    for it in range(a.steps):
        r = step(a.warmup + it)
    barrier()
    dt = time.perf_counter() - t0
    if use_async:
        r = tr.read_scalars()
    # per-launch HIP events for the roofline object: the same step is run (eagerly) right after the timed region with
    # every recurrent launch bracketed by events on its launch stream, so the timed steps carry no instrumentation
    psteps = max(0, a.profile_steps)
    ops.Profiler.start(("rnn", "gemm") if a.profile_gemm else ("rnn",))
    for it in range(psteps):
        tr.train_step(ny, cl, a.warmup + a.steps + it, log_norms=False)
    torch.cuda.synchronize()
    prof = ops.Profiler.stop()
    assert not ops.rnn_timeout_flag(), "persistent RNN kernel hit its spin timeout"
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        ms = 1000.0 * dt / a.steps
        value = world * N_PER * T / (dt / a.steps)
        if not prof:   # --profile-steps 0 (PMC passes): no per-launch events, no roofline object
            print(json.dumps({"metric": "AAS train-step frames/sec (80-dim LMFB, batch 30 per GPU)", "value": value, "unit": "frames/s",
                              "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "roofline": None, "cpu_baseline": None}))
            if world > 1:
                dist.barrier()
                dist.destroy_process_group()
            return
        dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        name, d = dom
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot
        # share a pass, and counters cannot be read from inside the process): the committed summary of those passes
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01c_pmc_traffic.json")))
            keys = {"lstm_bwd": ["rnn_bwd_rs_kernel<1, 32, 4>"], "lstm_fwd": ["rnn_split_kernel<0, 1, 4>", "rnn_fwd32_kernel<0, 2, 0>"],
                    "gru_bwd": ["rnn_bwd_rs_kernel<3, 32, 8>"], "gru_fwd": ["rnn_fwd32_kernel<2, 4, 2>"]}.get(name, [])
            ks = [pmc["kernels"][k] for k in keys if k in pmc["kernels"]]
            if ks:  # dispatch-weighted mean over the kernel's instantiations
                traffic = sum(k["hbm_bytes_per_launch"] * k["dispatches"] for k in ks) / sum(k["dispatches"] for k in ks)
        except Exception:  # noqa: BLE001
            traffic = None
        achieved = d["flops_per_launch"] / (d["avg_ms"] * 1e-3) / 1e12
        out = {
            "metric": "AAS train-step frames/sec (80-dim LMFB, batch 30 per GPU)", "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "trainer_AAS step, BASELINE configs[1]: E=D=4x500 BiLSTM, frozen A=2xconv1d+5x1000 BiGRU+CTC, "
                                   "N=30/GPU, T=200, F=80, L=20 labels/utt, schedule=%s, %s" % (a.schedule, "hipGraph replay" if use_graph else ("eager launches, no per-step host read-back" if use_async else "eager launches")),
                       "global_batch": world * N_PER, "frames_per_utt": T, "parallelism": "dp%d" % world,
                       "last_losses": {k: r[k] for k in ("l_adv_ny_G", "l_adv_cl", "l_ctc", "kt")}},
            "roofline": {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                         "traffic_note": "bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from separate rocprofv3 --pmc passes "
                                         "(profiles/r01c_pmc_traffic.json; L2<->fabric bytes, i.e. including the in-launch exchange ring that the Infinity Cache absorbs); "
                                         "algorithmic bytes per launch 0.26 GB (N=30) / 0.53 GB (N=60)",
                         "timing": ("HIP events around each launch on its launch stream, %d eager steps run right after the "
                                    "timed region" % psteps),
                         "avg_launch_ms": d["avg_ms"], "launches_per_step": d["count"] / psteps,
                         "algorithmic_flops_per_launch": d["flops_per_launch"],
                         "busy_ms_per_step": d["total_ms"] / psteps,
                         "kernels": {k: {"avg_ms": v["avg_ms"], "per_step": v["count"] / psteps,
                                         "tflops": v["flops_per_launch"] / (v["avg_ms"] * 1e-3) / 1e12,
                                         "busy_ms_per_step": v["total_ms"] / psteps} for k, v in prof.items()}},
        }
        out["cpu_baseline"] = None
        if world == 1 and not a.no_cpu_baseline:
            # run in a child process under a hard timeout so a slow host can never hang the bench
            import subprocess
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True,
                                    text=True, timeout=240, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
                out["cpu_baseline"] = json.loads(cp.stdout.strip().splitlines()[-1])
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": usable_cores(), "kind": "port",
                                       "sample": "FAILED: %r" % (e,)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
