"""Stress check of the step schedules: a serial reference step sequence, then SCHED_REPS fresh trainers on the two-chain schedule
(SCHED_INTERLEAVE=1 alternating / 0 chain after chain) at BASELINE config-2 sizes; every run must reproduce the serial scalars,
outputs and parameters to 2e-5.  (Found the address-keyed cache of frozen weight planes handing a new model another
layer's planes.)  SCHED_FLAGS = aas_set_debug_flags value."""
import os, sys
import numpy as np, torch, torch.nn as nn
sys.path.insert(0, ".")
from tests.test_gpu_step import cfg, LABELS, load_sd
from aas_enhancement_amd import knobs, prng, ops, _lib
from aas_enhancement_amd.model import DeepSpeech, stackedBRNN
from aas_enhancement_amd.trainer_AAS import Trainer
_lib.lib().aas_set_debug_flags(int(os.environ.get("SCHED_FLAGS", "0")))
N, F, T, H, HA, M, L = 30, 80, 200, 500, 1000, 128, 20
reps = int(os.environ.get("SCHED_REPS", "6"))
modes = [("serial", dict(OVERLAP_ASR=False))] + [("alt%d" % i, dict(OVERLAP_ASR=True, INTERLEAVE=os.environ.get("SCHED_INTERLEAVE", "1") == "1")) for i in range(reps)]
res = {}
for mode, env in modes:
    for k_, v_ in env.items():
        knobs.set(k_, v_)
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
        out += [r[k] for k in ("l_adv_ny_G", "l_adv_cl", "l_ctc", "kt")] + [float(r["enhanced"].detach().double().abs().sum()), float(r["prob"].detach().double().abs().sum())]
    torch.cuda.synchronize()
    out += [float(sum(p.double().abs().sum() for p in m.parameters())) for m in (G, D)]
    res[mode] = np.asarray(out)
ref = res["serial"]
bad = 0
for k, v in res.items():
    d = np.abs(v - ref) / np.maximum(np.abs(ref), 1e-30)
    if d.max() > 2e-5:
        bad += 1
        print("%-8s MISMATCH max %.1e  %s" % (k, d.max(), np.array2string(d, precision=0)))
print("flags", os.environ.get("SCHED_FLAGS", "0"), "interleave", os.environ.get("SCHED_INTERLEAVE", "1"), "planes", knobs.get("PLANES_PRE"), ": %d of %d runs mismatch; timeout flag %s" % (bad, reps, ops.rnn_timeout_flag()))
