import sys, torch
sys.path.insert(0, ".")
from aas_enhancement_amd import knobs, ops
from tests.test_gpu_round5 import _layer_run
shapes = [("lstm", 200, 30, 500), ("gru", 85, 30, 1000), ("lstm", 200, 60, 500), ("lstm", 5, 7, 500), ("gru", 4, 33, 512)]
res = []
for managed in (False, False, True, True):
    with knobs.override(MANAGED_XCHG=managed):
        out = [_layer_run(ops, k, T, N, H, 100 + i) for i, (k, T, N, H) in enumerate(shapes)]
        torch.cuda.synchronize()
        res.append(out)
names = ["y", "dx", "dwih", "dwhh", "dwih_r", "dwhh_r"]
for i, sh in enumerate(shapes):
    for j, nm in enumerate(names):
        e = [torch.equal(res[0][i][j], res[k][i][j]) for k in (1, 2, 3)]
        if not all(e):
            d = (res[0][i][j] != res[2][i][j])
            print(sh, nm, "legacy2 / managed1 / managed2 equal to legacy1:", e, "maxdiff %.3e" % float((res[0][i][j] - res[2][i][j]).abs().max()),
                  "n diff", int(d.sum()), "first idx", d.nonzero()[:4].tolist())
print("timeout", ops.rnn_timeout_flag(), ops.rnn_timeout_layers())
