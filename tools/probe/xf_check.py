import sys, torch, torch.nn as nn
sys.path.insert(0, ".")
from aas_enhancement_amd import ops, _lib
L = _lib.lib()
torch.manual_seed(0)
for (T, N, H) in ((200, 30, 500), (37, 30, 500), (60, 3, 16), (50, 8, 256), (9, 5, 100), (1, 2, 12), (40, 30, 128)):
    res = {}
    for fl in (0, 1073741824):
        L.aas_set_debug_flags(fl)
        torch.manual_seed(1)
        x = (torch.randn(T, N, H) * 0.5).cuda()
        w = [(torch.randn(4 * H, H) / H ** 0.5).cuda() for _ in range(4)]
        hout, gact, cst = ops._birnn_fwd("lstm", x, *w)
        torch.cuda.synchronize()
        res[fl] = (hout.clone(), gact.clone(), cst.clone())
    L.aas_set_debug_flags(0)
    e = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(res[0], res[1073741824])]
    print(T, N, H, "max rel diff hout/gact/cst:", e, "timeout", ops.rnn_timeout_flag())
