#!/usr/bin/env python3
"""LMFB kernel timing with phase ablations (debug flags 1 staging loads, 2 fold, 4 MFMA, 8 mel, 16 output stores; 31 = loop
skeleton only) and the scalar-FMA kernel it replaced.  Usage: python tools/lmfb_bench.py [N utterances]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aas_enhancement_amd import _lib, prng
from aas_enhancement_amd.lmfb import LMFB


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    mod = LMFB(n_mels=80).cuda()
    w = torch.from_numpy(prng.normal(126, (64, 31840), 0.0, 0.1)).cuda().repeat(n // 64, 1).contiguous()
    L = _lib.lib()
    for fl in (0, 32, 1, 2, 4, 8, 16, 31):
        L.aas_set_debug_flags(fl)
        for _ in range(3):
            mod(w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            mod(w)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("flags %2d: %.3f ms  (%.0f GB/s algorithmic)" % (fl, ms, n * 191040 / ms / 1e6), flush=True)
    L.aas_set_debug_flags(0)
    for _ in range(2):
        mod(w, force_scalar=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        mod(w, force_scalar=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("scalar-FMA kernel (round 1): %.3f ms  (%.0f GB/s algorithmic)" % (ms, n * 191040 / ms / 1e6), flush=True)


if __name__ == "__main__":
    main()
