"""Portable synthetic-batch builder shared by tools/make_goldens.py-style fixtures and the tests
(identical construction to make_goldens.make_batch; kept here so tests never import /root/reference)."""
import numpy as np

from aas_enhancement_amd import prng


def make_batch(N, Fdim, lens, seed, label_lens=None, lab_seed=None):
    T = max(lens)
    x = np.zeros((N, Fdim, T), np.float32)
    mask = np.zeros((N, 1, T), np.uint8)
    pct = np.zeros(N, np.float32)
    for n in range(N):
        x[n, :, :lens[n]] = prng.uniform(seed + 17 * n, (Fdim, lens[n]), 0.0, 6.0)
        mask[n, :, lens[n]:] = 1
        pct[n] = lens[n] / float(T)
    out = dict(inputs=x, mask=mask, pct=pct)
    if label_lens is not None:
        tg = []
        for n in range(N):
            tg.extend(prng.randint(lab_seed + n, (label_lens[n],), 1, 28).tolist())
        out["targets"] = np.asarray(tg, np.int32)
        out["target_sizes"] = np.asarray(label_lens, np.int32)
    return out
