"""Portable, dependency-free PRNG (splitmix64) used for synthetic weights/inputs.

The same generator is used by the golden-fixture generator (tools/make_goldens.py,
run next to the reference), by the tests and by bench.py, so that the GPU box can
regenerate bit-identical inputs/weights from a seed instead of shipping tensors.
SURVEY.md section 8(d) "Synthetic inputs" defines the distributions.
"""
import numpy as np

_MASK = (1 << 64) - 1


def splitmix64(seed, n):
    """n uint64 outputs of splitmix64 started at `seed` (vectorised)."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed & _MASK) + idx * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, shape):
    """float64 U[0,1) with 53 random bits."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return u.reshape(shape)


def uniform(seed, shape, lo=-1.0, hi=1.0):
    return (lo + (hi - lo) * uniform01(seed, shape)).astype(np.float32)


def normal(seed, shape, mean=0.0, std=1.0):
    """Box-Muller on two uniform streams (float64 math, cast to f32)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(seed, (n,))
    u2 = uniform01(seed ^ 0x5DEECE66D, (n,))
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


def randint(seed, shape, lo, hi):
    """integers in [lo, hi] inclusive."""
    n = int(np.prod(shape)) if len(shape) else 1
    r = splitmix64(seed, n) % np.uint64(hi - lo + 1)
    return (r.astype(np.int64) + lo).reshape(shape)


def fill_state_dict(state_dict, seed, conv_std=None):
    """Deterministically fill a torch state_dict in key order:
    RNN / linear / pointwise-conv weights U(+-1/sqrt(fan_in)), biases U(+-0.05), DeepSpeech conv ~N(0,0.1),
    BN gamma ~N(1,0.01), beta 0, running stats left at defaults.
    Returns {key: np.ndarray} for float entries it filled."""
    out = {}
    for i, (k, v) in enumerate(state_dict.items()):
        shp = tuple(v.shape)
        s = (seed * 1000003 + i * 7919) & _MASK
        if k.endswith("num_batches_tracked") or "running_" in k:
            continue
        is_bn = (".batch_norm." in k) or k.startswith("conv.1.") or k.startswith("conv.4.") \
            or k.startswith("fc.0.module.0.")
        if is_bn:
            if k.endswith("weight"):
                out[k] = normal(s, shp, 1.0, 0.01)
            else:
                out[k] = np.zeros(shp, np.float32)
        elif k.startswith("conv.") and conv_std is not None:
            if k.endswith("weight"):
                out[k] = normal(s, shp, 0.0, conv_std)
            else:
                out[k] = np.zeros(shp, np.float32)
        else:
            if len(shp) >= 2:
                fan = shp[1] * (shp[2] if len(shp) > 2 else 1)
                b = 1.0 / np.sqrt(fan)
            else:
                b = 0.05
            out[k] = uniform(s, shp, -b, b)
    return out
