"""Shared helpers for oracle and GPU parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LABELS = "_'abcdefghijklmnopqrstuvwxyz "


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def sub(z, prefix):
    """{key-without-prefix: tensor} for npz keys starting with prefix."""
    return {k[len(prefix):]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(prefix)}


def load_sd(module, sd, strict=True):
    cur = module.state_dict()
    fixed = {}
    for k, v in sd.items():
        if k in cur:
            fixed[k] = v.to(cur[k].dtype).reshape(cur[k].shape)
    missing = [k for k in cur if k not in fixed]
    if strict:
        assert not missing, missing
    module.load_state_dict(fixed, strict=strict)


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def batch_from(z, prefix, device="cpu"):
    """(inputs, targets, pct, target_sizes, mask) in _collate_fn order."""
    g = lambda k: torch.from_numpy(np.asarray(z[prefix + k]))
    inputs, mask, pct = g("inputs").to(device), g("mask").to(device), g("pct")
    if prefix + "targets" in z.files:
        return (inputs, g("targets"), pct, g("target_sizes"), mask)
    return (inputs, None, pct, None, mask)


def grad_close(a, b, rtol=1e-4, atol=1e-6):
    """|a-b| <= rtol*max|b| + atol (a conv bias feeding a train-mode BatchNorm has an exactly-zero
    true gradient, so its computed gradient is pure rounding noise ~1e-7)."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max()) <= rtol * float(b.abs().max()) + atol


# parameters whose gradient is identically zero in exact arithmetic (bias before train-mode BN):
# Adam turns their rounding-noise gradients into +-lr steps, so they are not comparable across
# implementations (and cannot influence any output).
NOISE_PARAMS = ("conv.0.bias", "conv.3.bias", "conv.1.running_mean", "conv.4.running_mean")  # running_mean absorbs the bias
