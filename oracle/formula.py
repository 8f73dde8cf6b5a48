"""Closed-form tensors for golden fixtures (TEST INFRASTRUCTURE; see oracle/__init__.py).

Weights and inputs of the fixtures are *computed*, never stored: every value is
`scale * u(hash(name, i))` (an integer hash mapped to a unit-variance uniform), rounded to fp32, so that the
fixtures do not depend on any RNG stream or torch version.  The same generator
is used by tools/make_golden.py (which loads it into the imported reference
modules) and by the tests (which load it into the oracle and into the HIP
modules).
"""
import math
import zlib

import numpy as np
import torch


def _seed(name):
    return np.uint64(zlib.crc32(name.encode())) * np.uint64(0x9E3779B97F4A7C15)


def _mix64(z):
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def wave(name, shape, scale=1.0, offset=0.0):
    """offset + scale * u, u uniform-looking in [-sqrt(3), sqrt(3)) (unit variance), from an
    integer hash of (name, index): exactly reproducible, no cancellation structure."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        h = _mix64(np.arange(n, dtype=np.uint64) * np.uint64(0xD6E8FEB86659FD93) + _seed(name))
    u = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)          # [0, 1)
    v = offset + scale * math.sqrt(3.0) * (2.0 * u - 1.0)
    return torch.from_numpy(v.astype(np.float32)).reshape(tuple(shape))


def param_value(name, shape):
    """Value of one state_dict entry by (reference) key name and shape.  The
    tokenizer is registered twice in the reference (vit.py:221): both names map
    to the same tensor, so both get the `patch_embed.` value."""
    if name.startswith("encoder.to_patch_embedding."):
        name = "patch_embed." + name[len("encoder.to_patch_embedding."):]
    leaf = name.split(".")[-1]
    is_norm = (".norm" in name or "_ln." in name or name.startswith("mlp_head.0.")
               or ".ln." in name)
    if is_norm and leaf == "weight":
        return wave(name, shape, scale=0.1, offset=1.0)
    if leaf in ("bias", "in_proj_bias"):
        return wave(name, shape, scale=0.05)
    if leaf == "W_seq":  # [out, N, R]: contraction over N*R
        fan_in = shape[1] * shape[2]
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
    else:
        fan_in = 1
    return wave(name, shape, scale=1.0 / math.sqrt(max(fan_in, 1)))


def fill_state_dict(sd):
    """New dict with every floating tensor of `sd` replaced by its formula value
    (integer buffers, i.e. the curve tables, are kept)."""
    out = {}
    for k, v in sd.items():
        if torch.is_floating_point(v):
            out[k] = param_value(k, tuple(v.shape)).to(v.dtype)
        else:
            out[k] = v.clone()
    return out


def image_batch(b, c, h, w, tag="img"):
    return wave(tag, (b, c, h, w), scale=1.0, offset=0.05)


def soft_targets(b, classes, lam=0.7):
    """Fixed-lambda MixUp of one-hot labels with the batch rolled by one
    (deterministic stand-in for src/training/train.py:148-160)."""
    y = torch.arange(b) * 3 % classes
    one = torch.nn.functional.one_hot(y, classes).float()
    return lam * one + (1.0 - lam) * one.roll(1, 0)
