"""Deterministic, name-keyed parameter fill.

There is no ``droid.pth`` checkpoint in this environment, so benchmarks and
parity tests use random-init weights.  Filling by parameter NAME (instead of
relying on torch's construction-order RNG) lets the reference's
``UpdateModule`` (droid_slam/droid_net.py:78-108), the oracle's restatement and
the HIP-backed module all receive bit-identical weights via ``load_state_dict``.
"""
import zlib
import numpy as np
import torch


def deterministic_state_dict(module, seed=1234, scale=1.0):
    sd = {}
    for name, p in module.state_dict().items():
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        shape = tuple(p.shape)
        if p.dim() > 1:
            fan_in = int(np.prod(shape[1:]))
        else:
            fan_in = max(int(shape[0]), 1) if p.dim() == 1 else 1
        bound = scale / np.sqrt(fan_in)
        w = rng.uniform(-bound, bound, size=shape).astype(np.float32)
        sd[name] = torch.from_numpy(w).to(dtype=p.dtype)
    return sd


def fill_deterministic(module, seed=1234, scale=1.0):
    module.load_state_dict(deterministic_state_dict(module, seed, scale))
    return module
