"""Deterministic, framework-independent parameter fill.

No checkpoints are reachable offline and a 3 M-parameter state_dict is too large for a test
fixture, so tests and benchmarks materialise weights from (key name, shape, seed) alone:
the same function fills the reference's model in the fixture generator and this package's
model on the GPU box, key for key."""
from __future__ import annotations

import hashlib

import numpy as np
import torch


def seeded_state_dict(shapes: dict, seed: int) -> dict:
    """shapes: {state_dict key: shape tuple} -> {key: tensor}.  Conv weights ~ He-normal,
    BN weight ~ U(0.5,1.5), BN/conv bias ~ U(-0.1,0.1), running_mean ~ N(0,0.1),
    running_var ~ U(0.5,1.5), num_batches_tracked = 0."""
    out = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        h = int.from_bytes(hashlib.sha256(f"{seed}:{key}".encode()).digest()[:8], "little")
        rng = np.random.Generator(np.random.PCG64(h))
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros(shape, dtype=torch.int64)
            continue
        if key.endswith("running_var"):
            a = rng.uniform(0.5, 1.5, shape)
        elif key.endswith("running_mean"):
            a = rng.normal(0.0, 0.1, shape)
        elif key.endswith("bias"):
            a = rng.uniform(-0.1, 0.1, shape)
        elif ".bn." in key and key.endswith("weight"):
            a = rng.uniform(0.5, 1.5, shape)
        else:  # conv / linear weight
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
            a = rng.normal(0.0, np.sqrt(2.0 / max(fan_in, 1)), shape)
        out[key] = torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape))
    return out
