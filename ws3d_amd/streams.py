"""One pool of HIP streams per device for everything in the package that wants a stream of its own.

A HIP stream may claim a hardware queue for as long as it lives, and beyond 23 queues in one process this runtime time-slices
them: every kernel of the process runs 1.4-1.6 x slower (measured).  So Stage1Pipeline's slots and the side
streams of the eager forward pass (ws3d_amd/fastpath.py) draw from the same numbered pool instead of creating streams of their
own: a second pipeline, a re-capture or an eager pass beside a pipeline reuses what exists.  Users of the same pool entry
simply serialise on it."""
from __future__ import annotations

import torch

_POOL = {}      # device index -> [streams]


def pooled_stream(device, j: int) -> torch.cuda.Stream:
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    pool = _POOL.setdefault(index, [])
    while len(pool) <= j:
        pool.append(torch.cuda.Stream(device=torch.device("cuda", index)))
    return pool[j]
