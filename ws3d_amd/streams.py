"""One pool of HIP streams per device for everything in the package that wants a stream of its own.

A HIP stream may claim a hardware queue for as long as it lives, and beyond 23 queues in one process this runtime time-slices
them: every kernel of the process runs 1.4-1.6 x slower (measured).  So Stage1Pipeline's slots and the side
streams of the eager forward pass (ws3d_amd/fastpath.py) draw from the same numbered pool instead of creating streams of their
own: a second pipeline, a re-capture or an eager pass beside a pipeline reuses what exists.  Users of the same pool entry
simply serialise on it.

Pipeline slots count up from entry 0, the eager pass's three side streams count DOWN from entry POOL_SIZE - 1 and skip any entry
that is capturing a hipGraph at that moment (``side_streams``): up to a depth of 17 the two never meet, and at the bench's depth
of 20 (20 slots + the null stream + nothing else is what the 24-queue limit leaves room for) an eager pass beside a capture takes
other entries instead of being swallowed by -- or aborting -- the capture.  Two Stage1Pipelines on one device share slot streams:
their batches serialise slot by slot, and latency figures taken beside a live pipeline are coupled to it."""
from __future__ import annotations

import torch

POOL_SIZE = 20      # Stage1Pipeline's largest default depth
_POOL = {}          # (device index, j) -> stream, created on first use


def pooled_stream(device, j: int) -> torch.cuda.Stream:
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    st = _POOL.get((index, j))
    if st is None:
        st = _POOL[(index, j)] = torch.cuda.Stream(device=torch.device("cuda", index))
    return st


def _capturing(st: torch.cuda.Stream) -> bool:
    with torch.cuda.stream(st):
        return torch.cuda.is_current_stream_capturing()


_PASS = {}          # (device index, main stream handle) -> the side streams the current forward pass of that stream resolved


def pass_streams(main: torch.cuda.Stream, count: int = 3, fresh: bool = False):
    """The side streams of ONE forward pass, resolved once: the first caller of a pass (fastpath._Geometry, `fresh=True`) picks them by
    the pool's capture state at that moment, later callers of the same pass (fastpath.rpn_forward's heads) get the SAME streams even if
    a capture started or ended on a pool entry in between -- two independent picks could differ and claim extra hardware queues
    (ADVICE round 4)."""
    key = (main.device.index, main.cuda_stream)
    got = None if fresh else _PASS.get(key)
    if got is None or len(got) < count:
        got = _PASS[key] = side_streams(main, count)
    return got[:count]


def side_streams(main: torch.cuda.Stream, count: int = 3):
    """`count` pool streams for side work of a pass whose own stream is `main`: from the top of the pool downwards, never `main`
    itself, never a stream that is being captured"""
    picked, j = [], POOL_SIZE - 1
    while len(picked) < count and j >= 0:
        st = pooled_stream(main.device, j)
        j -= 1
        if st.cuda_stream != main.cuda_stream and not _capturing(st):
            picked.append(st)
    if len(picked) < count:
        raise RuntimeError("no free side stream in the pool (every entry is the caller's stream or capturing)")
    return tuple(picked)
