"""Isolated timing of the two-layer FP modules of the c3 step (FP0: 131072 rows x 128 -> 128, FP1: 32768 rows x 256 -> 256):
ws3d_qinterp_gemm (round 4) against ws3d_chain_fp (round 6) on arguments captured from a real forward pass; bit-equality checked;
workgroup sweep.   python scripts/r06/bench_fp.py [kind]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch
from bench_c3 import C3
from ws3d_amd import compat, fastpath

kind = sys.argv[1] if len(sys.argv) > 1 else "hdl64"
wl = C3(8, 0, 1, kind, depth=1)
captured = []
orig = compat.chain_fp


def cap(*a, **kw):
    captured.append((a, kw))
    return orig(*a, **kw)


compat.chain_fp = cap
with fastpath.geometry_ahead(False):
    wl.model.rpn_forward({'pts_input': wl.pts})
torch.cuda.synchronize()
compat.chain_fp = orig
assert captured, "the forward pass did not reach chain_fp"


def timeit(fn, iters=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for a, kw in captured:
    q, idx, weight, w2, b2, r2, ticket = a
    rows, C, O = idx.size(0) * idx.size(1), q.size(2), w2.size(1)
    flop = 2.0 * rows * C * O
    kw2 = {k: v for k, v in kw.items()}
    old = compat.qinterp_gemm(q, idx, weight, w2, b2, r2, **kw2)
    t = torch.zeros_like(ticket)
    new = orig(q, idx, weight, w2, b2, r2, t, **kw2)
    print("FP module rows %d, %d -> %d (%s): bit-identical %s" % (rows, C, O, "lin" if kw.get("lin") is not None else "skip", bool(torch.equal(old, new))))
    tz = timeit(lambda: t.zero_())
    to = timeit(lambda: compat.qinterp_gemm(q, idx, weight, w2, b2, r2, **kw2))
    print("  ws3d_qinterp_gemm        %.1f us  %.1f TFLOP/s (%.2f of 157.3)" % (to, flop / to / 1e6, flop / to / 1e6 / 157.3))
    for wgs in (0, 256, 192, 128, 96, 64, 32):
        compat.CHAIN_FP_WORKGROUPS = wgs
        tn = timeit(lambda: (t.zero_(), orig(q, idx, weight, w2, b2, r2, t, **kw2))) - tz
        print("  ws3d_chain_fp wgs=%-4d    %.1f us  %.1f TFLOP/s (%.2f of 157.3)" % (wgs, tn, flop / tn / 1e6, flop / tn / 1e6 / 157.3))
    compat.CHAIN_FP_WORKGROUPS = 0
