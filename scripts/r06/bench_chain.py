"""Isolated timing of SA2's SharedMLP over compact rows on the headline batch: ws3d_compact_mlp_pair(3) (round 5) against
ws3d_chain_mlp3 (round 6), the arguments captured from a real forward pass of the c3 network on 8 hdl64 scenes; bit-equality of
the pooled rows checked; also the workgroup-count sweep of the new kernel.
    python scripts/r06/bench_chain.py [kind]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch
from bench_c3 import C3
from ws3d_amd import compat, fastpath

kind = sys.argv[1] if len(sys.argv) > 1 else "hdl64"
wl = C3(8, 0, 1, kind, depth=1)
captured = []
orig_chain, orig_pair = compat.chain_mlp3, compat.compact_mlp_pair


def cap_chain(scales, ticket):
    captured.append([dict(s) for s in scales])
    return orig_chain(scales, ticket)


compat.chain_mlp3 = cap_chain
fastpath.PAIR_DISPATCH = "compact"
with fastpath.geometry_ahead(False):
    wl.model.rpn_forward({'pts_input': wl.pts})
torch.cuda.synchronize()
compat.chain_mlp3 = orig_chain
assert captured, "the forward pass did not reach chain_mlp3"
scales = captured[0]
out = scales[0]["out2d"]
totals = [int(s["pairs"][2].item()) for s in scales]
flop = sum(2.0 * t * (3 * 64 + 64 * s["w2t"].size(1) + s["w2t"].size(1) * 128) for t, s in zip(totals, scales))
print("SA2 on %s: compact rows %s, widths %s, useful GFLOP %.3f" % (kind, totals, [(s["o1"], s["w2t"].size(1), s["w3t"].size(1)) for s in scales], flop / 1e9))
ticket = torch.zeros(compat.chain_ticket_ints(), dtype=torch.int32, device="cuda")


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def run_old():
    out.zero_()
    assert orig_pair(3, scales, max_lds=64 * 1024)


def run_new():
    out.zero_(); ticket.zero_()
    assert orig_chain(scales, ticket)


t_zero = timeit(lambda: (out.zero_(), ticket.zero_()))
run_old(); torch.cuda.synchronize(); want = out.clone()
run_new(); torch.cuda.synchronize(); got = out.clone()
print("bit-identical pooled rows:", bool(torch.equal(want, got)), " max |diff| %.3e" % float((want - got).abs().max()))
t_old = timeit(run_old) - t_zero
print("ws3d_compact_mlp_pair(3)   %.1f us  %.1f TFLOP/s useful (%.2f of 157.3)" % (t_old, flop / t_old / 1e6, flop / t_old / 1e6 / 157.3))
for wgs in (0, 256, 192, 128, 96, 64, 32):
    compat.CHAIN_WORKGROUPS = wgs
    t_new = timeit(run_new) - t_zero
    print("ws3d_chain_mlp3 wgs=%-4d   %.1f us  %.1f TFLOP/s useful (%.2f of 157.3)" % (wgs, t_new, flop / t_new / 1e6, flop / t_new / 1e6 / 157.3))
compat.CHAIN_WORKGROUPS = 0
