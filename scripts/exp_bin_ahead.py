"""A/B (latency mode, eager pass with geometry ahead): fastpath.BIN_INPUT_AHEAD -- the binned copy of the input cloud on the search stream
beside the first level's sampling kernel instead of behind it.    python scripts/exp_bin_ahead.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch
from bench_c3 import C3
from ws3d_amd import fastpath

ref = None
for kind in ("hdl64", "lidar"):
    wl = C3(8, 0, 1, kind, depth=1)
    for rep in range(3):
        for flag in (False, True):
            fastpath.BIN_INPUT_AHEAD = flag
            lat, detail = wl.latency_mode(n=30)
            wl.step(eager=True)
            torch.cuda.synchronize()
            o = {k: wl.last[0][k].detach().clone() for k in ("rpn_cls", "rpn_reg")}
            if rep == 0 and not flag:
                ref = o
            print("%-6s BIN_INPUT_AHEAD=%-5s latency %.3f ms   outputs bit-equal: %s" % (kind, flag, detail["eager_side_streams_ms"], all(torch.equal(ref[k], o[k]) for k in o)), flush=True)
    wl.release()
