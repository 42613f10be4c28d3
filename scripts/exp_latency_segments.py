"""Where the eager latency-mode pass spends its time on the device, WITHOUT a profiler: HIP events recorded on the caller's stream
around the modules of fastpath.backbone_forward (sampling + geometry issue, SA1..SA4, FP4..FP1) and the rest of the step; median of n
passes, for fastpath.BIN_INPUT_AHEAD off / on.    python scripts/exp_latency_segments.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
import torch
from bench_c3 import C3
from ws3d_amd import fastpath

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
marks = []


def mark(tag):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((tag, e))


orig_sa, orig_fp, orig_geo = fastpath.sa_forward, fastpath.fp_forward, fastpath._Geometry.__init__


def sa(sa_, xyz, feats, geo=None, level=0, zeros=None):
    r = orig_sa(sa_, xyz, feats, geo, level, zeros)
    mark("SA%d" % (level + 1))
    return r


def fp(*a, **k):
    r = orig_fp(*a, **k)
    mark("FP")
    return r


def geo_init(self, *a, **k):
    mark("start")
    orig_geo(self, *a, **k)


fastpath.sa_forward, fastpath.fp_forward, fastpath._Geometry.__init__ = sa, fp, geo_init
wl = C3(8, 0, 1, "hdl64", depth=1)
for flag in (False, True, False, True):
    fastpath.BIN_INPUT_AHEAD = flag
    rows = []
    for it in range(n + 3):
        marks.clear()
        wl.step(eager=True)
        mark("end")
        torch.cuda.synchronize()
        if it >= 3:
            rows.append([marks[i][1].elapsed_time(marks[i + 1][1]) for i in range(len(marks) - 1)])
    med = np.median(np.asarray(rows), axis=0)
    tags = ["%s->%s" % (marks[i][0], marks[i + 1][0]) for i in range(len(marks) - 1)]
    print("BIN_INPUT_AHEAD=%s  total %.3f ms: " % (flag, med.sum()) + "  ".join("%s %.3f" % (t, v) for t, v in zip(tags, med)), flush=True)
