"""EXPERIMENT (not a measurement of the product): throughput mode of the c3 step with the level-1 sampling kernel replaced by a copy of
its precomputed result -- what the kernel costs the 20-deep pipeline beyond the CU time it uses (it needs EMPTY compute units: 16
waves x 128 registers fill one).  python scripts/exp_no_fps.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch
from bench_c3 import C3
from ws3d_amd import pn2_ops, fastpath

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80


def run(tag, wl):
    for _ in range(3):
        wl.step()
    assert wl.capture(), wl._graph_err
    for _ in range(2):
        wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-40s %.4f ms per batch  %.0f scenes/s" % (tag, dt / steps * 1e3, wl.scenes() * steps / dt), flush=True)
    wl.release()


wl = C3(8, 0, 1, "hdl64", depth=20)
run("plain", wl)
orig = pn2_ops.furthest_point_sample_gather
cache = {}


def fake(xyz, npoint):
    if xyz.size(1) != 16384:
        return orig(xyz, npoint)
    key = (xyz.data_ptr(), npoint)
    if key not in cache:                      # first sight (eager priming): the real kernel, result kept
        idx, nx = orig(xyz, npoint)
        cache[key] = (idx.clone(), nx.clone())
    idx, nx = cache[key]
    return idx.clone(), nx.clone()            # a copy kernel in the graph instead of the sampling kernel


pn2_ops.furthest_point_sample_gather = fake
run("level-1 sampling replaced by a copy", C3(8, 0, 1, "hdl64", depth=20, model=wl.model))
pn2_ops.furthest_point_sample_gather = orig
run("plain again", C3(8, 0, 1, "hdl64", depth=20, model=wl.model))
