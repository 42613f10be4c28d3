"""c3 throughput against the pipeline depth (batches in flight = streams = hardware queues) with graph-only slots: the 24-queue limit
of this runtime was measured with the eager pass's side streams alive (round 3); is it the same for 1 stream per slot?
    python scripts/exp_depth.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import warnings
warnings.simplefilter("ignore")
import torch
from ws3d_amd import streams
streams.POOL_SIZE = 32
from bench_c3 import C3

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
model = None
for rep in range(2):
    for depth in (12, 16, 20, 22, 26, 30):
        wl = C3(8, 0, 1, "hdl64", depth=depth, model=model)
        model = wl.model
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
        print("depth %2d   %.4f ms per batch   %.0f scenes/s" % (depth, dt / steps * 1e3, wl.scenes() * steps / dt), flush=True)
        wl.release()
