"""What does ONE more (tiny) kernel launch per step cost the 20-deep pipeline?  The c3 step with K extra one-element fills captured into
every slot's graph, ABAB: the slope is the price of a launch, i.e. what merging launches can buy.    python scripts/exp_launch_cost.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch
from bench_c3 import C3
from ws3d_amd import pipeline

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 160
orig_body = pipeline.Stage1Pipeline.body
EXTRA = [0]


def body(self, pts):
    res = orig_body(self, pts)
    pad = getattr(self, "_pad", None)
    if pad is None:
        pad = self._pad = torch.zeros(64, device=pts.device)
    for k in range(EXTRA[0]):
        pad[k % 64:k % 64 + 1].fill_(float(k))          # one tiny dependent launch each
    return res


pipeline.Stage1Pipeline.body = body
model = None
for rep in range(2):
    for extra in (0, 10, 20, 40):
        EXTRA[0] = extra
        wl = C3(8, 0, 1, "hdl64", depth=20, model=model)
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
        print("extra launches per step %3d: %.4f ms per batch  %.0f scenes/s" % (extra, dt / steps * 1e3, 8 * steps / dt), flush=True)
        wl.release()
