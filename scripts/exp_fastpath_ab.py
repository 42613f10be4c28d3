"""A/B of one module attribute of ws3d_amd.fastpath (two exact forms of the same function) on the c3 step: throughput mode (20 in
flight) and latency mode, ABAB on one box, both generators; outputs compared to the first run.
    python scripts/exp_fastpath_ab.py FUSED_COMPACT3_MAX_LDS 0 65536 [steps [reps [kinds]]]"""
import ast, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch
from bench_c3 import C3
from ws3d_amd import fastpath

attr, values = sys.argv[1], [ast.literal_eval(v) for v in sys.argv[2].split(",")] if "," in sys.argv[2] else [ast.literal_eval(v) for v in sys.argv[2:4]]
if "," in sys.argv[2]:          # "v1,v2,v3,.." as ONE argument: any number of values; the remaining arguments move up by one
    sys.argv.insert(3, None)
target = fastpath
tune_name = None
if attr.startswith("tune:"):    # "tune:mlp2_wgs": a launch-geometry knob of the library (compat.tune / ws3d_tune)
    from ws3d_amd import compat as _compat
    tune_name = attr[5:]
    assert tune_name in _compat.TUNE_KEYS, tune_name
elif "." in attr:                 # "compat.CHAIN_WORKGROUPS": an attribute of another module of the package
    import importlib
    modname, attr = attr.rsplit(".", 1)
    target = importlib.import_module("ws3d_amd." + modname)
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 80
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
kinds = sys.argv[6].split(",") if len(sys.argv) > 6 else ["hdl64", "lidar"]
rates = {}
assert tune_name or hasattr(target, attr), attr
model, ref = None, {}


def run(value, kind):
    global model
    if tune_name:
        _compat.tune(tune_name, value)
    else:
        setattr(target, attr, value)
    wl = C3(8, 0, 1, kind, depth=20, model=model)
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
    lat, detail = wl.latency_mode(n=20)
    wl.step(eager=True)
    torch.cuda.synchronize()
    o = wl.last[0]
    res = {k: o[k].detach().clone() for k in ("rpn_cls", "rpn_reg")}
    r0 = ref.setdefault(kind, res)
    diff = max(float((r0[k] - res[k]).abs().max()) for k in res)
    print("%s=%-8s %-6s %.4f ms per batch  %.0f scenes/s   latency %.3f ms   max |output - first run| %.2e" %
          (attr, value, kind, dt / steps * 1e3, wl.scenes() * steps / dt, lat, diff), flush=True)
    rates.setdefault((kind, value), []).append(wl.scenes() * steps / dt)
    wl.release()


for kind in kinds:
    for rep in range(reps):
        for v in values:
            run(v, kind)
for (kind, value), r in rates.items():
    print("mean %-6s %s=%-8s %.0f scenes/s over %d runs" % (kind, attr, value, sum(r) / len(r), len(r)))
