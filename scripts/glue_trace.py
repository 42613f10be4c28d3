"""Which Python lines of the eager Stage-1 step launch the small library kernels (copies, fills, cats, adds) between our kernels?
torch.profiler with stacks over ONE step (serial order, as the pipeline's graphs capture it): aten ops that launch a kernel, with
the innermost ws3d_amd / bench frame that issued them.    python scripts/glue_trace.py"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from bench_c3 import C3
from ws3d_amd import fastpath

wl = C3(8, 0, 1, "hdl64", depth=1)
with fastpath.geometry_ahead(False):
    for _ in range(3):
        wl._body()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        wl._body()
        torch.cuda.synchronize()
rows = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 and not ev.name in ("aten::copy_", "aten::fill_", "aten::zero_"):
        continue
    if ev.name in ("aten::mm", "aten::addmm", "aten::_addmm_activation"):
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") for c in ev.cpu_children):
        continue            # count leaves only
    frame = next((f for f in (ev.stack or []) if ("ws3d_amd" in f or "bench" in f) and "glue_trace" not in f), (ev.stack or ["?"])[0] if ev.stack else "?")
    rows[(ev.name, frame.split("/")[-1][:110], str(ev.input_shapes)[:70])] += 1
for (name, frame, shapes), n in sorted(rows.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%2d x %-28s %-112s %s" % (n, name, frame, shapes))
