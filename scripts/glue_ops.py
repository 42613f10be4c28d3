"""Which torch (aten) operators does ONE Stage-1 step issue besides the C-ABI launches, and from which source line?
torch.profiler over one eager step of the bench's c3 body (fast path, eval): operator, calls, device time, innermost ws3d_amd / bench frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from bench_c3 import C3

wl = C3(8, 0, 1, "hdl64", depth=1)
for _ in range(3):
    wl.step(eager=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    wl.step(eager=True)
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 and not ev.kernels:
        continue
    if not ev.kernels:
        continue
    where = next((s for s in ev.stack if "ws3d_amd" in s or "bench" in s), ev.stack[0] if ev.stack else "?")
    key = (ev.name, where.split("/")[-1][:70], str(ev.input_shapes)[:60])
    r = rows.setdefault(key, [0, 0.0, set()])
    r[0] += 1
    r[1] += sum(k.duration for k in ev.kernels)
    r[2].update(k.name[:50] for k in ev.kernels)
tot = 0
for (name, where, shp), (n, us, ks) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%-28s x%-2d %7.1f us  %-70s %s  -> %s" % (name, n, us, where, shp, sorted(ks)[0]))
    tot += n
print("aten operators with a device kernel in one step:", tot)
