import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench_c3.py") else os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch
from bench_c3 import C3
from ws3d_amd import fastpath
wl = C3(8, 0, 1, "hdl64", depth=1)
keys = fastpath.primed_compact_scales(wl.model.rpn.backbone_net, wl.pts)
with fastpath.geometry_ahead(False), fastpath.compact_only_scales(keys):
    for _ in range(2):
        wl.model.rpn_forward({'pts_input': wl.pts})
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
        out = wl.model.rpn_forward({'pts_input': wl.pts, 'defer_reg_join': True})
        from ws3d_amd.stage1 import proposals_from_rpn
        proposals_from_rpn(out, wl.cfg, with_pool_boxes=True, with_packed=True)
        torch.cuda.synchronize()
print("arena fallbacks", fastpath.LAST_ARENA_FALLBACKS)
for e in prof.events():
    if "fill" in e.name.lower() or "zero" in e.name.lower():
        print(e.name, e.input_shapes if hasattr(e, "input_shapes") else "", [str(f) for f in (e.stack or [])[:6]])
