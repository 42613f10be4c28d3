"""EXPERIMENT (not the product): the level-1 sampling kernel taken OUT of the slots' graphs and launched eagerly on a few streams of its
own -- normal or HIGH priority -- with the graphs reading its result from static buffers.  Question: does a pending sampling workgroup (it
needs an empty compute unit) get its CU sooner when its queue outranks the others?   python scripts/exp_fps_priority.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch
from bench_c3 import C3
from ws3d_amd import pn2_ops, compat

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
orig = pn2_ops.furthest_point_sample_gather
model = None


def measure(tag, depth, ext_streams, prio):
    global model
    cache, order = {}, []

    def fake(xyz, npoint):
        if xyz.size(1) != 16384:
            return orig(xyz, npoint)
        key = (xyz.data_ptr(), npoint)
        if key not in cache:
            idx, nx = orig(xyz, npoint)
            cache[key] = (idx.clone(), nx.clone(), xyz.clone())
            order.append(key)
        idx, nx, _ = cache[key]
        return idx.clone(), nx.clone()

    if ext_streams:
        pn2_ops.furthest_point_sample_gather = fake
    try:
        wl = C3(8, 0, 1, "hdl64", depth=depth, model=model)
        model = wl.model
        for _ in range(3):
            wl.step()
        assert wl.capture(), wl._graph_err
        for _ in range(depth + 2):
            wl.step()
        torch.cuda.synchronize()
        ext = [torch.cuda.Stream(priority=-1 if prio else 0) for _ in range(ext_streams)]
        # the graphs' sampling inputs in slot order: the eager priming of every slot met a fresh xyz address; the captured ones are the LAST depth keys
        keys = order[-depth:] if ext_streams else []
        slots = wl.pipe.slots

        def step(s):
            if ext_streams:
                j = wl.pipe.submitted % depth
                idx, nx, xyz = cache[keys[j]]
                p = ext[s % ext_streams]
                p.wait_event(slots[j]["done"])
                with torch.cuda.stream(p):
                    compat.furthest_point_sampling_gather(8, 16384, 4096, xyz, None, idx, nx)
                    ev = torch.cuda.Event()
                    ev.record(p)
                slots[j]["stream"].wait_event(ev)
            wl.step()
        for s in range(depth):
            step(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            step(s)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-72s %.4f ms per batch  %.0f scenes/s" % (tag, dt / steps * 1e3, wl.scenes() * steps / dt), flush=True)
        wl.release()
    finally:
        pn2_ops.furthest_point_sample_gather = orig


for rep in range(2):
    measure("plain: sampling inside the graphs, 20 slots", 20, 0, False)
    measure("sampling outside, 2 normal-priority streams, 20 slots", 20, 2, False)
    measure("sampling outside, 2 HIGH-priority streams, 20 slots", 20, 2, True)
    measure("sampling outside, 3 HIGH-priority streams, 18 slots", 18, 3, True)
