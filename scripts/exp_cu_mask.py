"""EXPERIMENT (not the product): what would the 20-deep pipeline gain if the level-1 sampling kernels ran on RESERVED compute units?
A sampling workgroup needs an EMPTY CU (16 waves x 128 registers), and CUs drain for it; scripts/exp_no_fps.py puts the cost at 0.11 ms
per batch for 0.05 ms of CU time.  Here every pipeline slot runs on a HIP stream created with a CU mask that EXCLUDES `reserved` CUs
(hipExtStreamCreateWithCUMask), the sampling kernel is taken out of the slots' graphs (replaced by a copy of its result, as in
exp_no_fps.py) and issued instead, once per submitted batch, on one more stream whose mask is exactly the reserved CUs.
    python scripts/exp_cu_mask.py [reserved CUs: 16] [steps: 80]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch
from bench_c3 import C3
from ws3d_amd import compat, pn2_ops, streams

reserved = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
hip = ctypes.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count
WORDS = (NCU + 31) // 32


def masked_stream(bits):
    mask = (ctypes.c_uint32 * WORDS)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(WORDS)])
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(WORDS), mask)
    assert rc == 0, "hipExtStreamCreateWithCUMask -> %d" % rc
    return torch.cuda.ExternalStream(h.value, device=torch.device("cuda", 0))


def run(tag, wl, fps_stream=None, fps_args=None):
    for _ in range(3):
        wl.step()
    assert wl.capture(), wl._graph_err
    for _ in range(2):
        wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
        if fps_stream is not None:
            with torch.cuda.stream(fps_stream):
                compat.furthest_point_sampling_gather(*fps_args)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-72s %.4f ms per batch  %.0f scenes/s" % (tag, dt / steps * 1e3, wl.scenes() * steps / dt), flush=True)
    wl.release()


print("device: %d CUs, mask words %d" % (NCU, WORDS))
wl = C3(8, 0, 1, "hdl64", depth=20)
model = wl.model
run("plain (unmasked streams, sampling inside the graphs)", wl)

# masks: the first `reserved` logical CUs for the sampling stream, the rest for the slots
res_bits = set(range(reserved))
rest_bits = set(range(NCU)) - res_bits
orig_pooled = streams.pooled_stream
pool = {}
streams.pooled_stream = lambda device, j: pool.setdefault(j, masked_stream(rest_bits))
import ws3d_amd.pipeline as pl
fps_stream = masked_stream(res_bits)
orig = pn2_ops.furthest_point_sample_gather
cache = {}


def fake(xyz, npoint):
    if xyz.size(1) != 16384:
        return orig(xyz, npoint)
    key = (xyz.data_ptr(), npoint)
    if key not in cache:
        idx, nx = orig(xyz, npoint)
        cache[key] = (idx.clone(), nx.clone())
    idx, nx = cache[key]
    return idx.clone(), nx.clone()


xyz = wl.pts[..., 0:3].contiguous()
idx = torch.empty((8, 4096), dtype=torch.int32, device="cuda")
nx = torch.empty((8, 4096, 3), device="cuda")
fps_args = (8, 16384, 4096, xyz, None, idx, nx)
pn2_ops.furthest_point_sample_gather = fake
run("masked slots (%d CUs), no sampling at all" % len(rest_bits), C3(8, 0, 1, "hdl64", depth=20, model=model))
run("masked slots (%d CUs) + one sampling launch per batch on %d reserved CUs" % (len(rest_bits), reserved), C3(8, 0, 1, "hdl64", depth=20, model=model), fps_stream, fps_args)
pn2_ops.furthest_point_sample_gather = orig
run("masked slots (%d CUs), sampling inside the graphs" % len(rest_bits), C3(8, 0, 1, "hdl64", depth=20, model=model))
