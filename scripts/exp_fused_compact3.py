"""A/B of ws3d_pgather_gemm3_compact (the whole SA2-4 SharedMLP of a scale over compact rows in ONE kernel) against the two-kernel form
(pgather_gemm2_compact + gemm_pool_compact): fastpath.FUSED_COMPACT3_MAX_LDS = 0 (off) / 64 KB (SA2) / 160 KB (SA2 + SA3), throughput
mode (20 in flight) and latency mode, ABAB on one box.    python scripts/exp_fused_compact3.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch
from bench_c3 import C3
from ws3d_amd import fastpath

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
model = None
ref = None


def run(tag, max_lds, kind="hdl64"):
    global model, ref
    fastpath.FUSED_COMPACT3_MAX_LDS = max_lds
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
    lat, detail = wl.latency_mode(n=10)
    wl.step(eager=True)
    torch.cuda.synchronize()
    o = wl.last[0]
    res = {k: o[k].detach().clone() for k in ("rpn_cls", "rpn_reg")}
    same = ""
    if kind == "hdl64":
        if ref is None:
            ref = res
        same = "  outputs bit-equal to the first run: %s" % all(torch.equal(ref[k], res[k]) for k in ref)
    print("%-34s %-6s %.4f ms per batch  %.0f scenes/s   latency %.3f ms%s" % (tag, kind, dt / steps * 1e3, wl.scenes() * steps / dt, lat, same), flush=True)
    wl.release()


for rep in range(2):
    run("two kernels per scale (off)", 0)
    run("fused at SA2 (64 KB)", 64 * 1024)
    run("fused at SA2 + SA3 (160 KB)", 160 * 1024)
for lds, tag in ((0, "two kernels per scale (off)"), (64 * 1024, "fused at SA2 (64 KB)"), (160 * 1024, "fused at SA2 + SA3 (160 KB)")):
    run(tag, lds, "lidar")
