#!/usr/bin/env python
"""bench.py -- WS3D hot-path benchmark on MI355X (driver contract: one JSON line on rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5|s2|t1] [--batch B]

With no flags it measures BASELINE.json's headline: workload **c3** = configs[2], the full Stage-1 RPN
forward + proposal NMS + roipool3d at batch 8 scenes/GPU (`value` = throughput mode, `latency_mode` =
one batch in flight, both in the same line), and -- outside the timed region -- the **c2** block
(configs[1]: FPS 16384->4096 + fused ball_query/group, the "FPS+group HBM GB/s" half of the metric)
with the `roofline` of the path's dominant kernel.

N > 1 runs one rank per GPU over RCCL, either way of starting it: under the driver's ``python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`` (RANK / WORLD_SIZE from the environment), or as plain ``python bench.py --gpus N``,
which starts its own N ranks (`_self_launch`).  A world size that differs from --gpus, or more RCCL ranks than devices, exits
non-zero; the line carries ``config.ranks_seen`` and the communicator's size.  Scenes are independent, so the path shards with NO data-path
collective: every rank processes its own ``batch`` scenes ("weak" scaling); the only exchange
is the fixed-shape all-gather of per-scene proposals in the c3 workload.

A "step" = one pass of the hot path over one batch of synthetic KITTI-shaped scenes that are
already resident in HBM:
  c2 : furthest_point_sample 16384->4096 (+fused gather) and fused ball_query+group
       (r=0.1, nsample=64, 3 xyz + 1 feature channel)             [BASELINE.json configs[1]]
  c3 : full Stage-1 RPN forward (Pointnet2MSG 4 SA + 4 FP + heads) + proposal NMS +
       roipool3d, batch 8 scenes/GPU                               [BASELINE.json configs[2]]
  c5 : N=65536, 512 proposals: roipool3d + rotated NMS, batch 8    [BASELINE.json configs[4]]
  s2 : the Stage-2 (RCNN) set-abstraction shapes of the same ops: 800 RoI clouds of 512 points,
       128 channels
  t1 : one Stage-1 RPN training iteration, batch 8 (bench_t1.py)

Extra objects on the JSON line (tier contract): "roofline" for the dominant kernel (FPS:
algorithmic bytes A_model = (M-1)*N*12 + M*4 per scene) with the per-launch
duration measured live with HIP events on the launch stream, and "cpu_baseline" (the CPU
oracle port, OpenMP over scenes, timed on this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4): streams that share a
# queue serialise.  The c3 pipeline keeps 20 batches in flight on 20 streams, so the cap is raised
# BEFORE the runtime starts (measured: 2,106 scenes/s with 4 queues / 3 in flight, 3,850 with 32 / 20).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver


def _self_launch():
    """``python bench.py --gpus N`` with N > 1 and no launcher around it (no WORLD_SIZE / RANK in the environment): start the
    N ranks here -- ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` on 127.0.0.1 and a free port, one rank per
    GPU -- pass their output through (rank 0 prints the one JSON line) and exit with their status.  Under a launcher
    (the driver's torch.distributed.run) this does nothing; a world size that differs from --gpus is an error in
    dist_setup(), never a silent one-GPU measurement.  Runs before torch is imported: the parent never touches HIP."""
    gpus, argv = 1, sys.argv[1:]
    for i, a in enumerate(argv):
        try:
            if a == "--gpus" and i + 1 < len(argv):
                gpus = int(argv[i + 1])
            elif a.startswith("--gpus="):
                gpus = int(a.split("=", 1)[1])
        except ValueError:
            return                                      # argparse reports it
    if gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ, WS3D_BENCH_SELF_LAUNCHED="1")
    if os.environ.get("WS3D_BENCH_LAUNCH_DRYRUN"):      # tests/test_bench_launch.py: the command, without running it
        print(json.dumps({"self_launch": cmd}))
        sys.exit(0)
    print("[bench] --gpus %d without a launcher: starting %d ranks: %s" % (gpus, gpus, " ".join(cmd[1:8])), file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


if __name__ == "__main__":
    _self_launch()
import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from bench_lib import *  # noqa: E402,F401,F403  (roofs, byte models, the c2 / c5 / s2 workloads, counter passes, kernel rows)
from bench_lib import _traffic_file  # noqa: E402,F401


def dist_setup(gpus):
    """-> (world, rank, local device, communicator description).  The world size MUST equal --gpus: a mismatch exits non-zero
    (``python bench.py --gpus N`` launches its own N ranks, _self_launch; under a launcher WORLD_SIZE says what was started)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    comm = {"backend": None, "size": 1, "ranks_seen": 1, "launcher": "none (single process)"}
    if world != gpus:
        if rank == 0:
            print(f"[bench] error: --gpus {gpus} but the launcher started WORLD_SIZE={world} rank(s); refusing to report a line for "
                  f"another world size than the one asked for", file=sys.stderr, flush=True)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # WS3D_DIST_BACKEND=gloo: control-flow test of the multi-rank path on a box with fewer GPUs
        # than ranks (ranks then share devices); the driver's runs use nccl (= RCCL over xGMI)
        backend = os.environ.get("WS3D_DIST_BACKEND", "nccl")
        ndev = torch.cuda.device_count()
        if backend == "nccl" and world > ndev:
            if rank == 0:
                print(f"[bench] error: --gpus {world} over RCCL needs {world} devices, this box shows {ndev} "
                      f"(WS3D_DIST_BACKEND=gloo is the control-flow test mode that lets ranks share a device)", file=sys.stderr, flush=True)
            sys.exit(3)
        local = local % max(ndev, 1)
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # every rank adds one: what the communicator actually spans, next to what it says about itself
        seen = torch.ones(1, dtype=torch.int32, device="cuda")
        dist.all_reduce(seen)
        devs = [None] * world
        dist.all_gather_object(devs, "%s:%d" % (os.uname().nodename, local))
        comm = {"backend": "rccl (torch 'nccl')" if backend == "nccl" else backend, "size": dist.get_world_size(), "ranks_seen": int(seen.item()),
                "devices": devs, "launcher": "bench.py self-launch (torch.distributed.run)" if os.environ.get("WS3D_BENCH_SELF_LAUNCHED")
                else "external (WORLD_SIZE/RANK from the environment)"}
        if comm["size"] != gpus or comm["ranks_seen"] != gpus:
            if rank == 0:
                print(f"[bench] error: communicator spans {comm['size']} rank(s), {comm['ranks_seen']} answered, --gpus {gpus}", file=sys.stderr, flush=True)
            sys.exit(4)
    else:
        torch.cuda.set_device(0)
    return world, rank, local, comm


def barrier_sync(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


LINE_BUDGET = 4096          # bytes of the ONE stdout line (the driver parses the tail of stdout: round 4's 24.8 KB line was not parsed)
DETAIL_FILE = "bench_detail.json"


def _short(s, n=200):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _num(x, nd=6):
    """floats to `nd` significant digits: the line is read by a parser and by people, not diffed"""
    if isinstance(x, float):
        return float("%.*g" % (nd, x))
    if isinstance(x, dict):
        return {k: _num(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, nd) for v in x]
    return x


def c3_step_roofline(out, kernels, wl, kind):
    """The tier contract's `roofline` for the c3 line: it describes the TIMED REGION of this line (one Stage-1 step over one batch,
    throughput mode), not another launch.  The step is nearer its matrix roof than its HBM roof, so the top-level
    bound is "mfma": useful f32 multiply-adds of the step (counted from the layer widths and the distinct pairs measured on the batch,
    bench_c3.matrix_work) / ms_per_step / the dense f32 MFMA peak.  `traffic` = HBM bytes per step from the committed rocprofv3 --pmc
    passes of the same step (profiles/traffic_c3.json: FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 corrections) when they were
    taken at this batch and generator, else null; `hbm` = that traffic against 8 TB/s; `dominant_kernel` = the level-1 sampling
    kernel AT THIS LINE'S BATCH, duration from HIP events around its launches in the eager steps after the timed region."""
    import bench_c3
    mw = out.get("throughput_mode", {}).get("matrix_roofline")
    tot = load_traffic("c3:_total_hbm_bytes_per_step")
    same = wl.scenes() == traffic_batch("c3:x") and traffic_kind("c3:x") == kind
    traffic = float(tot) if (tot and same) else None
    sec = out["ms_per_step"] * 1e-3
    a_min = 88550656 * wl.scenes()                 # compulsory bytes of the custom ops of one scene's Stage-1 forward
    r = {"measured_in": "the timed region of this line (%s, batch %d per GPU, %s)" % (wl.name, wl.scenes(), kind),
         "bound": "mfma", "unit": "TFLOP/s", "peak": bench_c3.FP32_MFMA_PEAK_TFLOPS,
         "achieved": mw["achieved"] if mw else None, "frac": mw["frac"] if mw else None,
         "gflop_per_step": mw["gflop_per_batch"] if mw else None, "traffic": traffic,
         "hbm": {"peak_GBps": HBM_PEAK / 1e9, "traffic_GBps": traffic / sec / 1e9 if traffic else None,
                 "frac": traffic / sec / HBM_PEAK if traffic else None,
                 "a_min_bytes_per_step": a_min, "a_min_frac": a_min / sec / HBM_PEAK,
                 "traffic_source": "profiles/traffic_c3.json (rocprofv3 --pmc, eager step)" if traffic else
                                   ("stale" if traffic_state("c3:x") == "stale" else
                                    "null: the committed --pmc pass is for batch %s / %s" % (traffic_batch("c3:x"), traffic_kind("c3:x")))}}
    if traffic is None and traffic_state("c3:x") == "stale":
        r["traffic_source"] = "stale"
    fps = next((k for k in kernels if k.get("bound") == "valu"), None)
    if fps is not None:
        us = fps.get("us_per_fps_step")
        e8 = fps_valu_pmc(wl.scenes(), kind)
        share = None if e8 is None else e8["sq_insts_valu_per_launch"] * 64.0 / (fps["ms_per_step"] * 1e-3) / (VALU_PEAK * wl.scenes() / 256.0)
        r["dominant_kernel"] = {"name": "fps_rounds2_kernel (level-1 furthest_point_sample 16384 -> 4096 + gather)", "measured_in": "this line's batch, eager steps, HIP events",
                                "ms_per_launch": fps["ms_per_step"] / max(fps["launches_per_step"], 1), "workgroups": wl.scenes(),
                                "us_per_sample": us, "clk_per_sample": None if us is None else us * 2400.0,
                                "bound": "cross-lane chain (neither HBM nor VALU issue)", "valu_frac_of_occupied_CUs": share,
                                "valu_frac_of_chip": fps.get("valu_frac"), "hbm_bytes_per_launch": (fps.get("traffic_bytes_per_launch") or {}).get("hbm_bytes")}
    return r


def compact_line(out, detail_path):
    """The ONE line the driver parses: the contract keys, `roofline`, `cpu_baseline` and a dozen scalars -- everything else
    (kernels[], the c2 block, list_fill, the other generator, notes) lives in `detail_path` and on stderr."""
    cfg = out["config"]
    comm = cfg.get("communicator") or {}
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype")}
    line["data"] = _short(out["data"], 160)
    line["config"] = {"workload": cfg["workload"], "batch_per_gpu": cfg["batch_per_gpu"], "n_points": cfg["n_points"],
                      "ranks_seen": cfg["ranks_seen"], "generator": out.get("generator"), "backend": comm.get("backend"),
                      "communicator_size": comm.get("size", 1),
                      "launcher": _short(comm.get("launcher", ""), 60)}
    for k in ("launch", "exchange", "hw_queues", "pipeline_depth", "untimed_before_clock", "m_points", "nsample", "radius", "proposals", "channels", "sampled"):
        if k in cfg and not isinstance(cfg[k], (dict, list)):
            line["config"][k] = _short(cfg[k], 120) if isinstance(cfg[k], str) else cfg[k]
    roof = dict(out["roofline"])
    for k in ("note", "frac_is"):
        if k in roof:
            roof[k] = _short(roof[k], 120)
    roof.pop("note", None)
    roof["kernel"] = _short(roof.get("kernel", roof.get("measured_in", "")), 160) if "kernel" in roof else None
    if roof["kernel"] is None:
        roof.pop("kernel")
    line["roofline"] = roof
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = {k: (_short(v, 280) if isinstance(v, str) else v) for k, v in cb.items() if k not in ("host", "ms_per_scene_by_part", "search_only")}
        if isinstance(cb.get("search_only"), dict):
            line["cpu_baseline"]["search_only_value"] = cb["search_only"].get("value")
        parts = cb.get("ms_per_scene_by_part")
        if isinstance(parts, dict):      # the two NMS figures side by side (the literal mask is a straw man as a CPU algorithm): ms per scene
            line["cpu_baseline"]["nms_ms_per_scene"] = {("literal_mask" if k.startswith("nms (literal") else "lazy_sweep"): v for k, v in parts.items() if k.startswith("nms (")}
        if isinstance(cb.get("host"), dict):
            line["cpu_baseline"]["host_cpu"] = _short(cb["host"].get("cpu", ""), 60)
    lat, c2 = out.get("latency_mode"), out.get("c2")
    if lat:
        line["latency_ms"], line["latency_scenes_per_s"] = lat["ms_per_batch"], lat["value"]
    if out.get("generator"):
        line["value_" + out["generator"]] = out["value"]
    og = out.get("other_generator")
    if og:
        line["value_" + og["generator"]] = og["value"]
        line["latency_ms_" + og["generator"]] = og.get("latency_ms_per_batch")
    if "value_all_rows" in out:
        line["value_all_rows"] = out["value_all_rows"]
    if "value_reference_layout" in out:       # CHANNELS_LAST_FASTPATH = False, eager, one batch in flight (beside latency_scenes_per_s)
        line["value_reference_layout"] = out["value_reference_layout"]
    if "ops_hbm_frac" in out:
        line["ops_hbm_frac"] = out["ops_hbm_frac"]
    if "value_steady" in out:                 # 160 more steps of the same replay loop, outside the contract's timed region
        line["value_steady"] = out["value_steady"]
    if isinstance(out.get("step_ms_percentiles"), dict):
        line["step_ms_percentiles"] = out["step_ms_percentiles"]
    if c2:
        line["c2_batch"], line["c2_scenes_per_s"] = c2["batch_per_gpu"], c2["scenes_per_s_per_gpu"]
        for k in c2["kernels"]:
            if k.get("bound") == "valu":
                line["c2_fps_ms"], line["c2_fps_us_per_sample"] = k["ms_per_step"], k.get("us_per_sample")
            elif k.get("bound") == "hbm":
                line["c2_query_group_ms"], line["c2_query_group_hbm_frac"] = k["ms_per_step"], k.get("frac_of_8TBps")
        pg = c2.get("path_gbps_per_gpu") or {}
        line["c2_a_min_hbm_frac"] = pg["a_min"] * 1e9 / HBM_PEAK if pg.get("a_min") else None
        line["c2_a_model_effective_GBps"] = pg.get("a_model")
        if isinstance(c2.get("cpu_baseline"), dict) and "value" in c2["cpu_baseline"]:
            line["c2_cpu_scenes_per_s"] = c2["cpu_baseline"]["value"]
    line["detail"] = os.path.basename(detail_path) if detail_path else None
    line = _num(line)
    text = json.dumps(line)
    if len(text) > LINE_BUDGET:                     # never lose the line to its own size again: drop the optional scalars first
        for k in [k for k in line if k.startswith(("c2_", "latency_ms_", "value_all"))]:
            line.pop(k)
        line["roofline"].pop("dominant_kernel", None)
    return line


def c2_block(batch, rank, kind, steps=10, warmup=2):
    """BASELINE configs[1] measured outside the timed region of the headline run: the 'FPS+group HBM GB/s'
    half of the metric, and the chip-filling launch of the path's dominant kernel (FPS)"""
    wl = C2(batch, rank, kind)
    for _ in range(warmup):
        wl.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step(timed=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_gpu = batch * steps / dt
    kernels = finish_kernel_rows(wl.kernel_table(), batch, kind)
    blk = {"workload": wl.name, "batch_per_gpu": batch, "steps": steps, "scenes_per_s_per_gpu": per_gpu, "ms_per_step": dt / steps * 1e3,
           "config": wl.config(), "kernels": kernels, "path_gbps_per_gpu": wl.path_gbps(per_gpu),
           "step_ms_percentiles": step_percentiles(wl)}
    return wl, blk


def c3_side_runs(wl, args, value, latency):
    """c3, one GPU, after the timed region: the same step (a) with the SharedMLPs over ALL m * nsample rows (the distinct-pairs form is
    exact, but how much it saves is a property of the data: `list_fill`), (b) on the other scene generator.  Every block is a fresh
    Stage1Pipeline over the same weights (captured graphs, same depth, same number of timed steps) -> dict merged into the line."""
    from bench_c3 import C3
    from ws3d_amd import fastpath

    def run(w):
        for _ in range(args.warmup):
            w.step()
        if not w.capture():
            return None
        for _ in range(max(2, w.depth)):
            w.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            w.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        lat, detail = w.latency_mode(n=10)
        w.release()
        return {"value": w.scenes() * args.steps / dt, "unit": "scenes/s", "ms_per_batch": dt / args.steps * 1e3, "latency_ms_per_batch": lat,
                "latency_launch": min(detail, key=detail.get)[:-3], "sharedmlp_rows": w.rows_mode, "steps": args.steps}

    out = {"list_fill": {"generator": wl.kind, "scales": wl.list_fill(),
                         "note": "distinct (centre, sample) pairs / all m * nsample list entries per ball-query scale on the timed batch; the "
                                 "SharedMLP of a scale runs over the distinct pairs iff its fill <= %.2f, decided on the device per batch (in the "
                                 "graphs: only where the priming batch's fill is above %.2f of that)" % (fastpath.COMPACT_MAX_FILL, fastpath.PRIMED_MARGIN)}}
    wl.release()
    # (c) what a drop-in caller of the reference layout gets: stage1.CHANNELS_LAST_FASTPATH = False -- the
    # network's modules as the reference composes them (pointnet2_modules.py:19-55,116-156 on (B, C, N) tensors: QueryAndGroup's
    # ball_query -> group -> sub -> group -> cat through the operators' API names, Conv2d SharedMLPs on the library, max_pool), the
    # same proposal stage and roipool3d behind it; eager launches, one batch in flight, median of 20 batches
    from ws3d_amd import stage1
    try:
        stage1.CHANNELS_LAST_FASTPATH = False
        w0 = C3(wl.B, wl.rank, 1, wl.kind, depth=1, model=wl.model)
        for _ in range(3):
            w0.step(eager=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            w0.step(eager=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        med = float(np.median(ts))
        out["reference_layout"] = {"value": w0.scenes() / med, "unit": "scenes/s", "ms_per_batch": med * 1e3, "batches_in_flight": 1, "batches_timed": 20,
                                   "what": "stage1.CHANNELS_LAST_FASTPATH = False: the reference's module composition on (B, C, N) tensors through "
                                           "the API-named operators; eager, one batch in flight (compare with latency_ms, not with value)"}
        out["value_reference_layout"] = out["reference_layout"]["value"]
    except Exception as e:      # a diagnostic figure: never a reason to lose the line
        out["reference_layout"] = {"error": repr(e)}
    finally:
        stage1.CHANNELS_LAST_FASTPATH = True
    saved = fastpath.COMPACT_PAIRS
    try:
        fastpath.COMPACT_PAIRS = False
        r = run(C3(wl.B, wl.rank, 1, wl.kind, depth=wl.depth, model=wl.model))
    finally:
        fastpath.COMPACT_PAIRS = saved
    if r is not None:
        out["all_rows"] = r
        out["value_all_rows"] = r["value"]
    other = "lidar" if wl.kind != "lidar" else "hdl64"
    w2 = C3(wl.B, wl.rank, 1, other, depth=wl.depth, model=wl.model)
    fill2 = w2.list_fill()
    r2 = run(w2)
    if r2 is not None:
        r2["list_fill"] = fill2
        try:
            fastpath.COMPACT_PAIRS = False
            r3 = run(C3(wl.B, wl.rank, 1, other, depth=wl.depth, model=wl.model))
        finally:
            fastpath.COMPACT_PAIRS = saved
        if r3 is not None:
            r2["value_all_rows"], r2["all_rows_latency_ms_per_batch"] = r3["value"], r3["latency_ms_per_batch"]
        out["other_generator"] = dict({"generator": other}, **r2)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps; default 80 for c3 (four rounds over the 20 slots: the fill and the drain of the pipeline are inside the timed region), 40 otherwise")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "c5", "s2", "t1", "ops"],
                    help="c3 (default) = BASELINE.json's headline: Stage-1 RPN forward + NMS + roipool3d, batch 8/GPU, with the c2 block")
    ap.add_argument("--batch", type=int, default=None, help="scenes per GPU (c3/c5 default 8, c2 default 512, s2 default 800)")
    ap.add_argument("--c2-batch", type=int, default=512, help="c3: scenes per launch of the embedded c2 block (0 = skip it)")
    ap.add_argument("--kind", default="hdl64", choices=["hdl64", "lidar", "uniform"],
                    help="synthetic scene generator: hdl64 (default) = ray-cast HDL-64E scan sub-sampled like the reference's dataset (KITTI's point "
                         "density); lidar = SURVEY 8d's sparse statistical model (the headline of rounds 1-2); uniform = iid in the scope box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=5.0, help="wall time of each CPU-oracle sample (the default line holds two)")
    ap.add_argument("--no-side-runs", action="store_true",
                    help="c3: skip the blocks measured after the timed region (all_rows = SharedMLPs over all m*nsample rows; the other generator)")
    ap.add_argument("--no-graph", action="store_true", help="c3: time eager launches instead of hipGraph replay")
    ap.add_argument("--detail", default=None, help="where the full record goes (kernels[], the c2 block, list_fill, the other generator, notes); "
                                                   "default bench_detail.json beside bench.py; '' = do not write it")
    ap.add_argument("--full-line", action="store_true", help="print the full record on stdout instead of the compact line (scripts/ that read kernels[])")
    ap.add_argument("--no-prefetch", action="store_true", help="t1: sample inside the step instead of one step ahead")
    ap.add_argument("--pipeline-depth", type=int, default=None,
                    help="c3: batches in flight (one HIP stream + graph each); default 20 on one GPU, 16 with --gpus > 1 (headroom for the "
                         "collective library's own hardware queues: beyond 23 queues per process the runtime time-slices, DESIGN.md 5.6)")
    ap.add_argument("--settle", type=float, default=1.0,
                    help="c3, graph mode: seconds of untimed replays between the priming replays and the timed region (the device's sustained state; 0: none)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 80 if args.workload == "c3" else 40

    from ws3d_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing
    world, rank, local, comm = dist_setup(args.gpus)

    if args.workload == "c3":
        from bench_c3 import C3
        if args.pipeline_depth is None:
            args.pipeline_depth = 20 if world == 1 else 16
        wl = C3(args.batch or 8, rank, world, args.kind, depth=args.pipeline_depth)
    elif args.workload == "c5":
        wl = C5(args.batch or 8, rank, args.kind)
    elif args.workload == "s2":
        wl = S2(args.batch or 800, rank, args.kind)
    elif args.workload == "t1":
        from bench_t1 import T1
        wl = T1(args.batch or 8, rank, world, args.kind, prefetch=not args.no_prefetch)
    elif args.workload == "ops":
        from bench_ops import OPS
        wl = OPS(args.batch or 8, rank, args.kind)
    else:
        wl = C2(args.batch or 512, rank, args.kind)

    for _ in range(args.warmup):
        wl.step()
    use_graph = hasattr(wl, "capture") and not args.no_graph and wl.capture()
    if use_graph:
        # every slot's graph replayed once, untimed, before the clock starts: the timed region then begins
        # with all `depth` batches' buffers, code objects and library workspaces touched -- steady state from its first step on
        for _ in range(max(2, getattr(wl, "depth", 2))):
            wl.step()
        # ... and the same replay loop for --settle seconds more, still untimed: the first bench process on a box that has been idle measured
        # 8,120 scenes/s in the 19 ms of the driver's 20 steps and 8,515 over the 160 steps right behind them (round 6, same process) --
        # the device had not reached the clocks it sustains.  The clock starts on a device in the state the loop keeps it in.
        if args.settle > 0 and args.workload == "c3":
            barrier_sync(world)
            t_s = time.perf_counter()
            while time.perf_counter() - t_s < args.settle:
                for _ in range(getattr(wl, "depth", 2)):
                    wl.step()
                barrier_sync(world)
    barrier_sync(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(timed=not use_graph)
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)

    # ---- everything below is outside the timed region
    total_scenes = wl.scenes() * world * args.steps
    value = total_scenes / dt
    ms_per_step = dt / args.steps * 1e3
    latency = None
    if hasattr(wl, "latency_mode"):
        # the same step with ONE batch in flight (submit, exchange, wait): what a caller that needs the
        # proposals of this batch before it submits the next one sees
        lat, lat_detail = wl.latency_mode(n=20)
        lat_ms = max_over_ranks(lat, world)
        latency = {"batches_in_flight": 1, "ms_per_batch": lat_ms, "value": wl.scenes() * world / (lat_ms * 1e-3),
                   "unit": getattr(wl, "unit", "scenes/s"), "batches_timed": 20, "statistic": "median",
                   "launch": min(lat_detail, key=lat_detail.get)[:-3], "rank0_ms_by_launch": lat_detail}
    steady = None
    if use_graph and args.workload == "c3" and args.steps >= 20 and not args.no_side_runs:
        # the same replay loop once more over 160 steps (eight rounds over the slots), OUTSIDE the contract's timed region: the
        # driver's flags time one pipeline fill (20 steps at depth 20); this says what the loop sustains
        barrier_sync(world)
        t1 = time.perf_counter()
        for _ in range(160):
            wl.step()
        barrier_sync(world)
        dt1 = max_over_ranks(time.perf_counter() - t1, world)
        steady = {"value": wl.scenes() * world * 160 / dt1, "ms_per_step": dt1 / 160 * 1e3, "steps": 160}
    if use_graph:
        # per-kernel HIP-event table from EAGER steps (events cannot be recorded inside a graph): >= 50 of them at the driver's flags
        # (median + p10 / p90 over >= 50 iterations); `value` above is the graph-replay throughput
        for _ in range(50 if args.steps >= 20 else min(args.steps, 5)):
            wl.step(timed=True)
        torch.cuda.synchronize()
    if os.environ.get("WS3D_BENCH_DUMP") and hasattr(wl, "dump"):
        wl.dump(os.environ["WS3D_BENCH_DUMP"])
    kernels = finish_kernel_rows(wl.kernel_table(), wl.scenes(), getattr(wl, 'kind', args.kind)) if rank == 0 else None
    side = {}
    if args.workload == "c3" and rank == 0:
        # distinct pairs per ball-query scale on the timed batch: what the matrix roofline of the step is counted from
        side["list_fill"] = {"generator": wl.kind, "scales": wl.list_fill()}
    if args.workload == "c3" and world == 1 and not args.no_side_runs and use_graph:
        side.update(c3_side_runs(wl, args, value, latency))
    c2wl = c2blk = None
    if args.workload == "c3" and args.c2_batch > 0:
        c2wl, c2blk = c2_block(args.c2_batch, rank, args.kind)
    barrier_sync(world)

    if rank == 0:
        dom = max((k for k in kernels if k["launches_per_step"] > 0), key=lambda k: k["ms_per_step"])
        per_gpu = value / world
        out = {
            "metric": getattr(wl, "metric", HEADLINE_METRIC), "value": value, "unit": getattr(wl, "unit", "scenes/s"),
            "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "generator": getattr(wl, 'kind', args.kind),
            "data": f"synthetic ({getattr(wl, 'kind', args.kind)}, seeded scenes: 1000*config + 100000*slot + scene, random-init weights)",
            "config": dict({"workload": wl.name, "batch_per_gpu": wl.scenes(), "n_points": N_PTS, "ranks_seen": comm["ranks_seen"],
                            "communicator": comm, "pipeline_depth": getattr(wl, "depth", None),
                            "multi_gpu_evidence": "no 1 -> 8 GPU scaling curve has been measured for this build (no multi-GPU node was available to it): "
                                                  "the N > 1 path is covered by 2- and 8-rank runs of this file with every rank on ONE GPU (gloo "
                                                  "exchange, tests/test_bench_contract.py) and by RCCL at world size 1"},
                           **wl.config()),  # (c5 overrides n_points)
        }
        if use_graph and args.workload == "c3":
            out["config"]["untimed_before_clock"] = "%d warmup steps, %d priming replays (one per slot), %.1f s of replays (--settle)" % (
                args.warmup, max(2, getattr(wl, "depth", 2)), args.settle)
        if steady is not None:
            out["steady_state"] = steady
            out["value_steady"] = steady["value"]
        if latency is not None:
            out["throughput_mode"] = {"batches_in_flight": getattr(wl, "depth", 1), "ms_per_batch": ms_per_step, "value": value,
                                      "unit": out["unit"]}
            out["latency_mode"] = latency
        if c2blk is not None:
            # BASELINE configs[1] beside the headline: its own block with its own roofline (the chip-filling launch of the sampling
            # kernel); the LINE's roofline below describes the line's timed region
            c2dom = max(c2blk["kernels"], key=lambda k: k["ms_per_step"])
            c2blk["roofline"] = roofline_of(c2dom, "c2 block of this run (batch %d per launch)" % c2blk["batch_per_gpu"])
            out["c2"] = c2blk
        if args.workload == "ops":
            # the copy operator FURTHEST from its HBM roof: SURVEY 8(d) applies the 40 % target to each of them literally
            dom = min((k for k in kernels if k.get("bound") == "hbm"), key=lambda k: k["frac_of_8TBps"])
            out["ops_hbm_frac"] = {k["name"].split(" ")[0]: k["frac_of_8TBps"] for k in kernels if k.get("bound") == "hbm"}
        if args.workload != "c3":
            out["roofline"] = roofline_of(dom, "the timed region of this line (HIP events on the launch stream)")
        out.update(side)
        if "list_fill" in out and "throughput_mode" in out and hasattr(wl, "cfg"):
            import bench_c3
            # the second roof of the step: the matrix work its GEMM-shaped kernels execute (own fp32-MFMA kernels + the library's
            # GEMMs) against the dense fp32 MFMA peak -- with 20 batches in flight the step is closer to this roof than to HBM's
            mw = bench_c3.matrix_work(wl.cfg, out["list_fill"]["scales"], wl.scenes())
            tf = mw["gflop_per_batch"] / ms_per_step
            mw.update({"bound": "mfma", "achieved": tf, "peak": bench_c3.FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / bench_c3.FP32_MFMA_PEAK_TFLOPS,
                       "note": "useful multiply-adds executed per batch (counted from the layer widths and the distinct pairs measured on the batch, "
                               "bench_c3.matrix_work; tile padding not counted) / ms_per_step of the timed region; f32 in, f32 accumulate "
                               "(v_mfma_f32_32x32x2_f32: the rate of the f32 vector unit, 1/16 of bf16)"})
            if "all_rows" in out:
                ar = bench_c3.matrix_work(wl.cfg, out["list_fill"]["scales"], wl.scenes(), compact=False)["gflop_per_batch"]
                mw["all_rows"] = {"gflop_per_batch": ar, "achieved": ar / out["all_rows"]["ms_per_batch"], "frac": ar / out["all_rows"]["ms_per_batch"] / bench_c3.FP32_MFMA_PEAK_TFLOPS}
            out["throughput_mode"]["matrix_roofline"] = mw
        if args.workload == "c3":
            out["roofline"] = c3_step_roofline(out, kernels, wl, getattr(wl, "kind", args.kind))
        out.update({"kernels": kernels, "step_ms_percentiles": step_percentiles(wl), "path_gbps_per_gpu": wl.path_gbps(per_gpu)})
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = wl.cpu_baseline(min_seconds=args.cpu_baseline_seconds)
                if c2wl is not None:
                    out["c2"]["cpu_baseline"] = c2wl.cpu_baseline(min_seconds=args.cpu_baseline_seconds)
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the line
                out.setdefault("cpu_baseline", {"error": repr(e)})
        detail_path = os.path.join(ROOT, DETAIL_FILE) if args.detail is None else args.detail
        if detail_path:
            try:
                with open(detail_path, "w") as f:
                    json.dump(out, f, indent=1)
            except OSError as e:
                print("bench.py: could not write %s: %r" % (detail_path, e), file=sys.stderr)
                detail_path = ""
        if args.full_line:
            print(json.dumps(out), flush=True)
        else:
            print("[bench detail] " + json.dumps(out), file=sys.stderr, flush=True)        # the full record, for the log
            print(json.dumps(compact_line(out, detail_path)), flush=True)                   # THE line (<= 4 KB)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
