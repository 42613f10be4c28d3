"""bench.py prints ONE JSON line per run with the driver's contract keys plus the tier's
`roofline` / `cpu_baseline` objects -- checked for every workload on a short run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


LINE_MAX = 8192          # the driver reads the TAIL of stdout: round 4's 24.8 KB line was not parsed (bench.py aims at 4 KB)


def check_line_size(text):
    """the ONE stdout line stays small enough for the driver's parser, and holds no prose"""
    assert len(text) < LINE_MAX, len(text)

    def walk(x, path="line"):
        if isinstance(x, dict):
            for k, v in x.items():
                walk(v, path + "." + k)
        elif isinstance(x, (list, tuple)):
            assert len(x) <= 16, (path, len(x))
            for v in x:
                walk(v, path + "[]")
        elif isinstance(x, str):
            assert len(x) <= 300, (path, len(x))
    walk(json.loads(text))


def run_bench(*flags, detail=None):
    """-> the compact line (dict); with `detail` (a path) the full record is written there and returned as the second value"""
    extra = ["--detail", str(detail)] if detail is not None else ["--detail", ""]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", *extra, *flags], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    check_line_size(lines[0])
    line = json.loads(lines[0])
    if detail is None:
        return line
    full = json.load(open(detail))
    # the line is an extract of the record, never a second measurement
    assert full["value"] == pytest.approx(line["value"], rel=1e-5) and full["ms_per_step"] == pytest.approx(line["ms_per_step"], rel=1e-5)
    return line, full


def test_default_line_is_the_baseline_headline(tmp_path):
    """no --workload: BASELINE.json's metric on configs[2] (c3) with latency + throughput modes and the c2 block in the record;
    the LINE is compact, and its roofline describes the line's own timed region (the c3 step), not the c2 launch"""
    line, out = run_bench("--cpu-baseline-seconds", "1", "--c2-batch", "64", "--pipeline-depth", "3", detail=tmp_path / "detail.json")
    assert KEYS | {"cpu_baseline"} <= set(line), KEYS - set(line)
    lr = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "measured_in"} <= set(lr)
    assert lr["bound"] == "mfma" and lr["unit"] == "TFLOP/s" and lr["peak"] == 157.3 and 0 < lr["frac"] < 1
    assert abs(lr["frac"] - lr["achieved"] / lr["peak"]) < 1e-4
    assert line["config"]["workload"] in lr["measured_in"] and "batch 8" in lr["measured_in"]        # the line's own workload
    assert lr["achieved"] == pytest.approx(lr["gflop_per_step"] / line["ms_per_step"], rel=1e-4)
    dk = lr["dominant_kernel"]
    assert "fps" in dk["name"] and dk["workgroups"] == 8 and dk["ms_per_launch"] > 0 and 0 < dk["us_per_sample"] < 5
    assert lr["hbm"]["a_min_frac"] > 0 and (lr["traffic"] is None) == (lr["hbm"]["frac"] is None)
    assert line["config"]["generator"] == "hdl64" and line["value_hdl64"] == line["value"] and line["value_lidar"] > 0
    assert line["latency_ms"] > 0 and line["c2_batch"] == 64 and line["c2_scenes_per_s"] > 0 and 0 < line["c2_query_group_hbm_frac"] <= 1
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"]) and line["cpu_baseline"]["kind"] in ("port", "reference")
    assert line["config"]["ranks_seen"] == 1 and line["config"]["communicator_size"] == 1
    for cpu in (out["cpu_baseline"], out["c2"]["cpu_baseline"]):
        assert "error" not in cpu and cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1, cpu
    assert out["c2"]["cpu_baseline"]["gpu_matches_oracle_on_sample"] is True
    assert out["value_all_rows"] == out["all_rows"]["value"] > 0 and out["other_generator"]["generator"] == "lidar"
    assert len(out["list_fill"]["scales"]) == 8 and all(0 < s_["fill"] <= 1 for s_ in out["list_fill"]["scales"])
    assert "hdl64" in out["data"]
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert out["metric"] == base["metric"] and out["config"]["workload"].startswith("c3") and out["config"]["batch_per_gpu"] == 8
    assert out["throughput_mode"]["value"] == out["value"] and out["throughput_mode"]["batches_in_flight"] == 3
    assert out["latency_mode"]["batches_in_flight"] == 1 and 0 < out["latency_mode"]["value"] <= out["value"] * 1.05
    mr = out["throughput_mode"]["matrix_roofline"]     # the step's GEMM-shaped work against the dense fp32 MFMA peak
    assert mr["bound"] == "mfma" and mr["peak"] == 157.3 and 0 < mr["frac"] < 1 and abs(mr["frac"] - mr["achieved"] / mr["peak"]) < 1e-9
    assert abs(mr["achieved"] - mr["gflop_per_batch"] / out["ms_per_step"]) < 1e-6 and mr["all_rows"]["gflop_per_batch"] > mr["gflop_per_batch"]
    c2 = out["c2"]
    assert c2["workload"].startswith("c2") and c2["batch_per_gpu"] == 64 and c2["scenes_per_s_per_gpu"] > 0
    assert {"a_model", "a_min", "a_model_bytes_per_scene", "a_min_bytes_per_scene"} <= set(c2["path_gbps_per_gpu"])
    roof = c2["roofline"]           # the c2 block's own roofline: the chip-filling launch of the sampling kernel
    assert roof["bound"] == "valu"
    if roof["frac"] is None:      # no committed --pmc pass of the kernel sources on disk for this batch: bench.py refuses a stale count
        assert roof["frac_is"].startswith("null") and 0 < roof["dense_equivalent_frac"] <= 1 and roof["achieved"] is None
    else:
        assert 0 < roof["frac"] <= 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["effective_frac"] > 0 and "fps" in roof["kernel"]
    hbm = [k for k in c2["kernels"] if k["bound"] == "hbm"]
    assert hbm and all(0 < k["frac_of_8TBps"] <= 1 for k in hbm)


@pytest.mark.parametrize("flags", [("--workload", "c2", "--batch", "8"), ("--workload", "c3", "--no-cpu-baseline", "--c2-batch", "0"),
                                   ("--workload", "c5", "--no-cpu-baseline"), ("--workload", "s2", "--batch", "64", "--no-cpu-baseline"),
                                   ("--workload", "t1", "--no-cpu-baseline")])
def test_bench_line_follows_the_contract(flags):
    out = run_bench(*flags)
    assert KEYS <= set(out), KEYS - set(out)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["scaling"] == "weak" and out["vs_baseline"] is None and out["dtype"] == "f32" and "synthetic" in out["data"]
    assert out["value"] > 0 and out["ms_per_step"] > 0 and "workload" in out["config"] and "model" not in out["config"]
    roof = out["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof)
    assert roof["bound"] in ("hbm", "valu", "mfma") and roof["unit"] in ("GB/s", "Tlane-instr/s", "TFLOP/s")
    assert "measured_in" in roof and "timed region of this line" in roof["measured_in"]
    if roof["frac"] is None:
        assert roof["bound"] == "valu" and roof["frac_is"].startswith("null") and 0 < roof["dense_equivalent_frac"] <= 1
    else:
        assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4 and 0 < roof["frac"] <= 1
    if "--no-cpu-baseline" not in flags:
        cpu = out["cpu_baseline"]
        assert {"value", "unit", "cores", "kind", "sample"} <= set(cpu), cpu
        assert cpu["kind"] in ("port", "reference") and cpu["value"] > 0 and cpu.get("gpu_matches_oracle_on_sample", True)


def _two_rank_bench(tmp_path, workload_flags, port, ranks=2):
    """`ranks` ranks of bench.py on ONE GPU (gloo for the exchange, every rank on cuda:0): the N>1 path with the real kernels"""
    env = dict(os.environ, WS3D_DIST_BACKEND="gloo", WS3D_BENCH_DUMP=str(tmp_path), WS3D_TUNE_GEMMS="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % ranks, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", *workload_flags]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=2400)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    check_line_size(lines[0])
    return json.loads(lines[0])


def test_two_ranks_c3_gather_equals_the_single_process_result(tmp_path):
    import numpy as np
    import torch
    out = _two_rank_bench(tmp_path, ("--workload", "c3", "--pipeline-depth", "2", "--c2-batch", "64"), 29631)
    assert out["n_gpus"] == 2 and out["config"]["workload"].startswith("c3") and out["scaling"] == "weak"
    assert "all_gather" in out["config"]["exchange"] and out["c2_batch"] == 64
    r0, r1 = (np.load(os.path.join(tmp_path, "proposals_rank%d.npz" % r)) for r in (0, 1))
    assert r0["gathered"].shape == (16, 100, 8) and r0["gathered_count"].shape == (16,)
    # every rank holds the same gathered tensor, in scene order: rank r's own scenes are rows [8r, 8r+8)
    assert np.array_equal(r0["gathered"], r1["gathered"]) and np.array_equal(r0["gathered_count"], r1["gathered_count"])
    for r, d in enumerate((r0, r1)):
        assert np.array_equal(d["gathered"][8 * r:8 * r + 8], d["local"]) and np.array_equal(d["gathered_count"][8 * r:8 * r + 8], d["local_count"])
    # ... and equals what ONE process computes for the same 16 scenes (two batches of 8, same seeds, same weights)
    os.environ.setdefault("WS3D_TUNE_GEMMS", "0")
    sys.path.insert(0, ROOT)
    from bench_c3 import C3
    from ws3d_amd import dist as wdist
    for r in (0, 1):
        wl = C3(8, r, 1, "hdl64", depth=1)
        wl.step()
        torch.cuda.synchronize()
        _, boxes, scores, count, _, _, _ = wl.last
        single = wdist.pack_proposals(boxes, scores).cpu().numpy()
        assert np.array_equal(count.cpu().numpy(), r0["gathered_count"][8 * r:8 * r + 8])
        np.testing.assert_allclose(single, r0["gathered"][8 * r:8 * r + 8], rtol=0, atol=1e-4)
    with open(os.path.join(ROOT, "gpurun_out", "two_rank_c3.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else os.devnull, "w") as f:
        json.dump(out, f)


def test_eight_ranks_batch_64_gather_equals_the_single_process_result(tmp_path):
    """BASELINE configs[3]'s SHAPE without its hardware: 8 ranks x 8 scenes (global batch 64), the launcher's default depth for
    world > 1 (16 batches in flight per rank), all eight ranks sharing the box's one GPU with gloo for the exchange.  Every rank ends
    with the same (64, 100, 8) tensor in scene order, and it equals what one process computes for the same 64 scenes.  (No 1 -> 8
    scaling curve exists for this build: no multi-GPU node was available to it.)"""
    import numpy as np
    import torch
    out = _two_rank_bench(tmp_path, ("--workload", "c3", "--batch", "8", "--c2-batch", "0"), 29634, ranks=8)
    assert out["n_gpus"] == 8 and out["config"]["ranks_seen"] == 8 and out["config"]["communicator_size"] == 8 and out["config"]["backend"] == "gloo"
    assert out["config"]["batch_per_gpu"] == 8 and out["config"]["pipeline_depth"] == 16 and "all_gather" in out["config"]["exchange"]
    d = [np.load(os.path.join(tmp_path, "proposals_rank%d.npz" % r)) for r in range(8)]
    assert d[0]["gathered"].shape == (64, 100, 8) and d[0]["gathered_count"].shape == (64,)
    for r in range(8):
        assert np.array_equal(d[r]["gathered"], d[0]["gathered"]) and np.array_equal(d[r]["gathered_count"], d[0]["gathered_count"])
        assert np.array_equal(d[0]["gathered"][8 * r:8 * r + 8], d[r]["local"]) and np.array_equal(d[0]["gathered_count"][8 * r:8 * r + 8], d[r]["local_count"])
    os.environ.setdefault("WS3D_TUNE_GEMMS", "0")
    sys.path.insert(0, ROOT)
    from bench_c3 import C3
    from ws3d_amd import dist as wdist
    model = None
    for r in range(8):
        wl = C3(8, r, 1, "hdl64", depth=1, model=model)
        model = wl.model
        wl.step()
        torch.cuda.synchronize()
        _, boxes, scores, count, _, _, _ = wl.last
        assert np.array_equal(count.cpu().numpy(), d[0]["gathered_count"][8 * r:8 * r + 8])
        np.testing.assert_allclose(wdist.pack_proposals(boxes, scores).cpu().numpy(), d[0]["gathered"][8 * r:8 * r + 8], rtol=0, atol=1e-4)
    with open(os.path.join(ROOT, "gpurun_out", "eight_rank_c3.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else os.devnull, "w") as f:
        json.dump(out, f)


def test_two_ranks_c2_shards_without_a_collective(tmp_path):
    out = _two_rank_bench(tmp_path, ("--workload", "c2", "--batch", "64"), 29632)
    assert out["n_gpus"] == 2 and out["config"]["workload"].startswith("c2") and out["config"]["batch_per_gpu"] == 64
    assert out["value"] > 0 and (out["roofline"]["frac"] is None or 0 < out["roofline"]["frac"] <= 1)
    with open(os.path.join(ROOT, "gpurun_out", "two_rank_c2.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else os.devnull, "w") as f:
        json.dump(out, f)


def test_gpus_2_without_a_launcher_starts_two_ranks(tmp_path):
    """``python bench.py --gpus 2`` -- the shape of the driver's command, NO torch.distributed.run around it -- starts its own
    two ranks (gloo here: both share the one GPU of the box) and reports n_gpus == 2 with both ranks seen"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WS3D_DIST_BACKEND="gloo", WS3D_TUNE_GEMMS="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--workload", "c3", "--pipeline-depth", "2", "--c2-batch", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks_seen"] == 2 and out["config"]["communicator_size"] == 2
    assert "self-launch" in out["config"]["launcher"] and "all_gather" in out["config"]["exchange"]


def test_more_rccl_ranks_than_devices_is_an_error():
    """two RCCL ranks on a one-GPU box cannot be measured: rc != 0 and no JSON line (never a silent one-GPU figure)"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 devices")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "WS3D_DIST_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--workload", "c2", "--batch", "8"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")], (p.stdout + p.stderr)[-2000:]
    assert "needs 2 devices" in p.stderr


def test_rccl_all_gather_runs_in_the_graph_mode_pipeline():
    """world size 1 under the nccl (= RCCL) backend with the collective forced: the exchange as the pipeline issues it
    (on the slot's side stream, after a graph replay) executes through RCCL on this 1-GPU box"""
    code = r'''
import os, sys, torch
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", RANK="0", WORLD_SIZE="1", GPU_MAX_HW_QUEUES="32")
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from bench_c3 import C3
from ws3d_amd import dist as wdist
wl = C3(8, 0, 1, "lidar", depth=2)
wl.step(); assert wl.capture(), wl._graph_err
for _ in range(3):
    ticket = wl.pipe.submit()
    slot = wl.pipe.slots[ticket %% 2]
    res = slot["out"]
    with torch.cuda.stream(slot["stream"]):
        packed = wdist.pack_proposals(res["boxes"], res["scores"])
        g, c = wdist.all_gather_proposals(packed, res["count"], 8, force=True)
    torch.cuda.synchronize()
    assert dist.get_backend() == "nccl" and torch.equal(g, packed) and torch.equal(c, res["count"]), "RCCL all-gather changed the data"
dist.destroy_process_group()
print("RCCL_OK")
''' % ROOT
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, (p.stdout + p.stderr)[-3000:]
