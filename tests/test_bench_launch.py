"""bench.py's own launcher (no GPU needed): ``python bench.py --gpus N`` starts N ranks by itself, and a world size that
differs from --gpus is an error instead of a silent one-GPU line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_LAUNCHER = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def _run(args, env):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_n_without_a_launcher_builds_a_torch_distributed_run_command():
    for flag in (["--gpus", "4"], ["--gpus=4"]):
        p = _run([*flag, "--steps", "3", "--warmup", "1"], dict(NO_LAUNCHER, WS3D_BENCH_LAUNCH_DRYRUN="1"))
        assert p.returncode == 0, p.stderr[-2000:]
        cmd = json.loads(p.stdout.strip().splitlines()[-1])["self_launch"]
        assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
        assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
        assert cmd[-len(flag) - 4:] == [*flag, "--steps", "3", "--warmup", "1"] and cmd[-len(flag) - 5].endswith("bench.py")


def test_under_a_launcher_nothing_is_started_and_a_world_mismatch_is_rc_nonzero():
    # WORLD_SIZE=1 in the environment (a launcher that started one rank) and --gpus 2: exit code != 0, no JSON line
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], dict(NO_LAUNCHER, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", WS3D_BENCH_LAUNCH_DRYRUN="1"))
    assert p.returncode != 0 and "self_launch" not in p.stdout
    assert "WORLD_SIZE=1" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_matrix_work_counts_the_multiply_adds_of_the_network_as_built():
    """bench_c3.matrix_work (the numerator of throughput_mode.matrix_roofline) against the weights of the Stage-1 network itself: with every
    list full, its count = the conv weights' sizes x the rows each is applied to in fastpath.py's formulation (first SA layers of
    levels 2-4 split into a per-point product + 3 xyz columns per row, first FP layers split into known-point / skip halves)"""
    sys.path.insert(0, ROOT)
    import bench_c3
    from ws3d_amd import stage1
    cfg = stage1.DEFAULT_CFG
    net = stage1.Stage1Net(mode="TEST", cfg=cfg)
    bb = net.rpn.backbone_net
    fill = [{"distinct_per_list": float(ns), "fill": 1.0} for nss in cfg.nsample for ns in nss]
    convs = lambda mod: [m.weight for m in mod.modules() if hasattr(m, "weight") and m.weight.dim() >= 3]     # (bn weights are 1-D)
    want, n = 0, cfg.num_points
    counts = [cfg.num_points] + list(cfg.npoints)
    for lvl, sa in enumerate(bb.SA_modules):
        for mlp, ns in zip(sa.mlps, cfg.nsample[lvl]):
            ws = convs(mlp)
            rows = sa.npoint * ns
            first = ws[0]
            if lvl == 0:
                want += rows * first.numel()
            else:
                want += n * first.size(0) * (first.size(1) - 3) + rows * 3 * first.size(0)
            want += rows * sum(w.numel() for w in ws[1:])
        n = sa.npoint
    for k, fp in enumerate(bb.FP_modules):
        ws = convs(fp.mlp)
        first = ws[0]
        skip = first.size(1) - (cfg.fp_mlps[k + 1][-1] if k + 1 < len(cfg.fp_mlps) else sum(m[-1] for m in cfg.mlps[-1]))
        want += counts[k + 1] * first.size(0) * (first.size(1) - skip) + counts[k] * first.size(0) * skip + counts[k] * sum(w.numel() for w in ws[1:])
    for head in (net.rpn.rpn_cls_layer, net.rpn.rpn_reg_layer):
        want += cfg.num_points * sum(w.numel() for w in convs(head))
    got = bench_c3.matrix_work(cfg, fill, batch=8, compact=False)
    assert abs(got["gflop_per_batch"] - 2.0 * want * 8 / 1e9) < 1e-6 * got["gflop_per_batch"], (got, 2.0 * want * 8 / 1e9)
    half = [dict(f, distinct_per_list=f["distinct_per_list"] / 4, fill=0.25) for f in fill]
    assert bench_c3.matrix_work(cfg, half, batch=8)["gflop_per_batch"] < got["gflop_per_batch"]
    assert bench_c3.matrix_work(cfg, half, batch=8, compact=False)["gflop_per_batch"] == got["gflop_per_batch"]


def test_self_launch_command_gives_every_rank_a_device_of_its_own():
    """``python bench.py --gpus 8``: the command it builds is ONE node, 8 processes -- torch.distributed.run then numbers LOCAL_RANK
    0..7, one per device -- and nothing in it pins or shares a device (no CUDA/HIP_VISIBLE_DEVICES edits, no --gpus rewrite)"""
    p = _run(["--gpus", "8", "--steps", "2", "--warmup", "1"], dict(NO_LAUNCHER, WS3D_BENCH_LAUNCH_DRYRUN="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    cmd = json.loads(p.stdout.strip().splitlines()[-1])["self_launch"]
    assert cmd.count("--nproc-per-node") == 1 and cmd[cmd.index("--nproc-per-node") + 1] == "8" and "--nnodes=1" in cmd
    assert cmd[cmd.index("--gpus") + 1] == "8" and not any("VISIBLE_DEVICES" in c for c in cmd)
    # LOCAL_RANK -> device is the identity in bench.dist_setup / ws3d_amd.dist.init over RCCL (more ranks than devices: an error)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if backend == "nccl" and world > ndev:' in src and "sys.exit(3)" in src


def test_compact_line_fits_the_drivers_parser():
    """bench.compact_line on a full record of round 4 (profiles/r04_bench_default_steps20_warmup5.json, 24.8 KB -- the line the
    driver could not parse): the extract is < 4 KB, holds the contract keys, and no string longer than 300 characters"""
    sys.path.insert(0, ROOT)
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_default_steps20_warmup5.json")).read().strip().splitlines()[-1])
    full.setdefault("generator", "hdl64")

    class W:
        name = full["config"]["workload"]

        def scenes(self):
            return 8
    full["roofline"] = bench.c3_step_roofline(full, full["kernels"], W(), "hdl64")
    line = bench.compact_line(full, "bench_detail.json")
    text = json.dumps(line)
    assert len(text) < bench.LINE_BUDGET < len(json.dumps(full))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert {"workload", "batch_per_gpu", "n_points", "ranks_seen", "generator", "backend"} <= set(line["config"])
    r = line["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - 0.302) < 0.01                                              # the verdict's own recomputation
    if bench.traffic_state("c3:x") == "fresh":        # the counter bytes are quoted only when the committed pass names the sources on disk
        assert 0.05 < r["hbm"]["frac"] < 0.2 and r["traffic"] > 5e8
    else:
        assert r["traffic"] is None and r["hbm"]["frac"] is None and r["traffic_source"] == "stale"
    assert "c3_stage1" in r["measured_in"] and r["dominant_kernel"]["workgroups"] == 8

    def strings(x):
        if isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, list):
            for v in x:
                yield from strings(v)
        elif isinstance(x, str):
            yield x
    assert max(len(t) for t in strings(line)) <= 300


def test_counter_figures_are_quoted_only_for_the_sources_they_were_taken_on(tmp_path, monkeypatch):
    """a profiles/traffic*.json carries the git blob hashes of the kernel sources it describes; on other sources
    bench.py quotes `traffic: null` + `traffic_source: stale` instead of a figure of code that no longer exists"""
    import json
    import bench
    import bench_lib
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench_lib, "_TRAFFIC_STATE", {})
    real_file = bench_lib._traffic_file

    def fake_file(key):
        p, k = real_file(key)
        return str(prof / os.path.basename(p)), k

    monkeypatch.setattr(bench_lib, "_traffic_file", fake_file)
    blobs = bench_lib.traffic_source_blobs("traffic_c5.json")
    assert set(blobs) == {"ws3d_amd/csrc/roipool3d.hip", "ws3d_amd/csrc/iou3d.hip", "ws3d_amd/csrc/common.h"} and all(len(v) == 40 for v in blobs.values())
    c3 = bench_lib.traffic_source_blobs("traffic_c3.json")
    assert "ws3d_amd/fastpath.py" in c3 and "ws3d_amd/csrc/chain_mlp.hip" in c3 and "ws3d_amd/csrc/gemm_pool.hip" in c3
    entry = {"kernel": "k", "hbm_bytes": 1.0e9}
    # fresh: named sources == the ones on disk
    (prof / "traffic_c5.json").write_text(json.dumps({"_scenes_per_launch": 8, "_source_blobs": blobs, "roipool3d_kernel": entry}))
    assert bench_lib.traffic_state("c5:roipool3d_kernel") == "fresh" and bench_lib.load_traffic("c5:roipool3d_kernel") == entry
    # stale: one hash differs / no hashes at all (a pass of rounds 1-5)
    for doc in ({"_scenes_per_launch": 8, "_source_blobs": dict(blobs, **{"ws3d_amd/csrc/roipool3d.hip": "0" * 40}), "roipool3d_kernel": entry},
                {"_scenes_per_launch": 8, "roipool3d_kernel": entry}):
        bench_lib._TRAFFIC_STATE.clear()
        (prof / "traffic_c5.json").write_text(json.dumps(doc))
        assert bench_lib.traffic_state("c5:roipool3d_kernel") == "stale" and bench_lib.load_traffic("c5:roipool3d_kernel") is None
        rows = bench_lib.finish_kernel_rows([{"name": "roipool3d", "ms_per_step": 0.25, "launches_per_step": 1, "alg_bytes_per_step": 1.1e9,
                                              "traffic_key": "c5:roipool3d_kernel", "bound": "hbm"}], 8)
        roof = bench_lib.roofline_of(rows[0], "test")
        assert roof["traffic"] is None and roof["traffic_source"] == "stale"
    bench_lib._TRAFFIC_STATE.clear()
    assert bench_lib.traffic_state("c3:x") == "absent"
    # the committed passes of THIS tree either name the sources on disk or are reported stale -- never quoted blindly
    monkeypatch.setattr(bench_lib, "_traffic_file", real_file)
    bench_lib._TRAFFIC_STATE.clear()
    for key in ("ball_query_grid_coop_kernel", "c5:roipool3d_kernel", "c3:_total_hbm_bytes_per_step"):
        st = bench_lib.traffic_state(key)
        assert st in ("fresh", "stale", "absent")
        assert (bench_lib.load_traffic(key) is not None) <= (st == "fresh")
    assert bench.traffic_state is bench_lib.traffic_state
