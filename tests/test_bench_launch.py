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
