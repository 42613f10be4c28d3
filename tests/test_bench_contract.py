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


def run_bench(*flags):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", *flags], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("flags", [("--workload", "c2", "--batch", "8"), ("--workload", "c3", "--no-cpu-baseline"),
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
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and roof["achieved"] > 0
    if "--no-cpu-baseline" not in flags:
        cpu = out["cpu_baseline"]
        assert {"value", "unit", "cores", "kind", "sample"} <= set(cpu), cpu
        assert cpu["kind"] in ("port", "reference") and cpu["value"] > 0 and cpu.get("gpu_matches_oracle_on_sample", True)
