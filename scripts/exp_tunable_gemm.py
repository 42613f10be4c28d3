"""EXPERIMENT: the library GEMMs of the c3 step (14 per batch: per-point products P / Q, FP skip halves + second layers; the largest
family by marginal cost in throughput mode) after a SECOND, longer TunableOp pass.  Stage1Pipeline already tunes during its priming runs
(pipeline._tunable), so the "plain" rows replay tuned solutions; here every shape is tuned again in an eager pass with a larger budget,
tuning is switched off, the pipeline is re-captured and timed.    python scripts/exp_tunable_gemm.py [steps] [results.csv]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch
import torch.cuda.tunable as tunable
from bench_c3 import C3

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
out_csv = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/tunableop_results.csv"


def run(tag, wl):
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
    print("%-44s %.4f ms per batch  %.0f scenes/s   latency %.3f ms (%s)" % (tag, dt / steps * 1e3, wl.scenes() * steps / dt, lat,
                                                                          min(detail, key=detail.get)), flush=True)
    wl.step(eager=True)
    torch.cuda.synchronize()
    o = wl.last[0]
    res = {k: o[k].detach().clone() for k in ("rpn_cls", "rpn_reg")}
    wl.release()
    return res


wl = C3(8, 0, 1, "hdl64", depth=20)
ref = run("plain (library heuristic)", wl)
tunable.set_filename(out_csv)
tunable.set_max_tuning_duration(30)
tunable.set_max_tuning_iterations(100)
tunable.enable(True)
tunable.tuning_enable(True)
w2 = C3(8, 0, 1, "hdl64", depth=20, model=wl.model)
t0 = time.perf_counter()
w2.step(eager=True)                              # every GEMM shape of the step is tuned on first sight
torch.cuda.synchronize()
print("tuning pass: %.1f s, %d results" % (time.perf_counter() - t0, len(tunable.get_results())), flush=True)
tunable.tuning_enable(False)
for r in tunable.get_results():
    print("   ", r, flush=True)
got = run("TunableOp (tuned, tuning off)", w2)
for k in ref:
    print("max |tuned - plain| %s: %.3e (max |plain| %.3e)" % (k, float((got[k] - ref[k]).abs().max()), float(ref[k].abs().max())), flush=True)
tunable.enable(False)
run("plain again", C3(8, 0, 1, "hdl64", depth=20, model=wl.model))
try:
    tunable.write_file(out_csv)
except Exception as e:
    print("write_file:", e)
