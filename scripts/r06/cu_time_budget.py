"""How much of the chip does one c3 step hold, kernel by kernel?  (the question behind DESIGN.md 10.3: is the 20-deep step bound by the
CU-time of its kernels or by how they pack?)

  rocprofv3 --kernel-trace --output-format csv -d <dir> -o ct -- python scripts/r06/cu_time_budget.py run     # eager steps, one batch in flight
  python scripts/r06/cu_time_budget.py report <dir>/..._kernel_trace.csv [ms_per_step_of_the_20_deep_line]

For every dispatch of the traced steps: workgroups, waves per workgroup, registers and LDS from the trace -> how many workgroups one CU
holds (512 VGPRs per SIMD lane incl. AGPRs, granule 8; 8 waves per SIMD at most; 160 KB LDS) -> the share of the chip the dispatch can
hold at once (1 when the grid exceeds 256 CUs x that) x its duration = CU-time, in ms of the WHOLE chip.  The sum over a step is the
time the step would take if its kernels packed perfectly and ran at their stand-alone speed."""
import csv
import os
import sys
from collections import defaultdict

STEPS, WARM = 6, 3


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from bench_c3 import C3
    w = C3(8, 0, 1, "hdl64", depth=1)
    for _ in range(WARM + STEPS):
        w.step(eager=True)
        torch.cuda.synchronize()
    print("traced %d + %d eager steps" % (WARM, STEPS))


def wgs_per_cu(wg_threads, vgpr, agpr, lds):
    waves = (wg_threads + 63) // 64
    regs = max(8, (vgpr + agpr + 7) // 8 * 8)
    per_simd = max(1, min(8, 512 // regs))
    by_waves = max(1, (4 * per_simd) // waves) if waves <= 4 * per_simd else 0
    by_lds = (160 * 1024) // lds if lds > 0 else 32
    return max(1, min(by_waves if by_waves else 1, by_lds, 32)), waves


def report(path, step_ms=None):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the traced process runs WARM + STEPS identical steps: keep the last STEPS / (WARM + STEPS) of the dispatches
    # the traced process runs priming passes and WARM + STEPS identical eager steps; every step launches the level-1 sampling kernel once:
    # keep the dispatches from the (STEPS)-th last of its launches on
    fps = [i for i, r in enumerate(rows) if "fps_rounds2_kernel" in r["Kernel_Name"]]
    rows = rows[fps[-STEPS]:] if len(fps) >= STEPS else rows
    fam = defaultdict(lambda: [0.0, 0.0, 0, 0.0])
    for r in rows:
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        nwg = max(1, g // max(wg, 1))
        per_cu, _ = wgs_per_cu(wg, int(r.get("VGPR_Count", 0) or 0), int(r.get("Accum_VGPR_Count", 0) or 0), int(r.get("LDS_Block_Size", 0) or 0))
        share = min(1.0, nwg / (256.0 * per_cu))
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        f = fam[r["Kernel_Name"].split("(")[0][:70]]
        f[0] += dur * share
        f[1] += dur
        f[2] += 1
        f[3] = max(f[3], share)
    tot = sum(f[0] for f in fam.values()) / STEPS
    print("# CU-time of one c3 step (8 scenes, hdl64), eager one batch in flight, mean of %d steps: share of the chip a dispatch can hold x its duration" % STEPS)
    print("# %-70s %6s %10s %10s %8s" % ("kernel", "calls", "dur ms", "chip ms", "share"))
    for k, f in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print("  %-70s %6.1f %10.4f %10.4f %8.3f" % (k, f[2] / STEPS, f[1] / STEPS, f[0] / STEPS, f[3]))
    print("# sum: %.4f ms of the whole chip per step (sum of durations %.3f ms)" % (tot, sum(f[1] for f in fam.values()) / STEPS))
    if step_ms:
        print("# the 20-deep line: %.4f ms per step -> the chip is held %.0f %% of the time by this accounting" % (step_ms, 100 * tot / step_ms))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else None)
