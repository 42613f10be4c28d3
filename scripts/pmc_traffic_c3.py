"""HBM traffic of ONE eager Stage-1 step (c3, batch 8) per launch family of bench_c3.py's kernel table, from two separate
rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel trace only, as MI355X_MICROARCH.md prescribes) over
``python bench.py --workload c3 --pipeline-depth 1 --no-graph --c2-batch 0 --no-cpu-baseline --no-side-runs``:

    python scripts/pmc_traffic_c3.py <fetch.db> <write.db> <out.json> <generator> <batch>

Every dispatch of every kernel is summed into its family; the number of steps traced = the dispatches of the level-1 sampling
kernel (one per step), so the figures are bytes PER STEP (= per batch of <batch> scenes).  FETCH_SIZE is kept raw (gfx950
under-reports wide 16 B/lane streams by 2x, other widths uncalibrated: the doubled figure is the upper bound); WRITE_SIZE matches
known byte counts to the byte (scripts/pmc_traffic.py).  bench.py reads the result as profiles/traffic_c3.json."""
import json
import os
import sqlite3
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# kernel-name fragment -> family name of bench_c3.FAMILIES (first match wins)
FAMILY_OF = [
    ("fps_rounds2_kernel", "fps level 1 (16384 -> 4096)"), ("fps_rounds_kernel", "fps level 1 (16384 -> 4096)"), ("fps_bucket_kernel", "fps level 1 (16384 -> 4096)"), ("fps_v3_kernel", "fps level 1 (16384 -> 4096)"),
    ("fps_nested_", "fps levels 2-4 (verified prefix)"),
    ("bin_points_", "binning (grid / x slabs / xz grid)"),
    ("ball_query_", "ball_query"),
    ("pair_", "pair compaction"),
    ("sa_mlp3_", "SharedMLP SA1 (3 layers + pool, own MFMA kernels)"),
    ("chain_mlp3_", "SharedMLP SA2 whole scale over compact rows (one own MFMA kernel)"), ("chain_pack_", "SharedMLP SA2 whole scale over compact rows (one own MFMA kernel)"),
    ("pgather_gemm3_", "SharedMLP SA2 whole scale over compact rows (one own MFMA kernel)"),
    ("pgather_", "SharedMLP SA2-4 layers 1+2 (gather + own MFMA kernels)"), ("gather_gemm", "SharedMLP SA2-4 layers 1+2 (gather + own MFMA kernels)"),
    ("gemm_pool_", "SharedMLP SA2-4 last layer + pool (own MFMA kernels)"), ("rowmax_", "SharedMLP SA2-4 last layer + pool (own MFMA kernels)"),
    ("three_nn", "three_nn (+ weights)"), ("nn_weights_kernel", "three_nn (+ weights)"),
    ("qinterp_rows_kernel", "FP first layer (interpolate + add; at FP1-2 with the second layer: own kernels)"), ("interp_gemm", "FP first layer (interpolate + add; at FP1-2 with the second layer: own kernels)"),
    ("three_interpolate", "FP first layer (interpolate + add; at FP1-2 with the second layer: own kernels)"),
    ("fp_fused", "FP module fused (interpolate + skip + 2 layers, own MFMA kernel)"),
    ("mlp2_rows_kernel", "heads (2 layers, own MFMA kernel)"),
    ("decode_center_boxes_kernel", "proposals: decode + top-k + gather + select"), ("topk_", "proposals: decode + top-k + gather + select"),
    ("gather_boxes_bev_kernel", "proposals: decode + top-k + gather + select"), ("select_proposals_kernel", "proposals: decode + top-k + gather + select"),
    ("nms_", "nms(mask+sweep)"), ("bev_frames_kernel", "nms(mask+sweep)"),
    ("roipool3d_", "roipool3d"),
]
LIBRARY = "library residual of rpn_forward: Tensile GEMMs + at::native glue"


def family(name):
    if "ws3d::" in name:
        for frag, fam in FAMILY_OF:
            if frag in name:
                return fam
        return "own, unattributed: " + name.split("(")[0][-60:]
    return LIBRARY


def per_dispatch(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name=? group by kernel_name",
                      (counter,)).fetchall()
    return {r[0]: (r[1] * 1024.0, r[2]) for r in rows}      # FETCH_SIZE / WRITE_SIZE are in KiB


def main(fetch_db, write_db, out, kind, batch):
    f, w = per_dispatch(fetch_db, "FETCH_SIZE"), per_dispatch(write_db, "WRITE_SIZE")
    steps_f = sum(n for k, (_, n) in f.items() if "fps_rounds2_kernel" in k or "fps_rounds_kernel" in k or "fps_bucket_kernel" in k) or 1
    steps_w = sum(n for k, (_, n) in w.items() if "fps_rounds2_kernel" in k or "fps_rounds_kernel" in k or "fps_bucket_kernel" in k) or 1
    res = {"_note": "c3 eager step, batch %s, generator %s: HBM bytes PER STEP per launch family (sum over the family's dispatches / steps "
                    "traced), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; fetch raw (gfx950: wide streams under-reported 2x)" % (batch, kind),
           "_scenes_per_launch": int(batch), "_kind": kind, "_steps_traced": [steps_f, steps_w]}
    import bench_lib
    res["_source_blobs"] = bench_lib.traffic_source_blobs(out)
    fam = {}
    for name in sorted(set(f) | set(w)):
        e = fam.setdefault(family(name), {"fetch_bytes_raw": 0.0, "write_bytes": 0.0, "launches_per_step": 0.0, "kernels": []})
        fb, nf = f.get(name, (0.0, 0))
        wb, nw = w.get(name, (0.0, 0))
        e["fetch_bytes_raw"] += fb / steps_f
        e["write_bytes"] += wb / steps_w
        e["launches_per_step"] += nf / steps_f
        e["kernels"].append(name.split("(")[0][:80])
    for e in fam.values():
        e["fetch_bytes_if_wide_stream_x2"] = 2 * e["fetch_bytes_raw"]
        e["hbm_bytes"] = e["fetch_bytes_raw"] + e["write_bytes"]
    res.update(fam)
    res["_total_hbm_bytes_per_step"] = sum(e["hbm_bytes"] for e in fam.values())
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: (v["hbm_bytes"] if isinstance(v, dict) else v) for k, v in res.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "hdl64", sys.argv[5] if len(sys.argv) > 5 else 8)
