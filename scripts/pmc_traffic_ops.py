"""HBM bytes per STEP of the stand-alone copy operators of `bench.py --workload ops` (gather_points, group_points x4, three_interpolate)
from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel trace only):

    python scripts/pmc_traffic_ops.py <fetch.db> <write.db> <out.json> <batch>

Steps traced = dispatches of the three_nn kernel (one per step, the set-up call excluded by taking the per-step majority).  FETCH_SIZE is
kept raw (gfx950 under-reports wide 16 B/lane streams by 2x: the doubled figure is the upper bound); bench.py reads the result as
profiles/traffic_ops.json (batch 8) / traffic_ops<batch>.json and quotes it only while `_source_blobs` names the sources on disk."""
import json
import os
import sqlite3
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KEYS = (("gather_points", "gather_points_kernel"), ("group_points", "group_points_kernel"), ("three_interpolate", "three_interpolate_kernel"))


def group_dispatches(db_path, counter):
    """the group_points_* dispatches in launch order: a step issues five of them -- gather_points_wrapper (nsample 1: the same kernels),
    then group_points_wrapper x4 -- so every fifth one, starting with the first, is the gather"""
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)").fetchall()]
    order = "dispatch_id" if "dispatch_id" in cols else "rowid"
    rows = db.execute("select kernel_name, value from counters_collection where counter_name=? order by %s" % order, (counter,)).fetchall()
    seq = [v * 1024.0 for k, v in rows if "group_points" in k and "grad" not in k]
    n = len(seq) // 5
    gather = sum(seq[5 * i] for i in range(n)) / max(n, 1)
    group = sum(sum(seq[5 * i + 1:5 * i + 5]) for i in range(n)) / max(n, 1)
    return gather, group, n


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name=? group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1] * 1024.0, r[2]) for r in rows}


def main(fetch_db, write_db, out, batch):
    import bench_lib
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    steps = {}
    for tag, d in (("f", f), ("w", w)):
        steps[tag] = max(1, sum(n for k, (_, n) in d.items() if "three_interpolate" in k and "grad" not in k))
    res = {"_note": "ops workload, batch %s: HBM bytes PER STEP (one call of the operator sequence) per copy operator, rocprofv3 --pmc FETCH_SIZE / "
                    "WRITE_SIZE in separate passes; fetch raw" % batch, "_scenes_per_launch": int(batch), "_steps_traced": [steps["f"], steps["w"]],
           "_source_blobs": bench_lib.traffic_source_blobs(out)}
    for frag, key in KEYS:
        fb = sum(v for k, (v, _) in f.items() if frag in k and "grad" not in k) / steps["f"]
        wb = sum(v for k, (v, _) in w.items() if frag in k and "grad" not in k) / steps["w"]
        names = sorted({k.split("(")[0][-70:] for k in set(f) | set(w) if frag in k and "grad" not in k})
        res[key] = {"kernel": " + ".join(names), "fetch_bytes_raw": fb, "fetch_bytes_if_wide_stream_x2": 2 * fb, "write_bytes": wb, "hbm_bytes": fb + wb}
    gf, grf, nf = group_dispatches(fetch_db, "FETCH_SIZE")
    gw, grw, nw = group_dispatches(write_db, "WRITE_SIZE")
    for key, fb, wb in (("gather_points_kernel", gf, gw), ("group_points_kernel", grf, grw)):
        res[key].update({"fetch_bytes_raw": fb, "fetch_bytes_if_wide_stream_x2": 2 * fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                         "split": "by launch order: of the five group_points_* dispatches of a step the first is gather_points_wrapper (%d / %d steps)" % (nf, nw)})
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else 8)
