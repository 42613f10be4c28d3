"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes) into profiles/traffic.json: HBM bytes per launch per kernel.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports WIDE coalesced
streams (16 B/lane) by exactly 2x; other access widths are uncalibrated.  The FPS kernel's only
reads are 4-byte per-lane gathers at start-up and the ball-query kernel reads 4-byte strided
triples, so the raw value is kept and the doubled value is recorded next to it as the upper
bound; WRITE_SIZE was calibrated here against a known byte count (fused query_and_group writes
exactly B*((3+C)*M*ns + M*ns)*4 bytes; the counter matches to the byte)."""
import json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? "
                      "group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1] * 1024.0, r[2]) for r in rows}

def short(name):
    if "fps_v3_kernel" in name:      # <PPT, NT, ZLDS, ONEX>: the two-scenes-per-CU instance is the one with ZLDS = true
        return "fps_v3_pair_kernel" if ", true, " in name else "fps_v3_kernel"
    if "roipool3d_binned_kernel" in name:                                  # scenes >= 16384 points (c5): the binned variant + its binning pass
        return "roipool3d_binned_kernel"
    if "roi_bin_kernel" in name:
        return "roi_bin_kernel"
    if "roipool3d_pipe_kernel" in name or "roipool3d_kernel" in name:      # the large-scene variant reports under the same key
        return "roipool3d_kernel"
    for k in ("fps_rounds2_kernel", "fps_rounds_kernel", "ball_query_grid_coop_kernel", "fps_bucket_kernel", "fps_zlds_kernel", "fps_reg_kernel", "fps_big_kernel", "ball_query_grid_kernel", "bin_points_grid_kernel", "ball_query_sorted_kernel", "bin_points_x_kernel", "ball_query_kernel", "nms_rot_mask_kernel", "nms_sweep_kernel", "bev_frames_kernel", "roipool3d_kernel",
              "three_nn_kernel", "three_interpolate_kernel", "group_points_kernel"):
        if k in name:
            return k
    return None

def main(fetch_db, write_db, out, note, scenes=512):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    import bench_lib
    res = {"_note": note, "_scenes_per_launch": int(scenes), "_source_blobs": bench_lib.traffic_source_blobs(out)}
    for name in sorted(set(f) | set(w)):
        k = short(name)
        if not k:
            continue
        fb, wb = f.get(name, (0, 0))[0], w.get(name, (0, 0))[0]
        res[k] = {"kernel": name, "fetch_bytes_raw": fb, "fetch_bytes_if_wide_stream_x2": 2 * fb,
                  "write_bytes": wb, "hbm_bytes": fb + wb, "launches_sampled": f.get(name, (0, 0))[1]}
    if "roipool3d_binned_kernel" in res and "roipool3d_kernel" not in res:
        # the binned variant is two launches (the scene's counting sort + the pooling): bench.py's c5 row times both and reads ONE key
        parts = [res[k] for k in ("roipool3d_binned_kernel", "roi_bin_kernel") if k in res]
        res["roipool3d_kernel"] = {"kernel": " + ".join(p_["kernel"].split("(")[0] for p_ in parts),
                                   "fetch_bytes_raw": sum(p_["fetch_bytes_raw"] for p_ in parts),
                                   "fetch_bytes_if_wide_stream_x2": sum(p_["fetch_bytes_if_wide_stream_x2"] for p_ in parts),
                                   "write_bytes": sum(p_["write_bytes"] for p_ in parts), "hbm_bytes": sum(p_["hbm_bytes"] for p_ in parts),
                                   "launches_sampled": parts[0]["launches_sampled"]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "", sys.argv[5] if len(sys.argv) > 5 else 512)
