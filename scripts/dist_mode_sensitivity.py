"""How often is the squared-distance convention (WS3D_DIST_MODE 0 / 1 / 2, csrc/common.h) visible in the outputs?
CPU only: runs the oracle (tests' checker; this script is a study tool, not product code) under the three conventions on the
lidar generator and reports, relative to mode 0, the fraction of FPS index sequences / ball-query rows / 3-NN rows that change.

    python scripts/dist_mode_sensitivity.py [scenes] > profiles/r02_dist_mode_sensitivity.json
"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
import oracle
from ws3d_amd import synth
S = int(sys.argv[1])
oracle.set_threads(min(oracle.max_threads(), 32))
out = {}
for kind in ("lidar", "uniform"):
    pcs = np.stack([(synth.lidar_cloud(16384, 4000 + s) if kind == "lidar" else synth.uniform_cloud(16384, 4000 + s))[:, :3] for s in range(S)])
    idx = oracle.furthest_point_sample(pcs, 4096)
    new = np.stack([pcs[b][idx[b]] for b in range(S)])
    # the SAME centres (mode 0's are passed in through a file) for the searches, so that only the search is compared
    out[kind] = {"fps": idx.tolist()}
    np.save(sys.argv[2] + "_%%s_new.npy" %% kind, new)
    cen = np.load(sys.argv[3] + "_%%s_new.npy" %% kind) if sys.argv[3] != "-" else new
    out[kind]["bq_r0.1"] = oracle.ball_query(0.1, 64, pcs, cen).tolist()
    out[kind]["bq_r0.5"] = oracle.ball_query(0.5, 32, pcs, cen).tolist()
    d2, i3 = oracle.three_nn_dist2(pcs[:, :4096], cen)
    out[kind]["nn3"] = i3.tolist()
json.dump(out, open(sys.argv[2] + ".json", "w"))
''' % ROOT


def main():
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    import tempfile
    tmp = tempfile.mkdtemp()
    res = {}
    for mode in (0, 1, 2):
        base = os.path.join(tmp, "m%d" % mode)
        ref = os.path.join(tmp, "m0") if mode else "-"
        subprocess.check_call([sys.executable, "-B", "-c", WORKER, str(scenes), base, ref], env=dict(os.environ, WS3D_DIST_MODE=str(mode)))
        res[mode] = json.load(open(base + ".json"))
    import numpy as np
    report = {"scenes": scenes, "n_points": 16384, "npoint": 4096,
              "note": "relative to mode 0 (the default); searches use mode 0's centres; see csrc/common.h for the three expressions"}
    for kind in ("lidar", "uniform"):
        r = {}
        for mode in (1, 2):
            a, b = res[0][kind], res[mode][kind]
            fps0, fps1 = np.array(a["fps"]), np.array(b["fps"])
            diff_scene = (fps0 != fps1).any(axis=1)
            first = [int(np.argmax(fps0[s] != fps1[s])) for s in range(scenes) if diff_scene[s]]
            m = {"fps_sequences_that_differ": float(diff_scene.mean()), "fps_first_differing_step": first,
                 "fps_same_SET_of_indices": float(np.mean([set(fps0[s]) == set(fps1[s]) for s in range(scenes)]))}
            for key in ("bq_r0.1", "bq_r0.5", "nn3"):
                x, y = np.array(a[key]), np.array(b[key])
                m[key + "_rows_that_differ"] = float((x != y).any(axis=-1).mean())
            r["mode%d" % mode] = m
        report[kind] = r
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
