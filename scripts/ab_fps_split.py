"""The level-1 sampling kernel split over G workgroups per scene (WS3D_FPS_SPLIT=2|4, fps_bucket.hip round 5) against the
one-workgroup kernel: same index tensors (incl. tie cases), time per launch at several batch sizes.
    python scripts/ab_fps_split.py            # spawns one child per G (the switch is read once per process)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [("hdl64", 8), ("lidar", 8), ("dups", 8), ("hdl64", 1), ("hdl64", 16), ("hdl64", 64)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from ws3d_amd import compat as c, synth
    out = {}
    for kind, B in CASES:
        pc = np.stack([synth.cloud("lidar" if kind == "dups" else kind, 16384, 7000 + s, dup_frac=0.02 if kind == "dups" else 0.0)[:, :3] for s in range(B)])
        if kind == "dups":
            pc[:, 8000:8200] = pc[:, 100:300]          # exact duplicates far apart in index
        xyz = torch.from_numpy(np.ascontiguousarray(pc)).cuda()
        idx = torch.empty((B, 4096), dtype=torch.int32, device="cuda"); new = torch.empty((B, 4096, 3), device="cuda")
        for _ in range(3):
            c.furthest_point_sampling_gather(B, 16384, 4096, xyz, None, idx, new)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            c.furthest_point_sampling_gather(B, 16384, 4096, xyz, None, idx, new)
        b.record(); torch.cuda.synchronize()
        out["%s_%d" % (kind, B)] = idx.cpu().numpy()
        out["new_%s_%d" % (kind, B)] = new.cpu().numpy()
        print("G=%s %-6s B=%-3d %.4f ms per launch  (%.3f us per sample)" % (os.environ.get("WS3D_FPS_SPLIT", "1"), kind, B, a.elapsed_time(b) / 10, a.elapsed_time(b) / 10 * 1e3 / 4095), flush=True)
    np.savez(sys.argv[2], **out)
    sys.exit(0)

import numpy as np
res = {}
for G in ("0", "4", "2"):
    f = "/tmp/fps_split_%s.npz" % G
    r = subprocess.run(["timeout", "300", sys.executable, __file__, "child", f], env=dict(os.environ, WS3D_FPS_SPLIT=G), capture_output=True, text=True)
    print(r.stdout, end="")
    if r.returncode != 0:
        print("G=%s FAILED rc=%d: %s" % (G, r.returncode, r.stderr[-800:]))
        continue
    res[G] = np.load(f)
for G in res:
    if G == "0":
        continue
    for k in res["0"].files:
        same = np.array_equal(res["0"][k], res[G][k])
        if not same:
            bad = np.argwhere(res["0"][k] != res[G][k])
            print("G=%s %s DIFFERS: %d elements, first at %s: %s vs %s" % (G, k, len(bad), bad[0], res["0"][k][tuple(bad[0])], res[G][k][tuple(bad[0])]))
    print("G=%s: %s" % (G, "all index / coordinate tensors equal the one-workgroup kernel's" if all(np.array_equal(res["0"][k], res[G][k]) for k in res["0"].files) else "MISMATCH"))
