"""End-to-end rate of ws3d_amd.infer_kitti on a synthetic KITTI tree: .bin scans on disk -> frustum filter +
16384-point sampler -> Stage-1 forward -> proposals -> KITTI result files, for a few (workers, depth) settings."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ws3d_amd.pipeline import ensure_hw_queues
ensure_hw_queues()
from ws3d_amd import infer_kitti, synth

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 96
with tempfile.TemporaryDirectory() as tmp:
    root = os.path.join(tmp, "kitti")
    synth.write_kitti_tree(root, [(i, 60000 + 500 * (i % 7), i) for i in range(n_scenes)])
    infer_kitti.run(root, "val", os.path.join(tmp, "warm"), batch=8, depth=2)          # library warm-up, file cache
    for workers, depth in ((0, 1), (0, 4), (8, 4), (16, 8)):
        t0 = time.perf_counter()
        files = infer_kitti.run(root, "val", os.path.join(tmp, "out_%d_%d" % (workers, depth)), batch=8, depth=depth, workers=workers)
        dt = time.perf_counter() - t0
        print("infer_kitti: %d scenes, workers %2d, %d batches in flight: %.1f scenes/s (%.2f s)" % (len(files), workers, depth, len(files) / dt, dt), flush=True)
