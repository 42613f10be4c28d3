"""Stage-1 inference over a KITTI directory: ``.bin`` scans in, KITTI-format result files out.

    python -m ws3d_amd.infer_kitti --root /data/KITTI/object --split val --out results/ [--ckpt x.pth]

The counterpart of the reference's ``tools/eval_auto.py`` / ``generate_box_dataset.py`` drivers
for the part of the pipeline this repository implements: ingest
(``ws3d_amd.kitti_io``) -> Stage-1 forward -> on-device proposal stage -> ``save_kitti_format``.
Weights come from a reference checkpoint (``model_state``, identical keys) or, without one, from
the seeded initialisation used by the benchmarks.
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from . import kitti_io, loader, stage1
from .pipeline import Stage1Pipeline, ensure_hw_queues


def run(root: str, split: str, out_dir: str, batch: int = 8, ckpt: str | None = None, npoints: int = 16384,
        seed: int = 666, device: str = "cuda:0", cfg: stage1.RPNConfig = stage1.DEFAULT_CFG, depth: int = 4,
        workers: int = 0) -> list:
    """returns the list of result files written (one per scene, possibly empty)"""
    dev = torch.device(device)
    model = stage1.Stage1Net(mode="TEST", cfg=cfg).to(dev).eval()
    if ckpt:
        state = torch.load(ckpt, map_location="cpu")
        model.load_state_dict(state.get("model_state", state), strict=True)
    else:
        from .seeded import seeded_state_dict
        model.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 0))
    rng = np.random.RandomState(seed)           # eval_auto.py:139 seeds numpy with 666 before sampling
    scenes = kitti_io.KittiScenes(root, split, npoints=npoints, rng=rng)
    os.makedirs(out_dir, exist_ok=True)
    written = []
    starts = list(range(0, len(scenes), batch))
    loaded = {}

    def batches():      # scans read + sampled in `workers` loader processes when asked for (ws3d_amd.loader)
        ids = [list(range(i0, min(i0 + batch, len(scenes)))) for i0 in starts]
        for i0, items in zip(starts, loader.item_batches(scenes, ids, workers, ahead=depth + 1, seed=seed)):
            loaded[i0] = items
            yield kitti_io.collate_scenes(items)["pts_input"]

    # `depth` batches in flight (ws3d_amd.pipeline): the next scans are read and sampled on the host while the device works
    pipe = Stage1Pipeline(model, cfg, batch=batch, n_points=npoints, depth=depth, device=dev)
    for i0, n_valid, out in pipe.map(batches()):
        boxes, scores, count = out["boxes"].cpu().numpy(), out["scores"].cpu().numpy(), out["count"].cpu().numpy()
        for j, s in enumerate(loaded.pop(i0)[:n_valid]):
            sid, k = s["sample_id"], int(count[j])
            written.append(kitti_io.save_kitti_format(sid, scenes.get_calib(sid), boxes[j, :k], out_dir,
                                                      scores[j, :k], scenes.get_image_shape(sid), "Car"))
    return written


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--root", required=True)
    ap.add_argument("--split", default="val")
    ap.add_argument("--out", required=True)
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--npoints", type=int, default=16384)
    ap.add_argument("--pipeline_depth", type=int, default=4, help="batches in flight on separate HIP streams")
    ap.add_argument("--workers", type=int, default=0, help="loader processes reading / sampling the next scans")
    a = ap.parse_args()
    ensure_hw_queues()
    files = run(a.root, a.split, a.out, a.batch, a.ckpt, a.npoints, depth=a.pipeline_depth, workers=a.workers)
    print(f"{len(files)} result files in {a.out}")


if __name__ == "__main__":
    main()
