#!/usr/bin/env python
"""Golden fixture for the KITTI ingest (ws3d_amd/kitti_io.py), produced by running the REFERENCE's
own data path on a synthetic KITTI directory: ``python -B tests/golden/make_golden_kitti.py``.

Runs only in the build container (imports /root/reference).  The directory is regenerated
deterministically by ``ws3d_amd.synth.write_kitti_tree`` (seeded), so the fixture holds only
expected OUTPUTS: hashes + sampled values of ``KittiRCNNDataset(mode='TEST')[i]['pts_input']``,
``Calibration`` transforms, ``Object3d`` fields and the text ``save_kitti_format`` writes.
``cv2`` is absent from the image and only referenced by a disabled method of the reference
(kitti_dataset.py:28-34): an empty module object stands in for the import.
"""
from __future__ import annotations

import ast
import json
import logging
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

import numpy as np  # noqa: E402

import make_golden as mg  # noqa: E402
from ws3d_amd import synth  # noqa: E402

SCENES = [(7, 60000, 1), (8, 9000, 2)]      # (sample id, points in the scan, seed): subsample path, tiling path
NP_SEED = 1234


def main():
    mg.install_reference_shims()
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    from lib.config import cfg, cfg_from_file
    cfg_from_file(os.path.join(REF, "tools", "cfgs", "weaklyRPN.yaml"))
    from lib.datasets.kitti_rcnn_dataset import KittiRCNNDataset
    import lib.utils.kitti_utils as ref_ku

    root = tempfile.mkdtemp(prefix="ws3d_kitti_")
    synth.write_kitti_tree(root, SCENES)
    logging.basicConfig(level=logging.WARNING)
    ds = KittiRCNNDataset(root_dir=root, npoints=16384, split="val", classes="Car", mode="TEST", random_select=True,
                          logger=logging.getLogger("golden"))
    out = {"generator": "tests/golden/make_golden_kitti.py", "scenes": SCENES, "np_seed": NP_SEED, "samples": []}
    np.random.seed(NP_SEED)
    for i in range(len(SCENES)):
        s = ds[i]
        p = np.ascontiguousarray(s["pts_input"])
        pos, val = mg.sample(p, 64, seed=i)
        out["samples"].append({"sample_id": int(s["sample_id"]), "shape": list(p.shape), "dtype": str(p.dtype),
                               "sha256": mg.sha(p), "pos": pos.tolist(), "val": [float(v) for v in val]})

    calib = ds.get_calib(7)
    rng = np.random.default_rng(3)
    pts = synth.velodyne_scan(50, 9)[:, :3]
    rect = calib.lidar_to_rect(pts)
    img, depth = calib.rect_to_img(rect)
    back = calib.img_to_rect(img[:, 0], img[:, 1], depth)
    out["calib"] = {"rect": rect.tolist(), "img": img.tolist(), "depth": depth.tolist(), "img_to_rect": back.tolist(),
                    "tx": float(calib.tx), "ty": float(calib.ty)}
    objs = ds.get_label(7)
    out["labels"] = [{"cls_type": o.cls_type, "level": int(o.level), "ry": o.ry, "score": o.score,
                      "pos": [float(v) for v in o.pos], "hwl": [o.h, o.w, o.l]} for o in objs]

    # save_kitti_format lives in a script that parses argv at import time: lift that one function
    src = open(os.path.join(REF, "tools", "eval_auto.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "save_kitti_format"][0]
    ns = {"np": np, "os": os, "kitti_utils": ref_ku, "cfg": cfg}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "eval_auto.py", "exec"), ns)
    boxes = synth.proposal_boxes(1, 12, 77)[0].astype(np.float32)
    boxes[:, 2] = np.abs(boxes[:, 2]) + 6.0
    boxes[3, 2] = 1.5                      # close to the camera: 2-D box taller than 80 % of the image -> dropped
    scores = rng.normal(0, 2, 12).astype(np.float32)
    outdir = tempfile.mkdtemp(prefix="ws3d_kitti_out_")
    ns["save_kitti_format"](7, calib, boxes, outdir, scores, ds.get_image_shape(7))
    out["result_file"] = {"boxes_config": [1, 12, 77], "text": open(os.path.join(outdir, "000007.txt")).read()}

    with open(os.path.join(HERE, "kitti_ingest.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote kitti_ingest.json:", [s["sha256"][:12] for s in out["samples"]], len(out["result_file"]["text"].splitlines()), "result lines")


if __name__ == "__main__":
    main()
