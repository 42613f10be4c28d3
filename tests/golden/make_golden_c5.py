#!/usr/bin/env python
"""Fixture for BASELINE.json configs[4] ("C5"): N = 65536 points, 512 proposals, roipool3d (512 sampled
points, 128 feature channels, extra width 1.0) + rotated NMS (threshold 0.7), through the REFERENCE's own
code: ``roipool_pc_cpu`` -> the reference's COMPILED C++ (roipool3d.cpp:97-195, built into oracle/_ref by
oracle/build_ref.py) for the pooling, ``iou3d_utils.nms_gpu`` (its sort / index-back, kernels backed by
the CPU oracle) for the NMS.  Build container only (same shims as make_golden.py):

    python -B tests/golden/make_golden_c5.py   ->  tests/golden/c5_roipool_nms.npz + c5_roipool_nms.json

Data only: seeds/params, the 512 empty flags, sha256 + 256 sampled values of the (512, 512, 131) pooled
tensor, and the NMS keep list."""
from __future__ import annotations

import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import make_golden as mg  # noqa: E402
from ws3d_amd import synth  # noqa: E402

CASE = {"config_id": 5, "n": 65536, "boxes": 512, "channels": 128, "sampled": 512, "extra_width": 1.0, "nms_thresh": 0.7,
        "feat_seed": 505}


def inputs():
    pc = synth.make_batch("lidar", 1, CASE["n"], CASE["config_id"])
    boxes = synth.proposal_boxes(1, CASE["boxes"], CASE["config_id"])
    boxes[0, 100:108, 0] += 500.0          # eight proposals far from every point: the empty-flag path
    feat = np.random.default_rng(CASE["feat_seed"]).standard_normal((1, CASE["n"], CASE["channels"])).astype(np.float32)
    scores = synth.distinct_scores(CASE["boxes"], CASE["config_id"])
    return pc[:, :, :3].copy(), boxes, feat, scores


def main():
    mg.install_reference_shims()
    from lib.utils import kitti_utils as ref_kitti
    from lib.utils.iou3d import iou3d_utils as ref_iou
    from lib.utils.roipool3d import roipool3d_utils as ref_roi
    xyz, boxes, feat, scores = inputs()
    # the reference's compiled CPU twin on the enlarged boxes (roipool3d_utils.py:62-110)
    enlarged = ref_kitti.enlarge_box3d(boxes[0], CASE["extra_width"])
    pts_p, feat_p, empty = ref_roi.roipool_pc_cpu(torch.from_numpy(xyz[0]), torch.from_numpy(feat[0]), torch.from_numpy(enlarged),
                                                  CASE["sampled"])
    pooled = np.concatenate([mg._np(pts_p), mg._np(feat_p)], axis=2).astype(np.float32)          # (512, 512, 3 + 128)
    # ... and the GPU-side wrapper (enlargement inside, leaf kernel = the CPU oracle) agrees with it
    pooled_w, empty_w = ref_roi.roipool3d_gpu(torch.from_numpy(xyz), torch.from_numpy(feat), torch.from_numpy(boxes),
                                              CASE["extra_width"], sampled_pt_num=CASE["sampled"])
    assert np.array_equal(mg._np(pooled_w)[0], pooled) and np.array_equal(mg._np(empty_w)[0].astype(np.int32), mg._np(empty).astype(np.int32))
    bev = ref_kitti.boxes3d_to_bev_torch(torch.from_numpy(boxes[0]))
    keep = mg._np(ref_iou.nms_gpu(bev, torch.from_numpy(scores), CASE["nms_thresh"])).astype(np.int64)
    pos, val = mg.sample(pooled, 256, seed=5)
    np.savez_compressed(os.path.join(HERE, "c5_roipool_nms.npz"), empty=mg._np(empty).astype(np.int32), pooled_pos=pos, pooled_val=val,
                        nms_keep=keep)
    meta = dict(CASE, generator="tests/golden/make_golden_c5.py", pooled_shape=list(pooled.shape), pooled_sha256=mg.sha(pooled),
                non_empty=int((mg._np(empty) == 0).sum()), kept=int(keep.size), roipool_by="reference compiled C++ (roipool3d.cpp:97-195)")
    json.dump(meta, open(os.path.join(HERE, "c5_roipool_nms.json"), "w"), indent=1)
    for f in ("c5_roipool_nms.npz", "c5_roipool_nms.json"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
    print(meta["non_empty"], "non-empty boxes,", meta["kept"], "kept by NMS")


if __name__ == "__main__":
    main()
