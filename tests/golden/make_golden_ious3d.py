#!/usr/bin/env python
"""Golden vectors for the rotated-BEV overlap / 3-D IoU from a reference-held CPU implementation that is NOT the CUDA kernel:
``lib/utils/gious.py`` (the reference's differentiable IoU of its Stage-2 losses, lib/net/train_functions.py:394) computes the
intersection of two rotated rectangles in pure PyTorch -- edge / corner vertices (`compute_vertex`, gious.py:17), angular sort
(`sort_vertex`, :300), shoelace area (`area_polygon`, :367) -- and `ious_3D` (:996) composes it with the height overlap into the 3-D
IoU.  Imported from /root/reference where it lies and run on the CPU here (build container only; the reference tree never travels):
``python -B tests/golden/make_golden_ious3d.py``.

Box convention: gious.py takes (x, y, z, sx, h, sz, ry) with sx / sz the extents along the box's own x / z axes; the pointnet boxes
of iou3d_utils.boxes_iou3d_gpu are (x, y, z, h, w, l, ry) with l along the box's x axis (kitti_utils.boxes3d_to_bev_torch), so
sx = l, sz = w, same heading (found by running both; the other three assignments disagree by > 0.2).

A different algorithm in float32, so the vectors pin VALUES to ~1e-4, not bits: they tie the oracle's restatement of
iou3d_kernel.cu:108-221 (and the HIP kernel) to a second implementation the reference itself ships.

Fixture = data only: the box pairs (A, B: (n, 7) float32) and the reference's IoU per pair."""
from __future__ import annotations

import importlib.util
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_FILE = "/root/reference/lib/utils/gious.py"

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ws3d_amd import synth  # noqa: E402


def to_gious(b):
    g = np.zeros_like(b)
    g[:, 0:3] = b[:, 0:3]
    g[:, 3], g[:, 4], g[:, 5], g[:, 6] = b[:, 5], b[:, 3], b[:, 4], b[:, 6]
    return torch.from_numpy(g)


def main():
    spec = importlib.util.spec_from_file_location("ref_gious", REF_FILE)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(2025)
    A, B = [], []
    # regimes: jittered copies (high IoU), shifted / rotated neighbours, boxes of different sizes, far pairs (IoU 0), contained boxes
    for k, (shift, rot, scale) in enumerate([(0.05, 0.02, 0.0), (0.5, 0.2, 0.0), (1.5, 1.0, 0.0), (0.3, 0.1, 0.3), (8.0, 3.0, 0.0), (0.1, 0.05, 0.6)]):
        n = 80
        a = synth.random_boxes3d(n, 100 + k)
        b = a.copy()
        b[:, 0] += rng.normal(0, shift, n)
        b[:, 2] += rng.normal(0, shift, n)
        b[:, 1] += rng.normal(0, 0.2, n)
        b[:, 6] += rng.normal(0, rot, n)
        b[:, 3:6] *= (1 + rng.uniform(-scale, scale, (n, 3))).astype(np.float32)
        A.append(a)
        B.append(b)
    A, B = np.concatenate(A).astype(np.float32), np.concatenate(B).astype(np.float32)
    with torch.no_grad():
        iou = ref.ious_3D()(to_gious(A), to_gious(B)).numpy()[:, 0].astype(np.float32)
    out = os.path.join(HERE, "ious3d_gious.npz")
    np.savez_compressed(out, A=A, B=B, iou3d=iou)
    print("wrote", out, A.shape, "IoU range", float(iou.min()), float(iou.max()), "pairs above 0.5:", int((iou > 0.5).sum()), "zero:", int((iou == 0).sum()))


if __name__ == "__main__":
    main()
