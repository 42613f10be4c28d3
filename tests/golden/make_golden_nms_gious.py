#!/usr/bin/env python
"""Golden vectors for the rotated-NMS DECISIONS (suppression bits + greedy keep list) from the reference's second, independent
rotated-rectangle intersection: ``lib/utils/gious.py`` ``rbbox_to_corners`` (:460-491) + ``rinter_area_compute`` (:572-585, i.e.
``compute_vertex`` / ``sort_vertex`` / ``area_polygon``), imported from /root/reference where it lies and run on the CPU here (build
container only; the reference tree never travels): ``python -B tests/golden/make_golden_nms_gious.py``.

What the CUDA path decides (iou3d_kernel.cu:250-292 + the host sweep iou3d.cpp:100-116): bit (i, j), j > i, of the mask is
``iou_bev(box_i, box_j) > thresh`` on score-sorted boxes; the sweep keeps i iff no kept earlier box suppresses it.  Here the SAME
decisions are derived from gious.py's intersection area (a different algorithm: corner-in-rectangle tests by dot products, edge
intersections by parametric solve, angular sort about the centroid, triangle-fan area) with
``iou = inter / max(area_i + area_j - inter, 1e-8)`` in float64, for every pair whose bounding circles touch (all other pairs are
disjoint: IoU 0).  A different algorithm in float32 agrees with the kernel to ~1e-5, not to the bit, so the box sets are CLEANED:
of every pair with ``|iou - thresh| <= 2e-3`` the lower-scored box is dropped (dropping a box creates no new pair), hence every
remaining decision has a margin above 2e-3 and MUST come out the same in the oracle and in the HIP kernels -- including the
pruning bounds of csrc/iou3d.hip that decide bits without computing the intersection.

Fixture = data only: the boxes (n, 7) [x, y_bottom, z, h, w, l, ry] in score order, thresh, the packed upper-triangle decision matrix
(small sets) or its sha256 + per-row popcounts (the 9000-box set), the smallest margin, and the keep list."""
from __future__ import annotations

import hashlib
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

CLEAN = 2e-3


def clustered(n_obj, per_obj, scattered, seed):
    """RPN-like proposals: per_obj jittered copies of n_obj car boxes (jitter drawn per proposal from tight / medium / loose, so that
    pairs land on both sides of the threshold) + scattered singles"""
    rng = np.random.default_rng(seed)
    cars = synth.random_boxes3d(n_obj, seed * 13 + 1)
    out = []
    for c in cars:
        p = np.repeat(c[None], per_obj, 0)
        s = rng.choice([0.04, 0.12, 0.35], per_obj)
        p[:, 0] += rng.normal(0, 1, per_obj) * s
        p[:, 2] += rng.normal(0, 1, per_obj) * s
        p[:, 6] += rng.normal(0, 1, per_obj) * s * 0.5 + rng.choice([0.0, np.pi], per_obj, p=[0.8, 0.2])     # heading flips
        p[:, 3:6] *= (1 + rng.uniform(-0.08, 0.08, (per_obj, 3)))
        out.append(p)
    out.append(synth.random_boxes3d(scattered, seed * 13 + 2))
    b = np.concatenate(out).astype(np.float32)
    return b[rng.permutation(len(b))]


SETS = [
    # name, thresh, boxes, how many to keep after the cleaning (the leading ones in score order)
    ("c5_512", 0.7, lambda: synth.proposal_boxes(1, 512, 5)[0], 512),            # the boxes of BASELINE configs[4] (497 stay: not a multiple of 64)
    ("rpn_1536", 0.8, lambda: clustered(24, 70, 240, 41), 1536),
    ("rpn_9000", 0.8, lambda: clustered(60, 180, 2600, 42), 9000),               # RPN_PRE_NMS_TOP_N boxes, TEST threshold
]


def reference_intersections(ref, b, i, j):
    """gious.py's intersection area of boxes b[i] and b[j] (aligned pairs), rbbox = (x, z, l, w, ry) as ious_3D feeds it (:1047-1048)"""
    out = np.zeros(len(i), np.float32)
    corners, area = ref.rbbox_to_corners(), ref.rinter_area_compute()
    rb = torch.from_numpy(np.ascontiguousarray(b[:, [0, 2, 5, 4, 6]]))
    with torch.no_grad():
        c = corners(rb)
        for s in range(0, len(i), 20000):
            out[s:s + 20000] = area(c[i[s:s + 20000]], c[j[s:s + 20000]]).numpy()
            print("   ", min(s + 20000, len(i)), "/", len(i), "pairs", flush=True)
    return out


def pairs_that_can_touch(b):
    x, z = b[:, 0].astype(np.float64), b[:, 2].astype(np.float64)
    rad = 0.5 * np.hypot(b[:, 4].astype(np.float64), b[:, 5].astype(np.float64))
    ii, jj = [], []
    for s in range(0, len(b), 512):
        d = np.hypot(x[s:s + 512, None] - x[None], z[s:s + 512, None] - z[None])
        a, c = np.nonzero(d < (rad[s:s + 512, None] + rad[None]) * 1.001 + 1e-3)
        a += s
        m = a < c
        ii.append(a[m])
        jj.append(c[m])
    return np.concatenate(ii), np.concatenate(jj)


def main():
    spec = importlib.util.spec_from_file_location("ref_gious", REF_FILE)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out, names = {}, []
    for name, thr, make, limit in SETS:
        b = make()
        scores = synth.distinct_scores(len(b), 900 + len(b))
        b = np.ascontiguousarray(b[np.argsort(-scores, kind="stable")])       # score order: what nms_gpu hands the kernel (iou3d_utils.py:67-69)
        i, j = pairs_that_can_touch(b)
        print(name, len(b), "boxes,", len(i), "pairs whose bounding circles touch", flush=True)
        inter = reference_intersections(ref, b, i, j).astype(np.float64)
        area = b[:, 4].astype(np.float64) * b[:, 5].astype(np.float64)
        iou = inter / np.maximum(area[i] + area[j] - inter, 1e-8)
        close = np.abs(iou - thr) <= CLEAN
        drop = np.unique(j[close])                                           # j > i: the lower-scored box of each close call
        keep_box = np.ones(len(b), bool)
        keep_box[drop] = False
        renum = np.cumsum(keep_box) - 1
        ok = keep_box[i] & keep_box[j]
        b, i, j, iou = b[keep_box], renum[i[ok]], renum[j[ok]], iou[ok]
        ok = j < limit                                                       # i < j: a prefix in score order keeps every decision among its boxes
        b, i, j, iou = b[:limit], i[ok], j[ok], iou[ok]
        n = len(b)
        margin = float(np.abs(iou - thr).min())
        above = iou > thr
        # decisions as an upper-triangle bit matrix, row i / column j, packed little-endian per 64 columns like the mask words
        dec = np.zeros((n, n), bool)
        dec[i[above], j[above]] = True
        # the greedy sweep of iou3d.cpp:100-116 over these decisions
        removed = np.zeros(n, bool)
        keep = []
        for r in range(n):
            if not removed[r]:
                keep.append(r)
                removed |= dec[r]
        keep = np.asarray(keep, np.int64)
        packed = np.packbits(dec, axis=1, bitorder="little")
        print("  ", n, "boxes after dropping", len(drop), "close calls; pairs above", int(above.sum()), "smallest margin %.4f" % margin,
              "kept", len(keep), flush=True)
        names.append(name)
        out[name + "_boxes"] = b
        out[name + "_thresh"] = np.float32(thr)
        out[name + "_keep"] = keep
        out[name + "_margin"] = np.float64(margin)
        out[name + "_rowcount"] = dec.sum(1).astype(np.int32)
        out[name + "_sha256"] = np.array(hashlib.sha256(packed.tobytes()).hexdigest())
        if n <= 2048:
            out[name + "_dec"] = packed
    np.savez_compressed(os.path.join(HERE, "nms_gious.npz"), sets=np.array(names), **out)


if __name__ == "__main__":
    main()
