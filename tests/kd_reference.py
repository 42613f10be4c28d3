"""An INDEPENDENT derivation of the two search operators the reference holds no second implementation of (pointnet2_utils.py:218-220:
CUDA only): ball_query (ball_query_gpu.cu:9-45) and three_nn (interpolate_gpu.cu:9-52), from float64 geometry through scipy's cKDTree
-- another algorithm (tree search, exact float64 distances) than the oracle's literal loops and the HIP kernels' grid / slab searches.

float64 geometry cannot reproduce the float32 kernels' decisions where a distance lies within float32 rounding of the decision
boundary, so the comparison is MARGIN-CLEANED (the method of tests/golden/make_golden_nms_gious.py): a centre / query whose outcome
depends on such a near-tie is reported as ambiguous and skipped; every other row must come out identical -- indices bit for bit.

    ball_query : in-ball test d2 < r2.  |d2 - r2| <= EPS_REL * r2 for any point that could enter the list -> the centre is ambiguous.
    three_nn   : the three smallest d2, ties to the lower index (strict '<' while scanning upwards).  Ambiguous when two of the four
                 smallest distances are closer than EPS_REL * d2 (+ EPS_ABS), i.e. float32 rounding could swap their order or a tie
                 could be broken the other way.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree

EPS_REL = 2e-5       # float32: the kernels' d2 carries <= ~4 ulp of 6e-8 relative error on each squared term, coordinates up to 80 m
EPS_ABS = 1e-9


def ball_query_kd(radius: float, nsample: int, xyz: np.ndarray, new_xyz: np.ndarray):
    """xyz (n, 3), new_xyz (m, 3) float32 -> (idx (m, nsample) int32, ambiguous (m,) bool).
    Reference semantics: scan k = 0 .. n-1, the first hit fills the whole row, hits are written in index order until nsample of
    them are found; a row without a hit stays 0."""
    x64, c64 = xyz.astype(np.float64), new_xyz.astype(np.float64)
    tree = cKDTree(x64)
    r2 = float(np.float32(radius) * np.float32(radius))          # the kernels compare with the float32 product radius * radius
    r_out = np.sqrt(r2 * (1 + EPS_REL)) + 1e-12
    cand = tree.query_ball_point(c64, r_out)                     # everything inside the ball enlarged by the margin
    m = new_xyz.shape[0]
    idx = np.zeros((m, nsample), dtype=np.int32)
    amb = np.zeros((m,), dtype=bool)
    for j in range(m):
        c = np.sort(np.asarray(cand[j], dtype=np.int64))
        if c.size == 0:
            continue
        d2 = ((x64[c] - c64[j]) ** 2).sum(1)
        inside = d2 < r2
        near = np.abs(d2 - r2) <= EPS_REL * r2
        hits = c[inside]
        take = hits[:nsample]
        if take.size:
            idx[j, :] = take[0]
            idx[j, :take.size] = take
        # a near-boundary point matters if it could enter the list: its index is below the last taken one, or the list is not full
        last = take[-1] if take.size == nsample else np.iinfo(np.int64).max
        amb[j] = bool((near & (c <= last)).any())
    return idx, amb


def three_nn_kd(unknown: np.ndarray, known: np.ndarray):
    """unknown (n, 3), known (m, 3) float32 -> (dist2 (n, 3) float64, idx (n, 3) int32, ambiguous (n,) bool)"""
    u64, k64 = unknown.astype(np.float64), known.astype(np.float64)
    tree = cKDTree(k64)
    kq = min(8, known.shape[0])
    d, i = tree.query(u64, k=kq)
    d2 = d * d
    # order by (distance, index): the scan keeps the lower index on an exact tie
    n = unknown.shape[0]
    idx = np.zeros((n, 3), dtype=np.int32)
    out_d = np.zeros((n, 3), dtype=np.float64)
    amb = np.zeros((n,), dtype=bool)
    for q in range(n):
        o = sorted(range(kq), key=lambda t: (d2[q, t], i[q, t]))
        dd, ii = d2[q, o], i[q, o]
        idx[q], out_d[q] = ii[:3], dd[:3]
        gaps = np.diff(dd[:4])
        amb[q] = bool((gaps <= EPS_REL * dd[1:4] + EPS_ABS).any())
        if kq == 8 and dd[7] - dd[3] <= EPS_REL * dd[7] + EPS_ABS:      # more than 8 near-equal neighbours: the query window may be short
            amb[q] = True
    return out_d, idx, amb
