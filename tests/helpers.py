"""Independent numpy re-derivations used to cross-check the C oracle (and, on the
GPU box, the HIP kernels).  Nothing here shares code with oracle/ws3d_oracle.c."""
from __future__ import annotations

import numpy as np


# ----------------------------------------------------------------------------- exact fmaf
def fmaf_np(a, b, c):
    """Correctly rounded float32 fma(a,b,c), vectorised.

    a*b is exact in float64 (24+24 <= 53 bits).  The float64 sum s = fl(p + c) is
    corrected to ROUND-TO-ODD with the exact TwoSum error term, after which the
    final float64 -> float32 rounding cannot double-round (53 >= 24 + 2)."""
    a = np.asarray(a, dtype=np.float32).astype(np.float64)
    b = np.asarray(b, dtype=np.float32).astype(np.float64)
    c = np.asarray(c, dtype=np.float32).astype(np.float64)
    p = a * b
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)  # exact: p + c = s + err
    s_arr = np.atleast_1d(s)
    e_arr = np.atleast_1d(err)
    si = s_arr.view(np.int64).copy()
    inexact = e_arr != 0
    even = (si & 1) == 0
    # the exact value lies strictly between s and its neighbour in the direction of err;
    # of those two doubles exactly one has an odd mantissa: pick it.
    toward_inf = (e_arr > 0) == (s_arr > 0)  # neighbour has larger magnitude
    adj = np.where(toward_inf, 1, -1)
    si = np.where(inexact & even, si + adj, si)
    out = si.view(np.float64).astype(np.float32)
    return out.reshape(np.shape(s))


def sqdist_np(p, q):
    """d = fmaf(dz,dz, fmaf(dx,dx, dy*dy)) with d* = p* - q* in float32."""
    p = np.asarray(p, dtype=np.float32)
    q = np.asarray(q, dtype=np.float32)
    dx = p[..., 0] - q[..., 0]
    dy = p[..., 1] - q[..., 1]
    dz = p[..., 2] - q[..., 2]
    return fmaf_np(dz, dz, fmaf_np(dx, dx, dy * dy))


# ----------------------------------------------------------------------------- FPS
def bitrev(v: np.ndarray, bits: int) -> np.ndarray:
    v = np.asarray(v, dtype=np.int64)
    r = np.zeros_like(v)
    for i in range(bits):
        r |= ((v >> i) & 1) << (bits - 1 - i)
    return r


def opt_n_threads(n: int) -> int:
    p = 1
    while p * 2 <= n:
        p *= 2
    return max(min(p, 1024), 1)


def fps_np(xyz: np.ndarray, m: int) -> np.ndarray:
    """Rank formulation of the reference FPS: every step takes the
    argmax of the running min-distance; exact ties go to the candidate whose owner thread
    (k mod bs) has the smallest bit-reversed id, then to the smallest k."""
    xyz = np.asarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    bs = opt_n_threads(n)
    bits = bs.bit_length() - 1
    k = np.arange(n)
    rank = bitrev(k % bs, bits) * (n // bs + 1) + k // bs  # smaller rank wins ties
    temp = np.full(n, 1e10, dtype=np.float32)
    idx = np.zeros(m, dtype=np.int32)
    old = 0
    for j in range(1, m):
        d = sqdist_np(xyz, xyz[old][None, :])
        temp = np.minimum(d, temp)
        mx = temp.max()
        cand = np.nonzero(temp == mx)[0]
        old = int(cand[np.argmin(rank[cand])])
        idx[j] = old
    return idx


# ----------------------------------------------------------------------------- ball query / knn
def ball_query_np(radius: float, nsample: int, xyz: np.ndarray, new_xyz: np.ndarray) -> np.ndarray:
    xyz = np.asarray(xyz, dtype=np.float32)
    new_xyz = np.asarray(new_xyz, dtype=np.float32)
    r2 = np.float32(radius) * np.float32(radius)
    out = np.zeros((new_xyz.shape[0], nsample), dtype=np.int32)
    for i, c in enumerate(new_xyz):
        d2 = sqdist_np(c[None, :], xyz)
        hit = np.nonzero(d2 < r2)[0][:nsample]
        if hit.size:
            out[i, :] = hit[0]
            out[i, :hit.size] = hit
    return out


def three_nn_np(unknown: np.ndarray, known: np.ndarray):
    unknown = np.asarray(unknown, dtype=np.float32)
    known = np.asarray(known, dtype=np.float32)
    n, m = unknown.shape[0], known.shape[0]
    d2 = np.full((n, 3), np.inf, dtype=np.float32)
    idx = np.zeros((n, 3), dtype=np.int32)
    for i, u in enumerate(unknown):
        d = sqdist_np(u[None, :], known)
        order = np.argsort(d, kind="stable")[:3]
        d2[i, :order.size] = d[order]
        idx[i, :order.size] = order
    return d2, idx


# ----------------------------------------------------------------------------- rotated boxes
def bev_corners(box):
    """[x1,y1,x2,y2,ry] -> 4 corners rotated like iou3d_kernel.cu:98-102 (float64)."""
    x1, y1, x2, y2, a = [float(v) for v in box]
    cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
    c, s = np.cos(a), np.sin(a)
    pts = []
    for (px, py) in ((x1, y1), (x2, y1), (x2, y2), (x1, y2)):
        dx, dy = px - cx, py - cy
        pts.append((dx * c + dy * s + cx, -dx * s + dy * c + cy))
    return pts


def _clip(subject, a, b):
    def inside(p):
        return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]) >= 0

    def inter(p, q):
        x1, y1, x2, y2 = p[0], p[1], q[0], q[1]
        x3, y3, x4, y4 = a[0], a[1], b[0], b[1]
        den = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
        t = ((x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)) / den
        return (x1 + t * (x2 - x1), y1 + t * (y2 - y1))

    out = []
    for i in range(len(subject)):
        cur, prev = subject[i], subject[i - 1]
        if inside(cur):
            if not inside(prev):
                out.append(inter(prev, cur))
            out.append(cur)
        elif inside(prev):
            out.append(inter(prev, cur))
    return out


def poly_area(p):
    a = 0.0
    for i in range(len(p)):
        a += p[i - 1][0] * p[i][1] - p[i][0] * p[i - 1][1]
    return abs(a) / 2


def overlap_sh(box_a, box_b) -> float:
    """Sutherland-Hodgman rotated-rectangle intersection area in float64 (independent of
    the reference's vertex-collect + atan2-sort + fan algorithm)."""
    pa, pb = bev_corners(box_a), bev_corners(box_b)

    def ccw(p):
        s = 0.0
        for i in range(len(p)):
            s += p[i - 1][0] * p[i][1] - p[i][0] * p[i - 1][1]
        return p if s > 0 else p[::-1]

    pa, pb = ccw(pa), ccw(pb)
    out = pa
    for i in range(4):
        if not out:
            return 0.0
        out = _clip(out, pb[i - 1], pb[i])
    return poly_area(out) if len(out) >= 3 else 0.0


def greedy_nms_from_iou(iou: np.ndarray, thresh: float) -> np.ndarray:
    n = iou.shape[0]
    removed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        removed[i + 1:] |= iou[i, i + 1:] > thresh
    return np.asarray(keep, dtype=np.int64)
