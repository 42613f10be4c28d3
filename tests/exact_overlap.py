"""Test infrastructure: the intersection area of two rotated BEV rectangles in float64 by Sutherland-Hodgman clipping -- a third,
independent implementation (neither the restated kernel iou3d_kernel.cu:108-212 nor the reference's gious.py) used to say WHICH of
the two float32 implementations a difference between them belongs to.  Boxes are (x, y_bottom, z, h, w, l, ry) with l along the box's
own x axis (kitti_utils.boxes3d_to_bev_torch, lib/utils/kitti_utils.py:134-147) and the kernel's rotation convention
(rotate_around_center, iou3d_kernel.cu:98-102)."""
import numpy as np


def _corners(b):
    x, z, l, w, ry = b[0], b[2], b[5], b[4], b[6]
    c, s = np.cos(ry), np.sin(ry)
    p = np.array([[-l / 2, -w / 2], [l / 2, -w / 2], [l / 2, w / 2], [-l / 2, w / 2]])
    return np.stack([x + p[:, 0] * c + p[:, 1] * s, z - p[:, 0] * s + p[:, 1] * c], 1)


def _signed_area2(p):
    return float(np.sum(p[:, 0] * np.roll(p[:, 1], -1) - p[:, 1] * np.roll(p[:, 0], -1)))


def _ccw(p):
    return p if _signed_area2(p) > 0 else p[::-1]


def _clip(subject, clipper):
    def inside(p, a, b):
        return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]) >= 0

    def cut(p, q, a, b):
        d1, d2 = q - p, b - a
        t = ((a[0] - p[0]) * d2[1] - (a[1] - p[1]) * d2[0]) / (d1[0] * d2[1] - d1[1] * d2[0])
        return p + t * d1
    out = list(subject)
    for i in range(len(clipper)):
        a, b = clipper[i], clipper[(i + 1) % len(clipper)]
        inp, out = out, []
        if not inp:
            break
        s = inp[-1]
        for e in inp:
            if inside(e, a, b):
                if not inside(s, a, b):
                    out.append(cut(s, e, a, b))
                out.append(e)
            elif inside(s, a, b):
                out.append(cut(s, e, a, b))
            s = e
    return out


def overlap_bev(a, b) -> float:
    """float64 intersection area of the BEV rectangles of boxes a and b"""
    p = _clip(_ccw(_corners(np.asarray(a, np.float64))), _ccw(_corners(np.asarray(b, np.float64))))
    return 0.0 if len(p) < 3 else 0.5 * abs(_signed_area2(np.array(p)))


def iou3d(a, b) -> float:
    """iou3d_utils.boxes_iou3d_gpu's composition (iou3d_utils.py:21-56) on the exact overlap"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    ov = overlap_bev(a, b)
    oh = max(min(a[1], b[1]) - max(a[1] - a[3], b[1] - b[3]), 0.0)
    return ov * oh / max(a[3] * a[4] * a[5] + b[3] * b[4] * b[5] - ov * oh, 1e-7)
