"""Synthetic KITTI-shaped inputs (no KITTI data is available offline).

Shapes and ranges follow the reference's Stage-1 input contract:
``(B, 16384, 4) float32`` = xyz in the rect-camera frame inside PC_AREA_SCOPE
x in [-40,40], y in [-3,3], z in [0,70.4] (tools/cfgs/weaklyRPN.yaml:18) plus
``intensity - 0.5`` (lib/datasets/kitti_rcnn_dataset.py:444).  Two generators
(SURVEY.md section 8d / BASELINE.md section 3), both seeded with
``numpy.random.Generator(PCG64(seed))``:

* ``uniform`` -- iid uniform in the box (worst case for ball-query early exit);
* ``lidar``   -- 70 % ground plane y ~ 1.65 +- 0.05 with planar density ~ 1/range,
  25 % on the surfaces of 15 car-sized boxes (CLS_MEAN_SIZE, weaklyRPN.yaml:19),
  5 % uniform clutter, shuffled; optional exact duplicates (KITTI scans with
  < 16384 points are padded by re-sampling, kitti_rcnn_dataset.py:435-441, so
  exact FPS ties are real).
"""
from __future__ import annotations

import numpy as np

PC_AREA_SCOPE = ((-40.0, 40.0), (-3.0, 3.0), (0.0, 70.4))
CLS_MEAN_SIZE = (1.52563191462, 1.62856739989, 3.88311640418)  # h, w, l


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(int(seed)))


def uniform_cloud(n: int, seed: int) -> np.ndarray:
    r = _rng(seed)
    lo = np.array([s[0] for s in PC_AREA_SCOPE], dtype=np.float64)
    hi = np.array([s[1] for s in PC_AREA_SCOPE], dtype=np.float64)
    xyz = r.uniform(lo, hi, size=(n, 3))
    inten = r.uniform(-0.5, 0.5, size=(n, 1))
    return np.concatenate([xyz, inten], axis=1).astype(np.float32)


def random_boxes3d(m: int, seed: int, jitter: float = 0.1) -> np.ndarray:
    """(m,7) [x, y_bottom, z, h, w, l, ry] car-sized boxes inside the scene."""
    r = _rng(seed)
    x = r.uniform(-35.0, 35.0, m)
    z = r.uniform(5.0, 65.0, m)
    yb = r.uniform(1.4, 1.9, m)
    hwl = np.asarray(CLS_MEAN_SIZE)[None, :] * r.uniform(1.0 - jitter, 1.0 + jitter, (m, 3))
    ry = r.uniform(-np.pi, np.pi, m)
    return np.concatenate([x[:, None], yb[:, None], z[:, None], hwl, ry[:, None]], 1).astype(np.float32)


def lidar_cloud(n: int, seed: int, n_cars: int = 15, dup_frac: float = 0.0,
                return_boxes: bool = False):
    r = _rng(seed)
    n_car = int(0.25 * n)
    n_clut = int(0.05 * n)
    n_gnd = n - n_car - n_clut
    # ground: planar density ~ 1/range  <=> range uniform, azimuth uniform
    rng_ = r.uniform(3.0, 75.0, n_gnd * 2)
    az = r.uniform(-np.pi / 4, np.pi / 4, n_gnd * 2) * 1.3
    gx, gz = rng_ * np.sin(az), rng_ * np.cos(az)
    ok = (np.abs(gx) < 40.0) & (gz > 0.0) & (gz < 70.4)
    gx, gz = gx[ok][:n_gnd], gz[ok][:n_gnd]
    if gx.shape[0] < n_gnd:  # top up (rare)
        extra = n_gnd - gx.shape[0]
        gx = np.concatenate([gx, r.uniform(-40, 40, extra)])
        gz = np.concatenate([gz, r.uniform(0, 70.4, extra)])
    gy = 1.65 + r.normal(0.0, 0.05, n_gnd)
    ground = np.stack([gx, gy, gz], 1)
    # cars: points on box surfaces
    boxes = random_boxes3d(n_cars, seed * 7919 + 13, jitter=0.1).astype(np.float64)
    which = r.integers(0, n_cars, n_car)
    u = r.uniform(-0.5, 0.5, (n_car, 3))
    face = r.integers(0, 3, n_car)
    sign = r.integers(0, 2, n_car) * 1.0 - 0.5
    u[np.arange(n_car), face] = sign
    b = boxes[which]
    lx, ly, lz = u[:, 0] * b[:, 5], u[:, 1] * b[:, 3], u[:, 2] * b[:, 4]  # l along x, h along y, w along z
    c, s = np.cos(b[:, 6]), np.sin(b[:, 6])
    cx = b[:, 0] + lx * c + lz * s
    cz = b[:, 2] - lx * s + lz * c
    cy = b[:, 1] - b[:, 3] / 2 + ly
    cars = np.stack([cx, cy, cz], 1)
    lo = np.array([s_[0] for s_ in PC_AREA_SCOPE])
    hi = np.array([s_[1] for s_ in PC_AREA_SCOPE])
    clutter = r.uniform(lo, hi, (n_clut, 3))
    xyz = np.concatenate([ground, cars, clutter], 0)
    xyz = np.clip(xyz, lo, hi)
    inten = r.uniform(-0.5, 0.5, (n, 1))
    pc = np.concatenate([xyz, inten], 1).astype(np.float32)
    perm = r.permutation(n)
    pc = pc[perm]
    if dup_frac > 0.0:
        nd = int(dup_frac * n)
        dst = r.choice(n, nd, replace=False)
        src = r.integers(0, n, nd)
        pc[dst] = pc[src]
    if return_boxes:
        return pc, boxes.astype(np.float32)
    return pc


def make_batch(kind: str, batch: int, n: int, config_id: int, dup_frac: float = 0.0) -> np.ndarray:
    """(batch, n, 4) float32; seed = 1000*config_id + scene index."""
    out = np.empty((batch, n, 4), dtype=np.float32)
    for s in range(batch):
        seed = 1000 * config_id + s
        if kind == "uniform":
            out[s] = uniform_cloud(n, seed)
        elif kind == "lidar":
            out[s] = lidar_cloud(n, seed, dup_frac=dup_frac)
        else:
            raise ValueError(kind)
    return out


def proposal_boxes(batch: int, m: int, config_id: int, near_cars: bool = True) -> np.ndarray:
    """(batch, m, 7) boxes for roipool/NMS benches: seeded centres near the synthetic
    car clusters of the matching ``lidar`` scene, sizes CLS_MEAN_SIZE*U(0.9,1.1)."""
    out = np.empty((batch, m, 7), dtype=np.float32)
    for s in range(batch):
        seed = 1000 * config_id + s
        r = _rng(seed * 31 + 5)
        bx = random_boxes3d(m, seed * 17 + 3)
        if near_cars:
            cars = random_boxes3d(15, seed * 7919 + 13)
            pick = r.integers(0, 15, m)
            bx[:, 0] = cars[pick, 0] + r.normal(0, 0.6, m)
            bx[:, 2] = cars[pick, 2] + r.normal(0, 0.6, m)
            bx[:, 6] = cars[pick, 6] + r.normal(0, 0.2, m)
        out[s] = bx
    return out


def distinct_scores(n: int, seed: int) -> np.ndarray:
    """n DISTINCT float32 scores in (0,1) (torch.sort is unstable, iou3d_utils.py:67)."""
    r = _rng(seed)
    s = (r.permutation(n).astype(np.float64) + 0.5) / n
    return s.astype(np.float32)


def boxes3d_to_bev(boxes3d: np.ndarray) -> np.ndarray:
    """numpy twin of kitti_utils.boxes3d_to_bev_torch (lib/utils/kitti_utils.py:134-147)."""
    b = np.asarray(boxes3d, dtype=np.float32)
    out = np.empty((b.shape[0], 5), dtype=np.float32)
    half_l, half_w = b[:, 5] / np.float32(2), b[:, 4] / np.float32(2)
    out[:, 0], out[:, 1] = b[:, 0] - half_l, b[:, 2] - half_w
    out[:, 2], out[:, 3] = b[:, 0] + half_l, b[:, 2] + half_w
    out[:, 4] = b[:, 6]
    return out
