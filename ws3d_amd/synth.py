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


# --------------------------------------------------------------------------- hdl64: a ray-cast Velodyne HDL-64E scan
# KITTI's sensor: 64 beams, +2 deg ... -8.33 deg at 1/3 deg (upper block) and -8.83 ... -24.33 deg at 1/2 deg (lower block),
# ~0.18 deg of azimuth per firing at 10 Hz, 1.73 m above the road; the camera of the rect frame sits 0.08 m below and 0.27 m
# in front of it (Tr_velo_to_cam of KITTI_CALIB_TEXT).  What the reference feeds the network is such a scan cropped to the
# camera image and PC_AREA_SCOPE and then sub-sampled to 16384 points with every point beyond 40 m kept
# (lib/datasets/kitti_rcnn_dataset.py:399-447, restated in kitti_io.sample_point_choice): near-range ground rings put dozens of
# points inside the r = 0.5 m ball of an SA1 centre -- the regime the `lidar` generator above (1-2 distinct neighbours per
# list) never reaches.
HDL64_ELEVATION_DEG = np.concatenate([2.0 - np.arange(32) / 3.0, -8.83 - 0.5 * np.arange(32)])
HDL64_AZIMUTH_STEP_DEG = 0.18
HDL64_SENSOR = (0.0, -0.08, -0.27)     # beam origin in the rect camera frame (x right, y down, z forward); road at y = 1.65
HDL64_ROAD_Y = 1.65
KITTI_P2 = (721.5377, 609.5593, 172.854, 1242, 375)   # fx = fy, cx, cy, image width, height


def _ray_boxes(o, d, boxes, az_lo=None, az_step=None):
    """nearest entry distance of the rays o + t d, d (64, A, 3) on a regular azimuth grid (column a covers azimuth
    az_lo + a az_step up to one step of per-beam offset), into oriented boxes (K,7) [x, y_bottom, z, h, w, l, ry] (the reference's
    frame: l along the box's x, w along its z, rotation about y) -> (64, A) t, inf where nothing is hit.  Each box is tested
    against the azimuth columns its bounding circle can cover only."""
    nb, na = d.shape[0], d.shape[1]
    best = np.full((nb, na), np.inf)
    for bx in boxes:
        c, s = np.cos(bx[6]), np.sin(bx[6])
        ctr = np.array([bx[0], bx[1] - bx[3] / 2, bx[2]])
        half = np.array([bx[5] / 2, bx[3] / 2, bx[4] / 2])                         # along the box's x (l), y (h), z (w)
        rel = o - ctr
        rho = np.hypot(rel[0], rel[2])                                              # ground distance sensor -> box centre
        rad = np.hypot(half[0], half[2])
        a0, a1 = 0, na
        if rho > rad * 1.05:
            mid, wid = np.arctan2(-rel[0], -rel[2]), np.arcsin(rad / rho) + 2.5 * az_step
            a0 = max(0, int(np.floor((mid - wid - az_lo) / az_step)))
            a1 = min(na, int(np.ceil((mid + wid - az_lo) / az_step)) + 1)
            if a1 <= a0:
                continue
        dd = d[:, a0:a1]
        ob = np.array([rel[0] * c - rel[2] * s, rel[1], rel[0] * s + rel[2] * c])   # origin in the box frame
        db = np.stack([dd[..., 0] * c - dd[..., 2] * s, dd[..., 1], dd[..., 0] * s + dd[..., 2] * c], -1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / db
            t1, t2 = (-half - ob) * inv, (half - ob) * inv
        tn, tf = np.minimum(t1, t2).max(-1), np.maximum(t1, t2).min(-1)
        hit = (tn <= tf) & (tn > 0.5)
        best[:, a0:a1] = np.minimum(best[:, a0:a1], np.where(hit, tn, np.inf))
    return best


def hdl64_scene_boxes(seed: int, n_cars: int = 15) -> np.ndarray:
    """the objects of one ray-cast scene: `n_cars` cars standing on the road (the proposals' ground truth, same seeding as
    lidar_cloud), two building fronts along the road and eight poles / trunks -> (n_cars + 10, 7)"""
    r = _rng(seed * 6151 + 7)
    cars = random_boxes3d(n_cars, seed * 7919 + 13, jitter=0.1).astype(np.float64)
    cars[:, 1] = HDL64_ROAD_Y
    walls = np.array([[side * r.uniform(9.0, 22.0), HDL64_ROAD_Y, r.uniform(25.0, 45.0), r.uniform(4.0, 9.0), r.uniform(50.0, 70.0), 1.0,
                       r.uniform(-0.06, 0.06)] for side in (-1.0, 1.0)])          # l = 1 m thick (x), w = 50-70 m long (z)
    poles = np.stack([r.uniform(-20.0, 20.0, 8), np.full(8, HDL64_ROAD_Y), r.uniform(6.0, 60.0, 8), r.uniform(2.5, 6.0, 8),
                      r.uniform(0.2, 0.5, 8), r.uniform(0.2, 0.5, 8), r.uniform(-np.pi, np.pi, 8)], 1)
    return np.concatenate([cars, walls, poles], 0)


def hdl64_scan(seed: int, n_cars: int = 15, return_boxes: bool = False, sweep: int = 0):
    """one ray-cast sweep of the forward quadrant, cropped like the reference's ``get_valid_flag`` (inside the 1242 x 375 image,
    in front of the camera, inside PC_AREA_SCOPE) -> (n, 4) float64 [x, y, z (rect camera frame), intensity in [0, 1)], n ~ 19-23 k.
    `sweep` > 0: another sweep over the SAME scene (same objects) with its own firing offsets and range noise"""
    r = _rng(seed + 104729 * sweep)
    elev = np.deg2rad(HDL64_ELEVATION_DEG)
    step = np.deg2rad(HDL64_AZIMUTH_STEP_DEG)
    az0 = np.arange(-43.0, 43.0, HDL64_AZIMUTH_STEP_DEG) * np.pi / 180.0      # the image frustum is +-40.7 deg wide
    # every laser fires at its own fixed azimuth offset inside a step (the sensor's staggered layout), plus firing jitter
    az = az0[None, :] + r.uniform(0.0, step, (64, 1)) + r.normal(0.0, 0.02 * step, (64, az0.size))
    el = elev[:, None] + r.normal(0.0, np.deg2rad(0.01), (64, az0.size))
    d = np.stack([np.cos(el) * np.sin(az), -np.sin(el), np.cos(el) * np.cos(az)], 2)              # (64, A, 3)
    o = np.asarray(HDL64_SENSOR, dtype=np.float64)
    boxes = hdl64_scene_boxes(seed, n_cars)
    with np.errstate(divide="ignore"):
        t_road = np.where(d[..., 1] > 1e-6, (HDL64_ROAD_Y - o[1]) / d[..., 1], np.inf)
    t = np.minimum(t_road, _ray_boxes(o, d, boxes, az0[0], step))
    ok = np.isfinite(t) & (t < 120.0)
    t, d = t[ok], d[ok]
    t = t + r.normal(0.0, 0.015, t.shape)                                       # range noise of the sensor
    pts = o[None, :] + t[:, None] * d
    fx, cx, cy, w, h = KITTI_P2
    z = pts[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u, v = fx * pts[:, 0] / z + cx, fx * pts[:, 1] / z + cy
    keep = (z >= 0) & (u >= 0) & (u < w) & (v >= 0) & (v < h)
    (x0, x1), (y0, y1), (z0, z1) = PC_AREA_SCOPE
    keep &= (pts[:, 0] >= x0) & (pts[:, 0] <= x1) & (pts[:, 1] >= y0) & (pts[:, 1] <= y1) & (z >= z0) & (z <= z1)
    pts = pts[keep]
    scan = np.concatenate([pts, r.uniform(0.0, 1.0, (pts.shape[0], 1))], 1)
    if return_boxes:
        return scan, boxes[:n_cars].astype(np.float32)
    return scan


def hdl64_cloud(n: int, seed: int, n_cars: int = 15, return_boxes: bool = False):
    """(n, 4) float32 network input from a ray-cast HDL-64E scan: the reference's sub-sampling to n points -- all points
    beyond 40 m kept, near ones drawn without replacement, a smaller scan tiled (kitti_rcnn_dataset.py:424-447 via
    kitti_io.sample_point_choice) on a seeded ``RandomState`` -- and intensity - 0.5.  The scan itself holds ~20 k points; for
    n above that (c5's 65536) four sweeps with independent noise are merged first (a denser sensor, not a tiled copy)."""
    from .kitti_io import sample_point_choice
    scan, boxes = hdl64_scan(seed, n_cars, return_boxes=True)
    k = 1
    while scan.shape[0] < n and k < 8:
        scan = np.concatenate([scan, hdl64_scan(seed, n_cars, sweep=k)], 0)
        k += 1
    rs = np.random.RandomState(seed % (2 ** 31))
    choice = sample_point_choice(scan[:, 2], n, rs)
    pc = np.concatenate([scan[choice, :3], scan[choice, 3:] - 0.5], 1).astype(np.float32)
    if return_boxes:
        return pc, boxes
    return pc


GENERATORS = ("hdl64", "lidar", "uniform")


def cloud(kind: str, n: int, seed: int, dup_frac: float = 0.0) -> np.ndarray:
    """(n, 4) float32 scene of one of the seeded generators: 'hdl64' (ray-cast HDL-64E scan, the reference's sub-sampling: KITTI's
    point density), 'lidar' (SURVEY 8d's sparse statistical model), 'uniform' (iid in the scope box)"""
    if kind == "uniform":
        return uniform_cloud(n, seed)
    if kind == "lidar":
        return lidar_cloud(n, seed, dup_frac=dup_frac)
    if kind == "hdl64":
        return hdl64_cloud(n, seed)
    raise ValueError(kind)


def make_batch(kind: str, batch: int, n: int, config_id: int, dup_frac: float = 0.0) -> np.ndarray:
    """(batch, n, 4) float32; seed = 1000*config_id + scene index."""
    out = np.empty((batch, n, 4), dtype=np.float32)
    for s in range(batch):
        out[s] = cloud(kind, n, 1000 * config_id + s, dup_frac)
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


def proposal_boxes_on_scene(pc: np.ndarray, cars: np.ndarray, m: int, seed: int, min_points: int = 16) -> np.ndarray:
    """(m, 7) proposals for roipool/NMS benches on a scene whose objects are only partly visible (hdl64: frustum crop + occlusion
    leave many of the 15 cars without a single return): centres near the cars that DO hold points -- at least `min_points` of
    `pc` (n, >=3) within 2 m of the car centre in the ground plane -- sizes CLS_MEAN_SIZE * U(0.9, 1.1) like `proposal_boxes`.
    SURVEY 8d: 'seeded random centres near the car clusters'; here after the crop, so that (almost) every RoI is non-empty."""
    r = _rng(seed * 31 + 11)
    bx = random_boxes3d(m, seed * 17 + 3)
    d2 = (pc[None, :, 0] - cars[:, None, 0]) ** 2 + (pc[None, :, 2] - cars[:, None, 2]) ** 2
    seen = np.nonzero((d2 < 4.0).sum(1) >= min_points)[0]
    if seen.size == 0:                       # no visible object: centre the proposals on points of the scene
        pick = r.integers(0, pc.shape[0], m)
        bx[:, 0], bx[:, 2] = pc[pick, 0], pc[pick, 2]
        return bx
    pick = seen[r.integers(0, seen.size, m)]
    bx[:, 0] = cars[pick, 0] + r.normal(0, 0.5, m)
    bx[:, 2] = cars[pick, 2] + r.normal(0, 0.5, m)
    bx[:, 6] = cars[pick, 6] + r.normal(0, 0.2, m)
    return bx.astype(np.float32)


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


# --------------------------------------------------------------------------- synthetic KITTI directory
KITTI_CALIB_TEXT = """P0: 7.215377e+02 0.000000e+00 6.095593e+02 0.000000e+00 0.000000e+00 7.215377e+02 1.728540e+02 0.000000e+00 0.000000e+00 0.000000e+00 1.000000e+00 0.000000e+00
P1: 7.215377e+02 0.000000e+00 6.095593e+02 -3.875744e+02 0.000000e+00 7.215377e+02 1.728540e+02 0.000000e+00 0.000000e+00 0.000000e+00 1.000000e+00 0.000000e+00
P2: 7.215377e+02 0.000000e+00 6.095593e+02 4.485728e+01 0.000000e+00 7.215377e+02 1.728540e+02 2.163791e-01 0.000000e+00 0.000000e+00 1.000000e+00 2.745884e-03
P3: 7.215377e+02 0.000000e+00 6.095593e+02 -3.395242e+02 0.000000e+00 7.215377e+02 1.728540e+02 2.199936e+00 0.000000e+00 0.000000e+00 1.000000e+00 2.729905e-03
R0_rect: 9.999239e-01 9.837760e-03 -7.445048e-03 -9.869795e-03 9.999421e-01 -4.278459e-03 7.402527e-03 4.351614e-03 9.999631e-01
Tr_velo_to_cam: 7.533745e-03 -9.999714e-01 -6.166020e-04 -4.069766e-03 1.480249e-02 7.280733e-04 -9.998902e-01 -7.631618e-02 9.998621e-01 7.523790e-03 1.480755e-02 -2.717806e-01
Tr_imu_to_velo: 9.999976e-01 7.553071e-04 -2.035826e-03 -8.086759e-01 -7.854027e-04 9.998898e-01 -1.482298e-02 3.195559e-01 2.024406e-03 1.482454e-02 9.998881e-01 -7.997231e-01
"""

KITTI_LABEL_TEXT = """Car 0.00 0 -1.58 587.01 173.33 614.12 200.12 1.65 1.67 3.64 -0.65 1.71 46.70 -1.59
Car 0.00 1 1.85 387.63 181.54 423.81 203.12 1.67 1.87 3.69 -16.53 2.39 58.49 1.57
Pedestrian 0.00 0 0.21 423.17 173.67 433.17 224.03 1.60 0.38 0.30 -5.87 1.63 23.11 -0.03
DontCare -1 -1 -10 503.89 169.71 590.61 190.13 -1 -1 -1 -1000 -1000 -1000 -10
"""


def _blank_png(width: int, height: int) -> bytes:
    """a valid all-black 8-bit RGB PNG, written with zlib only"""
    import struct
    import zlib

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    raw = (b"\x00" + b"\x00" * (3 * width)) * height
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 2, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def velodyne_scan(n: int, seed: int) -> np.ndarray:
    """(n,4) float32 velodyne-frame points: x forward, y left, z up; a forward wedge wider than the
    camera frustum with a ground sheet and some elevated clutter, reflectance in [0,1)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = 1.0 + 77.0 * rng.uniform(0, 1, n) ** 2.5        # lidar returns thin out with range
    y = rng.uniform(-0.75, 0.75, n) * (x + 4.0)
    ground = rng.uniform(0, 1, n) < 0.7
    z = np.where(ground, -1.73 + rng.normal(0, 0.03, n), rng.uniform(-1.7, 1.2, n))
    refl = rng.uniform(0, 1, n)
    return np.stack((x, y, z, refl), axis=1).astype(np.float32)


def write_kitti_tree(root: str, scenes) -> None:
    """scenes: iterable of (sample_id, n_points, seed).  Lays out root/ImageSets/val.txt and
    root/training/{velodyne,calib,image_2,label_2}/%06d.* like the KITTI object benchmark."""
    import os
    for sub in ("velodyne", "calib", "image_2", "label_2"):
        os.makedirs(os.path.join(root, "training", sub), exist_ok=True)
    os.makedirs(os.path.join(root, "ImageSets"), exist_ok=True)
    ids = []
    for sample_id, n, seed in scenes:
        ids.append("%06d" % sample_id)
        velodyne_scan(n, seed).tofile(os.path.join(root, "training", "velodyne", "%06d.bin" % sample_id))
        with open(os.path.join(root, "training", "calib", "%06d.txt" % sample_id), "w") as f:
            f.write(KITTI_CALIB_TEXT)
        with open(os.path.join(root, "training", "image_2", "%06d.png" % sample_id), "wb") as f:
            f.write(_blank_png(1242, 375))
        with open(os.path.join(root, "training", "label_2", "%06d.txt" % sample_id), "w") as f:
            f.write(KITTI_LABEL_TEXT)
    with open(os.path.join(root, "ImageSets", "val.txt"), "w") as f:
        f.write("\n".join(ids) + "\n")


def roi_clouds(batch: int, n: int, config_id: int) -> np.ndarray:
    """(batch, n, 3) float32 Stage-2 inputs: the points pooled for one proposal, in the box's
    canonical frame (car-sized box enlarged by 1 m: |x| < 2.9, y in (-2.5, 0.5), |z| < 1.8), denser
    on the object's surface than in the margin"""
    rng = np.random.Generator(np.random.PCG64(1000 * config_id + 17))
    half = np.array([2.94, 1.5, 1.81], dtype=np.float64)
    centre = np.array([0.0, -1.0, 0.0])
    u = rng.uniform(-1, 1, (batch, n, 3))
    shell = rng.uniform(0, 1, (batch, n, 1)) < 0.6
    axis = rng.integers(0, 3, (batch, n))
    snap = np.where(np.arange(3)[None, None, :] == axis[:, :, None], np.sign(u) * 0.66, u * 0.66)
    pts = np.where(shell, snap, u) * half + centre
    return pts.astype(np.float32)
