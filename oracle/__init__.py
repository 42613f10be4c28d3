"""CPU oracle for the WS3D hot path -- TEST INFRASTRUCTURE ONLY.

numpy front-end to ``oracle/libws3d_oracle.so`` (built from ``ws3d_oracle.c`` by
``oracle/Makefile``; every C function cites the reference file:line it restates).

Import policy: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package.  Nothing under
``ws3d_amd/`` does.

All arrays are C-contiguous; indices are int32 and data float32, exactly the
dtypes of the reference's CUDA extension.  Outputs are
allocated here with the same initial contents the reference's Python wrappers
give them (zeros for ``ball_query`` idx, 1e10 for the FPS ``temp`` scratch...).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# WS3D_DIST_MODE=1|2 (the variable ws3d_amd reads too): the oracle built under the same alternative convention
DIST_MODE = int(os.environ.get("WS3D_DIST_MODE", "0") or 0)
_SO_NAME = "libws3d_oracle.so" if DIST_MODE == 0 else "libws3d_oracle_dm%d.so" % DIST_MODE
_SO = os.path.join(_HERE, _SO_NAME)

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds).  Returns the path of the .so."""
    src = os.path.join(_HERE, "ws3d_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", _SO_NAME], stdout=subprocess.DEVNULL)
    return _SO


def build_all() -> None:
    """all three conventions (the default library plus _dm1 / _dm2): they travel to the GPU box prebuilt"""
    src = os.path.join(_HERE, "ws3d_oracle.c")
    for name in ("libws3d_oracle.so", "libws3d_oracle_dm1.so", "libws3d_oracle_dm2.so"):
        so = os.path.join(_HERE, name)
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", name], stdout=subprocess.DEVNULL)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.ws3d_oracle_box_overlap_pair.restype = C.c_float
        _lib.ws3d_oracle_iou_bev_pair.restype = C.c_float
        _lib.ws3d_oracle_iou_normal_pair.restype = C.c_float
        _lib.ws3d_oracle_sqdist.restype = C.c_float
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def set_threads(t: int) -> None:
    lib().ws3d_oracle_set_threads(int(t))


def max_threads() -> int:
    return int(lib().ws3d_oracle_max_threads())


def dist_mode() -> int:
    return int(lib().ws3d_oracle_dist_mode())


def opt_n_threads(n: int) -> int:
    return int(lib().ws3d_oracle_opt_n_threads(int(n)))


# --------------------------------------------------------------------------- pointnet2
def furthest_point_sample(xyz: np.ndarray, npoint: int, return_temp: bool = False):
    """xyz (B,N,3) f32 -> idx (B,npoint) i32.  (sampling_gpu.cu:93-209)"""
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    temp = np.full((B, N), 1e10, dtype=np.float32)
    idx = np.zeros((B, npoint), dtype=np.int32)
    rc = lib().ws3d_oracle_fps(px, temp.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p),
                               B, N, int(npoint))
    if rc != 0:
        raise ValueError("ws3d_oracle_fps rc=%d" % rc)
    return (idx, temp) if return_temp else idx


def gather_operation(features: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """features (B,C,N), idx (B,M) -> (B,C,M).  (sampling_gpu.cu:8-24)"""
    features, pf = _f(features)
    idx, pi = _i(idx)
    B, Cc, N = features.shape
    M = idx.shape[1]
    out = np.empty((B, Cc, M), dtype=np.float32)
    lib().ws3d_oracle_gather_points(pf, pi, out.ctypes.data_as(_f32p), B, Cc, N, M)
    return out


def gather_operation_grad(grad_out: np.ndarray, idx: np.ndarray, N: int) -> np.ndarray:
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, Cc, M = grad_out.shape
    out = np.zeros((B, Cc, N), dtype=np.float32)
    lib().ws3d_oracle_gather_points_grad(pg, pi, out.ctypes.data_as(_f32p), B, Cc, int(N), M)
    return out


def ball_query(radius: float, nsample: int, xyz: np.ndarray, new_xyz: np.ndarray) -> np.ndarray:
    """xyz (B,N,3), new_xyz (B,M,3) -> idx (B,M,nsample) i32.  (ball_query_gpu.cu:9-45)"""
    xyz, px = _f(xyz)
    new_xyz, pn = _f(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), dtype=np.int32)
    lib().ws3d_oracle_ball_query(pn, px, idx.ctypes.data_as(_i32p), B, N, M,
                                 C.c_float(radius), int(nsample))
    return idx


def grouping_operation(features: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """features (B,C,N), idx (B,M,ns) -> (B,C,M,ns).  (group_points_gpu.cu:47-66)"""
    features, pf = _f(features)
    idx, pi = _i(idx)
    B, Cc, N = features.shape
    _, M, ns = idx.shape
    out = np.empty((B, Cc, M, ns), dtype=np.float32)
    lib().ws3d_oracle_group_points(pf, pi, out.ctypes.data_as(_f32p), B, Cc, N, M, ns)
    return out


def grouping_operation_grad(grad_out: np.ndarray, idx: np.ndarray, N: int) -> np.ndarray:
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, Cc, M, ns = grad_out.shape
    out = np.zeros((B, Cc, N), dtype=np.float32)
    lib().ws3d_oracle_group_points_grad(pg, pi, out.ctypes.data_as(_f32p), B, Cc, int(N), M, ns)
    return out


def three_nn_dist2(unknown: np.ndarray, known: np.ndarray):
    """unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3) f32 (SQUARED), idx (B,n,3) i32.
    (interpolate_gpu.cu:9-52)"""
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.empty((B, n, 3), dtype=np.float32)
    idx = np.empty((B, n, 3), dtype=np.int32)
    lib().ws3d_oracle_three_nn(pu, pk, dist2.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p),
                               B, n, m)
    return dist2, idx


def three_nn(unknown: np.ndarray, known: np.ndarray):
    """As the Python wrapper returns it: sqrt(dist2), idx (pointnet2_utils.py:98)."""
    d2, idx = three_nn_dist2(unknown, known)
    return np.sqrt(d2), idx


def three_interpolate(features: np.ndarray, idx: np.ndarray, weight: np.ndarray) -> np.ndarray:
    """features (B,C,m), idx/weight (B,n,3) -> (B,C,n).  (interpolate_gpu.cu:77-97)"""
    features, pf = _f(features)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, Cc, m = features.shape
    n = idx.shape[1]
    out = np.empty((B, Cc, n), dtype=np.float32)
    lib().ws3d_oracle_three_interpolate(pf, pi, pw, out.ctypes.data_as(_f32p), B, Cc, m, n)
    return out


def three_interpolate_grad(grad_out: np.ndarray, idx: np.ndarray, weight: np.ndarray, m: int):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, Cc, n = grad_out.shape
    out = np.zeros((B, Cc, m), dtype=np.float32)
    lib().ws3d_oracle_three_interpolate_grad(pg, pi, pw, out.ctypes.data_as(_f32p), B, Cc, n, int(m))
    return out


# --------------------------------------------------------------------------- roipool3d
def pts_in_boxes3d(pts: np.ndarray, boxes3d: np.ndarray) -> np.ndarray:
    """pts (N,3), boxes3d (M,7) -> flag (M,N) int64.  (roipool3d.cpp:97-124)"""
    pts, pp = _f(pts)
    boxes3d, pb = _f(boxes3d)
    M, N = boxes3d.shape[0], pts.shape[0]
    flag = np.zeros((M, N), dtype=np.int64)
    lib().ws3d_oracle_pts_in_boxes3d(flag.ctypes.data_as(_i64p), pp, pb, M, N)
    return flag


def roipool3d(xyz, boxes3d, pts_feature, sampled_pt_num=512, return_idx=False):
    """xyz (B,N,3), boxes3d (B,M,7) [already enlarged], pts_feature (B,N,C) ->
    pooled (B,M,S,3+C) f32, empty (B,M) i32.  (roipool3d_kernel.cu:97-237)"""
    xyz, px = _f(xyz)
    boxes3d, pb = _f(boxes3d)
    pts_feature, pf = _f(pts_feature)
    B, N, _ = xyz.shape
    M = boxes3d.shape[1]
    Cc = pts_feature.shape[2]
    S = int(sampled_pt_num)
    pooled = np.zeros((B, M, S, 3 + Cc), dtype=np.float32)
    empty = np.zeros((B, M), dtype=np.int32)
    sel = np.zeros((B, M, S), dtype=np.int32)
    lib().ws3d_oracle_roipool3d(px, pb, pf, pooled.ctypes.data_as(_f32p),
                                empty.ctypes.data_as(_i32p), sel.ctypes.data_as(_i32p),
                                B, N, M, Cc, S)
    return (pooled, empty, sel) if return_idx else (pooled, empty)


def roipool3d_cpu(pts, boxes3d, pts_feature, sampled_pt_num=512):
    """Single scene, split outputs (roipool3d.cpp:127-195)."""
    pts, pp = _f(pts)
    boxes3d, pb = _f(boxes3d)
    pts_feature, pf = _f(pts_feature)
    M, N, Cc, S = boxes3d.shape[0], pts.shape[0], pts_feature.shape[1], int(sampled_pt_num)
    pooled_pts = np.zeros((M, S, 3), dtype=np.float32)
    pooled_feat = np.zeros((M, S, Cc), dtype=np.float32)
    empty = np.zeros((M,), dtype=np.int64)
    lib().ws3d_oracle_roipool3d_cpu(pp, pb, pf, pooled_pts.ctypes.data_as(_f32p),
                                    pooled_feat.ctypes.data_as(_f32p),
                                    empty.ctypes.data_as(_i64p), M, N, Cc, S)
    return pooled_pts, pooled_feat, empty


# --------------------------------------------------------------------------- iou3d
def boxes_overlap_bev(boxes_a: np.ndarray, boxes_b: np.ndarray) -> np.ndarray:
    """(Na,5),(Nb,5) [x1,y1,x2,y2,ry] -> overlap area (Na,Nb).  (iou3d_kernel.cu:108-234)"""
    boxes_a, pa = _f(boxes_a)
    boxes_b, pb = _f(boxes_b)
    out = np.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=np.float32)
    lib().ws3d_oracle_boxes_overlap_bev(pa, pb, out.ctypes.data_as(_f32p),
                                        boxes_a.shape[0], boxes_b.shape[0])
    return out


def boxes_iou_bev(boxes_a: np.ndarray, boxes_b: np.ndarray) -> np.ndarray:
    boxes_a, pa = _f(boxes_a)
    boxes_b, pb = _f(boxes_b)
    out = np.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=np.float32)
    lib().ws3d_oracle_boxes_iou_bev(pa, pb, out.ctypes.data_as(_f32p),
                                    boxes_a.shape[0], boxes_b.shape[0])
    return out


def nms_mask(boxes: np.ndarray, thresh: float, normal: bool = False) -> np.ndarray:
    """boxes (n,5) score-sorted -> mask (n, ceil(n/64)) u64.  (iou3d_kernel.cu:250-348)"""
    boxes, pb = _f(boxes)
    n = boxes.shape[0]
    cb = (n + 63) // 64
    mask = np.zeros((n, cb), dtype=np.uint64)
    lib().ws3d_oracle_nms_mask(pb, mask.ctypes.data_as(_u64p), n, C.c_float(thresh), int(normal))
    return mask


def nms_sweep(mask: np.ndarray) -> np.ndarray:
    """Greedy host sweep (iou3d.cpp:100-116) -> kept row indices (int64)."""
    mask = np.ascontiguousarray(mask, dtype=np.uint64)
    n = mask.shape[0]
    keep = np.zeros((max(n, 1),), dtype=np.int64)
    cnt = lib().ws3d_oracle_nms_sweep(mask.ctypes.data_as(_u64p), keep.ctypes.data_as(_i64p), n)
    return keep[:cnt].copy()


def nms_sorted(boxes: np.ndarray, thresh: float, normal: bool = False) -> np.ndarray:
    """ext-level nms_gpu / nms_normal_gpu (iou3d.cpp:73-170): boxes already sorted."""
    boxes, pb = _f(boxes)
    n = boxes.shape[0]
    keep = np.zeros((max(n, 1),), dtype=np.int64)
    cnt = lib().ws3d_oracle_nms(pb, keep.ctypes.data_as(_i64p), n, C.c_float(thresh), int(normal))
    return keep[:cnt].copy()


def nms_sorted_lazy(boxes: np.ndarray, thresh: float, normal: bool = False, max_keep: int = 0) -> np.ndarray:
    """nms_sorted's keep list (its first max_keep entries when max_keep > 0) from IoUs against the kept boxes only -- no (n, n/64)
    mask; the CPU-side baseline a caller would actually run (ws3d_oracle_nms_lazy)."""
    boxes, pb = _f(boxes)
    n = boxes.shape[0]
    keep = np.zeros((max(n, 1),), dtype=np.int64)
    cnt = lib().ws3d_oracle_nms_lazy(pb, keep.ctypes.data_as(_i64p), n, C.c_float(thresh), int(normal), int(max_keep))
    return keep[:cnt].copy()


def nms(boxes: np.ndarray, scores: np.ndarray, thresh: float, normal: bool = False) -> np.ndarray:
    """Python-level nms_gpu (iou3d_utils.py:59-90) with a STABLE descending sort."""
    order = np.argsort(-np.asarray(scores, dtype=np.float32), kind="stable")
    keep = nms_sorted(np.asarray(boxes, dtype=np.float32)[order], thresh, normal)
    return order[keep]


def radius_nms_sorted(centers_xz: np.ndarray, radius: float) -> np.ndarray:
    """centres (n,2) sorted by descending score -> kept indices (generate_box_dataset.py:127-140)"""
    c, pc = _f(centers_xz)
    n = c.shape[0]
    keep = np.zeros((max(n, 1),), dtype=np.int64)
    cnt = lib().ws3d_oracle_radius_nms(pc, keep.ctypes.data_as(_i64p), n, C.c_float(radius))
    return keep[:cnt].copy()


def box_overlap_pair(a, b) -> float:
    a, pa = _f(a)
    b, pb = _f(b)
    return float(lib().ws3d_oracle_box_overlap_pair(pa, pb))


def pt_in_box3d(p, box) -> int:
    p, pp = _f(p)
    box, pb = _f(box)
    return int(lib().ws3d_oracle_pt_in_box3d(pp, pb))


def sqdist(a, b) -> float:
    a, pa = _f(a)
    b, pb = _f(b)
    return float(lib().ws3d_oracle_sqdist(pa, pb))
