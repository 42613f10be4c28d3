"""Drop-in replacements for the reference's three pybind11 extension modules.

The reference's Python wrappers do ``import pointnet2_cuda`` / ``import iou3d_cuda`` /
``import roipool3d_cuda`` and call functions that take PRE-ALLOCATED torch tensors
(pointnet2_api.cpp:10-24, iou3d.cpp:174-179, roipool3d.cpp:198-203).  The three
namespaces below export exactly those function names with the same positional
arguments, forwarding ``tensor.data_ptr()`` + the current torch stream to the C ABI of
``libws3d_hip.so`` (include/ws3d_ops.h).  ``install()`` registers them in
``sys.modules`` so that the reference's own ``pointnet2_utils.py`` / ``iou3d_utils.py`` /
``roipool3d_utils.py`` run unmodified on an MI355X.

Differences from the reference, all deliberate:
  * errors raise ``Ws3dError`` instead of ``exit(-1)`` (sampling_gpu.cu:39-43);
  * every launch goes to torch's CURRENT stream (the reference's iou3d/roipool3d use
    the null stream and cudaMalloc/cudaFree per call);
  * ``nms_gpu`` sweeps on the device; the only host sync is the copy of the result
    into the caller's CPU ``keep`` tensor, which the reference signature demands.
"""
from __future__ import annotations

import sys
import types

import os

import torch

from . import _lib
from ._lib import Ws3dError, check


def _dev(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise Ws3dError("ws3d_amd ops need HIP ('cuda') tensors; there is no CPU fallback")
        if not t.is_contiguous():
            raise Ws3dError("ws3d_amd ops need contiguous tensors (the reference asserts the same)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise Ws3dError("tensors live on different devices")
    return dev


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    """raw handle of the current HIP stream of the current device (the C call behind torch.cuda.current_stream().cuda_stream: the
    wrapper objects cost ~9 us per launch on the issuing thread, and an eager Stage-1 step makes ~100 launches)"""
    return _raw_stream(_cur_device())


try:
    _raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:                                      # (another torch build: the public, slower way)
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream

    def _cur_device():
        return torch.cuda.current_device()


class _SameDevice:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_SAME_DEVICE = _SameDevice()


def _on(dev):
    """context that makes `dev` the current device for the launch -- nothing to do (and nothing paid) when it already is"""
    return _SAME_DEVICE if dev.index == _cur_device() else torch.cuda.device(dev)


def _f32(t, name):
    if t.dtype != torch.float32:
        raise Ws3dError(f"{name} must be float32, got {t.dtype}")


def _i32(t, name):
    if t.dtype != torch.int32:
        raise Ws3dError(f"{name} must be int32, got {t.dtype}")


# ------------------------------------------------------------------ pointnet2_cuda
def furthest_point_sampling_wrapper(b, n, m, points_tensor, temp_tensor, idx_tensor):
    """sampling.cpp:36-46"""
    dev = _dev(points_tensor, temp_tensor, idx_tensor)
    _f32(points_tensor, "xyz"); _i32(idx_tensor, "idx")
    with _on(dev):
        check(_lib.load().ws3d_furthest_point_sampling(b, n, m, _p(points_tensor), _p(temp_tensor),
                                                        _p(idx_tensor), _stream()), "furthest_point_sampling")
    return 1


def furthest_point_sampling_gather(b, n, m, xyz, temp, idx, new_xyz):
    """fused a1+a2 (ws3d extension; no reference counterpart)"""
    dev = _dev(xyz, temp, idx, new_xyz)
    _f32(xyz, "xyz"); _i32(idx, "idx")
    with _on(dev):
        check(_lib.load().ws3d_furthest_point_sampling_gather(b, n, m, _p(xyz), _p(temp), _p(idx),
                                                               _p(new_xyz), _stream()), "fps_gather")
    return 1


def furthest_point_sampling_nested(b, n, m, xyz, idx, new_xyz):
    """FPS + gather of a cloud that is already in sampling order (the centres of the previous set-abstraction level):
    same results as furthest_point_sampling_gather, found by a parallel check instead of m dependent steps.  ws3d extension."""
    dev = _dev(xyz, idx, new_xyz)
    _f32(xyz, "xyz"); _i32(idx, "idx"); _f32(new_xyz, "new_xyz")
    with _on(dev):
        check(_lib.load().ws3d_furthest_point_sampling_nested(b, n, m, _p(xyz), _p(idx), _p(new_xyz), _stream()), "fps_nested")
    return 1


def furthest_point_sampling_nested_chain(xyz, npoints):
    """xyz (B, N, 3) in sampling order, npoints = the non-increasing sample counts of the levels below it -> [(idx, new_xyz), ...], one
    pair per level (level l samples level l-1's new_xyz): what len(npoints) calls of furthest_point_sampling_nested return, in four
    launches (ws3d_furthest_point_sampling_nested_chain).  ws3d extension."""
    import ctypes as _ct
    dev = _dev(xyz)
    _f32(xyz, "xyz")
    B, N = xyz.size(0), xyz.size(1)
    L = len(npoints)
    idx = [torch.empty((B, int(m)), dtype=torch.int32, device=dev) for m in npoints]
    new = [torch.empty((B, int(m), 3), dtype=torch.float32, device=dev) for m in npoints]
    ms = (_ct.c_int * L)(*[int(m) for m in npoints])
    ip = (_ct.c_void_p * L)(*[t.data_ptr() for t in idx])
    np_ = (_ct.c_void_p * L)(*[t.data_ptr() for t in new])
    with _on(dev):
        check(_lib.load().ws3d_furthest_point_sampling_nested_chain(B, N, L, _ct.cast(ms, _ct.c_void_p), _p(xyz), _ct.cast(ip, _ct.c_void_p),
                                                                     _ct.cast(np_, _ct.c_void_p), _stream()), "fps_nested_chain")
    return list(zip(idx, new))


def gather_points_wrapper(b, c, n, npoints, points_tensor, idx_tensor, out_tensor):
    """sampling.cpp:11-20"""
    dev = _dev(points_tensor, idx_tensor, out_tensor)
    _f32(points_tensor, "points"); _i32(idx_tensor, "idx")
    with _on(dev):
        check(_lib.load().ws3d_gather_points(b, c, n, npoints, _p(points_tensor), _p(idx_tensor),
                                              _p(out_tensor), _stream()), "gather_points")
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out_tensor, idx_tensor, grad_points_tensor):
    """sampling.cpp:23-33"""
    dev = _dev(grad_out_tensor, idx_tensor, grad_points_tensor)
    _f32(grad_out_tensor, "grad_out"); _i32(idx_tensor, "idx")
    with _on(dev):
        check(_lib.load().ws3d_gather_points_grad(b, c, n, npoints, _p(grad_out_tensor), _p(idx_tensor),
                                                   _p(grad_points_tensor), _stream()), "gather_points_grad")
    return 1


SORTED_MIN_N = 2048  # below this the LDS-tiled brute-force scan is already cheap


BQ_FINE_GRID = True   # sort_points_x builds the fine (x, z) grid (ws3d_sort_points_grid); False: x slabs (A/B runs)
TOPK_SEGMENTS = True   # topk_sorted: a scene's sort spread over its CUs (ws3d_topk_sorted_ws); False: one workgroup per scene (tests flip it)


def sort_points_x(xyz, min_n=None, grid=None):
    """(B,N,3) -> opaque uint8 buffer for the ball-query entries (per scene: N float4 {x,y,z,bits(index)} binned into a
    fine (x, z) grid -- or, grid=False, into x slabs -- a header and a cell-start table), or None when the binned path
    does not apply (N < min_n, default 2048 -- where the binned ball query starts to pay -- or N > 16384).
    ws3d extension."""
    dev = _dev(xyz)
    _f32(xyz, "xyz")
    b, n = xyz.size(0), xyz.size(1)
    lib = _lib.load()
    nbytes = lib.ws3d_sorted_points_bytes(b, n)
    if n < (SORTED_MIN_N if min_n is None else min_n) or nbytes == 0:
        return None
    out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with _on(dev):
        if BQ_FINE_GRID if grid is None else grid:
            check(lib.ws3d_sort_points_grid(b, n, _p(xyz), _p(out), _stream()), "sort_points_grid")
        else:
            check(lib.ws3d_sort_points_x(b, n, _p(xyz), _p(out), _stream()), "sort_points_x")
    return out


def sort_points_xz(xyz, min_n=256):
    """(B,N,3) -> the same kind of buffer as sort_points_x, binned into an (x, z) grid: for three_nn's
    ``sorted_known`` only (NOT for the ball query).  None when N < min_n or N > 16384.  ws3d extension."""
    dev = _dev(xyz)
    _f32(xyz, "xyz")
    b, n = xyz.size(0), xyz.size(1)
    lib = _lib.load()
    nbytes = lib.ws3d_sorted_points_bytes(b, n)
    if n < min_n or nbytes == 0:
        return None
    out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with _on(dev):
        check(lib.ws3d_sort_points_xz(b, n, _p(xyz), _p(out), _stream()), "sort_points_xz")
    return out


SORT_JOBS_MAX = 8     # = BIN_MAX_JOBS of csrc/bin_kernels.h


def sort_points_jobs(jobs):
    """jobs: [(xyz (B,N_i,3), kind)] with kind "grid" (sort_points_x's fine-grid buffer, the ball query's) or "xz" (sort_points_xz's,
    three_nn's) -> the list of buffers, all binned by ONE launch (ws3d_sort_points_jobs); same bytes as one call per job.  None for a
    job whose cloud the binned searches do not take (N > 16384).  ws3d extension."""
    if not jobs:
        return []
    import ctypes as C
    dev = _dev(*[x for x, _ in jobs])
    lib = _lib.load()
    b = jobs[0][0].size(0)
    outs, n_arr, k_arr, x_arr, o_arr = [], [], [], [], []
    for xyz, kind in jobs:
        _f32(xyz, "xyz")
        if xyz.size(0) != b or kind not in ("grid", "xz"):
            raise ValueError("sort_points_jobs: every job takes the same batch and kind 'grid' or 'xz'")
        nbytes = lib.ws3d_sorted_points_bytes(b, xyz.size(1))
        out = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
        outs.append(out)
        if out is not None:
            n_arr.append(xyz.size(1)); k_arr.append(0 if kind == "grid" else 1); x_arr.append(xyz.data_ptr()); o_arr.append(out.data_ptr())
    # at most SORT_JOBS_MAX jobs per launch (csrc/bin_kernels.h BIN_MAX_JOBS: the kernel's argument block); a deeper backbone -- five
    # or more levels of >= 256 points make nine or more jobs -- takes one launch per chunk instead of an E_INVALID
    for c0 in range(0, len(n_arr), SORT_JOBS_MAX):
        k = min(SORT_JOBS_MAX, len(n_arr) - c0)
        sl = slice(c0, c0 + k)
        with _on(dev):
            check(lib.ws3d_sort_points_jobs(b, k, (C.c_int * k)(*n_arr[sl]), (C.c_int * k)(*k_arr[sl]), (C.c_void_p * k)(*x_arr[sl]),
                                            (C.c_void_p * k)(*o_arr[sl]), _stream()), "sort_points_jobs")
    return outs


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz_tensor, xyz_tensor, idx_tensor, sorted_xyz=None):
    """ball_query.cpp:14-25 (sorted_xyz: optional output of sort_points_x for this xyz)"""
    dev = _dev(new_xyz_tensor, xyz_tensor, idx_tensor, sorted_xyz)
    _f32(xyz_tensor, "xyz"); _f32(new_xyz_tensor, "new_xyz"); _i32(idx_tensor, "idx")
    with _on(dev):
        check(_lib.load().ws3d_ball_query(b, n, m, float(radius), nsample, _p(new_xyz_tensor),
                                           _p(xyz_tensor), _p(idx_tensor), _p(sorted_xyz), _stream()), "ball_query")
    return 1


def ball_query_lists(radius, nsample, xyz, new_xyz, sorted_xyz=None):
    """(B, M, nsample) int32 ball-query lists into a fresh (uncleared) tensor: rows of centres without a hit come out as zeros, as in
    a zero-initialised idx of ball_query_wrapper.  ws3d extension."""
    dev = _dev(xyz, new_xyz, sorted_xyz)
    _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz")
    B, N, M = xyz.size(0), xyz.size(1), new_xyz.size(1)
    idx = torch.empty((B, M, nsample), dtype=torch.int32, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_ball_query_fill(B, N, M, float(radius), nsample, _p(new_xyz), _p(xyz), _p(idx), _p(sorted_xyz), _stream()),
              "ball_query_fill")
    return idx


def ball_query_pairs(radius, nsample, xyz, new_xyz, sorted_grid, total=None):
    """ball_query_lists + compact_pairs in ONE launch: -> (idx (B, M, nsample) int32, (rowc, rowsrc, total)), or None when the
    kernel that emits the pairs does not cover the call (no fine-grid buffer, nsample > 64).  `total`: a 1-element int32 tensor
    that is ZERO (e.g. a slice of one cleared buffer shared by several calls); allocated and cleared here when None.  ws3d extension."""
    if sorted_grid is None or nsample > 64 or not BQ_FINE_GRID:
        return None
    dev = _dev(xyz, new_xyz, sorted_grid)
    _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz")
    B, N, M = xyz.size(0), xyz.size(1), new_xyz.size(1)
    idx = torch.empty((B, M, nsample), dtype=torch.int32, device=dev)
    rowc = torch.empty(B * M * nsample, dtype=torch.int32, device=dev)
    rowsrc = torch.empty(B * M * nsample, dtype=torch.int32, device=dev)
    if total is None:
        total = torch.zeros(1, dtype=torch.int32, device=dev)
    with _on(dev):
        rc = _lib.load().ws3d_ball_query_pairs(B, N, M, float(radius), nsample, _p(new_xyz), _p(xyz), _p(idx), _p(sorted_grid), _p(rowc), _p(rowsrc),
                                               _p(total), _stream())
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "ball_query_pairs")
    return idx, (rowc, rowsrc, total)


def ball_query_pairs2(radii, nsamples, xyz, new_xyz, sorted_grid, totals=None):
    """ball_query_pairs for the two scales of a level in ONE launch -> [(idx, (rowc, rowsrc, total)), (idx, (rowc, rowsrc, total))], or
    None when the dual kernel does not cover the call (the caller then takes ball_query_pairs per scale).  `totals`: two 1-element int32
    tensors that are ZERO, or None.  ws3d extension."""
    if sorted_grid is None or max(nsamples) > 64 or not BQ_FINE_GRID or len(radii) != 2:
        return None
    dev = _dev(xyz, new_xyz, sorted_grid)
    _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz")
    B, N, M = xyz.size(0), xyz.size(1), new_xyz.size(1)
    outs = []
    for k in range(2):
        ns = int(nsamples[k])
        outs.append((torch.empty((B, M, ns), dtype=torch.int32, device=dev), torch.empty(B * M * ns, dtype=torch.int32, device=dev),
                     torch.empty(B * M * ns, dtype=torch.int32, device=dev),
                     totals[k] if totals is not None else torch.zeros(1, dtype=torch.int32, device=dev)))
    with _on(dev):
        rc = _lib.load().ws3d_ball_query_pairs2(B, N, M, _p(new_xyz), _p(xyz), _p(sorted_grid),
                                                float(radii[0]), int(nsamples[0]), _p(outs[0][0]), _p(outs[0][1]), _p(outs[0][2]), _p(outs[0][3]),
                                                float(radii[1]), int(nsamples[1]), _p(outs[1][0]), _p(outs[1][1]), _p(outs[1][2]), _p(outs[1][3]), _stream())
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "ball_query_pairs2")
    return [(o[0], (o[1], o[2], o[3])) for o in outs]


def group_points_wrapper(b, c, n, npoints, nsample, points_tensor, idx_tensor, out_tensor):
    """group_points.cpp:25-36"""
    dev = _dev(points_tensor, idx_tensor, out_tensor)
    _f32(points_tensor, "points"); _i32(idx_tensor, "idx")
    with _on(dev):
        check(_lib.load().ws3d_group_points(b, c, n, npoints, nsample, _p(points_tensor), _p(idx_tensor),
                                             _p(out_tensor), _stream()), "group_points")
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out_tensor, idx_tensor, grad_points_tensor):
    """group_points.cpp:11-22"""
    dev = _dev(grad_out_tensor, idx_tensor, grad_points_tensor)
    _f32(grad_out_tensor, "grad_out"); _i32(idx_tensor, "idx")
    with _on(dev):
        check(_lib.load().ws3d_group_points_grad(b, c, n, npoints, nsample, _p(grad_out_tensor),
                                                  _p(idx_tensor), _p(grad_points_tensor), _stream()),
              "group_points_grad")
    return 1


def query_and_group(b, n, m, c, radius, nsample, use_xyz, xyz, new_xyz, features, idx_out, out, sorted_xyz=None):
    """fused a5 (ws3d extension)"""
    dev = _dev(xyz, new_xyz, features, idx_out, out, sorted_xyz)
    _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz")
    with _on(dev):
        check(_lib.load().ws3d_query_and_group(b, n, m, c, float(radius), nsample, int(bool(use_xyz)),
                                                _p(xyz), _p(new_xyz), _p(features), _p(idx_out), _p(out),
                                                _p(sorted_xyz), _stream()), "query_and_group")
    return 1


def query_and_group_nlc(radius, nsample, xyz, new_xyz, features_nlc, use_xyz=True, sorted_xyz=None, idx_out=None):
    """channels-last fused QueryAndGroup: xyz (B,N,3), new_xyz (B,M,3), features_nlc (B,N,C) or None
    -> (B, M, nsample, 3*use_xyz + C) rows [dx,dy,dz, features...] (ws3d extension)"""
    dev = _dev(xyz, new_xyz, features_nlc, sorted_xyz)
    _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz")
    B, N, M = xyz.size(0), xyz.size(1), new_xyz.size(1)
    C = 0 if features_nlc is None else features_nlc.size(2)
    if features_nlc is not None:
        _f32(features_nlc, "features")
    out = torch.empty((B, M, nsample, (3 if use_xyz else 0) + C), dtype=torch.float32, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_query_and_group_nlc(B, N, M, C, float(radius), nsample, int(bool(use_xyz)), _p(xyz),
                                                   _p(new_xyz), _p(features_nlc), _p(idx_out), _p(out),
                                                   _p(sorted_xyz), _stream()), "query_and_group_nlc")
    return out


def three_nn_jobs(jobs):
    """jobs: [(unknown (B,N_k,3), known (B,M_k,3), sorted_known (sort_points_xz / _x of known), sorted_unknown or None)] ->
    [(idx (B,N_k,3) int32, weight (B,N_k,3))]: three_nn_with_weights of every job in ONE launch (ws3d_three_nn_jobs), or None when a
    job does not fit it (more than 4 jobs, a known set that is not binned or larger than 4096 points).  ws3d extension."""
    import ctypes as C
    if not jobs or len(jobs) > 4 or any(sk is None or kn.size(1) > 4096 or kn.size(1) < 3 or (su is not None and un.size(1) > 16384) for un, kn, sk, su in jobs):
        return None
    dev = _dev(*[t for un, kn, sk, su in jobs for t in (un, kn, sk)])
    B = jobs[0][0].size(0)
    k = len(jobs)
    outs = []
    for un, kn, sk, su in jobs:
        _f32(un, "unknown"); _f32(kn, "known")
        if un.size(0) != B or kn.size(0) != B:
            raise ValueError("three_nn_jobs: every job takes the same batch")
        N = un.size(1)
        outs.append((torch.empty((B, N, 3), dtype=torch.float32, device=dev), torch.empty((B, N, 3), dtype=torch.int32, device=dev),
                     torch.empty((B, N, 3), dtype=torch.float32, device=dev)))
    arr = lambda vals: (C.c_void_p * k)(*vals)
    with _on(dev):
        check(_lib.load().ws3d_three_nn_jobs(B, k, (C.c_int * k)(*[j[0].size(1) for j in jobs]), (C.c_int * k)(*[j[1].size(1) for j in jobs]),
                                             arr([j[0].data_ptr() for j in jobs]), arr([j[2].data_ptr() for j in jobs]),
                                             arr([o[0].data_ptr() for o in outs]), arr([o[1].data_ptr() for o in outs]), arr([o[2].data_ptr() for o in outs]),
                                             arr([None if j[3] is None else j[3].data_ptr() for j in jobs]), _stream()), "three_nn_jobs")
    return [(o[1], o[2]) for o in outs]


def three_nn_with_weights(unknown, known, sorted_known=None, sorted_unknown=None):
    """unknown (B,N,3), known (B,M,3) -> (idx (B,N,3) int32, weight (B,N,3)): three_nn + the FP module's
    normalised inverse-distance weights in ONE launch (ws3d_three_nn_w: the weights are the search kernel's epilogue; bit-identical
    to three_nn_wrapper + ws3d_three_nn_weights).  sorted_unknown: a binned copy of `unknown` (sort_points_x / sort_points_xz of the
    same tensor): the queries are then taken in cell order (ws3d_three_nn_wq; same rows).  ws3d extension."""
    dev = _dev(unknown, known)
    _f32(unknown, "unknown"); _f32(known, "known")
    B, N, M = unknown.size(0), unknown.size(1), known.size(1)
    d2 = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
    idx = torch.empty((B, N, 3), dtype=torch.int32, device=dev)
    w = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
    lib = _lib.load()
    with _on(dev):
        if sorted_unknown is not None and sorted_known is not None:
            check(lib.ws3d_three_nn_wq(B, N, M, _p(unknown), _p(known), _p(d2), _p(idx), _p(w), _p(sorted_known), _p(sorted_unknown),
                                       _stream()), "three_nn_wq")
        else:
            check(lib.ws3d_three_nn_w(B, N, M, _p(unknown), _p(known), _p(d2), _p(idx), _p(w),
                                      _p(sorted_known) if sorted_known is not None else None, _stream()), "three_nn_w")
    return idx, w


def three_interpolate_nlc(feats_nlc, idx, weight, out=None):
    """feats_nlc (B,M,C), idx/weight (B,N,3) -> (B,N,C); `out` may be a (B,N,>=C) buffer whose first
    C columns are written (row stride = out.size(2)).  ws3d extension."""
    dev = _dev(feats_nlc, idx, weight)
    _f32(feats_nlc, "feats"); _i32(idx, "idx"); _f32(weight, "weight")
    B, M, C = feats_nlc.shape
    N = idx.size(1)
    if out is None:
        out = torch.empty((B, N, C), dtype=torch.float32, device=dev)
    _f32(out, "out")
    with _on(dev):
        check(_lib.load().ws3d_three_interpolate_nlc(B, C, M, N, _p(feats_nlc), _p(idx), _p(weight), _p(out),
                                                     out.size(2), _stream()), "three_interpolate_nlc")
    return out


def rowmax_rows(y, ns, out=None, col0=0):
    """y (R*ns, O) contiguous -> max over each group of ns consecutive rows; written into
    out[:, col0:col0+O] of a (R, >=O) buffer when given.  ws3d extension."""
    dev = _dev(y)
    _f32(y, "y")
    rows, O = y.shape
    R = rows // ns
    if out is None:
        out = torch.empty((R, O), dtype=torch.float32, device=dev)
    view = out[:, col0:col0 + O]
    with _on(dev):
        check(_lib.load().ws3d_rowmax_rows(R, ns, O, _p(y), view.data_ptr(), out.size(1), _stream()), "rowmax_rows")
    return out


def bn_relu_train_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, relu=True, num_batches_tracked=None):
    """x (B,C,...) contiguous fp32 -> (y, save_mean (C), save_invstd (C)): training-mode BatchNorm
    (+ReLU), running stats (or None) updated in place, the int64 scalar num_batches_tracked (or None)
    incremented on the device.  ws3d extension."""
    dev = _dev(x, gamma, beta, running_mean, running_var, num_batches_tracked)
    if num_batches_tracked is not None and num_batches_tracked.dtype != torch.int64:
        raise TypeError("num_batches_tracked must be int64")
    _f32(x, "x"); _f32(gamma, "gamma"); _f32(beta, "beta")
    b, c = x.size(0), x.size(1)
    l = x.numel() // max(b * c, 1)
    y = torch.empty_like(x)
    mean = torch.empty((c,), dtype=torch.float32, device=dev)
    invstd = torch.empty((c,), dtype=torch.float32, device=dev)
    lib = _lib.load()
    nbytes = lib.ws3d_bn_workspace_bytes(b, c, l)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=dev)
    with _on(dev):
        check(lib.ws3d_bn_relu_train_fwd(b, c, l, _p(x), _p(gamma), _p(beta), float(eps), float(momentum), int(bool(relu)),
                                         _p(running_mean), _p(running_var), _p(num_batches_tracked), _p(y), _p(mean), _p(invstd), _p(ws), nbytes,
                                         _stream()), "bn_relu_train_fwd")
    return y, mean, invstd


def bn_relu_train_bwd(x, dy, gamma, beta, save_mean, save_invstd, relu=True):
    """-> (dx like x, dgamma (C), dbeta (C)).  ws3d extension."""
    dev = _dev(x, dy, gamma, beta, save_mean, save_invstd)
    _f32(x, "x"); _f32(dy, "dy")
    b, c = x.size(0), x.size(1)
    l = x.numel() // max(b * c, 1)
    dx = torch.empty_like(x)
    dgamma = torch.empty((c,), dtype=torch.float32, device=dev)
    dbeta = torch.empty((c,), dtype=torch.float32, device=dev)
    lib = _lib.load()
    nbytes = lib.ws3d_bn_workspace_bytes(b, c, l)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=dev)
    with _on(dev):
        check(lib.ws3d_bn_relu_train_bwd(b, c, l, _p(x), _p(dy), _p(gamma), _p(beta), _p(save_mean), _p(save_invstd),
                                         int(bool(relu)), _p(dx), _p(dgamma), _p(dbeta), _p(ws), nbytes, _stream()),
              "bn_relu_train_bwd")
    return dx, dgamma, dbeta


def conv1x1_wgrad(grad_out, x, shape=None):
    """grad_out (B, O, ...), x (B, C, ...) contiguous, same trailing shape -> grad_w (O, C) of a 1x1 convolution
    (allocated as `shape`, e.g. (O, C, 1, 1), when given); deterministic.  ws3d extension."""
    dev = _dev(grad_out, x)
    _f32(grad_out, "grad_out"); _f32(x, "x")
    b, o, c = x.size(0), grad_out.size(1), x.size(1)
    l = x.numel() // max(b * c, 1)
    if grad_out.numel() != b * o * l:
        raise Ws3dError("conv1x1_wgrad: grad_out and x disagree in shape")
    gw = torch.empty((o, c) if shape is None else tuple(shape), dtype=torch.float32, device=dev)
    lib = _lib.load()
    nbytes = lib.ws3d_conv1x1_wgrad_workspace_bytes(b, o, c, l)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
    with _on(dev):
        check(lib.ws3d_conv1x1_wgrad(b, o, c, l, _p(grad_out), _p(x), _p(gw), _p(ws), nbytes, _stream()), "conv1x1_wgrad")
    return gw


def _gate(gate):
    """(total tensor, limit) -> (pointer, limit) of a dense kernel's launch gate (run iff *total > limit); None -> always run"""
    if gate is None:
        return None, 0
    total, limit = gate
    return total.data_ptr(), int(limit)


def gemm_pool(x_rows, wt, bias, relu, nsample, out, col0=0, gate=None):
    """out[:, col0:col0+O] = act(max over each group of nsample rows of x_rows @ wt + bias); False when the
    shape is not covered by the fused kernel (rows, O multiples of 64, K of 4, nsample 16 / 32).  gate = (total, limit):
    device-side dispatch, the kernel runs iff *total > limit (ws3d_ops.h "launch gates").  ws3d extension."""
    dev = _dev(x_rows, wt, bias)
    _f32(x_rows, "x_rows"); _f32(wt, "wt")
    rows, k = x_rows.shape
    o = wt.size(1)
    if rows % 64 or rows // 64 > 65535 or o % 64 or k % 4 or nsample not in (16, 32) or wt.size(0) != k:
        return False
    view = out[:, col0:col0 + o]
    with _on(dev):
        check(_lib.load().ws3d_gemm_pool(rows, nsample, k, o, _p(x_rows), _p(wt), _p(bias), int(bool(relu)), view.data_ptr(),
                                         out.size(1), *_gate(gate), _stream()), "gemm_pool")
    return True


def gather_gemm(feats, xyz, new_xyz, nbr, wt_feat_then_xyz, bias, relu):
    """first SharedMLP layer with the grouping fused in: feats (B,N,C), xyz (B,N,3), new_xyz (B,M,3), nbr (B,M,ns) int32,
    wt (C+3, O) with the xyz rows LAST -> (B*M*ns, O), or None when the shape is not covered (C % 4, O % 64, rows % 64).
    ws3d extension."""
    dev = _dev(feats, xyz, new_xyz, nbr, wt_feat_then_xyz)
    _f32(feats, "feats"); _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz"); _i32(nbr, "nbr"); _f32(wt_feat_then_xyz, "wt")
    B, N, C = feats.shape
    M, ns = nbr.size(1), nbr.size(2)
    O = wt_feat_then_xyz.size(1)
    rows = B * M * ns
    if C % 4 or O % 64 or rows % 64 or rows // 64 > 65535 or wt_feat_then_xyz.size(0) != C + 3 or not feats.is_contiguous():
        return None
    out = torch.empty((rows, O), dtype=torch.float32, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_gather_gemm(B, N, M, ns, C, O, _p(feats), _p(xyz), _p(new_xyz), _p(nbr), _p(wt_feat_then_xyz), _p(bias),
                                           int(bool(relu)), _p(out), _stream()), "gather_gemm")
    return out


def gather_gemm2(feats, xyz, new_xyz, nbr, w1t_feat_then_xyz, b1, relu1, w2t, b2, relu2):
    """the first TWO SharedMLP layers with the grouping fused in (the first activation stays on chip): w1t (C+3, O1) with the
    xyz rows LAST, w2t (O1, O2) -> (B*M*ns, O2), or None when the shape is not covered (C % 4, O1 in {64,128,256}, O2 % 4,
    rows % 64).  ws3d extension."""
    dev = _dev(feats, xyz, new_xyz, nbr, w1t_feat_then_xyz, w2t)
    _f32(feats, "feats"); _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz"); _i32(nbr, "nbr"); _f32(w1t_feat_then_xyz, "w1t"); _f32(w2t, "w2t")
    B, N, C = feats.shape
    M, ns = nbr.size(1), nbr.size(2)
    O1, O2 = w1t_feat_then_xyz.size(1), w2t.size(1)
    rows = B * M * ns
    if (C % 4 or O1 not in (64, 128, 256) or O2 % 4 or rows % 64 or w1t_feat_then_xyz.size(0) != C + 3 or w2t.size(0) != O1 or
            not feats.is_contiguous() or not w2t.is_contiguous() or not w1t_feat_then_xyz.is_contiguous()):
        return None
    out = torch.empty((rows, O2), dtype=torch.float32, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_gather_gemm2(B, N, M, ns, C, O1, O2, _p(feats), _p(xyz), _p(new_xyz), _p(nbr), _p(w1t_feat_then_xyz), _p(b1),
                                            int(bool(relu1)), _p(w2t), _p(b2), int(bool(relu2)), _p(out), _stream()), "gather_gemm2")
    return out


def pgather_gemm2(pmat, col0, o1, xyz, new_xyz, nbr, w1x, b1, relu1, w2t, b2, relu2, out=None, gate=None):
    """layer 1 from the per-point product P = feats @ W_f (pmat (B*N, W) row-major, this scale's columns col0 .. col0 + o1) +
    the xyz term (w1x (3, o1)) + bias + ReLU, then layer 2 (w2t (o1, O2)): -> (B*M*ns, O2), or None when the shape is not
    covered (o1 in {64, 128}, O2 % 4, M*ns % 64).  ws3d extension."""
    dev = _dev(pmat, xyz, new_xyz, nbr, w1x, w2t)
    _f32(pmat, "pmat"); _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz"); _i32(nbr, "nbr"); _f32(w1x, "w1x"); _f32(w2t, "w2t")
    B, N = xyz.size(0), xyz.size(1)
    M, ns = nbr.size(1), nbr.size(2)
    O2 = w2t.size(1)
    if (o1 not in (64, 128) or O2 % 4 or (M * ns) % 64 or pmat.dim() != 2 or pmat.size(0) != B * N or pmat.stride(1) != 1 or
            col0 < 0 or col0 + o1 > pmat.size(1) or tuple(w1x.shape) != (3, o1) or w2t.size(0) != o1 or not w1x.is_contiguous() or not w2t.is_contiguous()):
        return None
    if out is None:
        out = torch.empty((B * M * ns, O2), dtype=torch.float32, device=dev)
    elif tuple(out.shape) != (B * M * ns, O2) or not out.is_contiguous():
        raise ValueError("pgather_gemm2: out must be a contiguous (B*M*ns, O2) tensor")
    with _on(dev):
        check(_lib.load().ws3d_pgather_gemm2(B, N, M, ns, o1, O2, pmat.data_ptr() + 4 * col0, pmat.stride(0), _p(xyz), _p(new_xyz), _p(nbr), _p(w1x), _p(b1),
                                             int(bool(relu1)), _p(w2t), _p(b2), int(bool(relu2)), _p(out), *_gate(gate), _stream()), "pgather_gemm2")
    return out


def pgather_rows(pmat, col0, o1, xyz, new_xyz, nbr, w1x, b1, relu1):
    """layer 1 alone from the per-point product: -> (B*M*ns, o1), or None when the shape is not covered (o1, col0, row stride % 4)."""
    dev = _dev(pmat, xyz, new_xyz, nbr, w1x)
    _f32(pmat, "pmat"); _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz"); _i32(nbr, "nbr"); _f32(w1x, "w1x")
    B, N = xyz.size(0), xyz.size(1)
    M, ns = nbr.size(1), nbr.size(2)
    if (o1 % 4 or col0 % 4 or pmat.dim() != 2 or pmat.size(0) != B * N or pmat.stride(1) != 1 or pmat.stride(0) % 4 or col0 < 0 or
            col0 + o1 > pmat.size(1) or tuple(w1x.shape) != (3, o1) or not w1x.is_contiguous() or pmat.data_ptr() % 16):
        return None
    out = torch.empty((B * M * ns, o1), dtype=torch.float32, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_pgather_rows(B, N, M, ns, o1, pmat.data_ptr() + 4 * col0, pmat.stride(0), _p(xyz), _p(new_xyz), _p(nbr), _p(w1x), _p(b1),
                                            int(bool(relu1)), _p(out), _stream()), "pgather_rows")
    return out


def qinterp_rows(q, idx, weight, lin=None, skip=None, wb=None, bias=None, relu=True):
    """first FP layer from the per-KNOWN-point product Q = known_feats @ W_a (B, M, O): interpolate Q's rows (idx / weight (B, N, 3)),
    add lin (B*N, O) = skip @ W_b + bias -- or, lin None and <= 4 skip channels, skip (B, N, C1) @ wb (C1, O) + bias -- and ReLU:
    -> (B*N, O), or None when the shape is not covered.  ws3d extension."""
    dev = _dev(q, idx, weight)
    _f32(q, "q"); _i32(idx, "idx"); _f32(weight, "weight")
    B, M, O = q.shape
    N = idx.size(1)
    c1 = 0 if skip is None else skip.size(2)
    if (O % 4 or not q.is_contiguous() or (lin is None and c1 > 4) or (lin is not None and (tuple(lin.shape) != (B * N, O) or not lin.is_contiguous())) or
            (lin is None and c1 > 0 and (wb is None or tuple(wb.shape) != (c1, O) or not wb.is_contiguous() or not skip.is_contiguous()))):
        return None
    out = torch.empty((B * N, O), dtype=torch.float32, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_qinterp_rows(B, N, M, O, _p(q), _p(idx), _p(weight), _p(lin), _p(skip), c1, _p(wb), _p(bias), int(bool(relu)), _p(out),
                                            _stream()), "qinterp_rows")
    return out


def qinterp_gemm(q, idx, weight, w2t, b2, relu2, lin=None, skip=None, wb=None, bias=None, relu=True):
    """qinterp_rows + the module's second layer in ONE kernel (the first layer's rows are built in the product's A operand, to the bit as
    qinterp_rows builds them): q (B, M, C), idx / weight (B, N, 3), lin (B*N, C) or skip (B, N, C1 <= 4) + wb (C1, C) + bias, w2t (C, O)
    -> (B*N, O), or None when the shape is not covered (B*N % 64, C % 16, O % 128).  ws3d extension."""
    dev = _dev(q, idx, weight, w2t)
    _f32(q, "q"); _i32(idx, "idx"); _f32(weight, "weight"); _f32(w2t, "w2t")
    B, M, C = q.shape
    N = idx.size(1)
    O = w2t.size(1)
    c1 = 0 if skip is None else skip.size(2)
    if ((B * N) % 64 or C % 16 or O % 128 or w2t.size(0) != C or not q.is_contiguous() or not w2t.is_contiguous() or not idx.is_contiguous() or
            not weight.is_contiguous() or (lin is None and c1 > 4) or (lin is not None and (tuple(lin.shape) != (B * N, C) or not lin.is_contiguous())) or
            (lin is None and c1 > 0 and (wb is None or tuple(wb.shape) != (c1, C) or not wb.is_contiguous() or not skip.is_contiguous()))):
        return None
    out = torch.empty((B * N, O), dtype=torch.float32, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_qinterp_gemm(B, N, M, C, O, _p(q), _p(idx), _p(weight), _p(lin), _p(skip), c1, _p(wb), _p(bias), int(bool(relu)), _p(w2t),
                                            _p(b2), int(bool(relu2)), _p(out), _stream()), "qinterp_gemm")
    return out


def compact_pairs(nbr, ordered=False):
    """(B, M, ns) ball-query lists -> (rowc, rowsrc, total): the DISTINCT (centre, source point) pairs as compact rows (int32
    tensors of B*M*ns entries, the first `total` -- a 1-element device tensor -- valid); a list's padding repeats its first hit
    and cannot change any layer's maximum over nsample.  ordered=False: one launch, rows of a centre contiguous, centres in
    arrival order of their workgroups; ordered=True: centres ascending (count kernel + prefix sum + placement kernel).
    ws3d extension."""
    if not ordered:
        dev = _dev(nbr)
        _i32(nbr, "nbr")
        B, M, ns = nbr.shape
        rowc = torch.empty(B * M * ns, dtype=torch.int32, device=dev)
        rowsrc = torch.empty(B * M * ns, dtype=torch.int32, device=dev)
        total = torch.zeros(1, dtype=torch.int32, device=dev)
        with _on(dev):
            check(_lib.load().ws3d_compact_pairs(B * M, ns, _p(nbr), _p(rowc), _p(rowsrc), _p(total), _stream()), "compact_pairs")
        return rowc, rowsrc, total
    dev = _dev(nbr)
    _i32(nbr, "nbr")
    B, M, ns = nbr.shape
    centres = B * M
    cnt = torch.empty(centres, dtype=torch.int32, device=dev)
    rowc = torch.empty(centres * ns, dtype=torch.int32, device=dev)
    rowsrc = torch.empty(centres * ns, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    with _on(dev):
        lib = _lib.load()
        check(lib.ws3d_compact_pairs_count(centres, ns, _p(nbr), _p(cnt), _stream()), "compact_pairs_count")
        incl = torch.cumsum(cnt, dim=0, dtype=torch.int32)
        check(lib.ws3d_compact_pairs_rows(centres, ns, _p(nbr), _p(cnt), _p(incl), _p(rowc), _p(rowsrc), _p(total), _stream()), "compact_pairs_rows")
    return rowc, rowsrc, total


def pgather_gemm2_compact(pmat, col0, o1, xyz, new_xyz, pairs, w1x, b1, relu1, w2t, b2, relu2, limit=-1):
    """pgather_gemm2 over the compact rows of compact_pairs: -> (B*M*ns, O2) of which the first `total` rows are written, or None
    when the shape is not covered (o1 in {64, 128, 256}, O2 % 4).  limit >= 0: runs iff total <= limit (launch gate).
    ws3d extension."""
    rowc, rowsrc, total = pairs
    dev = _dev(pmat, xyz, new_xyz, rowc, w1x, w2t)
    _f32(pmat, "pmat"); _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz"); _f32(w1x, "w1x"); _f32(w2t, "w2t")
    B, N = xyz.size(0), xyz.size(1)
    M = new_xyz.size(1)
    O2 = w2t.size(1)
    rows = rowc.numel()
    if (o1 not in (64, 128, 256) or O2 % 4 or pmat.dim() != 2 or pmat.size(0) != B * N or pmat.stride(1) != 1 or col0 < 0 or col0 + o1 > pmat.size(1) or
            tuple(w1x.shape) != (3, o1) or w2t.size(0) != o1 or not w1x.is_contiguous() or not w2t.is_contiguous()):
        return None
    out = torch.empty((rows, O2), dtype=torch.float32, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_pgather_gemm2_compact(B, N, M, rows, o1, O2, pmat.data_ptr() + 4 * col0, pmat.stride(0), _p(xyz), _p(new_xyz), _p(rowc),
                                                     _p(rowsrc), _p(total), _p(w1x), _p(b1), int(bool(relu1)), _p(w2t), _p(b2), int(bool(relu2)), _p(out),
                                                     int(limit), _stream()), "pgather_gemm2_compact")
    return out


def pgather_gemm3_compact(pmat, col0, o1, xyz, new_xyz, pairs, w1x, b1, relu1, w2t, b2, relu2, w3t, b3, out2d, col_offset, limit=-1, max_lds=160 * 1024):
    """pgather_gemm2_compact + gemm_pool_compact in ONE kernel (layer 2's tile stays in LDS): the whole SharedMLP of a scale over the
    compact rows, reduced by atomic max into out2d[:, col_offset : col_offset + O3] -- ZERO on entry; the last layer ends in a ReLU.
    True, or False when the shape is not covered (o1 in {64, 128}, O2 % 4, O3 % 128, the two tiles within `max_lds` bytes of LDS).
    Bit-identical to the two-kernel form.  ws3d extension."""
    rowc, rowsrc, total = pairs
    dev = _dev(pmat, xyz, new_xyz, rowc, w1x, w2t, w3t, out2d)
    _f32(pmat, "pmat"); _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz"); _f32(w1x, "w1x"); _f32(w2t, "w2t"); _f32(w3t, "w3t"); _f32(out2d, "out2d")
    B, N = xyz.size(0), xyz.size(1)
    M = new_xyz.size(1)
    O2, O3 = w2t.size(1), w3t.size(1)
    rows = rowc.numel()
    lds = 4 * ((o1 + (O2 + 15) // 16 * 16) * 65 + 2 * 16 * 64)
    if (o1 not in (64, 128) or O2 % 4 or O3 % 128 or lds > min(max_lds, 160 * 1024 - 1024) or pmat.dim() != 2 or pmat.size(0) != B * N or pmat.stride(1) != 1 or
            col0 < 0 or col0 + o1 > pmat.size(1) or tuple(w1x.shape) != (3, o1) or w2t.size(0) != o1 or w3t.size(0) != O2 or not w1x.is_contiguous() or
            not w2t.is_contiguous() or not w3t.is_contiguous() or out2d.dim() != 2 or out2d.stride(1) != 1 or col_offset < 0 or col_offset + O3 > out2d.size(1)):
        return False
    with _on(dev):
        check(_lib.load().ws3d_pgather_gemm3_compact(B, N, M, rows, o1, O2, O3, pmat.data_ptr() + 4 * col0, pmat.stride(0), _p(xyz), _p(new_xyz), _p(rowc),
                                                     _p(rowsrc), _p(total), _p(w1x), _p(b1), int(bool(relu1)), _p(w2t), _p(b2), int(bool(relu2)), _p(w3t), _p(b3),
                                                     out2d.data_ptr() + 4 * col_offset, out2d.stride(0), int(limit), _stream()), "pgather_gemm3_compact")
    return True


def gemm_pool_compact(x_rows, pairs, wt, bias, out2d, col_offset, limit=-1):
    """last SharedMLP layer (+ bias + ReLU) over the compact rows + max over each centre's rows, by atomic max into
    out2d[:, col_offset : col_offset + O] -- which must be ZERO on entry.  True, or False when the shape is not covered
    (K % 4, O % 64).  ws3d extension."""
    rowc, rowsrc, total = pairs
    dev = _dev(x_rows, rowc, wt, out2d)
    _f32(x_rows, "x_rows"); _f32(wt, "wt"); _f32(out2d, "out2d")
    rows, k = x_rows.shape
    o = wt.size(1)
    if (k % 4 or o % 64 or wt.size(0) != k or rows != rowc.numel() or not x_rows.is_contiguous() or not wt.is_contiguous() or out2d.dim() != 2 or
            out2d.stride(1) != 1 or col_offset < 0 or col_offset + o > out2d.size(1)):
        return False
    with _on(dev):
        check(_lib.load().ws3d_gemm_pool_compact(rows, k, o, _p(x_rows), _p(rowc), _p(total), _p(wt), _p(bias), out2d.data_ptr() + 4 * col_offset,
                                                 out2d.stride(0), int(limit), _stream()), "gemm_pool_compact")
    return True


def compact_mlp_pair(kind, scales, max_lds=160 * 1024, mids=None):
    """The two scales of a set-abstraction level in ONE launch (ws3d_compact_mlp_pair).  scales: two dicts with the arguments of the
    single-scale calls -- pmat, col0, o1, xyz, new_xyz, pairs, w1x, b1, relu1, w2t, b2, relu2, w3t, b3, out2d, col_offset.
    kind 3: pgather_gemm3_compact of both -> True / False (shape not covered: nothing launched); kind 2: pgather_gemm2_compact of both
    -> the two (rows, O2) tensors, or None; kind 1: gemm_pool_compact of `mids` (kind 2's result) into out2d -> True / False.  Both
    scales must have the same first-layer width.  Same results as the single-scale calls, always ungated (limit -1).  ws3d extension."""
    import ctypes as C
    if len(scales) != 2 or scales[0]["o1"] != scales[1]["o1"]:
        return None if kind == 2 else False
    blocks, keep, outs = [], [], []
    for si, a in enumerate(scales):
        rowc, rowsrc, total = a["pairs"]
        pmat, xyz, new_xyz, w1x, w2t, w3t, out2d, o1, col0, col = a["pmat"], a["xyz"], a["new_xyz"], a["w1x"], a["w2t"], a["w3t"], a["out2d"], a["o1"], a["col0"], a["col_offset"]
        dev = _dev(pmat, xyz, new_xyz, rowc, w1x, w2t, w3t, out2d)
        for t_, nm in ((pmat, "pmat"), (xyz, "xyz"), (new_xyz, "new_xyz"), (w1x, "w1x"), (w2t, "w2t"), (w3t, "w3t"), (out2d, "out2d")):
            _f32(t_, nm)
        B, N, M = xyz.size(0), xyz.size(1), new_xyz.size(1)
        O2, O3 = w2t.size(1), w3t.size(1)
        rows = rowc.numel()
        lds3 = 4 * ((o1 + (O2 + 15) // 16 * 16) * 65 + 2 * 16 * 64)
        ok = (O2 % 4 == 0 and pmat.dim() == 2 and pmat.size(0) == B * N and pmat.stride(1) == 1 and col0 >= 0 and col0 + o1 <= pmat.size(1) and
              tuple(w1x.shape) == (3, o1) and w2t.size(0) == o1 and w3t.size(0) == O2 and w1x.is_contiguous() and w2t.is_contiguous() and w3t.is_contiguous() and
              out2d.dim() == 2 and out2d.stride(1) == 1 and col >= 0 and col + O3 <= out2d.size(1))
        if kind == 3:
            ok = ok and o1 in (64, 128) and O3 % 128 == 0 and lds3 <= min(max_lds, 160 * 1024 - 1024)
        elif kind == 2:
            ok = ok and o1 in (64, 128, 256)
        else:
            ok = ok and O3 % 64 == 0 and mids is not None and tuple(mids[si].shape) == (rows, O2) and mids[si].is_contiguous()
        if not ok:
            return None if kind == 2 else False
        mid = mids[si] if kind == 1 else (torch.empty((rows, O2), dtype=torch.float32, device=dev) if kind == 2 else None)
        outs.append(mid)
        keep.append((pmat, xyz, new_xyz, rowc, rowsrc, total, w1x, w2t, w3t, out2d, mid, a["b1"], a["b2"], a["b3"]))
        blocks.append(_lib.CompactMlpArgs(B, N, M, rows, o1, O2, O3, pmat.data_ptr() + 4 * col0, pmat.stride(0), _p(xyz), _p(new_xyz), _p(rowc), _p(rowsrc),
                                          _p(total), _p(w1x), _p(a["b1"]), int(bool(a["relu1"])), _p(w2t), _p(a["b2"]), int(bool(a["relu2"])), _p(w3t), _p(a["b3"]),
                                          _p(mid), out2d.data_ptr() + 4 * col, out2d.stride(0), -1))
    with _on(dev):
        rc = _lib.load().ws3d_compact_mlp_pair(int(kind), C.byref(blocks[0]), C.byref(blocks[1]), _stream())
    if rc == _lib.E_UNSUPPORTED:        # a condition the C side checks and this restatement does not (alignment of w2t / w3t / mid, the o1
        return None if kind == 2 else False      # rule per kind): nothing was launched, the caller's per-scale path runs
    check(rc, "compact_mlp_pair")
    return outs if kind == 2 else True


CHAIN_WORKGROUPS = 0      # ws3d_chain_mlp3: workgroups per launch (0 = the library's choice, ws3d_tune key 0)
TUNE_KEYS = {"chain_wgs": 0, "mlp2_wgs": 1, "sa1_wgs": 2, "fp_wgs": 3, "pair_wgs": 4, "bq_wide_nw": 5}


def tune(name: str, value: int = -1) -> int:
    """ws3d_tune: set (value >= 0; 0 = built-in) or read (value < 0) a launch-geometry knob of the persistent kernels -> previous value"""
    rc = _lib.load().ws3d_tune(TUNE_KEYS[name], int(value))
    if rc < 0:
        check(rc, "tune")
    return rc


_CHAIN_BLOBS = {}         # (data pointers + versions of a scale's six weight tensors) -> (packed blob, the tensors: kept alive so that the pointers stay theirs)


def chain_ticket_ints() -> int:
    """int32 elements of chain_mlp3's (zeroed) ticket tensor"""
    return int(_lib.load().ws3d_chain_mlp3_ticket_ints())


def _chain_blob(block, tensors, dev):
    import ctypes as C
    key = tuple((t.data_ptr(), t._version) if t is not None else None for t in tensors)
    hit = _CHAIN_BLOBS.get(key)
    if hit is None:
        n = int(_lib.load().ws3d_chain_mlp3_blob_floats(block.o2))
        blob = torch.empty(n, dtype=torch.float32, device=dev)
        with _on(dev):
            check(_lib.load().ws3d_chain_mlp3_pack(C.byref(block), _p(blob), _stream()), "chain_mlp3_pack")
        if torch.cuda.is_current_stream_capturing():        # packed inside a capture: valid for this graph's replays only, not cached
            return blob
        hit = _CHAIN_BLOBS[key] = (blob, tensors)         # never evicted: a captured hipGraph may hold the blob's address (50-76 KB per weight set)
    return hit[0]


def chain_mlp3(scales, ticket):
    """The whole SharedMLP (three layers + pool) of one or two scales over their compact rows on the register-chained kernel
    (ws3d_chain_mlp3, csrc/chain_mlp.hip).  scales: compact_mlp_pair's dicts (one or two); ticket: a ZEROED int32 tensor of
    chain_ticket_ints() elements, consumed.  The scales' weights are packed once per weight set (ws3d_chain_mlp3_pack, cached here).
    True, or False when a shape is not covered (o1 = 64, o2 <= 96, o3 = 128): nothing launched.  Bit-identical to
    compact_mlp_pair(3, ..) / pgather_gemm3_compact.  ws3d extension."""
    import ctypes as C
    if len(scales) not in (1, 2) or ticket is None or ticket.dtype != torch.int32 or ticket.numel() < chain_ticket_ints() or not ticket.is_contiguous():
        return False
    blocks, keep, blobs = [], [], []
    for a in scales:
        rowc, rowsrc, total = a["pairs"]
        pmat, xyz, new_xyz, w1x, w2t, w3t, out2d, o1, col0, col = a["pmat"], a["xyz"], a["new_xyz"], a["w1x"], a["w2t"], a["w3t"], a["out2d"], a["o1"], a["col0"], a["col_offset"]
        dev = _dev(pmat, xyz, new_xyz, rowc, w1x, w2t, w3t, out2d, ticket)
        for t_, nm in ((pmat, "pmat"), (xyz, "xyz"), (new_xyz, "new_xyz"), (w1x, "w1x"), (w2t, "w2t"), (w3t, "w3t"), (out2d, "out2d")):
            _f32(t_, nm)
        B, N, M = xyz.size(0), xyz.size(1), new_xyz.size(1)
        O2, O3 = w2t.size(1), w3t.size(1)
        ok = (o1 == 64 and O2 <= 96 and O3 == 128 and pmat.dim() == 2 and pmat.size(0) == B * N and pmat.stride(1) == 1 and pmat.stride(0) % 4 == 0 and
              col0 >= 0 and col0 % 4 == 0 and col0 + o1 <= pmat.size(1) and tuple(w1x.shape) == (3, o1) and w2t.size(0) == o1 and w3t.size(0) == O2 and
              w1x.is_contiguous() and w2t.is_contiguous() and w3t.is_contiguous() and out2d.dim() == 2 and out2d.stride(1) == 1 and col >= 0 and
              col % 4 == 0 and col + O3 <= out2d.size(1) and xyz.is_contiguous() and new_xyz.is_contiguous())
        if not ok:
            return False
        keep.append((pmat, xyz, new_xyz, rowc, rowsrc, total, w1x, w2t, w3t, out2d, a["b1"], a["b2"], a["b3"]))
        blocks.append(_lib.CompactMlpArgs(B, N, M, rowc.numel(), o1, O2, O3, pmat.data_ptr() + 4 * col0, pmat.stride(0), _p(xyz), _p(new_xyz), _p(rowc), _p(rowsrc),
                                          _p(total), _p(w1x), _p(a["b1"]), int(bool(a["relu1"])), _p(w2t), _p(a["b2"]), int(bool(a["relu2"])), _p(w3t), _p(a["b3"]),
                                          None, out2d.data_ptr() + 4 * col, out2d.stride(0), int(a.get("limit", -1))))
        blobs.append(_chain_blob(blocks[-1], (w1x, a["b1"], w2t, a["b2"], w3t, a["b3"]), dev))
    with _on(dev):
        rc = _lib.load().ws3d_chain_mlp3(C.byref(blocks[0]), C.byref(blocks[1]) if len(blocks) > 1 else None, _p(blobs[0]), _p(blobs[1]) if len(blobs) > 1 else None,
                                         _p(ticket), int(CHAIN_WORKGROUPS), _stream())
    if rc == _lib.E_UNSUPPORTED:
        return False
    check(rc, "chain_mlp3")
    return True


def interp_gemm(known_feats, unknown_feats, idx, weight, wt, bias, relu):
    """first FP-module layer with the interpolation + skip concat fused in: known_feats (B,M,C2), unknown_feats (B,N,C1) or
    None, idx / weight (B,N,3), wt (C2+C1, O) -> (B*N, O), or None when the shape is not covered.  ws3d extension."""
    dev = _dev(known_feats, idx, weight, wt)
    _f32(known_feats, "known_feats"); _i32(idx, "idx"); _f32(weight, "weight"); _f32(wt, "wt")
    B, M, C2 = known_feats.shape
    N = idx.size(1)
    C1 = 0 if unknown_feats is None else unknown_feats.size(2)
    O = wt.size(1)
    rows = B * N
    if (C2 % 4 or O % 64 or rows % 64 or rows // 64 > 65535 or wt.size(0) != C2 + C1 or not known_feats.is_contiguous() or
            (unknown_feats is not None and not unknown_feats.is_contiguous())):
        return None
    out = torch.empty((rows, O), dtype=torch.float32, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_interp_gemm(B, N, M, C2, C1, O, _p(known_feats), _p(unknown_feats), _p(idx), _p(weight), _p(wt), _p(bias),
                                           int(bool(relu)), _p(out), _stream()), "interp_gemm")
    return out


def mlp2_rows(x2d, w1t, b1, relu1, w2t, b2, relu2, ticket=None):
    """two pointwise layers on rows in one kernel: x2d (R,128) @ w1t (128,128) -> @ w2t (128,O2 <= 64) -> (R,O2), or None when
    the shape is not covered (the caller runs two GEMMs).  ticket: a zeroed int32 element (consumed) for the dynamic hand-out
    of row tiles; None = a fresh one.  ws3d extension."""
    dev = _dev(x2d, w1t, w2t)
    _f32(x2d, "x2d"); _f32(w1t, "w1t"); _f32(w2t, "w2t")
    R, K = x2d.shape
    O1, O2 = w1t.size(1), w2t.size(1)
    if (K != 128 or O1 != 128 or w1t.size(0) != 128 or w2t.size(0) != 128 or O2 > 64 or R % 32 or not x2d.is_contiguous() or
            not w1t.is_contiguous() or not w2t.is_contiguous()):
        return None
    out = torch.empty((R, O2), dtype=torch.float32, device=dev)
    if ticket is None:
        ticket = torch.zeros(1, dtype=torch.int32, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_mlp2_rows(R, K, O1, O2, _p(x2d), _p(w1t), _p(b1), int(bool(relu1)), _p(w2t), _p(b2), int(bool(relu2)),
                                         _p(out), _p(ticket), _stream()), "mlp2_rows")
    return out


def pool_nsample(x):
    """x (..., nsample) contiguous fp32 -> (max over the last axis (...), position of the maximum u8);
    F.max_pool2d(kernel=[1, nsample]) scan rule (first maximum, NaN propagates).  ws3d extension."""
    dev = _dev(x)
    _f32(x, "x")
    ns = x.size(-1)
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=dev)
    arg = torch.empty(x.shape[:-1], dtype=torch.uint8, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_pool_nsample(out.numel(), ns, _p(x), _p(out), _p(arg), _stream()), "pool_nsample")
    return out, arg


def pool_nsample_grad(grad_out, arg, nsample):
    """grad_out (...), arg (...) u8 -> grad_x (..., nsample): grad_out at arg, zero elsewhere.  ws3d extension."""
    dev = _dev(grad_out, arg)
    _f32(grad_out, "grad_out")
    if arg.dtype != torch.uint8 or not arg.is_contiguous():
        raise TypeError("arg must be a contiguous uint8 tensor")
    grad_x = torch.empty(tuple(grad_out.shape) + (nsample,), dtype=torch.float32, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_pool_nsample_grad(grad_out.numel(), nsample, _p(grad_out), _p(arg), _p(grad_x), _stream()),
              "pool_nsample_grad")
    return grad_x


SA_MLP3_SHAPES = {(16, 16, 32), (32, 32, 64)}


def sa_mlp3_pool(x_rows4, nsample, layers, out, col0=0):
    """x_rows4 (R*nsample, 4); layers = [(W^T (in,out), bias, relu)] * 3 -> out[:, col0:col0+c3] of the
    (R, >=c3) buffer `out`; returns False when there is no fused kernel for these widths."""
    (w1, b1, r1), (w2, b2, r2), (w3, b3, r3) = layers
    widths = (w1.size(1), w2.size(1), w3.size(1))
    if (x_rows4.size(1) != 4 or widths not in SA_MLP3_SHAPES or nsample not in (16, 32) or not (r1 and r2)
            or b1 is None or b2 is None or b3 is None):
        return False
    dev = _dev(x_rows4, out)
    view = out[:, col0:col0 + widths[2]]
    with _on(dev):
        check(_lib.load().ws3d_sa_mlp3_pool(x_rows4.size(0), nsample, widths[0], widths[1], widths[2], _p(x_rows4),
                                            _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), int(bool(r3)),
                                            view.data_ptr(), out.size(1), _stream()), "sa_mlp3_pool")
    return True


def sa_mlp3_pool_compact(xyz, new_xyz, feat1, pairs, layers, out, col0=0, limit=-1):
    """sa_mlp3_pool over the compact pairs of compact_pairs: rows [x_j - c, f_j] built from xyz (B,N,3), new_xyz (B,M,3) and the one
    feature channel feat1 (B,N,1) or (B,N); atomic max into out[:, col0:col0+c3], which must be ZERO on entry.  False when there
    is no kernel for these widths / flags."""
    (w1, b1, r1), (w2, b2, r2), (w3, b3, r3) = layers
    widths = (w1.size(1), w2.size(1), w3.size(1))
    if (widths not in ((16, 16, 32), (32, 32, 64)) or w1.size(0) != 4 or not (r1 and r2 and r3) or b1 is None or b2 is None or b3 is None or
            feat1.numel() != xyz.size(0) * xyz.size(1) or not feat1.is_contiguous()):
        return False
    rowc, rowsrc, total = pairs
    dev = _dev(xyz, new_xyz, feat1, rowc, out)
    _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz"); _f32(feat1, "feat1"); _f32(out, "out")
    view = out[:, col0:col0 + widths[2]]
    with _on(dev):
        check(_lib.load().ws3d_sa_mlp3_pool_compact(xyz.size(0), xyz.size(1), new_xyz.size(1), rowc.numel(), widths[0], widths[1], widths[2], _p(xyz),
                                                    _p(new_xyz), _p(feat1), _p(rowc), _p(rowsrc), _p(total), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3),
                                                    _p(b3), view.data_ptr(), out.size(1), int(limit), _stream()), "sa_mlp3_pool_compact")
    return True


def sa_mlp3_pool_lists(xyz, new_xyz, feat1, nbr, layers, out, col0=0, gate=None):
    """the first level's three layers + pool over ALL rows of the neighbour lists nbr (B, M, ns), rows [x_j - c, f_j] built inside
    (no grouped tensor), the pooled rows STORED into out[:, col0:col0+c3]: == query_and_group_nlc + sa_mlp3_pool, bit for bit.
    gate = (total, limit): runs iff *total > limit.  False when there is no kernel for these widths / flags."""
    (w1, b1, r1), (w2, b2, r2), (w3, b3, r3) = layers
    widths = (w1.size(1), w2.size(1), w3.size(1))
    ns = nbr.size(2)
    if (widths not in ((16, 16, 32), (32, 32, 64)) or w1.size(0) != 4 or ns not in (16, 32) or not (r1 and r2) or b1 is None or b2 is None or b3 is None or
            feat1.numel() != xyz.size(0) * xyz.size(1) or not feat1.is_contiguous() or nbr.numel() % 32):
        return False
    dev = _dev(xyz, new_xyz, feat1, nbr, out)
    _f32(xyz, "xyz"); _f32(new_xyz, "new_xyz"); _f32(feat1, "feat1"); _i32(nbr, "nbr"); _f32(out, "out")
    view = out[:, col0:col0 + widths[2]]
    with _on(dev):
        check(_lib.load().ws3d_sa_mlp3_pool_lists(xyz.size(0), xyz.size(1), new_xyz.size(1), ns, widths[0], widths[1], widths[2], _p(xyz), _p(new_xyz),
                                                  _p(feat1), _p(nbr), _p(w1), _p(b1), _p(w2), _p(b2), _p(w3), _p(b3), int(bool(r3)), view.data_ptr(),
                                                  out.size(1), *_gate(gate), _stream()), "sa_mlp3_pool_lists")
    return True


def three_nn_wrapper(b, n, m, unknown_tensor, known_tensor, dist2_tensor, idx_tensor, sorted_known=None):
    """interpolate.cpp:14-23 (+ optional x-binned copy of `known` from sort_points_x: same result)"""
    dev = _dev(unknown_tensor, known_tensor, dist2_tensor, idx_tensor)
    _f32(unknown_tensor, "unknown"); _f32(known_tensor, "known"); _i32(idx_tensor, "idx")
    with _on(dev):
        check(_lib.load().ws3d_three_nn(b, n, m, _p(unknown_tensor), _p(known_tensor), _p(dist2_tensor),
                                         _p(idx_tensor), _p(sorted_known) if sorted_known is not None else None,
                                         _stream()), "three_nn")


def three_interpolate_wrapper(b, c, m, n, points_tensor, idx_tensor, weight_tensor, out_tensor):
    """interpolate.cpp:26-39"""
    dev = _dev(points_tensor, idx_tensor, weight_tensor, out_tensor)
    _f32(points_tensor, "points"); _i32(idx_tensor, "idx"); _f32(weight_tensor, "weight")
    with _on(dev):
        check(_lib.load().ws3d_three_interpolate(b, c, m, n, _p(points_tensor), _p(idx_tensor),
                                                  _p(weight_tensor), _p(out_tensor), _stream()),
              "three_interpolate")


def three_interpolate_grad_wrapper(b, c, n, m, grad_out_tensor, idx_tensor, weight_tensor, grad_points_tensor):
    """interpolate.cpp:41-53"""
    dev = _dev(grad_out_tensor, idx_tensor, weight_tensor, grad_points_tensor)
    _f32(grad_out_tensor, "grad_out"); _i32(idx_tensor, "idx"); _f32(weight_tensor, "weight")
    with _on(dev):
        check(_lib.load().ws3d_three_interpolate_grad(b, c, n, m, _p(grad_out_tensor), _p(idx_tensor),
                                                       _p(weight_tensor), _p(grad_points_tensor), _stream()),
              "three_interpolate_grad")


def group_points_grad_det(b, c, n, npoints, nsample, grad_out_tensor, idx_tensor, grad_points_tensor):
    """deterministic group_points_grad / gather_points_grad (nsample=1): fixed summation order
    (ascending slot), bit-identical run to run and to the sequential CPU loop.  ws3d extension."""
    dev = _dev(grad_out_tensor, idx_tensor, grad_points_tensor)
    _f32(grad_out_tensor, "grad_out"); _i32(idx_tensor, "idx"); _f32(grad_points_tensor, "grad_points")
    lib = _lib.load()
    nbytes = lib.ws3d_scatter_workspace_bytes(b, c, n, npoints * nsample)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
    with _on(dev):
        check(lib.ws3d_group_points_grad_det(b, c, n, npoints, nsample, _p(grad_out_tensor), _p(idx_tensor),
                                             _p(grad_points_tensor), _p(ws), nbytes, _stream()), "group_points_grad_det")
    return 1


def three_interpolate_grad_det(b, c, n, m, grad_out_tensor, idx_tensor, weight_tensor, grad_points_tensor):
    """deterministic three_interpolate_grad (see group_points_grad_det).  ws3d extension."""
    dev = _dev(grad_out_tensor, idx_tensor, weight_tensor, grad_points_tensor)
    _f32(grad_out_tensor, "grad_out"); _i32(idx_tensor, "idx"); _f32(weight_tensor, "weight")
    lib = _lib.load()
    nbytes = lib.ws3d_scatter_workspace_bytes(b, c, m, n * 3)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
    with _on(dev):
        check(lib.ws3d_three_interpolate_grad_det(b, c, n, m, _p(grad_out_tensor), _p(idx_tensor), _p(weight_tensor),
                                                  _p(grad_points_tensor), _p(ws), nbytes, _stream()),
              "three_interpolate_grad_det")
    return 1


def topk_sorted(scores, k, spread=None, sigmoid=False):
    """scores (B,N) float32, N <= 16384 -> (values (B,k) descending, indices (B,k) int64); ties in
    ascending index order (ws3d extension).  sigmoid: `scores` holds logits, the sort runs over torch.sigmoid's fp32 expression of
    them (evaluated in the kernel: same values and order as topk_sorted(torch.sigmoid(scores), k), one launch fewer).  spread: the sort of a scene over several workgroups (ws3d_topk_sorted_ws: 107 -> ~25 us
    at 16384 scores, for ~1.5x the CU time); None: when the call is not being captured into a hipGraph -- a lone batch leaves the
    chip idle beside a one-workgroup sort, the graphs of a full pipeline do not (measured: -2.3 % latency, -1.1 % throughput)."""
    dev = _dev(scores)
    _f32(scores, "scores")
    B, N = scores.shape
    vals = torch.empty((B, k), dtype=torch.float32, device=dev)
    idx = torch.empty((B, k), dtype=torch.int64, device=dev)
    lib = _lib.load()
    if spread is None:
        spread = TOPK_SEGMENTS and not torch.cuda.is_current_stream_capturing()
    need = int(lib.ws3d_topk_workspace_bytes(B, N)) if spread else 0
    with _on(dev):
        if need:        # the sort of a scene spread over its CUs: 2048-key segments side by side, then ranked against each other
            ws = torch.empty(need // 8, dtype=torch.int64, device=dev)
            check((lib.ws3d_topk_sorted_sigmoid_ws if sigmoid else lib.ws3d_topk_sorted_ws)(B, N, k, _p(scores), _p(vals), _p(idx), _p(ws), need,
                                                                                            _stream()), "topk_sorted")
        else:
            check((lib.ws3d_topk_sorted_sigmoid if sigmoid else lib.ws3d_topk_sorted)(B, N, k, _p(scores), _p(vals), _p(idx), _stream()), "topk_sorted")
    return vals, idx


def decode_center_boxes(xyz, rpn_reg, loc_scope, loc_bin_size, mean_size):
    """xyz (B,N,3), rpn_reg (B,N,4*bins) -> proposal rows (B,N,7) (ws3d extension, see ws3d_ops.h)"""
    dev = _dev(xyz, rpn_reg)
    _f32(xyz, "xyz"); _f32(rpn_reg, "rpn_reg")
    B, N = xyz.size(0), xyz.size(1)
    bins = int(loc_scope / loc_bin_size) * 2
    if rpn_reg.size(2) != 4 * bins or not rpn_reg.is_contiguous() or not xyz.is_contiguous():
        raise ValueError("decode_center_boxes: rpn_reg must be contiguous (B,N,%d)" % (4 * bins))
    boxes = torch.empty((B, N, 7), dtype=torch.float32, device=dev)
    h, w, l = mean_size
    with _on(dev):
        check(_lib.load().ws3d_decode_center_boxes(B, N, bins, float(loc_scope), float(loc_bin_size), float(h), float(w),
                                                   float(l), _p(xyz), _p(rpn_reg), _p(boxes), _stream()), "decode_center_boxes")
    return boxes


def decode_gather_boxes_bev(xyz, rpn_reg, order, loc_scope, loc_bin_size, mean_size):
    """xyz (B,N,3), rpn_reg (B,N,4*bins), order (B,top) int64 -> (proposal rows of the points `order` names (B,top,7), their BEV
    rectangles (B,top,5)): decode_center_boxes + gather_boxes_bev in one launch, only `top` rows decoded (ws3d extension)"""
    dev = _dev(xyz, rpn_reg, order)
    _f32(xyz, "xyz"); _f32(rpn_reg, "rpn_reg")
    B, N, top = xyz.size(0), xyz.size(1), order.size(1)
    bins = int(loc_scope / loc_bin_size) * 2
    if rpn_reg.size(2) != 4 * bins or not rpn_reg.is_contiguous() or not xyz.is_contiguous():
        raise ValueError("decode_gather_boxes_bev: rpn_reg must be contiguous (B,N,%d)" % (4 * bins))
    if order.dtype != torch.int64 or not order.is_contiguous():
        raise ValueError("decode_gather_boxes_bev: order must be a contiguous int64 tensor")
    out = torch.empty((B, top, 7), dtype=torch.float32, device=dev)
    bev = torch.empty((B, top, 5), dtype=torch.float32, device=dev)
    h, w, l = mean_size
    with _on(dev):
        check(_lib.load().ws3d_decode_gather_boxes_bev(B, N, top, bins, float(loc_scope), float(loc_bin_size), float(h), float(w), float(l),
                                                       _p(xyz), _p(rpn_reg), _p(order), _p(out), _p(bev), _stream()), "decode_gather_boxes_bev")
    return out, bev


def split_points_clear(pointcloud, clear=None):
    """pointcloud (B,N,C>=3) contiguous -> (xyz (B,N,3), feats (B,N,C-3) or None), and `clear` (a contiguous tensor whose byte size
    is a multiple of 16, or None) zeroed -- one launch (ws3d_split_points_clear: the step's prologue).  ws3d extension."""
    dev = _dev(pointcloud) if clear is None else _dev(pointcloud, clear)
    _f32(pointcloud, "pointcloud")
    if not pointcloud.is_contiguous() or pointcloud.dim() != 3 or pointcloud.size(2) < 3:
        raise ValueError("split_points_clear: pointcloud must be a contiguous (B,N,C>=3) tensor")
    B, N, Cc = pointcloud.shape
    xyz = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
    feats = torch.empty((B, N, Cc - 3), dtype=torch.float32, device=dev) if Cc > 3 else None
    nbytes = 0 if clear is None else clear.numel() * clear.element_size()
    if clear is not None and (not clear.is_contiguous() or nbytes % 16 or clear.data_ptr() % 16):
        raise ValueError("split_points_clear: `clear` must be contiguous, 16-byte aligned and a multiple of 16 bytes long")
    with _on(dev):
        check(_lib.load().ws3d_split_points_clear(B * N, Cc, _p(pointcloud), _p(xyz), _p(feats), _p(clear), nbytes, _stream()), "split_points_clear")
    return xyz, feats


def gather_boxes_bev(box, order):
    """box (B,N,7), order (B,top) int64 -> (box[order] (B,top,7), its BEV rectangles (B,top,5)) in one launch (ws3d extension)"""
    dev = _dev(box, order)
    _f32(box, "box")
    if order.dtype != torch.int64 or not order.is_contiguous() or not box.is_contiguous():
        raise ValueError("gather_boxes_bev: box must be contiguous fp32 and order contiguous int64")
    B, N, top = box.size(0), box.size(1), order.size(1)
    box_sorted = torch.empty((B, top, 7), dtype=torch.float32, device=dev)
    bev = torch.empty((B, top, 5), dtype=torch.float32, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_gather_boxes_bev(B, N, top, _p(box), _p(order), _p(box_sorted), _p(bev), _stream()), "gather_boxes_bev")
    return box_sorted, bev


def select_proposals(box_sorted, scores_sorted, keep, num, k, extra_width=None, packed=False):
    """first min(num, k) NMS survivors of score-sorted boxes -> (boxes (B,k,7), scores (B,k), count (B,) int64,
    boxes enlarged by extra_width for RoI pooling or None), zero padded; one launch (ws3d extension).  packed: a fifth result, the
    (B,k,8) rows box + score (ws3d_amd.dist.pack_proposals' tensor) written by the same launch.  packed = a float32 tensor
    (>= B, k * 8 + 1): the send buffer of ws3d_amd.dist.ProposalExchange -- the rows and, behind each scene's rows, its count go straight
    into it (ws3d_select_proposals_send) and the fifth result is a (B,k,8) VIEW of that buffer: nothing is allocated or packed later"""
    dev = _dev(box_sorted, scores_sorted, keep, num)
    _f32(box_sorted, "box_sorted"); _f32(scores_sorted, "scores")
    if keep.dtype != torch.int64 or num.dtype != torch.int32:
        raise ValueError("select_proposals: keep must be int64 and num int32 (ws3d_nms_batched outputs)")
    B, top = box_sorted.size(0), box_sorted.size(1)
    boxes = torch.empty((B, k, 7), dtype=torch.float32, device=dev)
    scores = torch.empty((B, k), dtype=torch.float32, device=dev)
    count = torch.empty((B,), dtype=torch.int64, device=dev)
    pooled = torch.empty((B, k, 7), dtype=torch.float32, device=dev) if extra_width is not None else None
    send = packed if torch.is_tensor(packed) else None
    if send is not None:
        if send.dtype != torch.float32 or send.device != dev or send.dim() != 2 or send.size(0) < B or send.size(1) != k * 8 + 1 or not send.is_contiguous():
            raise ValueError("select_proposals: the send buffer must be a contiguous float32 (>= %d, %d) tensor on %s, got %s %s"
                             % (B, k * 8 + 1, dev, tuple(send.shape), send.dtype))
        with _on(dev):
            check(_lib.load().ws3d_select_proposals_send(B, top, keep.size(1), k, _p(box_sorted.contiguous()), _p(scores_sorted.contiguous()),
                                                         _p(keep), _p(num), float(extra_width or 0.0), _p(boxes), _p(scores), _p(count), _p(pooled),
                                                         _p(send), k * 8 + 1, _stream()), "select_proposals")
        return boxes, scores, count, pooled, send[:B, :k * 8].unflatten(1, (k, 8))
    pk = torch.empty((B, k, 8), dtype=torch.float32, device=dev) if packed else None
    with _on(dev):
        if packed:
            check(_lib.load().ws3d_select_proposals_packed(B, top, keep.size(1), k, _p(box_sorted.contiguous()), _p(scores_sorted.contiguous()),
                                                           _p(keep), _p(num), float(extra_width or 0.0), _p(boxes), _p(scores), _p(count), _p(pooled),
                                                           _p(pk), _stream()), "select_proposals")
        else:
            check(_lib.load().ws3d_select_proposals(B, top, keep.size(1), k, _p(box_sorted.contiguous()), _p(scores_sorted.contiguous()),
                                                    _p(keep), _p(num), float(extra_width or 0.0), _p(boxes), _p(scores), _p(count), _p(pooled),
                                                    _stream()), "select_proposals")
    return (boxes, scores, count, pooled, pk) if packed else (boxes, scores, count, pooled)


def bias_act_inplace(y, bias, relu=True):
    """y (B,O,L...) contiguous: y = relu?(y + bias[o]) in place, one pass (ws3d extension)"""
    dev = _dev(y, bias)
    _f32(y, "y"); _f32(bias, "bias")
    B, O = y.size(0), y.size(1)
    L = y.numel() // max(B * O, 1)
    with _on(dev):
        check(_lib.load().ws3d_bias_act_inplace(B, O, L, int(bool(relu)), _p(y), _p(bias), _stream()), "bias_act")
    return y


def rowmax_bias_act(y, bias=None, relu=True):
    """y (B,O,M,S) contiguous -> (B,O,M): relu?(max over S + bias[o]) in one pass (ws3d extension)"""
    dev = _dev(y) if bias is None else _dev(y, bias)
    _f32(y, "y")
    if bias is not None:
        _f32(bias, "bias")
    B, O, M, S = y.shape
    out = torch.empty((B, O, M), dtype=torch.float32, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_rowmax_bias_act(B, O, M, S, int(bool(relu)), _p(y), _p(bias) if bias is not None else None,
                                               _p(out), _stream()), "rowmax_bias_act")
    return out


# ------------------------------------------------------------------ iou3d_cuda
def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    """iou3d.cpp:31-50"""
    dev = _dev(boxes_a, boxes_b, ans_overlap)
    _f32(boxes_a, "boxes_a"); _f32(boxes_b, "boxes_b"); _f32(ans_overlap, "ans")
    with _on(dev):
        check(_lib.load().ws3d_boxes_overlap_bev(boxes_a.size(0), _p(boxes_a), boxes_b.size(0), _p(boxes_b),
                                                  _p(ans_overlap), _stream()), "boxes_overlap_bev")
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    """iou3d.cpp:52-71"""
    dev = _dev(boxes_a, boxes_b, ans_iou)
    _f32(boxes_a, "boxes_a"); _f32(boxes_b, "boxes_b"); _f32(ans_iou, "ans")
    with _on(dev):
        check(_lib.load().ws3d_boxes_iou_bev(boxes_a.size(0), _p(boxes_a), boxes_b.size(0), _p(boxes_b),
                                              _p(ans_iou), _stream()), "boxes_iou_bev")
    return 1


def nms_device(boxes, thresh, normal=False, max_keep=0):
    """Device-resident NMS: boxes (n,5) score-sorted -> (keep int64 (n,) DEVICE, num int32 (1,)
    DEVICE).  No host synchronisation (ws3d extension used by the Stage-1 pipeline).
    max_keep > 0: stop after that many survivors (== the reference's keep[:max_keep])."""
    dev = _dev(boxes)
    _f32(boxes, "boxes")
    n = boxes.size(0)
    lib = _lib.load()
    ws_bytes = lib.ws3d_nms_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    with _on(dev):
        check(lib.ws3d_nms(n, _p(boxes), float(thresh), int(bool(normal)), int(max_keep), _p(ws), ws_bytes,
                           _p(keep), _p(num), _stream()), "nms")
    return keep, num


def nms_device_batched(boxes, thresh, normal=False, max_keep=0):
    """boxes (B,n,5), every scene score-sorted -> keep (B,n) int64, num (B,) int32, both on the
    device; one mask launch + one sweep launch for the whole batch, no host synchronisation."""
    dev = _dev(boxes)
    _f32(boxes, "boxes")
    B, n = boxes.size(0), boxes.size(1)
    lib = _lib.load()
    ws_bytes = lib.ws3d_nms_workspace_bytes(n) * max(B, 1)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    keep = torch.empty((B, max(n, 1)), dtype=torch.int64, device=dev)
    num = torch.empty(B, dtype=torch.int32, device=dev)          # (every scene's count is written by the first sweep, or cleared when n == 0)
    with _on(dev):
        check(lib.ws3d_nms_batched(B, n, _p(boxes), float(thresh), int(bool(normal)), int(max_keep), _p(ws),
                                   ws_bytes, _p(keep), _p(num), _stream()), "nms_batched")
    return keep, num


def radius_nms_device_batched(centers, radius, max_keep=0):
    """centers (B,n,2) = (x,z), every scene sorted by descending score -> keep (B,n) int64,
    num (B,) int32 on the device (ws3d extension)."""
    dev = _dev(centers)
    _f32(centers, "centers")
    B, n = centers.size(0), centers.size(1)
    lib = _lib.load()
    ws_bytes = lib.ws3d_nms_workspace_bytes(n) * max(B, 1)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    keep = torch.empty((B, max(n, 1)), dtype=torch.int64, device=dev)
    num = torch.zeros(B, dtype=torch.int32, device=dev)
    with _on(dev):
        check(lib.ws3d_radius_nms_batched(B, n, _p(centers), float(radius), int(max_keep), _p(ws), ws_bytes,
                                          _p(keep), _p(num), _stream()), "radius_nms")
    return keep, num


def _nms_into_cpu_keep(boxes, keep, thresh, normal):
    if keep.is_cuda or keep.dtype != torch.int64 or not keep.is_contiguous():
        raise Ws3dError("keep must be a contiguous CPU int64 tensor (iou3d.cpp:73-82)")
    keep_dev, num = nms_device(boxes, thresh, normal)
    n_keep = int(num.item())  # the reference's signature returns the count to the host
    keep[:n_keep] = keep_dev[:n_keep].cpu()
    return n_keep


def nms_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d.cpp:73-120"""
    return _nms_into_cpu_keep(boxes, keep, nms_overlap_thresh, False)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d.cpp:123-170"""
    return _nms_into_cpu_keep(boxes, keep, nms_overlap_thresh, True)


def nms_mask(boxes, thresh, normal=False, full_grid=False):
    """K12/K13 mask only -> (n, ceil(n/64)) int64 tensor holding the uint64 words."""
    dev = _dev(boxes)
    _f32(boxes, "boxes")
    n = boxes.size(0)
    mask = torch.zeros((n, (n + 63) // 64), dtype=torch.int64, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_nms_mask(n, _p(boxes), float(thresh), int(bool(normal)), int(bool(full_grid)),
                                         _p(mask), _stream()), "nms_mask")
    return mask


# ------------------------------------------------------------------ roipool3d_cuda
def _roipool3d(fill, xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag, pts_idx):
    dev = _dev(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag, pts_idx)
    _f32(xyz, "xyz"); _f32(boxes3d, "boxes3d"); _f32(pts_feature, "pts_feature")
    _f32(pooled_features, "pooled_features"); _i32(pooled_empty_flag, "pooled_empty_flag")
    lib = _lib.load()
    B, N = xyz.size(0), xyz.size(1)
    with _on(dev):
        # large scenes: a scratch buffer for the binned copy of the scene (ws3d_roipool3d_ws; from the caching allocator: stream-ordered,
        # capturable); small ones: 0 bytes, the scanning kernels
        nbytes = lib.ws3d_roipool3d_workspace_bytes(B, N)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
        check(lib.ws3d_roipool3d_ws(B, N, boxes3d.size(1), pts_feature.size(2), pooled_features.size(2), _p(xyz), _p(boxes3d),
                                    _p(pts_feature), _p(pooled_features), _p(pooled_empty_flag), _p(pts_idx), int(fill),
                                    _p(ws), nbytes, _stream()), "roipool3d")
    return 1


def roipool3d_forward(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag, pts_idx=None):
    """roipool3d.cpp:48-79 (forward) == :15-44 (forward_slow)"""
    return _roipool3d(0, xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag, pts_idx)


def roipool3d_forward_fill(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag, pts_idx=None):
    """roipool3d_forward that writes every output element (outputs need not be pre-zeroed).  ws3d extension."""
    return _roipool3d(1, xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag, pts_idx)


def pts_in_boxes3d_device(pts, boxes3d):
    """device twin of pts_in_boxes3d_cpu: (N,3),(M,7) -> (M,N) int64 on the device"""
    dev = _dev(pts, boxes3d)
    _f32(pts, "pts"); _f32(boxes3d, "boxes3d")
    flag = torch.empty((boxes3d.size(0), pts.size(0)), dtype=torch.int64, device=dev)
    with _on(dev):
        check(_lib.load().ws3d_pts_in_boxes3d(boxes3d.size(0), pts.size(0), _p(pts), _p(boxes3d), _p(flag),
                                               _stream()), "pts_in_boxes3d")
    return flag


def _default_device():
    if not torch.cuda.is_available():
        raise Ws3dError("no HIP device: the reference's *_cpu entry points are served by the MI355X "
                        "kernels here (H2D -> kernel -> D2H); ws3d_amd has no CPU implementation")
    return torch.device("cuda", torch.cuda.current_device())


def pts_in_boxes3d_cpu(pts_flag, pts, boxes3d):
    """roipool3d.cpp:97-124 -- CPU tensors in/out, computed on the device."""
    dev = _default_device()
    flag = pts_in_boxes3d_device(pts.float().contiguous().to(dev), boxes3d.float().contiguous().to(dev))
    pts_flag.copy_(flag.cpu())
    return 1


def roipool3d_cpu(pts, boxes3d, pts_feature, pooled_pts, pooled_features, pooled_empty_flag):
    """roipool3d.cpp:127-195 -- CPU tensors in/out, computed on the device."""
    dev = _default_device()
    S, C_ = pooled_pts.size(1), pts_feature.size(1)
    M = boxes3d.size(0)
    out = torch.zeros((1, M, S, 3 + C_), dtype=torch.float32, device=dev)
    empty = torch.zeros((1, M), dtype=torch.int32, device=dev)
    roipool3d_forward(pts.float().contiguous().to(dev)[None], boxes3d.float().contiguous().to(dev)[None],
                      pts_feature.float().contiguous().to(dev)[None], out, empty)
    out = out[0].cpu()
    pooled_pts.copy_(out[:, :, :3])
    pooled_features.copy_(out[:, :, 3:])
    pooled_empty_flag.copy_(empty[0].cpu().to(pooled_empty_flag.dtype))
    return 1


def _namespace(name, **fns):
    m = types.ModuleType(name)
    m.__dict__.update(fns)
    m.__doc__ = f"ws3d_amd drop-in for the reference's `{name}` pybind11 extension"
    return m


pointnet2_cuda = _namespace(
    "pointnet2_cuda",
    furthest_point_sampling_wrapper=furthest_point_sampling_wrapper,
    gather_points_wrapper=gather_points_wrapper,
    gather_points_grad_wrapper=gather_points_grad_wrapper,
    ball_query_wrapper=ball_query_wrapper,
    group_points_wrapper=group_points_wrapper,
    group_points_grad_wrapper=group_points_grad_wrapper,
    three_nn_wrapper=three_nn_wrapper,
    three_interpolate_wrapper=three_interpolate_wrapper,
    three_interpolate_grad_wrapper=three_interpolate_grad_wrapper,
)
iou3d_cuda = _namespace(
    "iou3d_cuda",
    boxes_overlap_bev_gpu=boxes_overlap_bev_gpu,
    boxes_iou_bev_gpu=boxes_iou_bev_gpu,
    nms_gpu=nms_gpu,
    nms_normal_gpu=nms_normal_gpu,
)
roipool3d_cuda = _namespace(
    "roipool3d_cuda",
    forward=roipool3d_forward,
    forward_slow=roipool3d_forward,
    pts_in_boxes3d_cpu=pts_in_boxes3d_cpu,
    roipool3d_cpu=roipool3d_cpu,
)


def install() -> None:
    """Register the three namespaces under the reference's module names."""
    sys.modules["pointnet2_cuda"] = pointnet2_cuda
    sys.modules["iou3d_cuda"] = iou3d_cuda
    sys.modules["roipool3d_cuda"] = roipool3d_cuda
