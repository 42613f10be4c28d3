"""Operator wrappers with the reference's names and signatures
(pointnet2_lib/pointnet2/pointnet2_utils.py), backed by the MI355X HIP kernels.

Same ``torch.autograd.Function`` surface: ``furthest_point_sample, gather_operation,
ball_query, grouping_operation, three_nn, three_interpolate, QueryAndGroup, GroupAll``
(reference lines 36, 73, 228, 197, 105, 153, 231, 267).  Outputs are allocated here
(like the reference does with ``torch.cuda.*Tensor``) and filled by the library.

Two fused entry points that the reference composes in Python are added:
``furthest_point_sample_gather`` (FPS + gather, pointnet2_modules.py:30-35) and
``query_and_group`` (ball_query + 2x grouping + centre subtraction + cat,
pointnet2_utils.py:241-264).  ``QueryAndGroup`` uses the fused kernel.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from . import compat as _C

# Backward of gather / group / three_interpolate: True = sorted-segment accumulation with a fixed
# summation order (bit-reproducible, equals the sequential CPU loop); False = float atomicAdd
# scatter like the reference (order, and so the low bits, vary from run to run).
DETERMINISTIC_BACKWARD = True


def _new(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        """xyz (B,N,3), N > npoint -> (B,npoint) int32 indices (pointnet2_utils.py:12-31)."""
        _dense(xyz=xyz)
        B, N, _ = xyz.size()
        output = _new((B, npoint), torch.int32, xyz)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        _C.furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


def _dense(**tensors):
    """the C ABI takes plain row-major device pointers: refuse strided views by name"""
    for name, t in tensors.items():
        if t is not None and not t.is_contiguous():
            raise AssertionError(f"{name} must be contiguous (got strides {tuple(t.stride())} for shape {tuple(t.shape)})")


def furthest_point_sample_gather(xyz: torch.Tensor, npoint: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fused FPS + gather: (idx (B,npoint) int32, new_xyz (B,npoint,3)).  new_xyz equals
    gather_operation(xyz^T, idx)^T bit for bit (pure copies).  No temp buffer is needed for
    N <= 16384 (the running min-distance lives in registers)."""
    _dense(xyz=xyz)
    B, N, _ = xyz.size()
    idx = _new((B, npoint), torch.int32, xyz)
    new_xyz = _new((B, npoint, 3), torch.float32, xyz)
    temp = None
    if N > 16384:
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
    with torch.no_grad():
        _C.furthest_point_sampling_gather(B, N, npoint, xyz, temp, idx, new_xyz)
    return idx, new_xyz


def furthest_point_sample_gather_nested(xyz: torch.Tensor, npoint: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """furthest_point_sample_gather for a cloud that is the new_xyz of a previous sampling pass (the input of every
    set-abstraction level after the first): same (idx, new_xyz), bit for bit, without the npoint dependent steps -- the
    kernel verifies that greedy selection returns the leading npoint points and falls back to the plain kernel per scene
    where it does not (ties).  Clouds above 4096 points take the plain path."""
    _dense(xyz=xyz)
    B, N, _ = xyz.size()
    if N > 4096 or npoint > N:
        return furthest_point_sample_gather(xyz, npoint)
    idx = _new((B, npoint), torch.int32, xyz)
    new_xyz = _new((B, npoint, 3), torch.float32, xyz)
    with torch.no_grad():
        _C.furthest_point_sampling_nested(B, N, npoint, xyz, idx, new_xyz)
    return idx, new_xyz


def furthest_point_sample_gather_nested_chain(xyz: torch.Tensor, npoints) -> list:
    """[(idx, new_xyz), ...] of the levels below `xyz` (a cloud in sampling order; level l samples level l-1's new_xyz): the results of
    chained furthest_point_sample_gather_nested calls, bit for bit, in four launches instead of three per level."""
    _dense(xyz=xyz)
    npoints = [int(m) for m in npoints]
    if not npoints:
        return []
    if xyz.size(1) > 4096 or npoints[0] > xyz.size(1) or any(b_ > a_ for a_, b_ in zip(npoints, npoints[1:])) or len(npoints) > 5:
        out, cur = [], xyz
        for m in npoints:
            out.append(furthest_point_sample_gather_nested(cur, m))
            cur = out[-1][1]
        return out
    with torch.no_grad():
        return _C.furthest_point_sampling_nested_chain(xyz, npoints)


def sampling_plan(xyz: torch.Tensor, npoints) -> list:
    """The backbone's chain of furthest-point samplings, ``[new_xyz_1 (B,npoints[0],3), new_xyz_2, ...]``
    with level k sampled from level k-1.  It depends on the coordinates only -- not on any weight --
    so a trainer can run it for the NEXT batch on a side HIP stream while the current step computes
    (train_rpn.DevicePrefetcher) and hand it to the SA modules through their ``new_xyz`` argument
    (pointnet2_modules.py:19-29 accepts it).  Same result as sampling inside the modules."""
    plan, cur = [], xyz
    for m in npoints:
        _, cur = furthest_point_sample_gather(cur, int(m))
        plan.append(cur)
    return plan


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint) -> (B,C,npoint) (pointnet2_utils.py:41-60)."""
        _dense(features=features)
        _dense(idx=idx)
        B, npoint = idx.size()
        _, C, N = features.size()
        output = _new((B, C, npoint), torch.float32, features)
        _C.gather_points_wrapper(B, C, N, npoint, features, idx, output)
        ctx.gather_shape = (idx, C, N)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.gather_shape
        B, npoint = idx.size()
        if DETERMINISTIC_BACKWARD:
            grad_features = torch.empty((B, C, N), dtype=torch.float32, device=grad_out.device)
            _C.group_points_grad_det(B, C, N, npoint, 1, grad_out.contiguous(), idx, grad_features)
        else:
            grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
            _C.gather_points_grad_wrapper(B, C, N, npoint, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor, sorted_known=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """unknown (B,N,3), known (B,M,3) -> dist (B,N,3) L2 distance, idx (B,N,3)
        (pointnet2_utils.py:79-99; the kernel returns squared distances, sqrt is taken here).
        sorted_known: optional ``sort_points_x(known)`` -- identical result, pruned search."""
        _dense(unknown=unknown)
        _dense(known=known)
        B, N, _ = unknown.size()
        m = known.size(1)
        dist2 = _new((B, N, 3), torch.float32, unknown)
        idx = _new((B, N, 3), torch.int32, unknown)
        _C.three_nn_wrapper(B, N, m, unknown, known, dist2, idx, sorted_known)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        """features (B,C,M), idx/weight (B,n,3) -> (B,C,n) (pointnet2_utils.py:110-131)."""
        _dense(features=features)
        _dense(idx=idx)
        _dense(weight=weight)
        B, c, m = features.size()
        n = idx.size(1)
        ctx.interp_args = (idx, weight, m)
        output = _new((B, c, n), torch.float32, features)
        _C.three_interpolate_wrapper(B, c, m, n, features, idx, weight, output)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight, m = ctx.interp_args
        B, c, n = grad_out.size()
        if DETERMINISTIC_BACKWARD:
            grad_features = torch.empty((B, c, m), dtype=torch.float32, device=grad_out.device)
            _C.three_interpolate_grad_det(B, c, n, m, grad_out.contiguous(), idx, weight, grad_features)
        else:
            grad_features = torch.zeros((B, c, m), dtype=torch.float32, device=grad_out.device)
            _C.three_interpolate_grad_wrapper(B, c, n, m, grad_out.contiguous(), idx, weight, grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)
        (pointnet2_utils.py:158-177)."""
        _dense(features=features)
        _dense(idx=idx)
        B, nfeatures, nsample = idx.size()
        _, C, N = features.size()
        output = _new((B, C, nfeatures, nsample), torch.float32, features)
        _C.group_points_wrapper(B, C, N, nfeatures, nsample, features, idx, output)
        ctx.group_shape = (idx, N)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        idx, N = ctx.group_shape
        B, C, npoint, nsample = grad_out.size()
        if DETERMINISTIC_BACKWARD:
            grad_features = torch.empty((B, C, N), dtype=torch.float32, device=grad_out.device)
            _C.group_points_grad_det(B, C, N, npoint, nsample, grad_out.contiguous(), idx, grad_features)
        else:
            grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
            _C.group_points_grad_wrapper(B, C, N, npoint, nsample, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        """xyz (B,N,3), new_xyz (B,npoint,3) -> idx (B,npoint,nsample) int32
        (pointnet2_utils.py:202-222)."""
        _dense(new_xyz=new_xyz)
        _dense(xyz=xyz)
        B, N, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = torch.zeros((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
        _C.ball_query_wrapper(B, N, npoint, radius, nsample, new_xyz, xyz, idx, _C.sort_points_x(xyz))
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class _QueryAndGroupFused(Function):
    """Fused QueryAndGroup.forward; backward scatters the feature-channel gradient with
    the saved neighbour lists (what GroupingOperation.backward does in the reference)."""

    @staticmethod
    def forward(ctx, radius, nsample, use_xyz, xyz, new_xyz, features, sorted_xyz=None):
        B, N, _ = xyz.size()
        M = new_xyz.size(1)
        C = 0 if features is None else features.size(1)
        cx = 3 if use_xyz else 0
        idx = _new((B, M, nsample), torch.int32, xyz)
        out = _new((B, cx + C, M, nsample), torch.float32, xyz)
        if sorted_xyz is None:
            sorted_xyz = _C.sort_points_x(xyz)
        _C.query_and_group(B, N, M, C, radius, nsample, use_xyz, xyz, new_xyz, features, idx, out, sorted_xyz)
        ctx.saved = (idx, N, cx, C)
        return out, idx

    @staticmethod
    def backward(ctx, grad_out, _grad_idx=None):
        idx, N, cx, C = ctx.saved
        grad_features = None
        if C > 0 and ctx.needs_input_grad[5]:
            B, _, M, ns = grad_out.size()
            g = grad_out[:, cx:].contiguous()
            if DETERMINISTIC_BACKWARD:
                grad_features = torch.empty((B, C, N), dtype=torch.float32, device=grad_out.device)
                _C.group_points_grad_det(B, C, N, M, ns, g, idx, grad_features)
            else:
                grad_features = torch.zeros((B, C, N), dtype=torch.float32, device=grad_out.device)
                _C.group_points_grad_wrapper(B, C, N, M, ns, g, idx, grad_features)
        return None, None, None, None, None, grad_features, None


class PoolNsample(Function):
    """max over the nsample axis of the grouped activation (B,C,npoint,nsample) -> (B,C,npoint):
    the reference's ``F.max_pool2d(x, kernel_size=[1, nsample]).squeeze(-1)``
    (pointnet2_modules.py:50-54) as one coalesced pass each way, same values / argmax / NaN rule."""

    @staticmethod
    def forward(ctx, x: torch.Tensor) -> torch.Tensor:
        x = x.contiguous()
        out, arg = _C.pool_nsample(x)
        ctx.save_for_backward(arg)
        ctx.nsample = x.size(-1)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (arg,) = ctx.saved_tensors
        return _C.pool_nsample_grad(grad_out.contiguous(), arg, ctx.nsample)


pool_nsample = PoolNsample.apply


def sort_points_x(xyz: torch.Tensor, min_n=None):
    """Per-scene x-sorted copy of xyz (or None when it does not pay): pass it to several
    ``query_and_group`` / ``QueryAndGroup`` calls on the same xyz (multi-scale grouping) or to
    ``three_nn`` as the binned copy of the known set."""
    return _C.sort_points_x(xyz, min_n)


def sort_points_xz(xyz: torch.Tensor, min_n: int = 256):
    """Per-scene copy of xyz binned into an (x, z) grid (or None for small sets): the ``sorted_known``
    of ``three_nn`` -- same neighbours, the search visits ~20 candidates per query.  Not for the ball query."""
    return _C.sort_points_xz(xyz, min_n)


def query_and_group(radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor,
                    features: torch.Tensor = None, use_xyz: bool = True, return_idx: bool = False,
                    sorted_xyz: torch.Tensor = None):
    """One kernel for ball_query + grouping(xyz) - centre + grouping(features) + cat:
    (B, 3+C, npoint, nsample) with channel order [dx,dy,dz, features...]."""
    _dense(xyz=xyz, new_xyz=new_xyz, features=features)
    assert use_xyz or features is not None, "Cannot have not features and not use xyz as a feature!"
    out, idx = _QueryAndGroupFused.apply(radius, nsample, use_xyz, xyz, new_xyz, features, sorted_xyz)
    return (out, idx) if return_idx else out


class QueryAndGroup(nn.Module):
    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        """pointnet2_utils.py:231-239"""
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None,
                sorted_xyz: torch.Tensor = None):
        """xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N) -> (B,3+C,npoint,nsample)
        (pointnet2_utils.py:241-264).  sorted_xyz: optional ``sort_points_x(xyz)`` shared by the
        scales of an MSG layer."""
        return query_and_group(self.radius, self.nsample, xyz, new_xyz, features, self.use_xyz,
                               sorted_xyz=sorted_xyz)


class GroupAll(nn.Module):
    def __init__(self, use_xyz: bool = True):
        """pointnet2_utils.py:267-270"""
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: torch.Tensor = None):
        """-> (B, C+3, 1, N) (pointnet2_utils.py:272-290): pure views/cat, no kernel."""
        coords = xyz.transpose(1, 2).unsqueeze(2)                 # (B, 3, 1, N): the whole cloud is the one group
        if features is None:
            return coords
        feats = features.unsqueeze(2)
        return torch.cat([coords, feats], dim=1) if self.use_xyz else feats
