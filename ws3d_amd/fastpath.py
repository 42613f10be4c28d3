"""Channels-last inference path for the Stage-1 network (same weights, same operators, other
memory layout).

The reference keeps point features as (B, C, N) and runs every SharedMLP as a 1x1 convolution.
On MI355X the natural layout is (B, N, C) -- "one row per point":
  * a SharedMLP layer is ONE row-major GEMM (B*L, C) x (C, O) over all scenes with bias + ReLU
    fused into the GEMM epilogue (``torch._addmm_activation`` -> hipBLASLt), instead of a batched
    GEMM followed by a separate bias/ReLU pass over the output (measured 20-35 % faster per layer,
    bit-identical results);
  * grouping, interpolation and RoI pooling gather contiguous rows (``ws3d_query_and_group_nlc``,
    ``ws3d_three_interpolate_nlc``; ``ws3d_roipool3d`` takes this layout already);
  * the heads' outputs (B, N, 1) / (B, N, 40) come out in the layout the reference transposes to.
Sampling, ball query and 3-NN are layout-independent and shared with the channels-first path.

``rpn_forward(model, pts)`` reads the folded weights of an eval-mode ``Stage1Net`` and returns the
same dict as ``Stage1Net.rpn_forward``; ``backbone_features`` is a (B, C, N) *view* of the (B, N, C)
result, so ``.transpose(1, 2).contiguous()`` on it costs nothing.  ``Stage1Net`` uses this path in
eval mode on the GPU unless ``stage1.CHANNELS_LAST_FASTPATH`` is cleared.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import compat as _C
from . import nn_blocks, pn2_ops


FUSED_SA_MLP = True   # ws3d_sa_mlp3_pool for the 4-channel SA level (clear to A/B against the GEMM chain)
FUSED_GEMM_POOL = True   # ws3d_gemm_pool: last layer of the other SA levels + pool on the matrix cores
FUSED_GATHER_GEMM = True  # ws3d_gather_gemm: grouping fused into the first layer's A operand (no grouped tensor in HBM)


def _row_weights(block):
    """(W^T (C,O) contiguous, bias (O,) or None, relu?) of a Conv+BN+ReLU block, cached on the block"""
    w, shift, act = block._folded()
    if act is not None and not isinstance(act, nn.ReLU):
        raise NotImplementedError("fast path supports ReLU / no activation only")
    cache = block.__dict__.get("_row_cache")
    if cache is None or cache[0] is not w:
        cache = (w, w.t().contiguous())
        if not torch.cuda.is_current_stream_capturing():
            block.__dict__["_row_cache"] = cache
    return cache[1], shift, act is not None


def _row_weights_xyz_last(block):
    """_row_weights with the three xyz rows moved behind the feature rows (the K order ws3d_gather_gemm reads), cached"""
    wt, bias, relu = _row_weights(block)
    cache = block.__dict__.get("_row_cache_xyz_last")
    if cache is None or cache[0] is not wt:
        cache = (wt, torch.cat((wt[3:], wt[:3]), dim=0).contiguous())
        if not torch.cuda.is_current_stream_capturing():
            block.__dict__["_row_cache_xyz_last"] = cache
    return cache[1], bias, relu


def _layer(x2d: torch.Tensor, block) -> torch.Tensor:
    wt, bias, relu = _row_weights(block)
    if bias is None:
        y = torch.mm(x2d, wt)
        return torch.relu_(y) if relu else y
    if relu:
        return torch._addmm_activation(bias, x2d, wt, use_gelu=False)
    return torch.addmm(bias, x2d, wt)


def _blocks(seq):
    """the Conv blocks of a SharedMLP / head Sequential (Dropout is the identity in eval mode)"""
    return [m for m in seq if isinstance(m, nn_blocks._ConvBlock)]


def mlp_rows(x2d: torch.Tensor, seq) -> torch.Tensor:
    for blk in _blocks(seq):
        x2d = _layer(x2d, blk)
    return x2d


def supported(model) -> bool:
    try:
        for m in model.modules():
            if isinstance(m, nn_blocks._ConvBlock):
                if not m._pointwise or not isinstance(getattr(m, "activation", None), (nn.ReLU, type(None))):
                    return False
        for sa in model.rpn.backbone_net.SA_modules:
            if sa.pool_method != "max_pool" or sa.npoint is None:
                return False
            for g, mlp in zip(sa.groupers, sa.mlps):
                if not isinstance(g, pn2_ops.QueryAndGroup) or any(b.conv.out_channels % 4 for b in _blocks(mlp)):
                    return False
        return True
    except AttributeError:
        return False


def sa_forward(sa, xyz: torch.Tensor, feats: torch.Tensor):
    """xyz (B,N,3), feats (B,N,C) or None -> new_xyz (B,M,3), new_feats (B,M,sum O)"""
    B = xyz.size(0)
    _, new_xyz = pn2_ops.furthest_point_sample_gather(xyz, sa.npoint)
    sorted_xyz = pn2_ops.sort_points_x(xyz)
    widths = [_blocks(mlp)[-1].conv.out_channels for mlp in sa.mlps]
    out = torch.empty((B * sa.npoint, sum(widths)), dtype=torch.float32, device=xyz.device)
    col = 0
    for grouper, mlp, width in zip(sa.groupers, sa.mlps, widths):
        blocks = _blocks(mlp)
        if (FUSED_GATHER_GEMM and feats is not None and feats.size(2) >= 16 and grouper.use_xyz and len(blocks) >= 2 and
                blocks[0].conv.out_channels % 64 == 0 and (B * sa.npoint * grouper.nsample) % 64 == 0):
            # neighbour lists only, then layer 1 gathers its own rows: the (rows, 3 + C) grouped tensor never exists
            nbr = torch.zeros((B, sa.npoint, grouper.nsample), dtype=torch.int32, device=xyz.device)
            _C.ball_query_wrapper(B, xyz.size(1), sa.npoint, grouper.radius, grouper.nsample, new_xyz, xyz, nbr, sorted_xyz)
            wt1, b1, r1 = _row_weights_xyz_last(blocks[0])
            y = _C.gather_gemm(feats, xyz, new_xyz, nbr, wt1, b1, r1)
            if y is not None:
                for blk in blocks[1:-1]:
                    y = _layer(y, blk)
                wt, bias, relu = _row_weights(blocks[-1])
                if not (FUSED_GEMM_POOL and _C.gemm_pool(y, wt, bias, relu, grouper.nsample, out, col)):
                    _C.rowmax_rows(_layer(y, blocks[-1]), grouper.nsample, out, col)
                col += width
                continue
        g = _C.query_and_group_nlc(grouper.radius, grouper.nsample, xyz, new_xyz, feats, grouper.use_xyz, sorted_xyz)
        rows = g.view(-1, g.size(3))
        # 4-channel level (dx,dy,dz,intensity): three layers + pool in one kernel, nothing but the
        # pooled rows leaves the chip; otherwise the GEMM chain + pool kernel
        if not (FUSED_SA_MLP and rows.size(1) == 4 and len(blocks) == 3 and
                _C.sa_mlp3_pool(rows, grouper.nsample, [_row_weights(b) for b in blocks], out, col)):
            y = rows
            for blk in blocks[:-1]:
                y = _layer(y, blk)                                     # (B*M*ns, C_k), bias + ReLU in the GEMM epilogue
            wt, bias, relu = _row_weights(blocks[-1])
            # last layer + pool in one matrix-core kernel (only the pooled rows reach HBM); shapes it does not
            # cover: GEMM with fused bias + ReLU, then the pool kernel into the column slice
            if not (FUSED_GEMM_POOL and _C.gemm_pool(y, wt, bias, relu, grouper.nsample, out, col)):
                _C.rowmax_rows(_layer(y, blocks[-1]), grouper.nsample, out, col)
        col += width
    return new_xyz, out.view(B, sa.npoint, -1)


def fp_forward(fp, unknown: torch.Tensor, known: torch.Tensor, unknown_feats, known_feats: torch.Tensor):
    """unknown (B,n,3), known (B,m,3), unknown_feats (B,n,C1) or None, known_feats (B,m,C2) -> (B,n,O)"""
    B, n = unknown.size(0), unknown.size(1)
    idx, weight = _C.three_nn_with_weights(unknown, known, pn2_ops.sort_points_xz(known))
    c2 = known_feats.size(2)
    c1 = 0 if unknown_feats is None else unknown_feats.size(2)
    cat = torch.empty((B, n, c2 + c1), dtype=torch.float32, device=unknown.device)
    _C.three_interpolate_nlc(known_feats, idx, weight, cat)         # left columns, any row stride (4-byte-aligned 16-byte stores)
    if c1:
        cat[:, :, c2:] = unknown_feats
    return mlp_rows(cat.view(B * n, c2 + c1), fp.mlp).view(B, n, -1)


@torch.no_grad()
def backbone_forward(net, pointcloud: torch.Tensor):
    """Pointnet2MSG.forward on channels-last tensors -> xyz (B,N,3), features (B,N,C)"""
    xyz = pointcloud[..., 0:3].contiguous()
    feats = pointcloud[..., 3:].contiguous() if pointcloud.size(-1) > 3 else None
    l_xyz, l_feats = [xyz], [feats]
    for sa in net.SA_modules:
        nx, nf = sa_forward(sa, l_xyz[-1], l_feats[-1])
        l_xyz.append(nx)
        l_feats.append(nf)
    for i in range(-1, -(len(net.FP_modules) + 1), -1):
        l_feats[i - 1] = fp_forward(net.FP_modules[i], l_xyz[i - 1], l_xyz[i], l_feats[i - 1], l_feats[i])
    return l_xyz[0], l_feats[0]


@torch.no_grad()
def rpn_forward(model, pts_input: torch.Tensor) -> dict:
    rpn = model.rpn
    xyz, feats = backbone_forward(rpn.backbone_net, pts_input)            # (B,N,3), (B,N,128)
    B, N, C = feats.shape
    rows = feats.view(B * N, C)
    rpn_cls = mlp_rows(rows, rpn.rpn_cls_layer).view(B, N, -1)
    rpn_reg = mlp_rows(rows, rpn.rpn_reg_layer).view(B, N, -1)
    return {"rpn_cls": rpn_cls, "rpn_reg": rpn_reg, "backbone_xyz": xyz,
            "backbone_features": feats.transpose(1, 2),                   # (B,C,N) view of the (B,N,C) tensor
            "backbone_features_nlc": feats}
