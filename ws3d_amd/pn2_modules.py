"""Set-abstraction / feature-propagation modules with the reference's class names and
constructor keywords (pointnet2_lib/pointnet2/pointnet2_modules.py:58-156), built on the
fused MI355X ops:

  SA layer = fused FPS+gather -> per scale [fused ball_query+group -> SharedMLP -> max over
             nsample] -> cat       (reference: 2 + 3*scales launches + 2*scales elementwise)
  FP layer = three_nn -> inverse-distance weights -> three_interpolate -> cat skip -> SharedMLP
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nn_blocks as pt_utils
from . import pn2_ops as pointnet2_utils


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None
        self.pool_method = 'max_pool'

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, new_xyz=None):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B,sum_k mlps[k][-1],npoint)
        (pointnet2_modules.py:19-55)."""
        if new_xyz is None and self.npoint is not None:
            _, new_xyz = pointnet2_utils.furthest_point_sample_gather(xyz, self.npoint)
        pooled = []
        # one x-sorted copy of the points serves every scale of the layer
        sorted_xyz = pointnet2_utils.sort_points_x(xyz) if self.npoint is not None else None
        for grouper, mlp in zip(self.groupers, self.mlps):
            if isinstance(grouper, pointnet2_utils.QueryAndGroup):
                grouped = grouper(xyz, new_xyz, features, sorted_xyz=sorted_xyz)   # (B, C, npoint, nsample)
            else:
                grouped = grouper(xyz, new_xyz, features)
            last = mlp[len(mlp) - 1]
            if (self.pool_method == 'max_pool' and hasattr(last, "forward_then_max") and last.fast_path_ok(grouped)
                    and isinstance(getattr(last, "activation", None), (nn.ReLU, type(None)))):
                # inference: pool BEFORE the last layer's bias+ReLU (exactly equal, nsample x less work)
                for i in range(len(mlp) - 1):
                    grouped = mlp[i](grouped)
                pooled.append(last.forward_then_max(grouped))
                continue
            grouped = mlp(grouped)                              # (B, mlp[-1], npoint, nsample)
            if self.pool_method == 'max_pool':
                if grouped.size(3) <= 255:
                    pooled.append(pointnet2_utils.pool_nsample(grouped))   # (B, mlp[-1], npoint), fwd + bwd kernels
                    continue
                grouped = F.max_pool2d(grouped, kernel_size=[1, grouped.size(3)])
            elif self.pool_method == 'avg_pool':
                grouped = F.avg_pool2d(grouped, kernel_size=[1, grouped.size(3)])
            else:
                raise NotImplementedError
            pooled.append(grouped.squeeze(-1))                  # (B, mlp[-1], npoint)
        return new_xyz, torch.cat(pooled, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Pointnet set abstraction layer with multiscale grouping (pointnet2_modules.py:58-92)"""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 bn: bool = True, use_xyz: bool = True, pool_method='max_pool', instance_norm=False):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                mlp_spec[0] += 3  # in place, like the reference (:88-89): callers see the widened spec
            self.mlps.append(pt_utils.SharedMLP(mlp_spec, bn=bn, instance_norm=instance_norm))
        self.pool_method = pool_method


class PointnetSAModule(PointnetSAModuleMSG):
    """Pointnet set abstraction layer (pointnet2_modules.py:95-113)"""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pool_method='max_pool', instance_norm=False):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn,
                         use_xyz=use_xyz, pool_method=pool_method, instance_norm=instance_norm)


class PointnetFPModule(nn.Module):
    """Propagates the features of one set to another (pointnet2_modules.py:116-156)"""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    def forward(self, unknown: torch.Tensor, known: torch.Tensor, unknow_feats: torch.Tensor,
                known_feats: torch.Tensor) -> torch.Tensor:
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n), known_feats (B,C2,m) -> (B,mlp[-1],n)"""
        if known is not None:
            # x-binned copy of the known set (None below 256 points): exact pruned 3-NN search
            dist, idx = pointnet2_utils.three_nn(unknown, known, pointnet2_utils.sort_points_xz(known))
            # inverse-distance weights of the three neighbours, normalised (same expression order as pointnet2_modules.py:140-142:
            # the quotient and the sum round identically)
            inv_d = 1.0 / (dist + 1e-8)
            w3 = inv_d / torch.sum(inv_d, dim=2, keepdim=True)
            upsampled = pointnet2_utils.three_interpolate(known_feats, idx, w3)
        else:           # no coordinates for the known set: one global feature vector, repeated for every point
            upsampled = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        stacked = upsampled if unknow_feats is None else torch.cat([upsampled, unknow_feats], dim=1)    # (B, C2 [+ C1], n)
        return self.mlp(stacked.unsqueeze(-1)).squeeze(-1)
