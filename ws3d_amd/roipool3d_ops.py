"""roipool3d operator wrappers with the reference's names and signatures
(lib/utils/roipool3d/roipool3d_utils.py:7-140) on the MI355X kernel."""
from __future__ import annotations

import numpy as np
import torch

from . import compat as _C
from . import kitti_utils


def _pool(pts, pts_feature, pooled_boxes3d, sampled_pt_num):
    batch_size, boxes_num, feature_len = pts.shape[0], pooled_boxes3d.shape[1], pts_feature.shape[2]
    # the kernel writes every element (zeros for empty boxes): no 214 MB memset per 8 scenes at S=512, C=128
    pooled_features = torch.empty((batch_size, boxes_num, sampled_pt_num, 3 + feature_len),
                                  dtype=torch.float32, device=pts.device)
    pooled_empty_flag = torch.empty((batch_size, boxes_num), dtype=torch.int32, device=pts.device)
    _C.roipool3d_forward_fill(pts.contiguous(), pooled_boxes3d.contiguous(), pts_feature.contiguous(),
                              pooled_features, pooled_empty_flag)
    return pooled_features, pooled_empty_flag


def roipool3d_gpu(pts, pts_feature, boxes3d, pool_extra_width, sampled_pt_num=512, enlarged=None):
    """pts (B,N,3), pts_feature (B,N,C), boxes3d (B,M,7) -> pooled (B,M,S,3+C), empty (B,M)
    (roipool3d_utils.py:7-28): boxes are enlarged by pool_extra_width first.  enlarged (ws3d extension): the already
    enlarged boxes (stage1.proposals_from_rpn(with_pool_boxes=True) emits them with the proposals)."""
    batch_size = pts.shape[0]
    if enlarged is not None:
        return _pool(pts, pts_feature, enlarged, sampled_pt_num)
    pooled_boxes3d = kitti_utils.enlarge_box3d(boxes3d.view(-1, 7), pool_extra_width).view(batch_size, -1, 7)
    return _pool(pts, pts_feature, pooled_boxes3d, sampled_pt_num)


def roipool3dball_gpu(pts, pts_feature, boxes3d, pool_extra_width, sampled_pt_num=512):
    """6 m cube around the centre instead of the box (roipool3d_utils.py:31-59)"""
    rng = boxes3d.new_zeros(boxes3d.shape[0], boxes3d.shape[1], 7)
    rng[..., 0] = boxes3d[..., 0]
    rng[..., 2] = boxes3d[..., 2]
    rng[..., 3:6] = 6.0
    return _pool(pts, pts_feature, rng, sampled_pt_num)


def pts_in_boxes3d_cpu(pts, boxes3d):
    """pts (N,3), boxes3d (M,7) -> list of M bool masks (N) (roipool3d_utils.py:62-80)"""
    if pts.is_cuda:
        raise NotImplementedError
    pts_flag = torch.zeros((boxes3d.size(0), pts.size(0)), dtype=torch.int64)
    _C.pts_in_boxes3d_cpu(pts_flag, pts.float().contiguous(), boxes3d.float().contiguous())
    return [pts_flag[k] > 0 for k in range(boxes3d.shape[0])]


def roipool_pc_cpu(pts, pts_feature, boxes3d, sampled_pt_num):
    """single scene, CPU tensors in/out (roipool3d_utils.py:83-100)"""
    pts = pts.cpu().float().contiguous()
    pts_feature = pts_feature.cpu().float().contiguous()
    boxes3d = boxes3d.cpu().float().contiguous()
    assert pts.shape[0] == pts_feature.shape[0] and pts.shape[1] == 3, '%s %s' % (pts.shape, pts_feature.shape)
    pooled_pts = torch.zeros((boxes3d.shape[0], sampled_pt_num, 3), dtype=torch.float32)
    pooled_features = torch.zeros((boxes3d.shape[0], sampled_pt_num, pts_feature.shape[1]), dtype=torch.float32)
    pooled_empty_flag = torch.zeros(boxes3d.shape[0], dtype=torch.int64)
    _C.roipool3d_cpu(pts, boxes3d, pts_feature, pooled_pts, pooled_features, pooled_empty_flag)
    return pooled_pts, pooled_features, pooled_empty_flag


def _rotate_pc_along_y(pc, rot_angle):
    """lib/utils/kitti_utils.py rotate_pc_along_y: rotate (x,z) by rot_angle"""
    cosval, sinval = np.cos(rot_angle), np.sin(rot_angle)
    rotmat = np.array([[cosval, -sinval], [sinval, cosval]])
    pc[:, [0, 2]] = np.dot(pc[:, [0, 2]], np.transpose(rotmat))
    return pc


def roipool3d_cpu(boxes3d, pts, pts_feature, pts_extra_input, pool_extra_width, sampled_pt_num=512,
                  canonical_transform=True):
    """numpy in/out variant with the canonical transform (roipool3d_utils.py:103-140)"""
    pooled_boxes3d = kitti_utils.enlarge_box3d(boxes3d, pool_extra_width)
    pts_feature_all = np.concatenate((pts_extra_input, pts_feature), axis=1)
    pooled_pts, pooled_features, pooled_empty_flag = roipool_pc_cpu(
        torch.from_numpy(pts), torch.from_numpy(pts_feature_all), torch.from_numpy(pooled_boxes3d), sampled_pt_num)
    extra = pts_extra_input.shape[1]
    sampled_pts_input = torch.cat((pooled_pts, pooled_features[:, :, 0:extra]), dim=2).numpy()
    sampled_pts_feature = pooled_features[:, :, extra:].numpy()
    if canonical_transform:
        roi_ry = boxes3d[:, 6] % (2 * np.pi)
        roi_center = boxes3d[:, 0:3]
        sampled_pts_input[:, :, 0:3] = sampled_pts_input[:, :, 0:3] - roi_center[:, np.newaxis, :]
        for k in range(sampled_pts_input.shape[0]):
            sampled_pts_input[k] = _rotate_pc_along_y(sampled_pts_input[k], roi_ry[k])
        return sampled_pts_input, sampled_pts_feature
    return sampled_pts_input, sampled_pts_feature, pooled_empty_flag.numpy()
