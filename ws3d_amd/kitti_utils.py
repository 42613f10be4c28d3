"""The two box helpers the op wrappers need (lib/utils/kitti_utils.py:134-160)."""
from __future__ import annotations

import numpy as np
import torch


def boxes3d_to_bev_torch(boxes3d: torch.Tensor) -> torch.Tensor:
    """(N,7) [x,y,z,h,w,l,ry] -> (N,5) [x1,y1,x2,y2,ry] in the x-z plane (kitti_utils.py:134-147)"""
    half_l, half_w = boxes3d[:, 5] / 2, boxes3d[:, 4] / 2
    cu, cv = boxes3d[:, 0], boxes3d[:, 2]
    return torch.stack((cu - half_l, cv - half_w, cu + half_l, cv + half_w, boxes3d[:, 6]), dim=1)


def enlarge_box3d(boxes3d, extra_width):
    """h,w,l += 2*extra_width; y_bottom += extra_width (kitti_utils.py:150-160)"""
    large = boxes3d.copy() if isinstance(boxes3d, np.ndarray) else boxes3d.clone()
    large[:, 3:6] += extra_width * 2
    large[:, 1] += extra_width
    return large
