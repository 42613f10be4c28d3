"""iou3d operator wrappers with the reference's names and signatures
(lib/utils/iou3d/iou3d_utils.py:6-90) on the MI355X kernels."""
from __future__ import annotations

import torch

from . import compat as _C
from . import kitti_utils


def boxes_iou_bev(boxes_a, boxes_b):
    """(M,5),(N,5) -> IoU (M,N) (iou3d_utils.py:6-18)"""
    ans_iou = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    _C.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7),(M,7) [x,y,z,h,w,l,ry] -> (iou2d, iou3d), each (N,M) (iou3d_utils.py:21-56)"""
    boxes_a_bev = kitti_utils.boxes3d_to_bev_torch(boxes_a)
    boxes_b_bev = kitti_utils.boxes3d_to_bev_torch(boxes_b)
    overlaps_bev = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    _C.boxes_overlap_bev_gpu(boxes_a_bev.contiguous(), boxes_b_bev.contiguous(), overlaps_bev)

    # height overlap: y is the box bottom, y - h the top (camera y points down)
    a_min, a_max = (boxes_a[:, 1] - boxes_a[:, 3]).view(-1, 1), boxes_a[:, 1].view(-1, 1)
    b_min, b_max = (boxes_b[:, 1] - boxes_b[:, 3]).view(1, -1), boxes_b[:, 1].view(1, -1)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)

    s_a = (boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    s_b = (boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    iou2d = overlaps_bev / torch.clamp(s_a + s_b - overlaps_bev, min=1e-7)

    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    iou3d = overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-7)
    return iou2d, iou3d


def _nms(boxes, scores, thresh, normal):
    # stable=True: equal scores keep their input order on every device (the reference's
    # unstable sort, iou3d_utils.py:67, is only reproducible for distinct scores)
    order = scores.sort(dim=0, descending=True, stable=True)[1]
    keep, num = _C.nms_device(boxes[order].contiguous(), thresh, normal)
    return order[keep[:int(num.item())]].contiguous()


def nms_gpu(boxes, scores, thresh):
    """rotated NMS: boxes (N,5) [x1,y1,x2,y2,ry], scores (N) -> kept indices (iou3d_utils.py:59-73)"""
    return _nms(boxes, scores, thresh, False)


def nms_normal_gpu(boxes, scores, thresh):
    """axis-aligned NMS (iou3d_utils.py:76-90)"""
    return _nms(boxes, scores, thresh, True)


def nms_gpu_padded(boxes, scores, thresh, max_out, normal=False):
    """Device-only variant for the data-parallel pipeline: fixed-shape (max_out,) index tensor
    (padded with -1) + count tensor; no host synchronisation."""
    order = scores.sort(dim=0, descending=True, stable=True)[1]
    keep, num = _C.nms_device(boxes[order].contiguous(), thresh, normal, max_keep=max_out)
    n = boxes.shape[0]
    out = torch.full((max_out,), -1, dtype=torch.int64, device=boxes.device)
    take = min(max_out, n)
    pos = torch.arange(take, device=boxes.device)
    cnt = torch.clamp(num.to(torch.int64), max=max_out)
    valid = pos < cnt
    out[:take] = torch.where(valid, order[keep[:take].clamp(min=0, max=max(n - 1, 0))], out[:take])
    return out, cnt


def nms_gpu_padded_batched(boxes, scores, thresh, max_out, normal=False, scores_sorted=False):
    """boxes (B,n,5), scores (B,n) -> (idx (B,max_out) int64 padded with -1, count (B,) int64);
    idx refers to the input order of each scene.  Whole batch in one launch pair.
    scores_sorted=True: every row of `scores` is already non-increasing (e.g. it comes from
    ``topk(sorted=True)``); the stable descending sort would be the identity and is skipped."""
    B, n = scores.shape
    if scores_sorted:
        order = torch.arange(n, device=scores.device).unsqueeze(0).expand(B, n)
        sorted_boxes = boxes.contiguous()
    else:
        order = scores.sort(dim=1, descending=True, stable=True)[1]                  # (B,n)
        sorted_boxes = torch.gather(boxes, 1, order.unsqueeze(-1).expand(B, n, boxes.shape[2])).contiguous()
    keep, num = _C.nms_device_batched(sorted_boxes, thresh, normal, max_keep=max_out)
    take = min(max_out, n)
    cnt = torch.clamp(num.to(torch.int64), max=max_out)
    pos = torch.arange(take, device=boxes.device).unsqueeze(0)
    valid = pos < cnt.unsqueeze(1)
    picked = torch.gather(order, 1, keep[:, :take].clamp(min=0, max=max(n - 1, 0)))
    out = torch.full((B, max_out), -1, dtype=torch.int64, device=boxes.device)
    out[:, :take] = torch.where(valid, picked, out[:, :take])
    return out, cnt
