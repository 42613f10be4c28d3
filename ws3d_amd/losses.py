"""Stage-1 training targets and losses, mirroring the reference's numerics.

  * ``gaussian_center_labels``  -- KittiRCNNDataset.generate_gaussian_training_labels
    (lib/datasets/kitti_rcnn_dataset.py:529-573): soft foreground label from the distance to the
    nearest annotated centre, regression target = offset to that centre in x / z
  * ``sigmoid_focal_loss``      -- SigmoidFocalClassificationLoss (lib/utils/loss_utils.py:25-90)
  * ``rpn_reg_loss``            -- get_rpn_reg_loss, bin classification + residual (:93-156)
  * ``rpn_loss``                -- get_rpn_loss (lib/net/train_functions.py:163-228), focal branch
Everything here is plain torch / numpy: the custom kernels enter through the network's forward
and (deterministic) backward.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

GAUSS_HEIGHT, GAUSS_STATUS, GAUSS_COV = 0.707, 0.7, 1.5     # tools/cfgs/weaklyRPN.yaml:32-34
FOCAL_ALPHA, FOCAL_GAMMA = 0.25, 2.0                        # weaklyRPN.yaml:61-62 (alpha[0])
LOSS_WEIGHT = (1.0, 1.0)                                    # weaklyRPN.yaml:64


def gaussian_center_labels(pts_rect: np.ndarray, gt_centers: np.ndarray, gauss_height: float = GAUSS_HEIGHT,
                           gauss_status: float = GAUSS_STATUS, gauss_cov: float = GAUSS_COV) -> Tuple[np.ndarray, np.ndarray]:
    """pts_rect (N,3), gt_centers (K,>=3) -> cls_label (N,) in [0,1], reg_label (N,3) [dx, 0, dz].

    Distance of a point to a centre: sqrt(dx^2 + (y * gauss_height)^2 + dz^2) -- the point's own
    height, not the difference, as the reference writes it.  cls = N(d'; 0, cov) / N(0; 0, cov)
    with d' = clip(min_k d_k - gauss_status, 0, 100); points closer than 4 m to some centre regress
    to the nearest one.  No centres: all-zero labels."""
    n, k = pts_rect.shape[0], gt_centers.shape[0]
    cls_label = np.zeros(n, dtype=np.float32)
    reg_label = np.zeros((n, 3), dtype=np.float32)
    if k == 0:
        return cls_label, reg_label
    dist = np.sqrt(np.power(pts_rect[:, 0:1] - gt_centers[None, :, 0], 2) +
                   np.power(pts_rect[:, 1:2] * gauss_height, 2) +
                   np.power(pts_rect[:, 2:3] - gt_centers[None, :, 2], 2)).astype(np.float32)          # (N,K)
    near = np.minimum(np.float32(100.0), np.clip(dist - gauss_status, 0, 100).min(axis=1)).astype(np.float32)
    norm = 1.0 / math.sqrt(2 * np.pi * gauss_cov)
    cls = (norm * np.exp(-0.5 * near.astype(np.float64) ** 2 / gauss_cov)) / norm     # scipy's pdf / its peak, float64
    target = dist.argmin(axis=1)
    close = dist.min(axis=1) < 4.0
    reg_label[close, 0] = gt_centers[target][close, 0] - pts_rect[close][:, 0]
    reg_label[close, 2] = gt_centers[target][close, 2] - pts_rect[close][:, 2]
    return cls, reg_label


def scene_augmentation(pts_rect: np.ndarray, gt_boxes3d: np.ndarray, rng=np.random, method_prob=(1.0, 1.0, 0.5),
                       rot_range: float = 18.0):
    """global rotation about y / scaling / x-flip of one scene and its annotations
    (KittiRCNNDataset.data_augmentation, lib/datasets/kitti_rcnn_dataset.py:223-255; AUG_METHOD_LIST,
    AUG_METHOD_PROB, AUG_ROT_RANGE of weaklyRPN.yaml:7-9).  Draws from `rng` in the reference's
    order: rand(3), then uniform(angle), uniform(scale) for the enabled methods.  Works on copies;
    returns (pts (N,3), boxes (K,>=3), [applied methods])."""
    pts = np.array(pts_rect, copy=True)
    boxes = np.array(gt_boxes3d, copy=True)
    enable = 1 - rng.rand(3)
    applied = []
    if enable[0] < method_prob[0]:
        angle = rng.uniform(-np.pi / rot_range, np.pi / rot_range)
        rot = np.array([[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]])
        pts[:, [0, 2]] = np.dot(pts[:, [0, 2]], rot.T)
        boxes[:, [0, 2]] = np.dot(boxes[:, [0, 2]], rot.T)
        applied.append(["rotation", angle])
    if enable[1] < method_prob[1]:
        scale = rng.uniform(0.95, 1.05)
        pts = pts * scale
        boxes[:, 0:6] = boxes[:, 0:6] * scale
        applied.append(["scaling", scale])
    if enable[2] < method_prob[2]:
        pts[:, 0] = -pts[:, 0]
        boxes[:, 0] = -boxes[:, 0]
        applied.append("flip")
    return pts, boxes, applied


def _sigmoid_cross_entropy_with_logits(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    loss = torch.clamp(logits, min=0) - logits * labels.type_as(logits)
    return loss + torch.log1p(torch.exp(-torch.abs(logits)))


def sigmoid_focal_loss(logits: torch.Tensor, targets: torch.Tensor, weights: torch.Tensor,
                       gamma: float = FOCAL_GAMMA, alpha: float = FOCAL_ALPHA) -> torch.Tensor:
    """element-wise focal cross entropy * weights (soft targets allowed), loss_utils.py:42-79"""
    ce = _sigmoid_cross_entropy_with_logits(logits, targets)
    p = torch.sigmoid(logits)
    p_t = targets * p + (1 - targets) * (1 - p)
    mod = torch.pow(1.0 - p_t, gamma) if gamma else 1.0
    a = targets * alpha + (1 - targets) * (1 - alpha) if alpha is not None else 1.0
    return mod * a * ce * weights


def rpn_reg_loss(pred_reg: torch.Tensor, reg_label: torch.Tensor, loc_scope: float, loc_bin_size: float, lazy: bool = False):
    """pred_reg (F, 4*bins), reg_label (F,3) [dx, 0, dz] -> (loss, {name: float}); bins for x and z
    (cross entropy) + smooth-L1 on the normalised in-bin residual of the labelled bin.
    lazy: the dict holds detached 0-dim tensors instead of floats (no host sync per entry)"""
    bins = int((loc_scope + 1e-3) / loc_bin_size) * 2
    assert pred_reg.shape[1] == 4 * bins, "%d vs %d" % (pred_reg.shape[1], 4 * bins)
    parts = {}
    for name, col, lo in (("x", 0, 0), ("z", 2, bins)):
        shift = torch.clamp(reg_label[:, col] + loc_scope, 0, loc_scope * 2 - 1e-3)
        bin_label = (shift / loc_bin_size).floor().long()
        loss_bin = F.cross_entropy(pred_reg[:, lo:lo + bins], bin_label)
        res_label = (shift - (bin_label.float() * loc_bin_size + loc_bin_size / 2)) / (loc_bin_size / 2)
        onehot = torch.zeros((bin_label.size(0), bins), dtype=pred_reg.dtype, device=pred_reg.device)
        onehot.scatter_(1, bin_label.view(-1, 1), 1)
        res_pred = (pred_reg[:, 2 * bins + lo:3 * bins + lo] * onehot).sum(dim=1)
        loss_res = F.smooth_l1_loss(res_pred, res_label)
        parts["loss_%s_bin" % name], parts["loss_%s_res" % name] = loss_bin, loss_res
    # the reference adds x_bin + z_bin first, then x_res + z_res
    total = (parts["loss_x_bin"] + parts["loss_z_bin"]) + (parts["loss_x_res"] + parts["loss_z_res"])
    return total, {k: (v.detach() if lazy else v.item()) for k, v in parts.items()}


def rpn_loss(rpn_cls: torch.Tensor, rpn_reg: torch.Tensor, cls_label: torch.Tensor, reg_label: torch.Tensor,
             loc_scope: float, loc_bin_size: float, gaussian_center: bool = True,
             loss_weight=LOSS_WEIGHT, lazy: bool = False) -> Tuple[torch.Tensor, Dict[str, float]]:
    """rpn_cls (B,N,1), rpn_reg (B,N,4*bins), cls_label (B,N) (soft if gaussian_center), reg_label
    (B,N,3) -> (loss, tb_dict) as get_rpn_loss with LOSS_CLS = SigmoidFocalLoss.
    lazy: the scalar entries of tb_dict stay on the device as detached 0-dim tensors (the reference
    reads every one back with .item(): five host syncs in the middle of the step); resolve them
    with ``resolve_scalars`` once the step is enqueued."""
    tb = {}
    read = (lambda t: t.detach()) if lazy else (lambda t: t.item())
    label = cls_label.reshape(-1)
    logits = rpn_cls.reshape(-1)
    fg_mask = label > 0
    if gaussian_center:
        target = label.float()
        pos, neg = label.float(), (1 - label).float()
    else:
        target = (label > 0.5).float()
        pos, neg = (label > 0.5).float(), (label < 0.5).float()
    weights = (pos + neg) / torch.clamp(pos.sum(), min=1.0)
    per_point = sigmoid_focal_loss(logits, target, weights)
    tb["rpn_loss_cls_pos"] = read((per_point * pos).sum())
    tb["rpn_loss_cls_neg"] = read((per_point * neg).sum())
    loss_cls = per_point.sum()
    point_num = rpn_reg.size(0) * rpn_reg.size(1)
    fg_sum = int(fg_mask.long().sum().item())
    if fg_sum != 0:
        loss_reg, _ = rpn_reg_loss(rpn_reg.reshape(point_num, -1)[fg_mask], reg_label.reshape(point_num, 3)[fg_mask],
                                   loc_scope, loc_bin_size, lazy=True)
    else:
        loss_reg = loss_cls * 0
    loss = loss_cls * loss_weight[0] + loss_reg * loss_weight[1]
    tb.update({"rpn_loss_cls": read(loss_cls), "rpn_loss_reg": read(loss_reg), "rpn_loss": read(loss), "rpn_fg_sum": fg_sum})
    return loss, tb


def resolve_scalars(tb: dict) -> dict:
    """read every 0-dim device tensor of a (lazy) tb_dict back with ONE host sync"""
    keys = [k for k, v in tb.items() if torch.is_tensor(v)]
    if keys:
        vals = torch.stack([tb[k].detach().float().reshape(()) for k in keys]).tolist()
        tb.update(zip(keys, vals))
    return tb
