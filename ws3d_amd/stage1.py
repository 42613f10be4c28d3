"""Stage-1 network of WS3D (the caller of the hot path) on the MI355X ops.

Counterpart of lib/net/pointnet2_msg.py:11-70 (``Pointnet2MSG``), lib/net/rpn.py:10-81
(``RPN``) and lib/net/point_rcnn.py:45-54 (``PointRCNN.rpn_forward``) for the
``tools/cfgs/weaklyRPN.yaml`` configuration, with IDENTICAL ``state_dict`` keys and shapes
(208 keys, 3,046,201 parameters: ``rpn.backbone_net.SA_modules.{k}.mlps.{s}.layer{i}.*``,
``rpn.backbone_net.FP_modules.{k}.mlp.layer{i}.*``, ``rpn.rpn_cls_layer.{0,2}.*``,
``rpn.rpn_reg_layer.{0,2}.*``) so that checkpoints trained with the reference load
key-for-key.  The EasyDict/YAML config tree is replaced by a frozen dataclass holding the
values of weaklyRPN.yaml:25-56.

Also here: ``decode_center_target`` (lib/utils/bbox_transform.py:24-61) and the on-device
proposal stage used by the data-parallel driver (score -> boxes -> rotated NMS -> roipool3d).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn

from . import iou3d_ops, nn_blocks as pt_utils, roipool3d_ops
from . import kitti_utils
from .pn2_modules import PointnetFPModule, PointnetSAModuleMSG


@dataclass(frozen=True)
class RPNConfig:
    """tools/cfgs/weaklyRPN.yaml:25-56 (+ TEST block :103-107, CLS_MEAN_SIZE :19)"""
    use_intensity: bool = True
    use_bn: bool = True
    num_points: int = 16384
    npoints: tuple = (4096, 1024, 256, 64)
    radius: tuple = ((0.1, 0.5), (0.5, 1.0), (1.0, 2.0), (2.0, 4.0))
    nsample: tuple = ((16, 32), (16, 32), (16, 32), (16, 32))
    mlps: tuple = (((16, 16, 32), (32, 32, 64)), ((64, 64, 128), (64, 96, 128)),
                   ((128, 196, 256), (128, 196, 256)), ((256, 256, 512), (256, 384, 512)))
    fp_mlps: tuple = ((128, 128), (256, 256), (512, 512), (512, 512))
    cls_fc: tuple = (128,)
    reg_fc: tuple = (128,)
    dp_ratio: float = 0.5
    loc_scope: float = 4.0
    loc_bin_size: float = 0.8
    focal_init_pi: float = 0.01
    score_thresh: float = 0.3
    rpn_pre_nms_top_n: int = 9000
    rpn_post_nms_top_n: int = 100
    rpn_nms_thresh: float = 0.8
    cls_mean_size: tuple = (1.52563191462, 1.62856739989, 3.88311640418)  # h, w, l
    roi_sampled_pts: int = 512
    roi_extra_width: float = 1.0


DEFAULT_CFG = RPNConfig()

# eval-mode GPU inference runs the channels-last pipeline of ws3d_amd/fastpath.py (same weights and
# operators, (B,N,C) feature rows, GEMM epilogues fused); clear to force the reference-layout path
CHANNELS_LAST_FASTPATH = True


class Pointnet2MSG(nn.Module):
    """4 SA-MSG + 4 FP layers (lib/net/pointnet2_msg.py:11-70)"""

    def __init__(self, input_channels=6, use_xyz=True, cfg: RPNConfig = DEFAULT_CFG):
        super().__init__()
        self.SA_modules = nn.ModuleList()
        channel_in = input_channels
        skip_channel_list = [input_channels]
        channel_out = channel_in
        for k in range(len(cfg.npoints)):
            mlps = [[channel_in] + list(m) for m in cfg.mlps[k]]
            channel_out = sum(m[-1] for m in mlps)
            self.SA_modules.append(PointnetSAModuleMSG(npoint=cfg.npoints[k], radii=list(cfg.radius[k]),
                                                       nsamples=list(cfg.nsample[k]), mlps=mlps,
                                                       use_xyz=use_xyz, bn=cfg.use_bn))
            skip_channel_list.append(channel_out)
            channel_in = channel_out
        self.FP_modules = nn.ModuleList()
        for k in range(len(cfg.fp_mlps)):
            pre_channel = cfg.fp_mlps[k + 1][-1] if k + 1 < len(cfg.fp_mlps) else channel_out
            self.FP_modules.append(PointnetFPModule(mlp=[pre_channel + skip_channel_list[k]] + list(cfg.fp_mlps[k])))

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def forward(self, pointcloud: torch.Tensor, sampling_plan=None):
        """sampling_plan: optional ``pointnet2_utils.sampling_plan(xyz, npoints)`` computed ahead of time
        (the SA modules then skip their own furthest point sampling)"""
        xyz, features = self._break_up_pc(pointcloud)
        l_xyz, l_features = [xyz], [features]
        for k, sa in enumerate(self.SA_modules):
            li_xyz, li_features = sa(l_xyz[-1], l_features[-1], sampling_plan[k] if sampling_plan else None)
            l_xyz.append(li_xyz)
            l_features.append(li_features)
        for i in range(-1, -(len(self.FP_modules) + 1), -1):
            l_features[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], l_features[i])
        return l_xyz[0], l_features[0]


class RPN(nn.Module):
    """backbone + point-wise cls / bin-based reg heads (lib/net/rpn.py:10-81; losses omitted:
    the training runtime is out of scope)"""

    def __init__(self, use_xyz=True, mode='TRAIN', cfg: RPNConfig = DEFAULT_CFG):
        super().__init__()
        self.training_mode = (mode == 'TRAIN')
        self.cfg = cfg
        self.backbone_net = Pointnet2MSG(input_channels=int(cfg.use_intensity), use_xyz=use_xyz, cfg=cfg)

        def head(fc_dims, out_channels):
            layers, pre = [], cfg.fp_mlps[0][-1]
            for width in fc_dims:
                layers.append(pt_utils.Conv1d(pre, width, bn=cfg.use_bn))
                pre = width
            layers.append(pt_utils.Conv1d(pre, out_channels, activation=None))
            if cfg.dp_ratio >= 0:
                layers.insert(1, nn.Dropout(cfg.dp_ratio))
            return nn.Sequential(*layers)

        per_loc_bin_num = int(cfg.loc_scope / cfg.loc_bin_size) * 2
        self.rpn_cls_layer = head(cfg.cls_fc, 1)
        self.rpn_reg_layer = head(cfg.reg_fc, per_loc_bin_num * 4)
        self.init_weights()

    def init_weights(self):
        pi = self.cfg.focal_init_pi
        nn.init.constant_(self.rpn_cls_layer[2].conv.bias, -np.log((1 - pi) / pi))
        nn.init.normal_(self.rpn_reg_layer[-1].conv.weight, mean=0, std=0.001)

    def forward(self, input_data):
        pts_input = input_data['pts_input']
        backbone_xyz, backbone_features = self.backbone_net(pts_input, input_data.get('sampling_plan'))  # (B,N,3), (B,C,N)
        rpn_cls = self.rpn_cls_layer(backbone_features).transpose(1, 2).contiguous()   # (B,N,1)
        rpn_reg = self.rpn_reg_layer(backbone_features).transpose(1, 2).contiguous()   # (B,N,40)
        return {'rpn_cls': rpn_cls, 'rpn_reg': rpn_reg,
                'backbone_xyz': backbone_xyz, 'backbone_features': backbone_features}


class Stage1Net(nn.Module):
    """``PointRCNN`` restricted to its Stage-1 half (lib/net/point_rcnn.py:10-54): attribute
    ``rpn`` keeps the checkpoint prefix."""

    def __init__(self, num_classes=2, use_xyz=True, mode='TEST', cfg: RPNConfig = DEFAULT_CFG):
        super().__init__()
        self.mode = mode
        self.cfg = cfg
        self.rpn = RPN(use_xyz=use_xyz, mode=mode, cfg=cfg)

    def rpn_forward(self, input_data):
        pts = input_data['pts_input']
        if CHANNELS_LAST_FASTPATH and not self.training and pts.is_cuda:
            from . import fastpath
            ok = self.__dict__.get("_fastpath_ok")
            if ok is None:
                ok = self.__dict__["_fastpath_ok"] = fastpath.supported(self)
            if ok:
                # 'defer_reg_join': see fastpath.rpn_forward (the caller promises to wait for out['rpn_reg_ready'] before reading rpn_reg)
                return fastpath.rpn_forward(self, pts, bool(input_data.get('defer_reg_join', False)))
        with torch.set_grad_enabled(self.training):
            return dict(self.rpn(input_data))

    forward = rpn_forward


def decode_center_target(roi_center, pred_reg, loc_scope, loc_bin_size):
    """(N,3) points + (N,4*bins) bin/residual logits -> (N,3) predicted centres with y = 0
    (lib/utils/bbox_transform.py:24-61)."""
    nb = int(loc_scope / loc_bin_size) * 2
    x_bin = torch.argmax(pred_reg[:, 0:nb], dim=1)
    z_bin = torch.argmax(pred_reg[:, nb:2 * nb], dim=1)
    pos_x = x_bin.float() * loc_bin_size + loc_bin_size / 2 - loc_scope
    pos_z = z_bin.float() * loc_bin_size + loc_bin_size / 2 - loc_scope
    x_res = torch.gather(pred_reg[:, 2 * nb:3 * nb], dim=1, index=x_bin.unsqueeze(1)).squeeze(1)
    z_res = torch.gather(pred_reg[:, 3 * nb:4 * nb], dim=1, index=z_bin.unsqueeze(1)).squeeze(1)
    pos_x = pos_x + x_res * (loc_bin_size / 2)
    pos_z = pos_z + z_res * (loc_bin_size / 2)
    # (the reference adds the point's x,z through `ret[:, [0, 2]] += ...`; the list index would
    # stage an index tensor through the host, which a hipGraph capture does not allow)
    return torch.stack((pos_x + roi_center[:, 0], torch.zeros_like(pos_x), pos_z + roi_center[:, 2]), dim=1)


def synthetic_orientation(n: int, device) -> torch.Tensor:
    """WS3D's Stage-1 predicts centres only; the op-level NMS/roipool configs
    need oriented boxes, so every point index gets a fixed pseudo-random heading in [-pi, pi)."""
    k = torch.arange(n, device=device, dtype=torch.int64)
    h = (k * 2654435761) % 4294967296
    return (h.double() / 4294967296.0 * (2 * math.pi) - math.pi).float()


@torch.no_grad()
def proposals_from_rpn(out: dict, cfg: RPNConfig = DEFAULT_CFG, with_pool_boxes: bool = False, fused: bool = True, with_packed: bool = False):
    """On-device proposal stage for the whole batch (config 3 of BASELINE.json): score =
    sigmoid(rpn_cls), centre = decode_center_target, box = centre + CLS_MEAN_SIZE + synthetic
    heading; top RPN_PRE_NMS_TOP_N by score -> rotated NMS (thresh 0.8) -> first
    RPN_POST_NMS_TOP_N survivors.  Returns boxes (B,K,7), scores (B,K), count (B,) -- fixed
    shapes, zero padded, batched torch ops + ONE NMS launch pair, no host synchronisation.
    with_pool_boxes: also return the (B,K,7) rows enlarged by cfg.roi_extra_width (what roipool3d_gpu would compute);
    with_packed: also (last) the (B,K,8) rows box + score that ws3d_amd.dist gathers across ranks; a tensor: the send buffer of a
    ``dist.ProposalExchange`` -- rows and counts are written straight into it and the last result is a view of it;
    fused=False keeps the torch composition of the sigmoid / decode / gather / BEV / padding steps (parity tests)."""
    xyz, reg, cls = out['backbone_xyz'], out['rpn_reg'], out['rpn_cls']
    B, N, _ = xyz.shape
    send = with_packed if torch.is_tensor(with_packed) else None
    with_packed = send is not None or bool(with_packed)
    K = cfg.rpn_post_nms_top_n
    h, w, l = cfg.cls_mean_size
    top = min(cfg.rpn_pre_nms_top_n, N)
    on_dev = xyz.is_cuda and top >= K and fused
    sc = order = score = None
    if xyz.is_cuda and N <= 16384:     # one LDS-resident sort per scene instead of topk's select + gather + merge sort
        from . import compat as _C
        if on_dev:                     # ... over torch.sigmoid's expression of the logits, evaluated while the keys are loaded: no score tensor
            sc, order = _C.topk_sorted(cls[:, :, 0].contiguous(), top, sigmoid=True)
        else:
            score = torch.sigmoid(cls[:, :, 0])                                           # (B,N)
            sc, order = _C.topk_sorted(score.contiguous(), top)      # (scores only: runs beside the regression head when that is deferred)
    else:
        score = torch.sigmoid(cls[:, :, 0])
    ready = out.get('rpn_reg_ready')
    if ready is not None:
        torch.cuda.current_stream(xyz.device).wait_event(ready)   # fastpath.rpn_forward(defer_reg_join=True): rpn_reg comes from a side stream
    if on_dev and order is not None:
        # four launches instead of ~30: the rows of the top points decoded in score order + their BEV rectangles, NMS, then the
        # padded survivors (+ the rows enlarged for RoI pooling / packed with their scores when asked for); bit-identical to the
        # composition below
        box_sorted, bev = _C.decode_gather_boxes_bev(xyz.contiguous(), reg.contiguous(), order, cfg.loc_scope, cfg.loc_bin_size, (h, w, l))
        keep_dev, num = _C.nms_device_batched(bev, cfg.rpn_nms_thresh, False, max_keep=K)
        res = _C.select_proposals(box_sorted, sc, keep_dev, num, K, cfg.roi_extra_width if with_pool_boxes else None, packed=send if send is not None else with_packed)
        return tuple(res[:3]) + ((res[3],) if with_pool_boxes else ()) + ((res[4],) if with_packed else ())
    if xyz.is_cuda:
        from . import compat as _C        # one kernel instead of ~25 tiny torch launches; bit-identical
        box = _C.decode_center_boxes(xyz.contiguous(), reg.contiguous(), cfg.loc_scope, cfg.loc_bin_size, (h, w, l))
    else:
        centre = decode_center_target(xyz.reshape(B * N, 3), reg.reshape(B * N, -1), cfg.loc_scope,
                                      cfg.loc_bin_size).view(B, N, 3)
        ry = synthetic_orientation(N, xyz.device).unsqueeze(0).expand(B, N)
        box = torch.stack((centre[..., 0], xyz[..., 1] + h / 2, centre[..., 2], torch.full_like(score, h),
                           torch.full_like(score, w), torch.full_like(score, l), ry), dim=2)     # (B,N,7)
    if sc is None:
        sc, order = torch.topk(score, top, dim=1, sorted=True)                            # (B,top)
    if on_dev:                 # (N > 16384: torch.topk above, then the fused gather / NMS / selection)
        box_sorted, bev = _C.gather_boxes_bev(box, order)
        keep_dev, num = _C.nms_device_batched(bev, cfg.rpn_nms_thresh, False, max_keep=K)
        res = _C.select_proposals(box_sorted, sc, keep_dev, num, K, cfg.roi_extra_width if with_pool_boxes else None, packed=send if send is not None else with_packed)
        return tuple(res[:3]) + ((res[3],) if with_pool_boxes else ()) + ((res[4],) if with_packed else ())
    box = torch.gather(box, 1, order.unsqueeze(-1).expand(B, top, 7))
    bev = kitti_utils.boxes3d_to_bev_torch(box.reshape(B * top, 7)).view(B, top, 5)
    keep, cnt = iou3d_ops.nms_gpu_padded_batched(bev, sc, cfg.rpn_nms_thresh, K, scores_sorted=True)   # (B,K), (B,)
    valid = keep >= 0
    safe = keep.clamp(min=0)
    boxes_out = torch.gather(box, 1, safe.unsqueeze(-1).expand(B, K, 7)) * valid.unsqueeze(-1)
    scores_out = torch.gather(sc, 1, safe) * valid
    res = (boxes_out, scores_out, cnt)
    if with_pool_boxes:
        res += (kitti_utils.enlarge_box3d(boxes_out.view(-1, 7), cfg.roi_extra_width).view(B, K, 7),)
    if with_packed:
        pk = torch.cat([boxes_out, scores_out.unsqueeze(-1)], dim=-1).contiguous()
        if send is not None:            # (the composition's form of ws3d_select_proposals_send: rows, then the count, per scene)
            send[:B, :K * 8] = pk.view(B, K * 8)
            send[:B, K * 8] = cnt.to(send.dtype)
            pk = send[:B, :K * 8].unflatten(1, (K, 8))
        res += (pk,)
    return res


@torch.no_grad()
def center_proposals(out: dict, cfg: RPNConfig = DEFAULT_CFG, prop_dist: float = 0.3, min_reg_dist: float = 0.2):
    """WS3D's live Stage-1 proposal stage (generate_box_dataset.py:92-140, tools/eval_auto.py:
    248-284) for ONE scene: sigmoid score, decode_center_target, keep points with
    score > SCORE_THRESH and |centre - point|_xz > 0.2, sort by score, greedy radius NMS (0.3 m).
    The reference's O(K^2) Python loop (one device sync per candidate) is one mask launch + one
    sweep launch here.  Returns (centres (K,3), sigmoid scores (K,), raw scores (K,))."""
    from . import compat as _C
    xyz = out['backbone_xyz'].reshape(-1, 3)
    raw = out['rpn_cls'].reshape(-1)
    norm = torch.sigmoid(raw)
    rois = decode_center_target(xyz, out['rpn_reg'].reshape(xyz.shape[0], -1), cfg.loc_scope, cfg.loc_bin_size)
    reg_dist = rois - xyz
    mask = (norm > cfg.score_thresh) & (torch.stack((reg_dist[:, 0], reg_dist[:, 2]), 1).pow(2).sum(-1).sqrt() > min_reg_dist)
    rois, raw, norm = rois[mask], raw[mask], norm[mask]          # dynamic shape, like the reference
    if rois.shape[0] == 0:
        return rois, norm, raw
    order = torch.argsort(-norm, stable=True)
    rois, raw, norm = rois[order], raw[order], norm[order]
    if rois.shape[0] > 1:
        cxz = torch.stack((rois[:, 0], rois[:, 2]), 1).contiguous().unsqueeze(0)
        keep, num = _C.radius_nms_device_batched(cxz, prop_dist)
        k = keep[0, :int(num[0])]
        rois, raw, norm = rois[k], raw[k], norm[k]
    return rois, norm, raw


@torch.no_grad()
def stage1_inference(model: Stage1Net, pts_input: torch.Tensor, cfg: RPNConfig = DEFAULT_CFG):
    """Stage-1 forward + proposals + RoI pooling for a batch of scenes (B,N,4)."""
    out = model.rpn_forward({'pts_input': pts_input})
    boxes, scores, count, enlarged = proposals_from_rpn(out, cfg, with_pool_boxes=True)
    feats = out['backbone_features'].transpose(1, 2).contiguous()  # (B,N,C)
    pooled, empty = roipool3d_ops.roipool3d_gpu(out['backbone_xyz'], feats, boxes, cfg.roi_extra_width,
                                                sampled_pt_num=cfg.roi_sampled_pts, enlarged=enlarged)
    return {'boxes': boxes, 'scores': scores, 'count': count, 'pooled': pooled, 'empty': empty, 'rpn': out}
