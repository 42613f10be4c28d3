#!/usr/bin/env python
"""Generate the committed golden fixtures by running the REFERENCE's own Python harness.

Runs ONLY in the build container (it imports from /root/reference, which does not exist
on the GPU box and is never copied): ``python -B tests/golden/make_golden.py``.

The reference's leaf kernels are CUDA-only, so its Python layer (pointnet2_utils /
pointnet2_modules / pytorch_utils / iou3d_utils / roipool3d_utils / lib.net.*) is imported
unmodified and run on CPU with the three extension modules it imports
(``pointnet2_cuda``, ``iou3d_cuda``, ``roipool3d_cuda``) backed by the CPU oracle -- and, for
roipool3d's CPU entry points, by the reference's own compiled C++ (oracle/_ref).  What
these fixtures pin is therefore every COMPOSITION rule above the leaf kernels:
QueryAndGroup's channel order and centre subtraction, the SA/FP module dataflow, the FP
inverse-distance weights, boxes_iou3d_gpu's height/volume math, nms_gpu's sort/index-back,
roipool3d_gpu's box enlargement, the Stage-1 network's layer wiring, state_dict key names
and decode_center_target.  (Import recipe: SURVEY.md appendix B.)

Fixtures are DATA only: seeds/params + expected outputs (index tensors, sampled floats).
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

import oracle  # noqa: E402
from oracle import build_ref  # noqa: E402
from ws3d_amd import synth  # noqa: E402
from ws3d_amd.seeded import seeded_state_dict  # noqa: E402


# --------------------------------------------------------------------------- shims
def _np(t):
    return t.detach().cpu().numpy()


def _fill(dst, arr):
    dst.copy_(torch.from_numpy(np.ascontiguousarray(arr)).to(dst.dtype))


def install_reference_shims():
    p2 = types.ModuleType("pointnet2_cuda")

    def furthest_point_sampling_wrapper(b, n, m, xyz, temp, idx):
        _fill(idx, oracle.furthest_point_sample(_np(xyz), m)); return 1

    def gather_points_wrapper(b, c, n, npoints, points, idx, out):
        _fill(out, oracle.gather_operation(_np(points), _np(idx))); return 1

    def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
        _fill(idx, oracle.ball_query(radius, nsample, _np(xyz), _np(new_xyz))); return 1

    def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
        _fill(out, oracle.grouping_operation(_np(points), _np(idx))); return 1

    def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
        d2, i = oracle.three_nn_dist2(_np(unknown), _np(known)); _fill(dist2, d2); _fill(idx, i)

    def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
        _fill(out, oracle.three_interpolate(_np(points), _np(idx), _np(weight)))

    for f in (furthest_point_sampling_wrapper, gather_points_wrapper, ball_query_wrapper, group_points_wrapper,
              three_nn_wrapper, three_interpolate_wrapper):
        setattr(p2, f.__name__, f)
    sys.modules["pointnet2_cuda"] = p2

    iou = types.ModuleType("iou3d_cuda")

    def boxes_overlap_bev_gpu(a, b, ans):
        _fill(ans, oracle.boxes_overlap_bev(_np(a), _np(b))); return 1

    def boxes_iou_bev_gpu(a, b, ans):
        _fill(ans, oracle.boxes_iou_bev(_np(a), _np(b))); return 1

    def nms_gpu(boxes, keep, thresh):
        k = oracle.nms_sorted(_np(boxes), thresh, False); keep[:len(k)] = torch.from_numpy(k); return len(k)

    def nms_normal_gpu(boxes, keep, thresh):
        k = oracle.nms_sorted(_np(boxes), thresh, True); keep[:len(k)] = torch.from_numpy(k); return len(k)

    for f in (boxes_overlap_bev_gpu, boxes_iou_bev_gpu, nms_gpu, nms_normal_gpu):
        setattr(iou, f.__name__, f)
    sys.modules["iou3d_cuda"] = iou

    roi = types.ModuleType("roipool3d_cuda")
    refmod = build_ref.load()

    def forward(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
        p, e = oracle.roipool3d(_np(xyz), _np(boxes3d), _np(pts_feature), pooled_features.shape[2])
        _fill(pooled_features, p); _fill(pooled_empty_flag, e); return 1

    roi.forward = forward
    roi.forward_slow = forward
    roi.pts_in_boxes3d_cpu = refmod.pts_in_boxes3d_cpu   # the reference's own compiled C++
    roi.roipool3d_cpu = refmod.roipool3d_cpu
    sys.modules["roipool3d_cuda"] = roi

    # legacy CUDA tensor constructors used by the wrappers -> CPU; .cuda() -> identity
    torch.cuda.IntTensor = torch.IntTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.Tensor.cuda = lambda self, *a, **k: self

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)

        __setattr__ = __setitem__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    _orig = yaml.load
    yaml.load = lambda f, Loader=yaml.FullLoader: _orig(f, Loader=Loader)
    sys.path[:0] = [REF, os.path.join(REF, "lib", "net")]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sample(a, k=96, seed=0):
    flat = np.ascontiguousarray(a).reshape(-1)
    pos = np.random.default_rng(seed).integers(0, flat.size, min(k, flat.size))
    return pos.astype(np.int64), flat[pos]


def main():
    install_reference_shims()
    from pointnet2_lib.pointnet2 import pointnet2_modules as ref_mod
    from pointnet2_lib.pointnet2 import pointnet2_utils as ref_utils
    from lib.config import cfg, cfg_from_file
    from lib.utils.iou3d import iou3d_utils as ref_iou
    from lib.utils.roipool3d import roipool3d_utils as ref_roi
    from lib.utils.bbox_transform import decode_center_target as ref_decode
    cfg_from_file(os.path.join(REF, "tools", "cfgs", "weaklyRPN.yaml"))
    from lib.net.point_rcnn import PointRCNN

    fx = {}
    meta = {"generator": "tests/golden/make_golden.py", "oracle_dist_mode": oracle.dist_mode(), "cases": {}}

    # ---- 1. QueryAndGroup / SA layer / FP layer compositions (small shapes)
    pc = synth.make_batch("lidar", 2, 1024, 31)
    xyz = torch.from_numpy(pc[:, :, :3].copy())
    feats = torch.from_numpy(np.ascontiguousarray(pc[:, :, 3:].transpose(0, 2, 1)))
    idx = ref_utils.furthest_point_sample(xyz, 128)
    new_xyz = ref_utils.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    qg = ref_utils.QueryAndGroup(1.0, 16, use_xyz=True)(xyz, new_xyz, feats)
    fx["qg_fps_idx"] = _np(idx).astype(np.int32)
    fx["qg_out"] = _np(qg)
    meta["cases"]["query_and_group"] = {"batch": 2, "n": 1024, "config_id": 31, "npoint": 128, "radius": 1.0, "nsample": 16}

    torch.manual_seed(0)
    sa = ref_mod.PointnetSAModuleMSG(npoint=128, radii=[0.5, 1.0], nsamples=[8, 16], mlps=[[1, 8, 16], [1, 8, 16]],
                                     use_xyz=True, bn=True).eval()
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in sa.state_dict().items()}, 5)
    sa.load_state_dict(sd)
    with torch.no_grad():
        sa_xyz, sa_feat = sa(xyz, feats)
    fx["sa_new_xyz"] = _np(sa_xyz)
    fx["sa_features"] = _np(sa_feat)
    meta["cases"]["sa_module"] = {"seed": 5, "npoint": 128, "radii": [0.5, 1.0], "nsamples": [8, 16],
                                  "mlps": [[1, 8, 16], [1, 8, 16]],
                                  "keys": {k: list(v.shape) for k, v in sa.state_dict().items()}}
    fp = ref_mod.PointnetFPModule(mlp=[32 + 1, 16, 8]).eval()
    fp.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in fp.state_dict().items()}, 6))
    with torch.no_grad():
        fp_out = fp(xyz, sa_xyz, feats, sa_feat)
    fx["fp_out"] = _np(fp_out)
    meta["cases"]["fp_module"] = {"seed": 6, "mlp": [33, 16, 8],
                                  "keys": {k: list(v.shape) for k, v in fp.state_dict().items()}}

    # ---- 2. iou3d compositions
    b3a = synth.proposal_boxes(1, 60, 41)[0]
    b3b = synth.proposal_boxes(1, 45, 41)[0] + np.array([0.3, 0.1, -0.2, 0, 0, 0, 0.1], dtype=np.float32)
    iou2d, iou3d = ref_iou.boxes_iou3d_gpu(torch.from_numpy(b3a), torch.from_numpy(b3b))
    fx["iou3d_a"], fx["iou3d_b"] = b3a, b3b
    fx["iou2d"], fx["iou3d"] = _np(iou2d), _np(iou3d)
    from lib.utils import kitti_utils as ref_kitti
    bev = ref_kitti.boxes3d_to_bev_torch(torch.from_numpy(b3a))
    fx["bev_a"] = _np(bev)
    scores = synth.distinct_scores(60, 41)
    fx["nms_scores"] = scores
    fx["nms_keep_rot"] = _np(ref_iou.nms_gpu(bev, torch.from_numpy(scores), 0.3)).astype(np.int64)
    fx["nms_keep_normal"] = _np(ref_iou.nms_normal_gpu(bev, torch.from_numpy(scores), 0.3)).astype(np.int64)
    fx["iou_bev"] = _np(ref_iou.boxes_iou_bev(bev, ref_kitti.boxes3d_to_bev_torch(torch.from_numpy(b3b))))

    # ---- 3. roipool3d wrapper (enlarge + pool) incl. the reference's compiled CPU twin
    rp_pc = synth.make_batch("lidar", 1, 2048, 43)
    rp_boxes = synth.proposal_boxes(1, 12, 43)
    rp_boxes[0, :6] = synth.random_boxes3d(15, 1000 * 43 * 7919 + 13)[:6]
    rp_feat = np.random.default_rng(43).standard_normal((1, 2048, 5)).astype(np.float32)
    pooled, empty = ref_roi.roipool3d_gpu(torch.from_numpy(rp_pc[:, :, :3].copy()), torch.from_numpy(rp_feat),
                                          torch.from_numpy(rp_boxes), 1.0, sampled_pt_num=64)
    fx["roi_boxes"], fx["roi_feat"] = rp_boxes, rp_feat
    fx["roi_pooled"], fx["roi_empty"] = _np(pooled), _np(empty).astype(np.int32)
    pp, pf, pe = ref_roi.roipool_pc_cpu(torch.from_numpy(rp_pc[0, :, :3].copy()), torch.from_numpy(rp_feat[0]),
                                        torch.from_numpy(ref_kitti.enlarge_box3d(rp_boxes[0], 1.0)), 64)
    assert np.array_equal(_np(pp), fx["roi_pooled"][0, :, :, :3]) and np.array_equal(_np(pf), fx["roi_pooled"][0, :, :, 3:])
    assert np.array_equal(_np(pe).astype(np.int32), fx["roi_empty"][0])
    meta["cases"]["roipool3d"] = {"config_id": 43, "n": 2048, "boxes": 12, "extra_width": 1.0, "sampled": 64,
                                  "checked_against_reference_cpu_twin": True}

    # ---- 4. Stage-1 network: state_dict layout + one full 16384-point forward
    model = PointRCNN(num_classes=2, use_xyz=True, mode='TEST').eval()
    keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    n_params = int(sum(p.numel() for p in model.parameters()))
    json.dump({"keys": keys, "n_params": n_params}, open(os.path.join(HERE, "stage1_state_dict.json"), "w"), indent=0)
    model.load_state_dict(seeded_state_dict({k: tuple(v) for k, v in keys.items()}, 7))
    pts = torch.from_numpy(synth.make_batch("lidar", 1, 16384, 3))
    taps = {}

    def tap_fps(orig):
        def f(xyz_, npoint):
            r = orig(xyz_, npoint)
            taps.setdefault("fps", []).append(_np(r).astype(np.int32))
            return r
        return f

    def tap_bq(orig):
        def f(radius, nsample, xyz_, new_xyz_):
            r = orig(radius, nsample, xyz_, new_xyz_)
            taps.setdefault("bq", []).append(sha(_np(r).astype(np.int32)))
            return r
        return f

    ref_utils.furthest_point_sample, ref_utils.ball_query = tap_fps(ref_utils.furthest_point_sample), tap_bq(ref_utils.ball_query)
    with torch.no_grad():
        out = model.rpn_forward({'pts_input': pts})
    s1 = {}
    for i, a in enumerate(taps["fps"]):
        s1[f"fps_idx_{i}"] = a
    meta["cases"]["stage1"] = {"config_id": 3, "seed": 7, "n_params": n_params, "n_keys": len(keys),
                               "ball_query_sha256": taps["bq"], "outputs": {}}
    for name in ("rpn_cls", "rpn_reg", "backbone_xyz", "backbone_features"):
        arr = _np(out[name])
        pos, val = sample(arr, 192, seed=len(name))
        s1[f"{name}_pos"], s1[f"{name}_val"] = pos, val
        meta["cases"]["stage1"]["outputs"][name] = {"shape": list(arr.shape), "abs_mean": float(np.abs(arr).mean())}
    dec = ref_decode(out["backbone_xyz"][0], out["rpn_reg"][0], cfg.RPN.LOC_SCOPE, cfg.RPN.LOC_BIN_SIZE)
    pos, val = sample(_np(dec), 192, seed=9)
    s1["decode_pos"], s1["decode_val"] = pos, val
    np.savez_compressed(os.path.join(HERE, "stage1_forward.npz"), **s1)
    np.savez_compressed(os.path.join(HERE, "compositions.npz"), **fx)
    json.dump(meta, open(os.path.join(HERE, "golden_meta.json"), "w"), indent=1)
    for f in ("stage1_forward.npz", "compositions.npz", "golden_meta.json", "stage1_state_dict.json"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
