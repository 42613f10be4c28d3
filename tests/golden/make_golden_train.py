#!/usr/bin/env python
"""Golden fixture for the Stage-1 training targets and losses (ws3d_amd/losses.py), produced by
the REFERENCE's own code on CPU: ``python -B tests/golden/make_golden_train.py``.

  * KittiRCNNDataset.generate_gaussian_training_labels  (lib/datasets/kitti_rcnn_dataset.py:529-573)
  * train_functions.model_joint_fn_decorator()'s model_fn -> get_rpn_loss -> loss_utils
    (lib/net/train_functions.py:18-228, lib/utils/loss_utils.py:25-156) with a stand-in "model"
    that returns seeded logits, so the fixture pins label generation, focal loss, bin/residual
    loss and their combination -- values and gradients.
Inputs are regenerated from seeds by the test; only expected outputs are stored.
"""
from __future__ import annotations

import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

import numpy as np  # noqa: E402
import torch  # noqa: E402

import make_golden as mg  # noqa: E402
from ws3d_amd import synth  # noqa: E402

CASES = [{"name": "two_scenes", "batch": 2, "n": 4096, "config_id": 61, "cars": 6},
         {"name": "no_centres", "batch": 1, "n": 1024, "config_id": 62, "cars": 0}]


def case_inputs(case):
    """shared with tests/test_train.py: points, annotated centres, seeded network outputs"""
    B, n = case["batch"], case["n"]
    pc = synth.make_batch("lidar", B, n, case["config_id"])
    centres = [synth.random_boxes3d(15, (1000 * case["config_id"] + b) * 7919 + 13)[:case["cars"], :3].astype(np.float32)
               for b in range(B)]
    rng = np.random.default_rng(case["config_id"])
    rpn_cls = rng.normal(-2.0, 1.5, (B, n, 1)).astype(np.float32)
    rpn_reg = rng.normal(0.0, 1.0, (B, n, 40)).astype(np.float32)
    return pc, centres, rpn_cls, rpn_reg


def main():
    mg.install_reference_shims()
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    from lib.config import cfg, cfg_from_file
    cfg_from_file(os.path.join(REF, "tools", "cfgs", "weaklyRPN.yaml"))
    cfg.RPN.ENABLED = True
    from lib.datasets.kitti_rcnn_dataset import KittiRCNNDataset
    import lib.net.train_functions as tf
    import lib.utils.loss_utils as lu

    out = {"generator": "tests/golden/make_golden_train.py", "cases": {}}
    model_fn = tf.model_joint_fn_decorator()
    for case in CASES:
        pc, centres, rpn_cls, rpn_reg = case_inputs(case)
        B, n = case["batch"], case["n"]
        cls_l, reg_l = [], []
        for b in range(B):
            c, r = KittiRCNNDataset.generate_gaussian_training_labels(pc[b, :, :3], centres[b])
            cls_l.append(np.asarray(c, dtype=np.float64)); reg_l.append(r)
        cls_label, reg_label = np.stack(cls_l), np.stack(reg_l)

        t_cls = torch.from_numpy(rpn_cls).requires_grad_(True)
        t_reg = torch.from_numpy(rpn_reg).requires_grad_(True)

        class FakeModel(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.rpn = types.SimpleNamespace(rpn_cls_loss_func=lu.SigmoidFocalClassificationLoss(
                    alpha=cfg.RPN.FOCAL_ALPHA[0], gamma=cfg.RPN.FOCAL_GAMMA))

            def forward(self, input_data):
                return {"rpn_cls": t_cls, "rpn_reg": t_reg}

        max_gt = max(len(c) for c in centres)
        gt = np.zeros((B, max(max_gt, 1), 3), dtype=np.float32)
        for b in range(B):
            gt[b, :len(centres[b])] = centres[b]
        data = {"pts_input": pc, "gt_centers": gt, "rpn_cls_label": cls_label, "rpn_reg_label": reg_label}
        ret = model_fn(FakeModel(), data)
        ret.loss.backward()
        pos_c, val_c = mg.sample(cls_label, 128, seed=1)
        pos_gc, val_gc = mg.sample(t_cls.grad.numpy(), 64, seed=2)
        g_reg = t_reg.grad.numpy() if t_reg.grad is not None else np.zeros_like(rpn_reg)   # no foreground: untouched
        pos_gr, val_gr = mg.sample(g_reg, 64, seed=3)
        nz = np.flatnonzero(g_reg.reshape(-1))[:64]
        out["cases"][case["name"]] = {
            "case": case,
            "cls_label": {"pos": pos_c.tolist(), "val": [float(v) for v in val_c], "sum": float(cls_label.sum()),
                          "fg": int((cls_label > 0).sum())},
            "reg_label": {"sha256": mg.sha(reg_label), "nonzero": int((reg_label != 0).sum())},
            "loss": float(ret.loss.item()), "tb": {k: float(v) for k, v in ret.tb_dict.items()},
            "grad_cls": {"pos": pos_gc.tolist(), "val": [float(v) for v in val_gc]},
            "grad_reg": {"pos": pos_gr.tolist() + nz.tolist(),
                         "val": [float(v) for v in val_gr] + [float(v) for v in g_reg.reshape(-1)[nz]]},
        }
        print(case["name"], "loss", ret.loss.item(), ret.tb_dict)
    # scene augmentation (rotation / scaling / flip) with the reference's random stream
    aug = []
    for seed in (0, 1, 2, 3):
        pts = synth.lidar_cloud(512, 900 + seed)[:, :3].astype(np.float64)
        boxes = synth.random_boxes3d(6, 77 + seed).astype(np.float64)
        np.random.seed(seed)
        a_pts, a_box, methods = KittiRCNNDataset.data_augmentation(None, pts.copy(), boxes.copy())
        aug.append({"seed": seed, "pts_sha256": mg.sha(np.ascontiguousarray(a_pts)), "box_sha256": mg.sha(np.ascontiguousarray(a_box)),
                    "methods": [m if isinstance(m, str) else [m[0], float(m[1])] for m in methods]})
    out["augmentation"] = aug
    with open(os.path.join(HERE, "train_losses.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
