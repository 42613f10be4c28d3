"""(part of bench.py) Config-3 workload: full Stage-1 RPN forward (Pointnet2MSG 4 SA + 4 FP +
heads, weaklyRPN cfg) + on-device proposal NMS + roipool3d on `batch` synthetic scenes per
GPU, plus the one all-gather of proposals when world > 1 (BASELINE.json configs[2]/[3])."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from ws3d_amd import dist as wdist
from ws3d_amd import roipool3d_ops, synth
from ws3d_amd.seeded import seeded_state_dict
from ws3d_amd.stage1 import DEFAULT_CFG, Stage1Net, proposals_from_rpn


class C3:
    name = "c3_stage1_rpn_forward_nms_roipool"

    def __init__(self, batch, rank, world, kind="lidar"):
        self.B, self.rank, self.world, self.cfg = batch, rank, world, DEFAULT_CFG
        self.pc_host = np.stack([synth.lidar_cloud(16384, 1000 * 3 + rank * batch + s) if kind == "lidar"
                                 else synth.uniform_cloud(16384, 1000 * 3 + rank * batch + s) for s in range(batch)])
        self.pts = torch.from_numpy(self.pc_host).cuda()
        model = Stage1Net(mode='TEST').eval()
        model.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 7))
        self.model = model.cuda()
        self.ev = []
        self.last = None

    def config(self):
        c = self.cfg
        return {"model": "Pointnet2MSG+RPN (weaklyRPN.yaml), 3,046,201 params, random-init (seeded)",
                "pre_nms": c.rpn_pre_nms_top_n, "nms_thresh": c.rpn_nms_thresh, "post_nms": c.rpn_post_nms_top_n,
                "roipool": {"sampled": c.roi_sampled_pts, "channels": 128, "extra_width": c.roi_extra_width},
                "exchange": "all_gather of (B,100,8) proposals" if self.world > 1 else "none (1 GPU)"}

    @torch.no_grad()
    def step(self, timed=False):
        e = None
        if timed:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
        out = self.model.rpn_forward({'pts_input': self.pts})
        if timed:
            e[1].record()
        boxes, scores, count = proposals_from_rpn(out, self.cfg)
        if timed:
            e[2].record()
        feats = out['backbone_features'].transpose(1, 2).contiguous()
        pooled, empty = roipool3d_ops.roipool3d_gpu(out['backbone_xyz'], feats, boxes, self.cfg.roi_extra_width,
                                                    sampled_pt_num=self.cfg.roi_sampled_pts)
        if timed:
            e[3].record()
            self.ev.append(e)
        gathered = wdist.all_gather_proposals(wdist.pack_proposals(boxes, scores), count, self.B * self.world)
        self.last = (out, boxes, scores, count, pooled, empty, gathered)

    def kernel_ms(self):
        fwd = float(np.mean([a[0].elapsed_time(a[1]) for a in self.ev]))
        rest = float(np.mean([a[1].elapsed_time(a[3]) for a in self.ev]))
        self.breakdown = {"rpn_forward_ms": fwd,
                          "proposals_nms_ms": float(np.mean([a[1].elapsed_time(a[2]) for a in self.ev])),
                          "roipool_ms": float(np.mean([a[2].elapsed_time(a[3]) for a in self.ev]))}
        return fwd, rest

    def scenes(self):
        return self.B

    def cpu_baseline(self):
        """The custom ops of ONE scene's Stage-1 forward on the CPU oracle port (the MLPs are
        torch/BLAS on both sides and are excluded): 4x FPS, 8x ball_query+group, 4x three_nn."""
        import oracle
        threads = max(1, min(oracle.max_threads(), len(os.sched_getaffinity(0))))
        oracle.set_threads(threads)
        ns = int(min(self.B, 2))
        xyz = np.ascontiguousarray(self.pc_host[:ns, :, :3])
        t0 = time.perf_counter()
        levels = [xyz]
        for k, m in enumerate(self.cfg.npoints):
            idx = oracle.furthest_point_sample(levels[-1], m)
            new = np.stack([levels[-1][b][idx[b]] for b in range(ns)])
            for r, s in zip(self.cfg.radius[k], self.cfg.nsample[k]):
                oracle.ball_query(r, s, levels[-1], new)
            levels.append(new)
        for k in range(4, 0, -1):
            oracle.three_nn_dist2(levels[k - 1], levels[k])
        dt = time.perf_counter() - t0
        oracle.set_threads(1)
        return {"value": ns / dt, "unit": "scenes/s", "cores": threads, "kind": "port",
                "sample": f"{ns} scenes: the search ops of the Stage-1 forward only (4 FPS, 8 ball queries, 4 three_nn) "
                          f"on oracle/ws3d_oracle.c with OpenMP, wall {dt:.2f} s; grouping copies and MLPs excluded"}
