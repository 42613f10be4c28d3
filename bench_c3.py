"""(part of bench.py) Config-3 workload: full Stage-1 RPN forward (Pointnet2MSG 4 SA + 4 FP +
heads, weaklyRPN cfg) + on-device proposal NMS + roipool3d on `batch` synthetic scenes per
GPU, plus the one all-gather of proposals when world > 1 (BASELINE.json configs[2]/[3])."""
from __future__ import annotations

import os

import numpy as np
import torch

from ws3d_amd import dist as wdist
from ws3d_amd import fastpath, roipool3d_ops, synth
from ws3d_amd.seeded import seeded_state_dict
from ws3d_amd.stage1 import DEFAULT_CFG, Stage1Net, proposals_from_rpn


# SURVEY.md 8(d) byte model of the custom ops of one Stage-1 forward (per scene, weaklyRPN shapes)
def _fps_model_bytes(cfg):
    n, tot = cfg.num_points, 0
    for m in cfg.npoints:
        tot += (m - 1) * n * 12 + m * 4
        n = m
    return tot


def _qg_bytes(cfg):
    n, c, tot = cfg.num_points, 1, 0
    for k, m in enumerate(cfg.npoints):
        for ns in cfg.nsample[k]:
            tot += (n + m) * 12 + m * ns * 4 + (3 + c) * n * 4 + (3 + c) * m * ns * 4
        c = sum(mm[-1] for mm in cfg.mlps[k])
        n = m
    return tot


def _interp_bytes(cfg):
    pts = [cfg.num_points] + list(cfg.npoints)
    chans = [sum(mm[-1] for mm in cfg.mlps[k]) for k in range(4)]      # 96, 256, 512, 1024
    known_c = [cfg.fp_mlps[1][-1], cfg.fp_mlps[2][-1], cfg.fp_mlps[3][-1], chans[3]]  # features interpolated FROM
    nn_b = interp_b = 0
    for k in range(4):
        n, m, c = pts[k], pts[k + 1], known_c[k]
        nn_b += (n + m) * 12 + n * 3 * 8
        interp_b += c * m * 4 + n * 3 * 8 + c * n * 4
    return nn_b, interp_b


class C3:
    name = "c3_stage1_rpn_forward_nms_roipool"
    # = BASELINE.json "metric"; `value` is its scenes/sec half at configs[2] (Stage-1 RPN forward incl. proposal NMS +
    # roipool3d, batch 8/GPU), the "FPS+group HBM GB/s" half is the c2 block of the same line
    metric = "KITTI scenes/sec (16384 pts) Stage-1 RPN fwd, 1/2/4/8 GPU; FPS+group HBM GB/s"

    def __init__(self, batch, rank, world, kind="lidar", depth=2):
        self.B, self.rank, self.world, self.cfg = batch, rank, world, DEFAULT_CFG
        self.depth = max(1, depth)
        self.pc_host = np.stack([synth.lidar_cloud(16384, 1000 * 3 + rank * batch + s) if kind == "lidar"
                                 else synth.uniform_cloud(16384, 1000 * 3 + rank * batch + s) for s in range(batch)])
        self.pts = torch.from_numpy(self.pc_host).cuda()
        self._i = 0
        model = Stage1Net(mode='TEST').eval()
        model.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 7))
        self.model = model.cuda()
        self.ev = []
        self.last = None
        self.op_ev = {}
        self._timed = False
        self._install_hooks()

    def _install_hooks(self):
        """HIP-event timers around every C-ABI launch family (only while a timed step runs)."""
        from ws3d_amd import compat
        fam = {"furthest_point_sampling_gather": "fps", "furthest_point_sampling_nested": "fps", "ball_query_wrapper": "ball_query+group", "query_and_group": "ball_query+group",
               "query_and_group_nlc": "ball_query+group", "three_interpolate_nlc": "three_interpolate",
               "three_nn_wrapper": "three_nn", "three_nn_with_weights": "three_nn", "three_interpolate_wrapper": "three_interpolate",
               "nms_device_batched": "nms(mask+sweep)", "roipool3d_forward": "roipool3d"}
        for fn_name, key in fam.items():
            orig = getattr(compat, fn_name)

            def wrapped(*a, __orig=orig, __key=key, **kw):
                if not self._timed:
                    return __orig(*a, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = __orig(*a, **kw)
                e1.record()
                self.op_ev.setdefault(__key, []).append((e0, e1))
                return r
            setattr(compat, fn_name, wrapped)

    def config(self):
        c = self.cfg
        return {"network": "Pointnet2MSG+RPN (weaklyRPN.yaml), 3,046,201 params, random-init (seeded)",
                "pre_nms": c.rpn_pre_nms_top_n, "nms_thresh": c.rpn_nms_thresh, "post_nms": c.rpn_post_nms_top_n,
                "roipool": {"sampled": c.roi_sampled_pts, "channels": 128, "extra_width": c.roi_extra_width},
                "exchange": "all_gather of (B,100,8) proposals" if self.world > 1 else "none (1 GPU)",
                "launch": ("hipGraph replay of the whole step, %d batches in flight on separate HIP streams" % self.depth)
                if getattr(self, "_graph", None) is not None
                else "eager (graph capture failed: %s)" % getattr(self, "_graph_err", "not attempted"),
                "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                # data-dependent work: the SharedMLPs run over the DISTINCT (centre, sample) pairs of the ball-query lists (bit-identical
                # to all m * nsample rows; these synthetic clouds: 1.0-2.1 distinct of 16 / 32, profiles/r02_neighbour_list_fill.txt)
                "sharedmlp_rows": ("distinct pairs of the ball-query lists (WS3D_COMPACT_PAIRS=0: all m*nsample rows)"
                                   if fastpath.COMPACT_PAIRS and fastpath.PER_POINT_L1 else "all m*nsample rows")}

    @torch.no_grad()
    def _body(self, pts=None):
        out = self.model.rpn_forward({'pts_input': self.pts if pts is None else pts})
        boxes, scores, count = proposals_from_rpn(out, self.cfg)
        feats = out['backbone_features'].transpose(1, 2).contiguous()
        pooled, empty = roipool3d_ops.roipool3d_gpu(out['backbone_xyz'], feats, boxes, self.cfg.roi_extra_width,
                                                    sampled_pt_num=self.cfg.roi_sampled_pts)
        return out, boxes, scores, count, pooled, empty

    def capture(self):
        """Throughput mode = the product's ``ws3d_amd.pipeline.Stage1Pipeline``: the whole step (all
        torch ops + every C-ABI launch: no entry point allocates or synchronises) captured into one
        hipGraph per slot, `depth` batches in flight on separate HIP streams; returns True on success."""
        from ws3d_amd.pipeline import Stage1Pipeline
        self.pipe = Stage1Pipeline(self.model, self.cfg, batch=self.B, n_points=self.pts.size(1), depth=self.depth,
                                   roipool=True, device=self.pts.device)
        for slot in self.pipe.slots:
            slot["inp"].copy_(self.pts)
        ok = self.pipe.capture_all()
        self._graph = True if ok else None
        self._graph_err = self.pipe.graph_error
        return ok

    @torch.no_grad()
    def step(self, timed=False, eager=False):
        if getattr(self, "_graph", None) is not None and not timed and not eager:
            ticket = self.pipe.submit()                    # inputs already resident in the slot's buffer
            slot = self.pipe.slots[ticket % self.depth]
            res = slot["out"]
            with torch.cuda.stream(slot["stream"]):
                gathered = wdist.all_gather_proposals(wdist.pack_proposals(res["boxes"], res["scores"]), res["count"],
                                                      self.B * self.world)
            self.last = (res["rpn"], res["boxes"], res["scores"], res["count"], res["pooled"], res["empty"], gathered)
            return
        self._timed = timed
        e = None
        from ws3d_amd import fastpath
        ahead = fastpath.GEOMETRY_AHEAD
        if timed:           # per-operator timers: one stream, so that an operator's time is its own
            fastpath.GEOMETRY_AHEAD = False
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
        try:
            out = self.model.rpn_forward({'pts_input': self.pts})
        finally:
            fastpath.GEOMETRY_AHEAD = ahead
        if timed:
            e[1].record()
        boxes, scores, count, enlarged = proposals_from_rpn(out, self.cfg, with_pool_boxes=True)
        if timed:
            e[2].record()
        feats = out['backbone_features'].transpose(1, 2).contiguous()
        pooled, empty = roipool3d_ops.roipool3d_gpu(out['backbone_xyz'], feats, boxes, self.cfg.roi_extra_width,
                                                    sampled_pt_num=self.cfg.roi_sampled_pts, enlarged=enlarged)
        if timed:
            e[3].record()
            self.ev.append(e)
        self._timed = False
        gathered = wdist.all_gather_proposals(wdist.pack_proposals(boxes, scores), count, self.B * self.world)
        self.last = (out, boxes, scores, count, pooled, empty, gathered)

    def latency_mode(self, n=10):
        """median ms per batch with ONE batch in flight (submit -> exchange -> wait), two ways of launching it:
        hipGraph replay of the whole step on one stream, and eager launches with the coordinate-only work (sampling chain of
        levels 2-4, ball-query lists, 3-NN) on side streams beside the GEMMs (ws3d_amd/fastpath.py _Geometry; a captured graph
        with such branches replays slower on this runtime, so the graph keeps one stream) -> (best ms, detail dict)"""
        import time

        def run(eager):
            torch.cuda.synchronize()
            for _ in range(2):
                self.step(eager=eager)
            torch.cuda.synchronize()
            ts = []
            for _ in range(n):                 # median of the per-batch times: eager launches are host-driven, one slow
                t0 = time.perf_counter()       # iteration (an allocator or interpreter hiccup) should not set the figure
                self.step(eager=eager)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            return float(np.median(ts))
        from ws3d_amd import fastpath
        detail = {"eager_side_streams_ms" if fastpath.GEOMETRY_AHEAD else "eager_ms": run(True)}
        if getattr(self, "_graph", None) is not None:
            detail["graph_replay_ms"] = run(False)
        return min(detail.values()), detail

    def dump(self, folder):
        """one more step, then this rank's own proposals and the gathered ones to <folder>/proposals_rank<r>.npz
        (tests/test_bench_contract.py compares them across ranks and with a single-process run)"""
        self.step()
        torch.cuda.synchronize()
        _, boxes, scores, count, _, _, gathered = self.last
        np.savez(os.path.join(folder, "proposals_rank%d.npz" % self.rank),
                 local=wdist.pack_proposals(boxes, scores).cpu().numpy(), local_count=count.cpu().numpy(),
                 gathered=gathered[0].cpu().numpy(), gathered_count=gathered[1].cpu().numpy())

    def _fps_lane_instr(self):
        from bench import fps_lane_instr
        n, tot = self.cfg.num_points, 0
        for m in self.cfg.npoints:
            tot += fps_lane_instr(n, m)
            n = m
        return tot

    def kernel_table(self):
        steps = max(len(self.ev), 1)
        cfg, B = self.cfg, self.B
        nn_b, interp_b = _interp_bytes(cfg)
        roi_b = cfg.rpn_post_nms_top_n * cfg.roi_sampled_pts * (3 + 128) * 4 + cfg.num_points * (3 + 128) * 4
        alg = {"fps": _fps_model_bytes(cfg) * B, "ball_query+group": _qg_bytes(cfg) * B, "three_nn": nn_b * B,
               "three_interpolate": interp_b * B, "roipool3d": roi_b * B,
               "nms(mask+sweep)": (cfg.rpn_pre_nms_top_n * 20 + cfg.rpn_pre_nms_top_n * 141 * 8) * B}
        rows = []
        for key, evs in self.op_ev.items():
            ms = float(sum(a.elapsed_time(b) for a, b in evs)) / steps
            row = {"name": key, "ms_per_step": ms, "launches_per_step": len(evs) / steps,
                   "alg_bytes_per_step": alg.get(key, 0), "traffic_key": None, "bound": "hbm"}
            if key == "fps":
                row.update({"bound": "valu", "lane_instr_per_step": self._fps_lane_instr() * B,
                            "comment": "4 levels 16384->4096->1024->256->64, one workgroup per scene (8 of 256 CUs).  lane_instr_per_step "
                                       "is the DENSE sweep's count (8 per point and step): level 1 runs the pruned kernel (fps_bucket.hip, "
                                       "~6 of 256 buckets updated per step), levels 2-4 the verified-prefix kernel (fps_nested.hip), so "
                                       "valu_frac here is a dense-equivalent rate, not issue-slot occupancy; the physical VALU roofline "
                                       "of the dense kernel is the c2 block's"})
            elif key in ("three_nn", "nms(mask+sweep)"):
                row["comment"] = "ALU-bound search / pair test; the HBM figure is for orientation only"
            rows.append(row)
        fwd = float(np.mean([a[0].elapsed_time(a[1]) for a in self.ev]))
        custom_in_fwd = sum(r["ms_per_step"] for r in rows if r["name"] in ("fps", "ball_query+group", "three_nn", "three_interpolate"))
        rows.append({"name": "torch (SharedMLP GEMMs with fused epilogues, BN-folded, cat, heads) [hipBLASLt/rocBLAS] + pool kernels",
                     "ms_per_step": max(fwd - custom_in_fwd, 0.0), "launches_per_step": 0, "alg_bytes_per_step": 0,
                     "traffic_key": None, "bound": "library"})
        self.breakdown = {"rpn_forward_ms": fwd,
                          "proposals_nms_ms": float(np.mean([a[1].elapsed_time(a[2]) for a in self.ev])),
                          "roipool_ms": float(np.mean([a[2].elapsed_time(a[3]) for a in self.ev]))}
        return rows

    def path_gbps(self, scenes_per_s_per_gpu):
        cfg = self.cfg
        am = _fps_model_bytes(cfg) + _qg_bytes(cfg) + sum(_interp_bytes(cfg))
        return {"custom_ops_a_model_bytes_per_scene": am, "a_model": am * scenes_per_s_per_gpu / 1e9,
                "a_model_frac_of_8TBs": am * scenes_per_s_per_gpu / 8.0e12, "breakdown_ms": getattr(self, "breakdown", None)}

    def scenes(self):
        return self.B

    def cpu_baseline(self):
        """The custom ops of ONE scene's Stage-1 forward on the CPU oracle port (the MLPs are
        torch/BLAS on both sides and are excluded): 4x FPS, 8x ball_query+group, 4x three_nn."""
        import oracle
        threads = max(1, min(oracle.max_threads(), len(os.sched_getaffinity(0))))
        oracle.set_threads(threads)
        from bench import host_info, repeat_for
        reps_of_batch = max(1, threads // max(self.B, 1))          # the batch tiled so that every core has a scene
        xyz = np.ascontiguousarray(np.tile(self.pc_host[:, :, :3], (reps_of_batch, 1, 1)))
        ns = xyz.shape[0]

        def one_pass():
            levels = [xyz]
            for k, m in enumerate(self.cfg.npoints):
                idx = oracle.furthest_point_sample(levels[-1], m)
                new = np.stack([levels[-1][b][idx[b]] for b in range(ns)])
                for r, s in zip(self.cfg.radius[k], self.cfg.nsample[k]):
                    oracle.ball_query(r, s, levels[-1], new)
                levels.append(new)
            for k in range(4, 0, -1):
                oracle.three_nn_dist2(levels[k - 1], levels[k])
        _, dt, reps = repeat_for(one_pass)
        oracle.set_threads(1)
        return {"value": ns * reps / dt, "unit": "scenes/s", "cores": threads, "kind": "port", "host": host_info(threads),
                "sample": f"{ns} scenes (the batch tiled {reps_of_batch}x) x {reps} pass(es): the search ops of the Stage-1 forward "
                          f"only (4 FPS, 8 ball queries, 4 three_nn) "
                          f"on oracle/ws3d_oracle.c with OpenMP, wall {dt:.2f} s; grouping copies and MLPs excluded"}
