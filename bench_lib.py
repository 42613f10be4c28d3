"""(part of bench.py) What the bench lines are computed from: the roofs and byte models of SURVEY.md 8(d), the op-level workloads
(c2: FPS + fused query / group; c5: roipool3d + NMS on dense scenes; s2: the Stage-2 SA shapes -- c3 and t1 live in bench_c3.py /
bench_t1.py), the committed counter passes (profiles/traffic*.json) and the per-kernel roofline rows.  bench.py re-exports
everything here (`from bench_lib import *`): scripts and tests keep addressing it as `bench.X`."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md (spec)
# VALU issue roof (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32;
# 256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz) in lane-instructions per second
VALU_PEAK = 256 * 4 * 32 * 2.4e9
# the FPS sweep's algorithmic VALU work per point and step: 3 v_sub + v_mul + 2 v_fma (squared distance),
# v_min (running min-distance), v_max (argmax candidate) -- DESIGN.md section 5.1
FPS_VALU_PER_POINT = 8
HEADLINE_METRIC = "KITTI scenes/sec (16384 pts) Stage-1 RPN fwd, 1/2/4/8 GPU; FPS+group HBM GB/s"   # = BASELINE.json "metric"


def fps_lane_instr(n, m):
    """algorithmic VALU lane-instructions of one scene's furthest_point_sample n -> m"""
    return (m - 1) * n * FPS_VALU_PER_POINT

# SURVEY.md 8(d) / BASELINE.md section 3: bytes per scene at config 2
N_PTS, M_PTS, NSAMPLE, RADIUS, C_FEAT = 16384, 4096, 64, 0.1, 1


def a_model_fps(n=N_PTS, m=M_PTS):
    return (m - 1) * n * 12 + m * 4


def a_min_fps(n=N_PTS, m=M_PTS):
    return n * 12 + m * 4


def a_rest(n=N_PTS, m=M_PTS, ns=NSAMPLE, c=C_FEAT):
    gather = 7 * m * 4
    bq = (n + m) * 12 + m * ns * 4
    group = m * ns * 4 + (3 + c) * n * 4 + (3 + c) * m * ns * 4
    return gather + bq + group


def host_info(threads):
    """the host the CPU leg ran on: model string, core counts, OpenMP threads used"""
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"cpu": model, "os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "omp_threads": threads}


def repeat_for(fn, min_seconds=8.0, max_reps=200):
    """run fn() until at least min_seconds of wall time have passed -> (first result, seconds, repetitions):
    the CPU leg is timed on a bounded sample of ~10 s whatever the workload's unit costs"""
    first, reps = None, 0
    t0 = time.perf_counter()
    while True:
        out = fn()
        reps += 1
        if first is None:
            first = out
        dt = time.perf_counter() - t0
        if dt >= min_seconds or reps >= max_reps:
            return first, dt, reps


class C2:
    """FPS + fused ball_query/group on `batch` scenes (pre-allocated outputs, compat-level calls)."""

    name = "c2_fps_ballquery_group"

    def __init__(self, batch, rank, kind="hdl64"):
        from ws3d_amd import compat, synth
        self.c = compat
        self.B = batch
        self.kind = kind
        pc = np.empty((batch, N_PTS, 4), dtype=np.float32)
        for s in range(batch):
            seed = 1000 * 2 + rank * batch + s
            pc[s] = synth.cloud(kind, N_PTS, seed)
        self.pc_host = pc
        self.xyz = torch.from_numpy(pc[:, :, :3].copy()).cuda()
        self.feat = torch.from_numpy(np.ascontiguousarray(pc[:, :, 3:].transpose(0, 2, 1))).cuda()
        self.idx = torch.empty((batch, M_PTS), dtype=torch.int32, device="cuda")
        self.new_xyz = torch.empty((batch, M_PTS, 3), dtype=torch.float32, device="cuda")
        self.nbr = torch.empty((batch, M_PTS, NSAMPLE), dtype=torch.int32, device="cuda")
        self.grouped = torch.empty((batch, 3 + C_FEAT, M_PTS, NSAMPLE), dtype=torch.float32, device="cuda")
        self.ev = []

    def step(self, timed=False):
        # (the grid binning needs the coordinates only, but issuing it on a side stream beside the sampling kernel does not pay
        # at this batch: two sampling workgroups fill a CU's LDS and wave slots, the binning workgroups queue behind them --
        # measured 7.27 vs 7.23 ms per step, and 6.96 vs 6.28 ms of sampling when the binning gets in first)
        c, B = self.c, self.B
        if timed:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
        c.furthest_point_sampling_gather(B, N_PTS, M_PTS, self.xyz, None, self.idx, self.new_xyz)
        if timed:
            e[1].record()
        c.query_and_group(B, N_PTS, M_PTS, C_FEAT, RADIUS, NSAMPLE, True, self.xyz, self.new_xyz, self.feat,
                          self.nbr, self.grouped, c.sort_points_x(self.xyz))
        if timed:
            e[2].record()
            self.ev.append(e)

    metric = "KITTI scenes/sec (16384 pts), fused FPS+ball_query+group path only (BASELINE configs[1]); FPS+group HBM GB/s"

    def kernel_table(self):
        fps = float(np.mean([a[0].elapsed_time(a[1]) for a in self.ev]))
        qg = float(np.mean([a[1].elapsed_time(a[2]) for a in self.ev]))
        # round 4: fps_rounds2_kernel (fps_bucket.hip) at every batch size -- one scene per CU, ceil(B / 256) waves of workgroups; 3.3 ms
        # per 512 scenes (round 3's fps_rounds_kernel 4.07, the dense two-scenes-per-CU kernel 6.10)
        pmc = fps_valu_pmc(self.B, self.kind)
        fps_row = {"name": "fps_rounds2_kernel (exact pruned sampling, two candidates per wave, up to 8 certified samples per record exchange; one scene per CU, "
                           "%d wave(s) of workgroups) (furthest_point_sample + gather)" % -(-self.B // 256), "ms_per_step": fps,
                   "launches_per_step": 1, "bound": "valu", "lane_instr_per_step": fps_lane_instr(N_PTS, M_PTS) * self.B,
                   "physical_lane_instr_per_step": None if pmc is None else pmc["sq_insts_valu_per_launch"] * 64.0,
                   "us_per_sample": fps * 1e3 / (M_PTS - 1) / -(-self.B // 256),
                   "alg_bytes_per_step": a_model_fps() * self.B,
                   "alg_bytes_min_per_step": a_min_fps() * self.B, "traffic_key": "fps_rounds2_kernel",
                   "comment": "chain-bound, not VALU-bound: a round of the kernel is box tests, bucket updates, a re-pick, a record exchange and the "
                              "certification of up to 8 samples (fps_bucket.hip); pruning leaves ~1/7 of the dense sweep's instructions.  valu_frac = "
                              "ISSUED wave64 VALU instructions x 64 lanes (SQ_INSTS_VALU of the committed --pmc pass of this batch and generator, "
                              "profiles/traffic_fps_valu.json) / the duration measured here / the VALU issue roof; dense_equivalent_valu_frac counts "
                              "the dense sweep's %d lane-instructions per point and step instead (what the VALU-bound dense kernel issues: 0.55-0.57 "
                              "of the roof at 6.1-6.2 ms per 512 scenes, 1.9 x slower than this kernel; above 1: faster than a VALU-bound "
                              "dense sweep could be)" % FPS_VALU_PER_POINT}
        return [
            fps_row,
            {"name": "bin_points_grid + ball_query_grid_coop_kernel<fused> (ball_query, one wave per centre, + group + centre + cat)", "ms_per_step": qg,
             "launches_per_step": 2, "bound": "hbm", "alg_bytes_per_step": a_rest() * self.B, "traffic_key": "ball_query_grid_coop_kernel",
             "comment": "A_model == A_min for this kernel (every input read once, every output written once)"},
        ]

    def path_gbps(self, scenes_per_s_per_gpu):
        am, an = (a_model_fps() + a_rest()), (a_min_fps() + a_rest())
        return {"a_model": am * scenes_per_s_per_gpu / 1e9, "a_model_frac_of_8TBs": am * scenes_per_s_per_gpu / HBM_PEAK,
                "a_min": an * scenes_per_s_per_gpu / 1e9, "a_model_bytes_per_scene": am, "a_min_bytes_per_scene": an,
                "fps_steps_per_s_per_scene": None if not self.ev else (M_PTS - 1) / (self.kernel_table()[0]["ms_per_step"] * 1e-3)}

    def scenes(self):
        return self.B

    def cpu_baseline(self, min_seconds=6.0):
        """CPU oracle port (same arithmetic, OpenMP over scenes/centres) on a bounded sample."""
        import oracle
        threads = max(1, min(oracle.max_threads(), len(os.sched_getaffinity(0))))
        oracle.set_threads(threads)
        ns = int(min(self.B, max(2, 4 * threads)))
        pc = self.pc_host[:ns]
        xyz = np.ascontiguousarray(pc[:, :, :3])
        feat = np.ascontiguousarray(pc[:, :, 3:].transpose(0, 2, 1))
        xyz_t = np.ascontiguousarray(xyz.transpose(0, 2, 1))

        def one_pass():
            idx = oracle.furthest_point_sample(xyz, M_PTS)
            new_xyz = np.stack([xyz[b][idx[b]] for b in range(ns)])
            nbr = oracle.ball_query(RADIUS, NSAMPLE, xyz, new_xyz)
            gx = oracle.grouping_operation(xyz_t, nbr)
            gx -= new_xyz.transpose(0, 2, 1)[..., None]
            gf = oracle.grouping_operation(feat, nbr)
            return idx, nbr, gx, gf
        (idx, nbr, gx, gf), dt, reps = repeat_for(one_pass, min_seconds)
        oracle.set_threads(1)
        # parity spot-check of the GPU result of the last step against the same oracle run
        ok = bool(np.array_equal(self.idx[:ns].cpu().numpy(), idx) and
                  np.array_equal(self.nbr[:ns].cpu().numpy(), nbr) and
                  np.array_equal(self.grouped[:ns, :3].cpu().numpy(), gx) and
                  np.array_equal(self.grouped[:ns, 3:].cpu().numpy(), gf))
        return {"value": ns * reps / dt, "unit": "scenes/s", "cores": threads, "kind": "port",
                "sample": f"{ns} of the {self.B} scenes of this workload x {reps} pass(es), wall {dt:.2f} s "
                          f"(oracle/ws3d_oracle.c, literal FPS emulation, OpenMP over scenes/centres)",
                "host": host_info(threads), "gpu_matches_oracle_on_sample": ok}


class C5:
    """BASELINE.json configs[4]: dense scan stress -- N=65536 points, 512 proposals per scene,
    roipool3d (S=512, C=128) + rotated NMS (thresh 0.7); `batch` scenes per launch."""

    name = "c5_roipool3d_nms_dense"
    metric = "scenes/sec, roipool3d + rotated NMS at N=65536, 512 proposals/scene; roipool HBM GB/s"
    N, M, C, S, THR = 65536, 512, 128, 512, 0.7

    def __init__(self, batch, rank, kind="lidar"):
        from ws3d_amd import compat, kitti_utils, synth
        self.c, self.B, self.kind = compat, batch, kind
        pc = np.stack([synth.cloud(kind, self.N, 1000 * 5 + rank * batch + s) for s in range(batch)])
        # proposals near the cars that hold points AFTER the generator's crop (hdl64: image frustum + occlusion leave about half of
        # the 15 cars without a return; round 3's boxes sat on all of them and 54 % of the RoIs were empty)
        boxes = np.stack([synth.proposal_boxes_on_scene(pc[s], synth.random_boxes3d(15, (1000 * 5 + rank * batch + s) * 7919 + 13), self.M,
                                                        1000 * 5 + rank * batch + s) for s in range(batch)])
        self.pc_host, self.boxes_host = pc, boxes
        self.xyz = torch.from_numpy(pc[:, :, :3].copy()).cuda()
        self.feat = torch.randn((batch, self.N, self.C), device="cuda")
        enl = kitti_utils.enlarge_box3d(torch.from_numpy(boxes).view(-1, 7), 1.0).view(batch, self.M, 7)
        self.boxes = enl.cuda().contiguous()
        self.pooled = torch.zeros((batch, self.M, self.S, 3 + self.C), device="cuda")
        self.empty = torch.zeros((batch, self.M), dtype=torch.int32, device="cuda")
        scores = np.stack([synth.distinct_scores(self.M, 50 + b) for b in range(batch)])
        order = np.argsort(-scores, axis=1, kind="stable")
        bev = np.stack([synth.boxes3d_to_bev(boxes[b])[order[b]] for b in range(batch)])
        self.bev_sorted = torch.from_numpy(np.ascontiguousarray(bev)).cuda()
        self.ev = []

    def config(self):
        return {"n_points": self.N, "proposals": self.M, "channels": self.C, "sampled": self.S, "nms_thresh": self.THR}

    def step(self, timed=False):
        if timed:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
        self.c.roipool3d_forward(self.xyz, self.boxes, self.feat, self.pooled, self.empty)
        if timed:
            e[1].record()
        self.keep, self.num = self.c.nms_device_batched(self.bev_sorted, self.THR, False, 0)
        if timed:
            e[2].record()
            self.ev.append(e)

    def scenes(self):
        return self.B

    def kernel_table(self):
        roi = float(np.mean([a[0].elapsed_time(a[1]) for a in self.ev]))
        nms = float(np.mean([a[1].elapsed_time(a[2]) for a in self.ev]))
        nonempty = int((self.empty == 0).sum().item())
        # bytes the kernel has to move, from what THIS batch holds (measured on the device): the rows of the non-empty RoIs are
        # written once (rows of empty RoIs stay untouched by contract), xyz is scanned once, and of the features only the rows of
        # points that were pooled by some RoI are needed (distinct pooled points, counted by their coordinates)
        distinct = sum(int(torch.unique(self.pooled[b, :, :, :3].reshape(-1, 3), dim=0).size(0)) for b in range(self.B))
        roi_bytes = nonempty * self.S * (3 + self.C) * 4 + self.B * self.N * 12 + distinct * (3 + self.C) * 4 + self.B * self.M * (28 + 4)
        survey_bytes = (self.M * self.S * (3 + self.C) * 4 + self.N * (3 + self.C) * 4) * self.B
        self._roi_bytes_per_scene = roi_bytes / self.B
        return [
            {"name": "roipool3d_kernel (select + wrap-pad + copy, fused)", "ms_per_step": roi, "launches_per_step": 1,
             "alg_bytes_per_step": roi_bytes, "traffic_key": "c5:roipool3d_kernel",
             "non_empty_rois": nonempty, "rois": self.B * self.M, "distinct_pooled_points": distinct,
             "survey_a_min_bytes_per_step": survey_bytes,
             "comment": "alg_bytes = rows of the %d non-empty RoIs of %d written once + xyz scanned once + the feature rows of the %d distinct "
                        "pooled points read once (measured on this batch); SURVEY 8d's A_min (171,720,704 B/scene) assumes every RoI "
                        "non-empty and every feature row needed" % (nonempty, self.B * self.M, distinct)},
            {"name": "nms_rot_mask_kernel + nms_sweep_kernel (n=512)", "ms_per_step": nms, "launches_per_step": 2,
             "alg_bytes_per_step": (self.M * 20 + self.M * 8 * 8) * self.B, "traffic_key": None,
             "comment": "ALU-bound: %d box pairs per scene" % (self.M * (self.M - 1) // 2)},
        ]

    def path_gbps(self, scenes_per_s_per_gpu):
        b = getattr(self, "_roi_bytes_per_scene", None)          # what this batch needs moved (kernel_table; non-empty RoIs only)
        survey = self.M * self.S * (3 + self.C) * 4 + self.N * (3 + self.C) * 4
        return {"roipool_bytes_per_scene_this_batch": b, "a_min": None if b is None else b * scenes_per_s_per_gpu / 1e9,
                "roipool_survey_a_min_bytes_per_scene": survey,
                "nms_pairs_per_s": self.M * (self.M - 1) // 2 * scenes_per_s_per_gpu}

    def cpu_baseline(self, min_seconds=6.0):
        import oracle
        threads = max(1, min(oracle.max_threads(), len(os.sched_getaffinity(0))))
        oracle.set_threads(threads)
        ns = int(min(self.B, 4))
        boxes, feat = self.boxes[:ns].cpu().numpy(), self.feat[:ns].cpu().numpy()
        bev = [self.bev_sorted[b].cpu().numpy() for b in range(ns)]

        def one_pass():
            pooled, empty = oracle.roipool3d(self.pc_host[:ns, :, :3], boxes, feat, self.S)
            return pooled, empty, [oracle.nms_sorted(bev[b], self.THR, False) for b in range(ns)]
        (pooled, empty, keep), dt, reps = repeat_for(one_pass, min_seconds)
        oracle.set_threads(1)
        ok = bool(np.array_equal(self.empty[:ns].cpu().numpy(), empty) and np.array_equal(self.pooled[:ns].cpu().numpy(), pooled) and
                  all(np.array_equal(self.keep[b, :int(self.num[b])].cpu().numpy(), keep[b]) for b in range(ns)))
        return {"value": ns * reps / dt, "unit": "scenes/s", "cores": threads, "kind": "port",
                "sample": f"{ns} scenes x {reps} passes (roipool3d 65536 pts x 512 boxes + NMS 512) on oracle/ws3d_oracle.c, "
                          f"OpenMP over boxes, wall {dt:.2f} s", "host": host_info(threads), "gpu_matches_oracle_on_sample": ok}


class S2:
    """Stage-2 (RCNN) set-abstraction shapes of the same ops (lib/config.py:122-129):
    `batch` RoI clouds of 512 points with 128 feature channels; SA1 = FPS 512->128 + fused
    ball_query/group (r=0.2, ns=64), SA2 = FPS 128->32 + fused query (r=0.4, ns=64) on 128-channel
    features, SA3 = GroupAll.  Thousands of tiny scenes: FPS runs one wave per cloud and the grouped
    tensors (4.3 MB per RoI) make this the HBM-write-bound regime of the grouping kernel."""

    name = "s2_rcnn_sa_shapes"
    unit = "RoI clouds/s"
    metric = "RoI clouds/sec (512 pts, 128 ch), Stage-2 SA ops (FPS + fused ball_query/group x2 + GroupAll); group HBM GB/s"
    N, C, M1, M2, NS, R1, R2 = 512, 128, 128, 32, 64, 0.2, 0.4

    def __init__(self, batch, rank, kind="lidar"):
        from ws3d_amd import compat, pn2_ops, synth
        self.c, self.pn, self.B = compat, pn2_ops, batch
        self.kind = "roi_clouds"     # synth.roi_clouds: points pooled for one proposal, whatever --kind says
        self.pts_host = synth.roi_clouds(batch, self.N, 6 + rank)
        B, N, C, M1, M2, NS = batch, self.N, self.C, self.M1, self.M2, self.NS
        self.xyz = torch.from_numpy(self.pts_host).cuda()
        g = torch.Generator().manual_seed(1234 + rank)
        self.feat = torch.randn((B, C, N), generator=g).cuda()
        self.feat2 = torch.randn((B, C, M1), generator=g).cuda()      # stands for the SA1 MLP output
        self.feat3 = torch.randn((B, 2 * C, M2), generator=g).cuda()  # ... SA2 MLP output
        self.idx1 = torch.empty((B, M1), dtype=torch.int32, device="cuda"); self.new1 = torch.empty((B, M1, 3), device="cuda")
        self.idx2 = torch.empty((B, M2), dtype=torch.int32, device="cuda"); self.new2 = torch.empty((B, M2, 3), device="cuda")
        self.nbr1 = torch.empty((B, M1, NS), dtype=torch.int32, device="cuda")
        self.nbr2 = torch.empty((B, M2, NS), dtype=torch.int32, device="cuda")
        self.out1 = torch.empty((B, 3 + C, M1, NS), device="cuda")
        self.out2 = torch.empty((B, 3 + C, M2, NS), device="cuda")
        self.group_all = self.pn.GroupAll(use_xyz=True)
        self.ev = []

    def config(self):
        return {"n_points": self.N, "channels": self.C, "npoint": [self.M1, self.M2, None], "radius": [self.R1, self.R2, 100],
                "nsample": self.NS}

    def step(self, timed=False):
        c, B = self.c, self.B
        if timed:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            e[0].record()
        c.furthest_point_sampling_gather(B, self.N, self.M1, self.xyz, None, self.idx1, self.new1)
        if timed: e[1].record()
        c.query_and_group(B, self.N, self.M1, self.C, self.R1, self.NS, True, self.xyz, self.new1, self.feat, self.nbr1,
                          self.out1, None)
        if timed: e[2].record()
        c.furthest_point_sampling_gather(B, self.M1, self.M2, self.new1, None, self.idx2, self.new2)
        if timed: e[3].record()
        c.query_and_group(B, self.M1, self.M2, self.C, self.R2, self.NS, True, self.new1, self.new2, self.feat2, self.nbr2,
                          self.out2, None)
        if timed: e[4].record()
        self.out3 = self.group_all(self.new2, None, self.feat3)
        if timed:
            e[5].record()
            self.ev.append(e)

    def scenes(self):
        return self.B

    def _bytes(self, n, m, c):
        return (3 + c) * m * self.NS * 4 + m * self.NS * 4 + (3 + c) * n * 4 + m * 12

    def kernel_table(self):
        t = [float(np.mean([a[i].elapsed_time(a[i + 1]) for a in self.ev])) for i in range(5)]
        B = self.B
        return [
            {"name": "ball_query_kernel<fused> SA1 (512 -> 128 x 64, 3+128 ch)", "ms_per_step": t[1], "launches_per_step": 1,
             "alg_bytes_per_step": self._bytes(self.N, self.M1, self.C) * B, "traffic_key": None,
             "comment": "A_model == A_min: grouped tensor + neighbour indices written once, xyz/features read once"},
            {"name": "ball_query_kernel<fused> SA2 (128 -> 32 x 64, 3+128 ch)", "ms_per_step": t[3], "launches_per_step": 1,
             "alg_bytes_per_step": self._bytes(self.M1, self.M2, self.C) * B, "traffic_key": None, "comment": ""},
            {"name": "fps_reg_kernel<*,64> SA1+SA2 (one wave per cloud)", "ms_per_step": t[0] + t[2], "launches_per_step": 2,
             "bound": "valu", "lane_instr_per_step": (fps_lane_instr(self.N, self.M1) + fps_lane_instr(self.M1, self.M2)) * B,
             "alg_bytes_per_step": ((self.M1 - 1) * self.N * 12 + (self.M2 - 1) * self.M1 * 12) * B, "traffic_key": None,
             "comment": "A_model (re-read per step) in alg_bytes; real traffic is the 6 KB cloud once; the physical fraction is valu_frac"},
            {"name": "GroupAll (torch cat)", "ms_per_step": t[4], "launches_per_step": 0,
             "alg_bytes_per_step": 2 * (3 + 2 * self.C) * self.M2 * 4 * B, "traffic_key": None, "comment": "not ours"},
        ]

    def path_gbps(self, rois_per_s_per_gpu):
        b = self._bytes(self.N, self.M1, self.C) + self._bytes(self.M1, self.M2, self.C)
        return {"group_a_min_bytes_per_roi": b, "a_min": b * rois_per_s_per_gpu / 1e9,
                "a_min_frac_of_8TBs": b * rois_per_s_per_gpu / HBM_PEAK}

    def cpu_baseline(self, min_seconds=6.0):
        import oracle
        threads = max(1, min(oracle.max_threads(), len(os.sched_getaffinity(0))))
        oracle.set_threads(threads)
        ns = int(min(self.B, 8 * threads))
        xyz = np.ascontiguousarray(self.pts_host[:ns])
        feat, feat2 = self.feat[:ns].cpu().numpy(), self.feat2[:ns].cpu().numpy()

        def one_pass():
            i1 = oracle.furthest_point_sample(xyz, self.M1)
            n1 = np.stack([xyz[b][i1[b]] for b in range(ns)])
            q1 = oracle.ball_query(self.R1, self.NS, xyz, n1)
            g1 = oracle.grouping_operation(feat, q1)
            i2 = oracle.furthest_point_sample(n1, self.M2)
            n2 = np.stack([n1[b][i2[b]] for b in range(ns)])
            q2 = oracle.ball_query(self.R2, self.NS, n1, n2)
            g2 = oracle.grouping_operation(feat2, q2)
            return i1, q1, g1, i2, q2, g2
        (i1, q1, g1, i2, q2, g2), dt, reps = repeat_for(one_pass, min_seconds)
        oracle.set_threads(1)
        ok = bool(np.array_equal(self.idx1[:ns].cpu().numpy(), i1) and np.array_equal(self.nbr1[:ns].cpu().numpy(), q1) and
                  np.array_equal(self.out1[:ns, 3:].cpu().numpy(), g1) and np.array_equal(self.idx2[:ns].cpu().numpy(), i2) and
                  np.array_equal(self.nbr2[:ns].cpu().numpy(), q2) and np.array_equal(self.out2[:ns, 3:].cpu().numpy(), g2))
        return {"value": ns * reps / dt, "unit": "RoI clouds/s", "cores": threads, "kind": "port",
                "sample": f"{ns} of the {self.B} RoI clouds x {reps} passes, both SA levels (FPS, ball query, feature grouping) on "
                          f"oracle/ws3d_oracle.c with OpenMP, wall {dt:.2f} s", "host": host_info(threads),
                "gpu_matches_oracle_on_sample": ok}


C2.config = lambda self: {"m_points": M_PTS, "radius": RADIUS, "nsample": NSAMPLE, "channels": "3 xyz + 1 feature"}


def step_percentiles(wl):
    """p10 / p50 / p90 of the per-step device time (first to last HIP event of a step) of the timed
    steps, when the workload records per-step events (SURVEY 8d timing method)"""
    ev = getattr(wl, "ev", None)
    if not ev:
        return None
    t = np.array([e[0].elapsed_time(e[-1]) for e in ev])
    return {"p10": float(np.percentile(t, 10)), "p50": float(np.percentile(t, 50)), "p90": float(np.percentile(t, 90)),
            "n": int(t.size)}


def git_blob_sha1(path):
    """what `git hash-object <path>` prints (sha1 of 'blob <size>\\0' + content)"""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


FPS_KERNEL_SOURCES = ("ws3d_amd/csrc/fps_bucket.hip", "ws3d_amd/csrc/common.h")   # what the instruction count of the level-1 kernel depends on
_PMC_STALE = []


def fps_valu_pmc(batch, kind):
    """SQ_INSTS_VALU (wave64 VALU instructions per launch) of the level-1 sampling kernel at this batch and generator from the committed
    --pmc pass (scripts/pmc_fps_valu.sh -> profiles/traffic_fps_valu.json), or None.  The count is a property of the data (same seeds
    here and there) AND of the kernel's code: the pass records the git blob hashes of the kernel's sources, and a count taken on other
    sources than the ones on disk is refused (None -> roofline.frac null, with a warning on stderr) instead of silently going stale."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "traffic_fps_valu.json")))
        want = j.get("_source_blobs") or {}
        have = {f: git_blob_sha1(os.path.join(ROOT, f)) for f in FPS_KERNEL_SOURCES}
        if want != have:
            if not _PMC_STALE:
                _PMC_STALE.append(True)
                print("bench.py: WARNING profiles/traffic_fps_valu.json was taken on other kernel sources (%s) than the ones on disk (%s): "
                      "the physical VALU fraction is not reported; re-run scripts/pmc_fps_valu.sh" % (want, have), file=sys.stderr, flush=True)
            return None
        e = j.get("b%d_%s" % (batch, kind))
        return e if e and e.get("sq_insts_valu_per_launch") else None
    except Exception:
        return None


def _traffic_file(kernel_key):
    """'fps_zlds_kernel' -> (profiles/traffic.json, key); 'c5:roipool3d_kernel' -> (profiles/traffic_c5.json, key);
    'fps_bucket_valu:hdl64' -> (profiles/traffic_fps_bucket_valu.json, 'hdl64')"""
    if kernel_key and ":" in kernel_key:
        tag, key = kernel_key.split(":", 1)
        return os.path.join(ROOT, "profiles", "traffic_%s.json" % tag), key
    return os.path.join(ROOT, "profiles", "traffic.json"), kernel_key


# The sources a counter pass describes: the pass records their git blob hashes (`_source_blobs`) and a figure
# taken on other sources than the ones on disk is NOT quoted -- `traffic: null`, `traffic_source: "stale"` -- instead of silently
# describing code that no longer exists.  c2 (traffic.json): sampling + ball query + binning; c5: roipool3d + iou3d; c3: every kernel
# source and the forward pass's dispatch logic.
_CSRC = "ws3d_amd/csrc/"
TRAFFIC_SOURCES = {
    "traffic.json": tuple(_CSRC + f for f in ("fps_bucket.hip", "ballquery_group.hip", "bin_kernels.h", "binning.h", "common.h")),
    "traffic_c5.json": tuple(_CSRC + f for f in ("roipool3d.hip", "iou3d.hip", "common.h")),
    "traffic_ops.json": tuple(_CSRC + f for f in ("ballquery_group.hip", "interpolate.hip", "fps.hip", "common.h")),
    "traffic_ops256.json": tuple(_CSRC + f for f in ("ballquery_group.hip", "interpolate.hip", "fps.hip", "common.h")),
    "traffic_c3.json": None,          # all of csrc/*.hip, csrc/*.h + ws3d_amd/fastpath.py (resolved in traffic_source_blobs)
}


def traffic_source_blobs(file_name):
    """{repo-relative path: git blob sha1} of the sources the counter file `file_name` (profiles/<file_name>) describes"""
    files = TRAFFIC_SOURCES.get(os.path.basename(file_name))
    if files is None:
        d = os.path.join(ROOT, "ws3d_amd", "csrc")
        files = tuple(sorted(_CSRC + f for f in os.listdir(d) if f.endswith((".hip", ".h")))) + ("ws3d_amd/fastpath.py",)
    return {f: git_blob_sha1(os.path.join(ROOT, f)) for f in files}


_TRAFFIC_STATE = {}


def traffic_state(kernel_key=None):
    """'fresh' (the pass names the sources on disk), 'stale' (it names others, or none: taken before round 6), 'absent'"""
    p = _traffic_file(kernel_key)[0]
    if p not in _TRAFFIC_STATE:
        try:
            j = json.load(open(p))
            want = j.get("_source_blobs")
            _TRAFFIC_STATE[p] = "fresh" if want and want == traffic_source_blobs(p) else "stale"
            if _TRAFFIC_STATE[p] == "stale":
                print("bench.py: WARNING %s was taken on other kernel sources than the ones on disk: its figures are not quoted "
                      "(traffic: null, traffic_source: stale); re-run scripts/gpu_refresh_lite.sh" % os.path.relpath(p, ROOT), file=sys.stderr, flush=True)
        except Exception:
            _TRAFFIC_STATE[p] = "absent"
    return _TRAFFIC_STATE[p]


def traffic_batch(kernel_key=None):
    try:
        return int(json.load(open(_traffic_file(kernel_key)[0])).get("_scenes_per_launch", 256))
    except Exception:
        return -1


def load_traffic(kernel_key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic*.json),
    already corrected as MI355X_MICROARCH.md prescribes; None when not measured OR measured on other sources (traffic_state)."""
    p, key = _traffic_file(kernel_key)
    if key and os.path.exists(p) and traffic_state(kernel_key) == "fresh":
        try:
            return json.load(open(p)).get(key)
        except Exception:
            return None
    return None


def traffic_kind(kernel_key=None):
    try:
        return json.load(open(_traffic_file(kernel_key)[0])).get("_kind")
    except Exception:
        return None


def finish_kernel_rows(kernels, scenes, kind=None):
    """derived figures of every kernel row: algorithmic GB/s, and the PHYSICAL roofline fraction of the
    bound that applies (VALU issue for FPS, HBM for the copy/search kernels)"""
    for k in kernels:
        sec = k["ms_per_step"] * 1e-3
        gbps = k["alg_bytes_per_step"] / sec / 1e9 if sec > 0 else 0.0
        if str(k.get("bytes_model", "")).startswith("reference op"):
            # the bytes are the REFERENCE operator's model, the time is that of own kernels that move far less: an
            # effective figure, not a fraction of the HBM roof
            k["effective_GBps"] = gbps
        elif k.get("bound") == "valu":
            # FPS: alg_bytes = SURVEY 8d's A_model (the scene re-read every step, the north-star's accounting); the kernel reads the
            # scene once, so this is an effective figure too and may exceed the roof
            k["effective_GBps_a_model"] = gbps
            k["effective_frac_a_model"] = gbps * 1e9 / HBM_PEAK
        else:
            k["achieved_GBps"] = gbps
            k["frac_of_8TBps"] = gbps * 1e9 / HBM_PEAK
        if k.get("bound") == "valu" and sec > 0:
            k["dense_equivalent_valu_frac"] = k["lane_instr_per_step"] / sec / VALU_PEAK
            phys = k.get("physical_lane_instr_per_step")
            # physical = instructions actually issued (hardware counter of the committed pass) / the duration measured in this run;
            # without a committed pass for this batch and generator only the dense-equivalent figure exists
            k["valu_frac_is"] = "physical (SQ_INSTS_VALU x 64 / duration / roof)" if phys else \
                "null: no committed --pmc pass of THESE kernel sources for this batch / generator (dense_equivalent_valu_frac is the model figure)"
            k["valu_lane_instr_per_s"] = phys / sec if phys else None
            k["valu_frac"] = phys / sec / VALU_PEAK if phys else None
        tkey = k.pop("traffic_key", None)
        tr = load_traffic(tkey)
        if tkey and tr is None:
            k["traffic_state"] = traffic_state(tkey)
        # the committed PMC passes were taken at one batch size: only comparable at that batch
        k["traffic_bytes_per_launch"] = tr if scenes == traffic_batch(tkey) and traffic_kind(tkey) in (None, kind) else None
    return kernels


def roofline_of(k, where):
    """the tier contract's roofline object for kernel row k (after finish_kernel_rows)"""
    launches = max(k["launches_per_step"], 1)
    traffic = k["traffic_bytes_per_launch"].get("hbm_bytes") if isinstance(k.get("traffic_bytes_per_launch"), dict) else None
    sec = k["ms_per_step"] * 1e-3
    r = {"kernel": k["name"], "measured_in": where, "ms_per_launch": k["ms_per_step"] / launches, "traffic": traffic}
    if traffic is None and k.get("traffic_state"):
        r["traffic_source"] = k["traffic_state"]
    if k.get("bound") == "valu":
        have = k["valu_lane_instr_per_s"] is not None
        r.update({"bound": "valu", "achieved": k["valu_lane_instr_per_s"] / 1e12 if have else None, "peak": VALU_PEAK / 1e12, "unit": "Tlane-instr/s",
                  "frac": k["valu_frac"], "frac_is": k.get("valu_frac_is"),
                  "lane_instr_per_launch": k["valu_lane_instr_per_s"] * sec / launches if have else None,
                  "dense_equivalent_frac": k.get("dense_equivalent_valu_frac"), "dense_sweep_lane_instr_per_launch": k["lane_instr_per_step"] / launches,
                  "us_per_sample": k.get("us_per_sample"), "valu_instr_per_point_and_step_of_the_dense_sweep": FPS_VALU_PER_POINT,
                  "effective_frac": k["effective_frac_a_model"], "effective_GBps_a_model": k["effective_GBps_a_model"],
                  "alg_bytes_per_launch_a_model": k["alg_bytes_per_step"] / launches,
                  "hbm_frac_physical": (traffic / (sec / launches) / HBM_PEAK) if traffic else None,
                  "note": "FPS never re-reads the scene, so HBM does not bound it; the roof that applies is VALU issue (MI355X_MICROARCH.md: a wave64 "
                          "instruction = 2 clk on a SIMD-32).  frac = VALU lane-instructions ISSUED (hardware counter) / duration / that roof: the "
                          "physical utilisation of a kernel that is deliberately NOT VALU-bound -- the sampling kernel prunes ~6/7 of the "
                          "dense sweep's instructions and is bound by its cross-lane chain; its figure of merit is us_per_sample.  dense_equivalent_frac "
                          "= the dense sweep's instruction count (8 per point and step) / duration / roof: what a VALU-bound kernel would need to reach "
                          "for the same speed.  effective_frac = SURVEY 8d's A_model bytes / duration / 8 TB/s (the north-star's accounting; exceeds 1 "
                          "because nothing is re-read).  traffic = FETCH_SIZE+WRITE_SIZE per launch from profiles/traffic*.json or null"})
    else:
        r.update({"bound": "hbm", "achieved": k["achieved_GBps"], "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": k["frac_of_8TBps"],
                  "alg_bytes_per_launch": k["alg_bytes_per_step"] / launches,
                  "note": "achieved = algorithmic bytes (SURVEY.md 8d byte model: every input read once, every output written once) / "
                          "measured duration; traffic = FETCH_SIZE+WRITE_SIZE per launch from profiles/traffic*.json or null"})
    return r


