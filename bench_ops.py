"""(part of bench.py) `--workload ops`: what a DROP-IN caller gets -- the API-named operators of the three pybind
modules called one by one through ws3d_amd.compat with the reference's positional arguments (pointnet2_api.cpp:10-24, iou3d.cpp:174-179),
at Stage-1 shapes, each timed on its own with HIP events:

  gather_points_wrapper      xyz^T (B, 3, 16384) by the 4096 sampled indices                      sampling_gpu.cu:8-24      (copy: HBM roof)
  group_points_wrapper       SA2's grouping as pointnet2_utils.py:241-264 issues it: xyz^T (3 ch) and the features (96 ch) of 4096 points,
                             1024 centres x nsample 16 and 32 -- four launches                    group_points_gpu.cu:47-66 (copy: HBM roof)
  three_interpolate_wrapper  FP0: 256 channels of 4096 known points onto 16384 points             interpolate_gpu.cu:77-97  (copy: HBM roof)
  three_nn_wrapper           16384 queries against 4096 known points                              interpolate_gpu.cu:9-52   (search: queries / s)
  ball_query_wrapper         16384 -> 4096, r = 0.1 / ns = 16 and r = 0.5 / ns = 32                ball_query_gpu.cu:9-45    (search)
  boxes_overlap_bev_gpu      512 x 512 rotated BEV boxes                                          iou3d_kernel.cu:223-235   (ALU: pairs / s)

For the copy operators SURVEY 8(d) applies the 40 % HBM target literally (A_model = A_min: every input read once, every output written
once); `traffic` = counter bytes from the committed --pmc passes of this workload (profiles/traffic_ops.json, per batch) when they name
the sources on disk.  A "step" = one call of each operator on the batch; `value` = scenes / s of that sequence."""
from __future__ import annotations

import numpy as np
import torch

from bench_lib import HBM_PEAK, host_info

N, M1, M2, C2, CF = 16384, 4096, 1024, 96, 256


class OPS:
    name = "ops_standalone_api_operators"
    metric = "KITTI scenes/sec (16384 pts), the API-named operators called one by one (drop-in callers); copy operators' HBM GB/s"

    def __init__(self, batch, rank, kind="hdl64"):
        from ws3d_amd import compat, synth
        self.c, self.B, self.kind = compat, batch, kind
        base = np.stack([synth.cloud(kind, N, 1000 * 6 + s)[:, :3] for s in range(min(batch, 8))])
        pc = np.ascontiguousarray(np.tile(base, (-(-batch // base.shape[0]), 1, 1))[:batch])
        self.pc_host = pc
        g = torch.Generator(device="cuda").manual_seed(6)
        B = batch
        self.xyz = torch.from_numpy(pc).cuda()
        self.xyz_t = self.xyz.transpose(1, 2).contiguous()                                   # (B, 3, N): what gather / group take
        self.idx1 = torch.empty((B, M1), dtype=torch.int32, device="cuda")
        self.xyz1 = torch.empty((B, M1, 3), device="cuda")
        compat.furthest_point_sampling_gather(B, N, M1, self.xyz, None, self.idx1, self.xyz1)
        self.idx2 = torch.empty((B, M2), dtype=torch.int32, device="cuda")
        self.xyz2 = torch.empty((B, M2, 3), device="cuda")
        compat.furthest_point_sampling_gather(B, M1, M2, self.xyz1, None, self.idx2, self.xyz2)
        self.xyz1_t = self.xyz1.transpose(1, 2).contiguous()
        self.feat1 = torch.randn((B, C2, M1), device="cuda", generator=g)                    # level-1 features, channels first
        self.gathered = torch.empty((B, 3, M1), device="cuda")
        self.nbr2 = {}
        self.grouped = {}
        for ns, r in ((16, 0.5), (32, 1.0)):                                                 # SA2's two scales
            nb = torch.zeros((B, M2, ns), dtype=torch.int32, device="cuda")
            compat.ball_query_wrapper(B, M1, M2, r, ns, self.xyz2, self.xyz1, nb)
            self.nbr2[ns] = nb
            self.grouped[(3, ns)] = torch.empty((B, 3, M2, ns), device="cuda")
            self.grouped[(C2, ns)] = torch.empty((B, C2, M2, ns), device="cuda")
        self.known_f = torch.randn((B, CF, M1), device="cuda", generator=g)                  # FP0: features of the 4096 known points
        self.dist2 = torch.empty((B, N, 3), device="cuda")
        self.nn_idx = torch.empty((B, N, 3), dtype=torch.int32, device="cuda")
        compat.three_nn_wrapper(B, N, M1, self.xyz, self.xyz1, self.dist2, self.nn_idx)
        d = torch.sqrt(self.dist2)
        w = 1.0 / (d + 1e-8)
        self.weight = (w / w.sum(dim=2, keepdim=True)).contiguous()
        self.interp = torch.empty((B, CF, N), device="cuda")
        self.bq = {16: torch.zeros((B, M1, 16), dtype=torch.int32, device="cuda"), 32: torch.zeros((B, M1, 32), dtype=torch.int32, device="cuda")}
        rng = np.random.default_rng(66)
        nb = 512
        cx, cz = rng.uniform(-20, 20, nb), rng.uniform(5, 45, nb)
        l_, w_ = rng.uniform(3.5, 4.3, nb), rng.uniform(1.5, 1.8, nb)
        bev = np.stack([cx - l_ / 2, cz - w_ / 2, cx + l_ / 2, cz + w_ / 2, rng.uniform(-np.pi, np.pi, nb)], axis=1).astype(np.float32)
        self.bev = torch.from_numpy(bev).cuda()
        self.overlap = torch.empty((nb, nb), device="cuda")
        self.ev = []
        self.ops = [
            ("gather_points_wrapper (xyz^T by 4096 indices)", "copy", lambda: compat.gather_points_wrapper(B, 3, N, M1, self.xyz_t, self.idx1, self.gathered),
             B * (1 + 2 * 3) * M1 * 4, "gather_points_kernel"),
            ("group_points_wrapper x4 (SA2: 3 + 96 channels, nsample 16 + 32)", "copy", self._group,
             B * sum(c * M2 * ns * 4 + M2 * ns * 4 + c * M1 * 4 for c in (3, C2) for ns in (16, 32)), "group_points_kernel"),
            ("three_interpolate_wrapper (FP0: 256 ch, 4096 -> 16384)", "copy",
             lambda: compat.three_interpolate_wrapper(B, CF, M1, N, self.known_f, self.nn_idx, self.weight, self.interp),
             B * (CF * M1 * 4 + N * 3 * 8 + CF * N * 4), "three_interpolate_kernel"),
            ("three_nn_wrapper (16384 queries, 4096 known)", "search", lambda: compat.three_nn_wrapper(B, N, M1, self.xyz, self.xyz1, self.dist2, self.nn_idx),
             B * ((N + M1) * 12 + N * 3 * 8), None),
            ("ball_query_wrapper x2 (16384 -> 4096: r 0.1 / ns 16, r 0.5 / ns 32)", "search", self._bq,
             B * sum((N + M1) * 12 + M1 * ns * 4 for ns in (16, 32)), None),
            ("boxes_overlap_bev_gpu (512 x 512)", "alu", lambda: compat.boxes_overlap_bev_gpu(self.bev, self.bev, self.overlap), 2 * nb * 20 + nb * nb * 4, None),
        ]

    def _group(self):
        for ns in (16, 32):
            self.c.group_points_wrapper(self.B, 3, M1, M2, ns, self.xyz1_t, self.nbr2[ns], self.grouped[(3, ns)])
            self.c.group_points_wrapper(self.B, C2, M1, M2, ns, self.feat1, self.nbr2[ns], self.grouped[(C2, ns)])

    def _bq(self):
        self.c.ball_query_wrapper(self.B, N, M1, 0.1, 16, self.xyz1, self.xyz, self.bq[16])
        self.c.ball_query_wrapper(self.B, N, M1, 0.5, 32, self.xyz1, self.xyz, self.bq[32])

    def step(self, timed=False):
        e = []
        for _, _, fn, _, _ in self.ops:
            if timed:
                a = torch.cuda.Event(enable_timing=True)
                a.record()
                e.append(a)
            fn()
        if timed:
            a = torch.cuda.Event(enable_timing=True)
            a.record()
            e.append(a)
            self.ev.append(e)

    def scenes(self):
        return self.B

    def config(self):
        return {"operators": [o[0] for o in self.ops], "n_points": N, "shapes": "Stage-1 (weaklyRPN.yaml): 16384 -> 4096 -> 1024, SA2 96 + 3 channels, FP0 256 channels"}

    def kernel_table(self):
        rows = []
        for i, (name, kind, _, nbytes, tkey) in enumerate(self.ops):
            t = np.array([e[i].elapsed_time(e[i + 1]) for e in self.ev])
            row = {"name": name, "ms_per_step": float(np.median(t)), "ms_p10": float(np.percentile(t, 10)), "ms_p90": float(np.percentile(t, 90)),
                   "launches_per_step": 4 if name.startswith("group") else 2 if name.startswith("ball") else 1,
                   "alg_bytes_per_step": nbytes, "bound": "hbm" if kind == "copy" else kind,
                   "traffic_key": (("ops:" if self.B == 8 else "ops%d:" % self.B) + tkey) if tkey else None}      # profiles/traffic_ops.json (batch 8), traffic_ops256.json, ..
            if kind == "search":
                row["queries_per_s"] = self.B * (N if name.startswith("three_nn") else 2 * M1) / (row["ms_per_step"] * 1e-3)
            if kind == "alu":
                row["pairs_per_s"] = 512 * 512 / (row["ms_per_step"] * 1e-3)
            rows.append(row)
        return rows

    def path_gbps(self, scenes_per_s_per_gpu):
        if not self.ev:
            return {}
        out = {}
        for r in self.kernel_table():
            if r["bound"] == "hbm":
                out[r["name"].split(" ")[0]] = {"GBps": r["alg_bytes_per_step"] / (r["ms_per_step"] * 1e-3) / 1e9,
                                                "frac_of_8TBps": r["alg_bytes_per_step"] / (r["ms_per_step"] * 1e-3) / HBM_PEAK}
        return out

    def cpu_baseline(self, min_seconds=6.0):
        """the same operator sequence on the oracle port (OpenMP), on a bounded sample of the batch"""
        import os
        import time
        import oracle
        threads = max(1, min(oracle.max_threads(), len(os.sched_getaffinity(0))))
        oracle.set_threads(threads)
        ns_ = int(min(self.B, 8))
        xyz = self.pc_host[:ns_]
        xyz1, xyz2 = self.xyz1[:ns_].cpu().numpy(), self.xyz2[:ns_].cpu().numpy()
        feat1, known = self.feat1[:ns_].cpu().numpy(), self.known_f[:ns_].cpu().numpy()
        xyz_t, xyz1_t = np.ascontiguousarray(xyz.transpose(0, 2, 1)), np.ascontiguousarray(xyz1.transpose(0, 2, 1))
        idx1 = self.idx1[:ns_].cpu().numpy()
        t0 = time.perf_counter()
        reps = 0
        while True:
            oracle.gather_operation(xyz_t, idx1)
            for ns, r in ((16, 0.5), (32, 1.0)):
                nb = oracle.ball_query(r, ns, xyz1, xyz2)
                oracle.grouping_operation(xyz1_t, nb)
                oracle.grouping_operation(feat1, nb)
            d2, nn = oracle.three_nn_dist2(xyz, xyz1)
            oracle.three_interpolate(known, nn, self.weight[:ns_].cpu().numpy())
            b16 = oracle.ball_query(0.1, 16, xyz, xyz1)
            oracle.ball_query(0.5, 32, xyz, xyz1)
            reps += 1
            if time.perf_counter() - t0 >= min_seconds:
                break
        dt = time.perf_counter() - t0
        oracle.set_threads(1)
        ok = bool(np.array_equal(self.nn_idx[:ns_].cpu().numpy(), nn) and np.array_equal(self.bq[16][:ns_].cpu().numpy(), b16))
        return {"value": ns_ * reps / dt, "unit": "scenes/s", "cores": threads, "kind": "port",
                "sample": "%d scenes x %d pass(es) of the same operator sequence (without the 512 x 512 overlap and with SA2's two searches), wall %.2f s; "
                          "oracle/ws3d_oracle.c, OpenMP" % (ns_, reps, dt), "host": host_info(threads), "gpu_matches_oracle_on_sample": ok}
