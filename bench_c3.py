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


def _bq_bytes(cfg):
    """lists only (+ their pair table): points and centres in, nsample indices per centre out, up to two ints per list entry of pairs"""
    n, tot = cfg.num_points, 0
    for k, m in enumerate(cfg.npoints):
        for ns in cfg.nsample[k]:
            tot += (n + m) * 12 + m * ns * 4 * 3
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


# HIP-event timers around every C-ABI launch family of the step (ws3d_amd.compat is what fastpath / stage1 / roipool3d_ops call; the
# wrappers are installed once and record only while the ACTIVE workload runs a timed step).  What is left of the forward pass after
# these families is library work: Tensile GEMMs and at::native glue.
FAMILIES = {
    "furthest_point_sampling_gather": "fps level 1 (16384 -> 4096)", "furthest_point_sampling_nested": "fps levels 2-4 (verified prefix)", "furthest_point_sampling_nested_chain": "fps levels 2-4 (verified prefix)",
    "split_points_clear": "prologue (input split + zero arena, one launch)",
    "sort_points_x": "binning (grid / x slabs / xz grid)", "sort_points_xz": "binning (grid / x slabs / xz grid)", "sort_points_jobs": "binning (grid / x slabs / xz grid)",
    "ball_query_wrapper": "ball_query", "ball_query_lists": "ball_query", "ball_query_pairs": "ball_query", "ball_query_pairs2": "ball_query", "query_and_group": "ball_query+group", "query_and_group_nlc": "ball_query+group",
    "compact_pairs": "pair compaction",
    "sa_mlp3_pool": "SharedMLP SA1 (3 layers + pool, own MFMA kernels)", "sa_mlp3_pool_compact": "SharedMLP SA1 (3 layers + pool, own MFMA kernels)",
    "sa_mlp3_pool_lists": "SharedMLP SA1 (3 layers + pool, own MFMA kernels)",
    "pgather_gemm2": "SharedMLP SA2-4 layers 1+2 (gather + own MFMA kernels)", "pgather_gemm2_compact": "SharedMLP SA2-4 layers 1+2 (gather + own MFMA kernels)",
    "pgather_gemm3_compact": "SharedMLP SA2 whole scale over compact rows (one own MFMA kernel)",
    "pgather_rows": "SharedMLP SA2-4 layers 1+2 (gather + own MFMA kernels)", "gather_gemm": "SharedMLP SA2-4 layers 1+2 (gather + own MFMA kernels)",
    "gather_gemm2": "SharedMLP SA2-4 layers 1+2 (gather + own MFMA kernels)",
    "gemm_pool": "SharedMLP SA2-4 last layer + pool (own MFMA kernels)", "gemm_pool_compact": "SharedMLP SA2-4 last layer + pool (own MFMA kernels)",
    "rowmax_rows": "SharedMLP SA2-4 last layer + pool (own MFMA kernels)",
    "compact_mlp_pair": "SharedMLP SA2-4 both scales per launch (compact rows; primed graphs only)",
    "chain_mlp3": "SharedMLP SA2-4 both scales per launch (compact rows; primed graphs only)",      # round 6: SA2 on the register-chained kernel
    "three_nn_wrapper": "three_nn (+ weights)", "three_nn_with_weights": "three_nn (+ weights)", "three_nn_jobs": "three_nn (+ weights)",
    "qinterp_gemm": "FP first layer (interpolate + add; at FP1-2 with the second layer: own kernels)",
    "qinterp_rows": "FP first layer (interpolate + add; at FP1-2 with the second layer: own kernels)", "interp_gemm": "FP first layer (interpolate + add; at FP1-2 with the second layer: own kernels)",
    "three_interpolate_nlc": "FP first layer (interpolate + add; at FP1-2 with the second layer: own kernels)", "three_interpolate_wrapper": "FP first layer (interpolate + add; at FP1-2 with the second layer: own kernels)",
    "mlp2_rows": "heads (2 layers, own MFMA kernel)",
    "decode_center_boxes": "proposals: decode + top-k + gather + select", "topk_sorted": "proposals: decode + top-k + gather + select",
    "gather_boxes_bev": "proposals: decode + top-k + gather + select", "decode_gather_boxes_bev": "proposals: decode + top-k + gather + select", "select_proposals": "proposals: decode + top-k + gather + select",
    "nms_device_batched": "nms(mask+sweep)", "roipool3d_forward": "roipool3d", "roipool3d_forward_fill": "roipool3d"}
# rows whose alg_bytes are the REFERENCE operator's byte model while the own kernels that replace it move far less:
# bench.py reports them as effective GB/s without a fraction of the HBM roof
EFFECTIVE_ROWS = {
    "fps levels 2-4 (verified prefix)": "A_model of three more sampling passes; fps_nested verifies the picks as a prefix of level 1's order and reads "
                                        "each level once",
    "FP first layer (interpolate + add; at FP1-2 with the second layer: own kernels)": "three_interpolate's (C, m) in / (C, n) out; the own kernels interpolate the per-known-point "
                                                       "products Q of the first layer instead (fewer channels, no (C, n) intermediate)"}
IN_FORWARD = lambda fam: not fam.startswith(("proposals", "nms", "roipool"))   # families inside rpn_forward (the rest follow it)
_ACTIVE = None
_HOOKED = False
_DOUBLE = os.environ.get("WS3D_BENCH_DOUBLE", "")


FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, exact f32 (no xf32 / TF32 on gfx950)


def matrix_work(cfg, fill_rows, batch, compact=True):
    """Multiply-adds the matrix cores EXECUTE for one batch of the c3 forward pass as ws3d_amd/fastpath.py runs it, counted from the
    layer widths of the configuration and the distinct (centre, sample) pairs measured on the batch (`fill_rows` = fastpath.list_fill):
      * SA level 1: three layers over the rows of each scale (distinct pairs when the scale's fill <= COMPACT_MAX_FILL, else all m * nsample);
      * SA levels 2-4: the first layer is linear in the gathered features, so its feature part is ONE product over the n source points
        (P = X W, per point) and only the 3 xyz columns are per row; layers 2 and 3 over the rows;
      * FP modules: the interpolated half of the first layer is a product over the KNOWN points (Q), the skip half and the second layer
        over the unknown points;  * heads: two layers each over every point.
    Padding of K / N to tile multiples is NOT counted (useful work only).  Returns per-batch GFLOP by family (2 flops per multiply-add)."""
    from ws3d_amd import fastpath
    n_in = cfg.num_points
    cin = int(cfg.use_intensity)
    mac = {"SA1 (three layers over rows)": 0.0, "SA2-4 first layer per point (P)": 0.0, "SA2-4 layers 2+3 over rows": 0.0,
           "FP known-point products (Q)": 0.0, "FP skip half + second layer": 0.0, "heads": 0.0}
    sa_out = [cin]
    it = iter(fill_rows)
    n = n_in
    for lvl, (npnt, specs, nsamples) in enumerate(zip(cfg.npoints, cfg.mlps, cfg.nsample)):
        c_prev = sa_out[-1]
        for spec, ns in zip(specs, nsamples):
            f = next(it)
            rows = npnt * (f["distinct_per_list"] if (compact and f["fill"] <= fastpath.COMPACT_MAX_FILL) else ns)
            c1, c2, c3 = spec
            if lvl == 0:
                mac["SA1 (three layers over rows)"] += rows * ((c_prev + 3) * c1 + c1 * c2 + c2 * c3)
            else:
                mac["SA2-4 first layer per point (P)"] += n * c_prev * c1
                mac["SA2-4 layers 2+3 over rows"] += rows * (3 * c1 + c1 * c2 + c2 * c3)
        sa_out.append(sum(s[-1] for s in specs))
        n = npnt
    counts = [n_in] + list(cfg.npoints)
    pre = sa_out[-1]
    for k in range(len(cfg.fp_mlps) - 1, -1, -1):           # FP module k: unknown = level k, known = level k + 1
        c1, c2 = cfg.fp_mlps[k]
        mac["FP known-point products (Q)"] += counts[k + 1] * pre * c1
        mac["FP skip half + second layer"] += counts[k] * (sa_out[k] * c1 + c1 * c2)
        pre = c2
    per_loc_bin_num = int(cfg.loc_scope / cfg.loc_bin_size) * 2
    mac["heads"] += n_in * (pre * cfg.cls_fc[0] + cfg.cls_fc[0] * 1 + pre * cfg.reg_fc[0] + cfg.reg_fc[0] * per_loc_bin_num * 4)
    g = {k: 2.0 * v * batch / 1e9 for k, v in mac.items()}
    return {"gflop_per_batch": sum(g.values()), "by_family_gflop": {k: round(v, 3) for k, v in g.items()}}


def _install_hooks():
    global _HOOKED
    if _HOOKED:
        return
    _HOOKED = True
    from ws3d_amd import compat
    if _DOUBLE == "library":                                   # the same diagnostic for the Tensile GEMMs of the forward pass
        for nm in ("mm", "addmm", "_addmm_activation"):
            orig_t = getattr(torch, nm)
            setattr(torch, nm, lambda *a, __o=orig_t, **kw: (__o(*a, **kw), __o(*a, **kw))[1])
    for fn_name, key in FAMILIES.items():
        orig = getattr(compat, fn_name)

        def wrapped(*a, __orig=orig, __key=key, __name=fn_name, **kw):
            wl = _ACTIVE
            if _DOUBLE and _DOUBLE in __key:
                # diagnostic (scripts/throughput_marginal.py): issue this family's launches twice -- same inputs, same outputs -- so
                # that the change of the throughput-mode step time is what the family costs with 20 batches in flight
                if __name in ("ball_query_pairs", "ball_query_pairs2"):
                    __orig(*a[:5], None)                       # its own cleared pair counter
                elif __name == "mlp2_rows" and len(a) >= 8 and a[7] is not None:
                    __orig(*a[:7], torch.zeros_like(a[7]))     # its own cleared ticket (round 5's table showed the heads at zero cost: the second launch found the ticket consumed and returned)
                elif __name == "chain_mlp3":
                    __orig(a[0], torch.zeros_like(a[1]))       # its own cleared tickets (the launch consumes them)
                else:
                    __orig(*a, **kw)
            if wl is None or not wl._timed:
                return __orig(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = __orig(*a, **kw)
            e1.record()
            wl.op_ev.setdefault(__key, []).append((e0, e1))
            return r
        setattr(compat, fn_name, wrapped)


class C3:
    name = "c3_stage1_rpn_forward_nms_roipool"
    # = BASELINE.json "metric"; `value` is its scenes/sec half at configs[2] (Stage-1 RPN forward incl. proposal NMS +
    # roipool3d, batch 8/GPU), the "FPS+group HBM GB/s" half is the c2 block of the same line
    metric = "KITTI scenes/sec (16384 pts) Stage-1 RPN fwd, 1/2/4/8 GPU; FPS+group HBM GB/s"

    def __init__(self, batch, rank, world, kind="hdl64", depth=2, model=None):
        self.B, self.rank, self.world, self.cfg = batch, rank, world, DEFAULT_CFG
        self.depth = max(1, depth)
        self.kind = kind
        self.pc_host = np.stack([synth.cloud(kind, 16384, 1000 * 3 + rank * batch + s) for s in range(batch)])
        self.pts = torch.from_numpy(self.pc_host).cuda()
        # every slot of the pipeline replays a batch of its OWN scenes (slot 0: the batch above, which the latency mode, the kernel
        # table and the CPU baseline use as well): the timed region sees depth x batch different clouds, and the device-side
        # choice between the SharedMLP forms is taken per batch
        self.slot_pts = [self.pts] + [torch.from_numpy(np.stack([synth.cloud(kind, 16384, 1000 * 3 + 100000 * j + rank * batch + s) for s in range(batch)])).cuda()
                                      for j in range(1, self.depth)]
        self._i = 0
        if model is None:
            model = Stage1Net(mode='TEST').eval()
            model.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 7))
            model = model.cuda()
        self.model = model
        self.ev = []
        self.last = None
        self.op_ev = {}
        self._timed = False
        self.rows_mode = ("distinct pairs / all rows per scale: the graphs keep only the compact kernels of the scales whose fill on the priming "
                          "batch is <= %.2f x %.2f (Stage1Pipeline pair_dispatch='primed'), the others and the eager pass choose per batch on the "
                          "device (launch gates, fill <= %.2f -> distinct pairs)" % (fastpath.PRIMED_MARGIN, fastpath.COMPACT_MAX_FILL, fastpath.COMPACT_MAX_FILL)
                          if fastpath.PAIR_DISPATCH == "device" else "distinct pairs of the ball-query lists, always") \
            if fastpath.COMPACT_PAIRS and fastpath.PER_POINT_L1 else "all m*nsample rows"
        _install_hooks()

    def release(self):
        """drop the pipeline (20 slots of intermediates + graphs) before another workload is built"""
        self.pipe = None
        self._graph = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    def list_fill(self):
        """distinct pairs / list entries per ball-query scale on this workload's batch (the statistic the launch gates act on)"""
        return fastpath.list_fill(self.model.rpn.backbone_net, self.pts)

    def config(self):
        c = self.cfg
        return {"network": "Pointnet2MSG+RPN (weaklyRPN.yaml), 3,046,201 params, random-init (seeded)",
                "pre_nms": c.rpn_pre_nms_top_n, "nms_thresh": c.rpn_nms_thresh, "post_nms": c.rpn_post_nms_top_n,
                "roipool": {"sampled": c.roi_sampled_pts, "channels": 128, "extra_width": c.roi_extra_width},
                "exchange": "all_gather of (B,100,8) proposals" if self.world > 1 else "none (1 GPU)",
                "launch": getattr(self, "_launch_desc", "eager (graph capture not attempted)"),
                "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                # data-dependent work: the SharedMLPs may run over the DISTINCT (centre, sample) pairs of the ball-query lists (bit-identical
                # to all m * nsample rows); `list_fill` in the line says how full this data's lists are, `all_rows` what the other form costs
                "sharedmlp_rows": self.rows_mode,
                "scenes": "%d distinct scenes per GPU in the timed region (every pipeline slot replays a batch of its own; latency mode, "
                          "kernels[] and the CPU baseline use slot 0's batch)" % (self.depth * self.B)}

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
                                   roipool=True, device=self.pts.device, exchange_global_batch=self.B * self.world if self.world > 1 else 0)
        for slot in self.pipe.slots:
            slot["inp"].copy_(self.pts)
        ok = self.pipe.capture_all()
        for slot, pts in zip(self.pipe.slots, self.slot_pts):          # (capture_all primes every slot on slot 0's batch)
            slot["inp"].copy_(pts)
        torch.cuda.synchronize()
        self._graph = True if ok else None
        self._graph_err = self.pipe.graph_error
        self._launch_desc = ("hipGraph replay of the whole step, %d batches in flight on separate HIP streams" % self.depth) if ok \
            else "eager (graph capture failed: %s)" % self._graph_err
        return ok

    @torch.no_grad()
    def step(self, timed=False, eager=False, pts=None):
        global _ACTIVE
        _ACTIVE = self
        if getattr(self, "_graph", None) is not None and not timed and not eager:
            ticket = self.pipe.submit(pts)                 # None: the slot's own batch, already resident in its input buffer
            slot = self.pipe.slots[ticket % self.depth]
            res = slot["out"]
            # world > 1: the slot's ProposalExchange gathered inside submit() -- one collective on resident buffers (the selection kernel
            # wrote the rows + counts into the send buffer); (proposals, float counts)
            gathered = res["gathered"] if "gathered" in res else (res["packed"], res["count"])
            self.last = (res["rpn"], res["boxes"], res["scores"], res["count"], res["pooled"], res["empty"], gathered)
            return
        self._timed = timed
        e = None
        from ws3d_amd import fastpath
        if timed:           # per-operator timers: one stream, so that an operator's time is its own
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
        if getattr(self, "_primed", None) is None:
            # the eager steps (per-operator timers, the rocprofv3 passes of scripts/) launch what the graphs replay: the scales whose fill
            # on this batch is far below the threshold keep only their compact kernels, as Stage1Pipeline(pair_dispatch="primed") captures
            # them -- paired per level, SA2 on ws3d_chain_mlp3 -- instead of the eager default's gated twins (set-up: synchronises once)
            self._primed = fastpath.primed_compact_scales(self.model.rpn.backbone_net, self.pts) if fastpath.PAIR_DISPATCH == "device" else frozenset()
        with fastpath.geometry_ahead(False if timed else fastpath.GEOMETRY_AHEAD), fastpath.compact_only_scales(self._primed):
            out = self.model.rpn_forward({'pts_input': self.pts, 'defer_reg_join': True})      # proposals_from_rpn waits for rpn_reg
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
        self.step(pts=self.pts)                            # (slot 0's batch, whichever slot is next)
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
        # roipool3d: bytes this batch needs moved, measured on the device like bench.py's c5 row -- rows of the NON-EMPTY RoIs written
        # once, xyz scanned once, the feature rows of the distinct pooled points read once (the proposals of a random-weight net
        # land where they land; rows of empty RoIs stay untouched by contract)
        roi_total = None
        if self.last is not None:
            pooled, empty = self.last[4], self.last[5]
            nonempty = int((empty == 0).sum().item())
            distinct = sum(int(torch.unique(pooled[b, :, :, :3].reshape(-1, 3), dim=0).size(0)) for b in range(B))
            roi_total = nonempty * cfg.roi_sampled_pts * (3 + 128) * 4 + B * cfg.num_points * 12 + distinct * (3 + 128) * 4
            self.roi_stats = {"non_empty_rois": nonempty, "rois": B * cfg.rpn_post_nms_top_n, "distinct_pooled_points": distinct}
        roi_b = (cfg.rpn_post_nms_top_n * cfg.roi_sampled_pts * (3 + 128) * 4 + cfg.num_points * (3 + 128) * 4) if roi_total is None else roi_total / B
        n1, m1 = cfg.num_points, cfg.npoints[0]
        fps1_b = (m1 - 1) * n1 * 12 + m1 * 4
        alg = {"fps level 1 (16384 -> 4096)": fps1_b * B, "fps levels 2-4 (verified prefix)": (_fps_model_bytes(cfg) - fps1_b) * B,
               "ball_query": _bq_bytes(cfg) * B, "ball_query+group": _qg_bytes(cfg) * B, "three_nn (+ weights)": nn_b * B,
               "FP first layer (interpolate + add; at FP1-2 with the second layer: own kernels)": interp_b * B, "roipool3d": roi_b * B,
               "nms(mask+sweep)": (cfg.rpn_pre_nms_top_n * 20 + cfg.rpn_pre_nms_top_n * 141 * 8) * B}
        rows = []
        for key, evs in self.op_ev.items():
            ms = float(sum(a.elapsed_time(b) for a, b in evs)) / steps
            row = {"name": key, "ms_per_step": ms, "launches_per_step": len(evs) / steps,
                   "alg_bytes_per_step": alg.get(key, 0), "traffic_key": "c3:" + key, "bound": "hbm"}
            if key in EFFECTIVE_ROWS:
                row["bytes_model"] = "reference op (effective): " + EFFECTIVE_ROWS[key]
            if key == "roipool3d" and getattr(self, "roi_stats", None):
                row.update(self.roi_stats)
            if key.startswith("fps level 1"):
                from bench import fps_lane_instr
                from bench import fps_valu_pmc
                e8 = fps_valu_pmc(B, self.kind)
                row.update({"bound": "valu", "lane_instr_per_step": fps_lane_instr(n1, m1) * B, "us_per_fps_step": ms * 1e3 / (m1 - 1),
                            "physical_lane_instr_per_step": None if e8 is None else e8["sq_insts_valu_per_launch"] * 64.0,
                            "comment": "one workgroup per scene (8 of 256 CUs): fps_rounds2_kernel (fps_bucket.hip), exact pruned sampling with two candidates per wave and up to 8 "
                                       "certified samples per record exchange.  Chain-bound: us_per_fps_step (time per sample) is the figure to "
                                       "watch; valu_frac = issued VALU lane-instructions (committed --pmc pass) / duration / the WHOLE chip's roof, "
                                       "lane_instr_per_step the dense sweep's count (8 per point and step)"})
            elif key.startswith("fps levels"):
                row["comment"] = "1024 + 256 + 64 picks verified as the leading prefix of the previous level's order (fps_nested.hip)"
            elif key.startswith(("three_nn", "nms", "ball_query")):
                row["comment"] = "ALU-bound search / pair test; the HBM figure is for orientation only"
            elif key.startswith(("SharedMLP", "heads", "FP first")):
                row["bound"] = "mfma" if not key.startswith("FP first") else "hbm"
            rows.append(row)
        rows.sort(key=lambda r: -r["ms_per_step"])
        fwd = float(np.mean([a[0].elapsed_time(a[1]) for a in self.ev]))
        custom_in_fwd = sum(r["ms_per_step"] for r in rows if IN_FORWARD(r["name"]))
        rows.append({"name": "library residual of rpn_forward: Tensile GEMMs (per-point products P / Q, skip products, second FP layers) + at::native "
                             "glue (fills, cat, copies)", "ms_per_step": max(fwd - custom_in_fwd, 0.0), "launches_per_step": 0, "alg_bytes_per_step": 0,
                     "traffic_key": "c3:library residual of rpn_forward: Tensile GEMMs + at::native glue", "bound": "library"})
        self.breakdown = {"rpn_forward_ms": fwd,
                          "proposals_nms_ms": float(np.mean([a[1].elapsed_time(a[2]) for a in self.ev])),
                          "roipool_ms": float(np.mean([a[2].elapsed_time(a[3]) for a in self.ev])),
                          "own_launches_per_step": float(sum(r["launches_per_step"] for r in rows))}
        return rows

    def path_gbps(self, scenes_per_s_per_gpu):
        cfg = self.cfg
        am = _fps_model_bytes(cfg) + _qg_bytes(cfg) + sum(_interp_bytes(cfg))
        return {"custom_ops_a_model_bytes_per_scene": am, "a_model": am * scenes_per_s_per_gpu / 1e9,
                "a_model_frac_of_8TBs": am * scenes_per_s_per_gpu / 8.0e12, "breakdown_ms": getattr(self, "breakdown", None)}

    def scenes(self):
        return self.B

    def cpu_baseline(self, min_seconds=6.0):
        """Two figures on this box's host cores (the reference has no CPU path for these operators, the oracle port stands in):
        `value` = the WHOLE step like for like (bench_cpu.whole_step_baseline: search + grouping + SharedMLPs on torch-CPU + proposal
        stage + NMS + roipool3d, the pooling also through the reference's own compiled `roipool3d_cpu`), and `search_only` = the custom
        search operators alone (4x FPS, 8x ball_query, 4x three_nn: the figure of rounds 1-4)."""
        import bench_cpu
        search = self.cpu_search_only(min_seconds)
        whole = bench_cpu.whole_step_baseline(self.model, self.cfg, self.pc_host, search, min_seconds)
        whole["search_only"] = search
        return whole

    def cpu_search_only(self, min_seconds=6.0):
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
        _, dt, reps = repeat_for(one_pass, min_seconds)
        oracle.set_threads(1)
        return {"value": ns * reps / dt, "unit": "scenes/s", "cores": threads, "kind": "port",
                "sample": f"{ns} scenes (the batch tiled {reps_of_batch}x) x {reps} pass(es): the search ops of the Stage-1 forward "
                          f"only (4 FPS, 8 ball queries, 4 three_nn) "
                          f"on oracle/ws3d_oracle.c with OpenMP, wall {dt:.2f} s; grouping copies and MLPs excluded"}
