"""Throughput mode of the Stage-1 inference step: several batches in flight.

One batch of 8 scenes cannot fill an MI355X: furthest point sampling is one workgroup per scene
(8 of 256 CUs for 4.8 ms) and most of the ~60 kernels after it are small.  ``Stage1Pipeline``
captures the whole step -- ``Stage1Net.rpn_forward`` + ``proposals_from_rpn`` (+ ``roipool3d``) --
into one hipGraph per slot, each slot on its own HIP stream with its own input buffer, and keeps
``depth`` batches in flight, so the sampling of one batch runs under the GEMMs / NMS / pooling of the
others.  Nothing in the C ABI allocates or synchronises, which is what makes the capture possible.

HIP multiplexes streams onto at most ``GPU_MAX_HW_QUEUES`` hardware queues (default 4) and streams
that share a queue serialise; the variable is read when the runtime starts, so set it (e.g. 32)
before the first device call of the process -- ``ensure_hw_queues()`` does it when still possible.
Measured (batch 8 x 16384 points, bench.py c3): 969 scenes/s with one batch in flight, 2,106 with 3
on the default 4 queues, 3,599 with 12 and 3,850 with 20 on 32 queues (24 in flight: slower again).

    pipe = Stage1Pipeline(model, cfg, batch=8, depth=6)
    for first_scene, n_valid, out in pipe.map(batches):      # batches: iterable of (<=B, N, 4) arrays / tensors
        boxes, scores, count = out["boxes"], out["scores"], out["count"]   # device tensors, valid until the slot is reused
"""
from __future__ import annotations

import os
from typing import Iterable, Iterator, Tuple

import numpy as np
import torch

from . import roipool3d_ops, stage1


def ensure_hw_queues(n: int = 32) -> bool:
    """raise the runtime's hardware-queue cap unless the user set it; True if the value can still
    take effect (no device context yet in this process).  Once the runtime has started the variable
    is left alone, so that it keeps saying what the runtime read."""
    if torch.cuda.is_initialized():
        return False
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(n))
    return True


def _hw_queue_cap() -> int:
    """the cap the runtime reads at start-up (its own default is 4)"""
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        return 4


def _slot_stream(device: torch.device, j: int) -> torch.cuda.Stream:
    """stream of slot j on `device`: from the package's per-device pool (ws3d_amd/streams.py), shared by every Stage1Pipeline of
    the process and by the eager forward's side streams -- a second pipeline (another model, a re-capture, bench.py's side runs)
    reuses the streams of the first instead of adding `depth` more hardware queues."""
    from .streams import pooled_stream
    return pooled_stream(device, j)


# Launch geometry of the persistent kernels while a pipeline with several batches in flight captures its graphs (ws3d_tune): a launch that
# holds a quarter of the chip at ~0.75 of its matrix pipes and leaves the rest to the other batches beats one that holds all of it at
# ~0.45 -- measured on the 20-deep c3 step: ws3d_chain_mlp3 64 workgroups +3-4 % against 256, ws3d_sa_mlp3_pool_compact 192 +2 % against 768,
# ws3d_mlp2_rows 128 +0.7 % against 256 (profiles/r06_chain_mlp3_workgroups.txt, r06_tune_workgroups.txt, r06_tune_workgroups2.txt).  A lone batch (depth < 4, eager callers) keeps the full-chip grids,
# which are 2.5 x faster in isolation.  The grids are baked into the graphs at capture; the library's knobs are restored behind it.
THROUGHPUT_GEOMETRY = {"chain_wgs": 64, "sa1_wgs": 192, "mlp2_wgs": 128, "bq_wide_nw": 8}      # (bq_wide_nw: the search launches on 8 waves x 8 centres per tile, +1 %: profiles/r06_bq_emit_anatomy.txt section 10)
THROUGHPUT_GEOMETRY_MIN_DEPTH = 4


class _LaunchGeometry:
    def __init__(self, on: bool):
        self.on, self.prev = on, {}

    def __enter__(self):
        if self.on:
            from . import compat
            for k, v in THROUGHPUT_GEOMETRY.items():
                self.prev[k] = compat.tune(k, v)
        return self

    def __exit__(self, *exc):
        if self.on:
            from . import compat
            for k, v in self.prev.items():
                compat.tune(k, v)
        return False


class Stage1Pipeline:
    def __init__(self, model: stage1.Stage1Net, cfg: stage1.RPNConfig = stage1.DEFAULT_CFG, batch: int = 8,
                 n_points: int = 16384, depth: int = 6, roipool: bool = False, device="cuda:0", use_graph: bool = True,
                 channels: int = 4, tune_gemms: bool = None, pair_dispatch: str = "primed", exchange_global_batch: int = None):
        """exchange_global_batch: the global batch of a multi-rank job (``batch`` x world for even shards).  Every slot then owns a
        ``dist.ProposalExchange`` -- send / receive buffers allocated HERE, once; the captured step's selection kernel writes the packed
        rows + counts straight into the slot's send buffer, and ``submit`` issues the step's ONE all-gather on the slot's stream right
        behind the replay: ``result(ticket)["gathered"]`` = ((global_batch, K, 8) proposals, (global_batch,) counts).  None: on when
        torch.distributed is initialised with more than one rank (global batch = batch x world), off otherwise.
        pair_dispatch: how a ball-query scale's SharedMLP chooses between its compact-pairs and its dense kernels (both exact,
        ws3d_amd/fastpath.py PAIR_DISPATCH).  "device": both forms in every graph, the batch's pair total decides in the kernels'
        prologues.  "primed" (default): the scales whose fill on the FIRST primed batch is <= fastpath.PRIMED_MARGIN x the threshold
        keep only their compact kernels in the graphs (ten launches fewer per step on LiDAR clouds); a later, denser batch still comes
        out identical -- the compact kernels are complete -- only slower than the dense form would have been."""
        if pair_dispatch not in ("primed", "device"):
            raise ValueError("Stage1Pipeline: pair_dispatch must be 'primed' or 'device', got %r" % (pair_dispatch,))
        from .streams import POOL_SIZE
        if max(1, int(depth)) > POOL_SIZE:          # (before any slot claims a pooled stream -- ADVICE round 5)
            raise ValueError("Stage1Pipeline: depth %d exceeds the per-device stream pool (ws3d_amd.streams.POOL_SIZE = %d): the slots "
                             "would share streams with the eager pass's side streams; beyond ~23 hardware queues in one process this "
                             "runtime time-slices the queues and every kernel slows down" % (int(depth), POOL_SIZE))
        self.pair_dispatch, self.compact_only, self.compact_only_source = pair_dispatch, None, None     # compact_only: the (level, scale) keys whose dense twin the graphs drop -- read it to see what priming decided
        self.model, self.cfg, self.B, self.depth, self.roipool = model.eval(), cfg, int(batch), max(1, int(depth)), roipool
        self.device = torch.device(device)
        if exchange_global_batch is None:
            import torch.distributed as tdist
            w = tdist.get_world_size() if (tdist.is_available() and tdist.is_initialized()) else 1
            exchange_global_batch = self.B * w if w > 1 else 0
        self.exchange_global_batch = int(exchange_global_batch)
        self.hw_queues_raised = ensure_hw_queues()      # False: the runtime already started with its own cap
        self.submitted = 0
        self.graph_error = None
        self.slots = []
        with torch.cuda.device(self.device):
            for j in range(self.depth):
                stream = _slot_stream(self.device, j)
                inp = torch.zeros((self.B, n_points, channels), dtype=torch.float32, device=self.device)
                exch = None
                if self.exchange_global_batch > 0:
                    from . import dist as wdist
                    exch = wdist.ProposalExchange(self.B, cfg.rpn_post_nms_top_n, self.exchange_global_batch, self.device)
                self.slots.append({"stream": stream, "inp": inp, "graph": None, "out": None,
                                   "done": torch.cuda.Event(), "primed": False, "exchange": exch})
        if tune_gemms is None:      # WS3D_TUNE_GEMMS=0: keep the library's heuristic GEMM solutions (identical in every process)
            tune_gemms = os.environ.get("WS3D_TUNE_GEMMS", "1") != "0"
        self.use_graph, self.tune_gemms = use_graph, tune_gemms
        if not self.hw_queues_raised and self.depth > _hw_queue_cap():
            import warnings
            warnings.warn("Stage1Pipeline: depth %d but the HIP runtime of this process already started with GPU_MAX_HW_QUEUES=%d; "
                          "streams that share a hardware queue serialise (measured: 2,106 scenes/s with 4 queues vs 3,850 with 32). "
                          "Export GPU_MAX_HW_QUEUES=32 (or call ws3d_amd.pipeline.ensure_hw_queues()) before the first device call."
                          % (self.depth, _hw_queue_cap()), RuntimeWarning, stacklevel=2)

    # ------------------------------------------------------------------ the step
    @torch.no_grad()
    def body(self, pts: torch.Tensor, exchange=None) -> dict:
        out = self.model.rpn_forward({"pts_input": pts, "defer_reg_join": True})     # (proposals_from_rpn waits for rpn_reg)
        boxes, scores, count, enlarged, packed = stage1.proposals_from_rpn(out, self.cfg, with_pool_boxes=True,
                                                                           with_packed=exchange.send if exchange is not None else True)
        res = {"rpn": out, "boxes": boxes, "scores": scores, "count": count, "packed": packed}      # packed (B,K,8): box + score, ws3d_amd.dist's rows
        if self.roipool:
            feats = out["backbone_features"].transpose(1, 2).contiguous()
            res["pooled"], res["empty"] = roipool3d_ops.roipool3d_gpu(out["backbone_xyz"], feats, boxes, self.cfg.roi_extra_width,
                                                                      sampled_pt_num=self.cfg.roi_sampled_pts, enlarged=enlarged)
        return res

    def _tunable(self, on: bool):
        """the GEMMs between our kernels run on hipBLASLt through torch; its TunableOp times the candidate
        solutions of each (m, n, k) once -- during the eager priming runs, so the captured graph replays
        the winners (measured +3 % on c3).  Process-wide switches: restored after the capture."""
        tun = getattr(torch.cuda, "tunable", None)
        if not self.tune_gemms or tun is None:
            return None
        try:
            if on:
                prev = (tun.is_enabled(), tun.tuning_is_enabled())
                tun.enable(True)
                tun.tuning_enable(True)
                if not os.environ.get("PYTORCH_TUNABLEOP_FILENAME"):      # results file: not into the working directory
                    import tempfile
                    tun.set_filename(os.path.join(tempfile.gettempdir(), "ws3d_amd_tunableop.csv"), True)
                return prev
            return None
        except Exception:           # an optimisation only
            return None

    def _prime(self, slot: dict) -> None:
        """first use of a slot: two eager runs (library workspaces, weight caches), then the capture"""
        slot["primed"] = True
        if not self.use_graph or self.graph_error is not None:
            return
        prev = self._tunable(True)
        try:
            with _LaunchGeometry(self.depth >= THROUGHPUT_GEOMETRY_MIN_DEPTH):
                self._prime_capture(slot)
        finally:
            if prev is not None:
                torch.cuda.tunable.enable(prev[0])
                torch.cuda.tunable.tuning_enable(prev[1])

    def _prime_capture(self, slot: dict) -> None:
        try:
            stream = slot["stream"]
            stream.wait_stream(torch.cuda.current_stream(self.device))
            from . import fastpath
            if self.compact_only is None:        # once per pipeline, on the first primed batch (synchronises: set-up)
                with torch.cuda.stream(stream):
                    # an all-zero input buffer (submit(None) before any batch was copied in) says nothing about the data: keep both
                    # forms of every scale in the graphs, as pair_dispatch="device" does, instead of deciding on padding
                    blank = not bool(slot["inp"].any().item())
                    self.compact_only = (fastpath.primed_compact_scales(self.model.rpn.backbone_net, slot["inp"])
                                         if (self.pair_dispatch == "primed" and not blank) else frozenset())
                    self.compact_only_source = "blank input: both forms kept" if blank else ("fills of the first primed batch" if self.pair_dispatch == "primed" else "pair_dispatch=device")
            # the priming runs stay on the slot's stream: the eager path's side streams would claim more hardware queues that
            # the graphs never use (a context variable: forward passes of other threads keep their own stream topology)
            with fastpath.geometry_ahead(False), fastpath.compact_only_scales(self.compact_only), torch.cuda.stream(stream):
                for _ in range(2):
                    self.body(slot["inp"], slot["exchange"])
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with fastpath.compact_only_scales(self.compact_only), torch.cuda.graph(graph, stream=stream):
                slot["out"] = self.body(slot["inp"], slot["exchange"])
            slot["graph"] = graph
        except Exception as exc:      # capture is an optimisation: fall back to eager launches on the slot streams
            self.graph_error = repr(exc)
            slot["graph"] = None

    def capture_all(self) -> bool:
        """prime every slot now (otherwise each is primed on first use); True if the graphs were captured"""
        for slot in self.slots:
            if not slot["primed"]:
                slot["inp"].copy_(self.slots[0]["inp"])
                self._prime(slot)
        torch.cuda.synchronize(self.device)
        return self.use_graph and self.graph_error is None

    # ------------------------------------------------------------------ submit / collect
    @torch.no_grad()
    def submit(self, pts=None) -> int:
        """start one batch on the next slot -> ticket.  pts (<=B, N, C) host or device (None: reuse the
        slot's input buffer); a short batch is padded by repeating its last scene.  The results of the
        batch submitted `depth` tickets ago become invalid."""
        slot = self.slots[self.submitted % self.depth]
        with torch.cuda.device(self.device):
            if pts is not None:
                t = torch.from_numpy(np.ascontiguousarray(pts, dtype=np.float32)) if not torch.is_tensor(pts) else pts
                if t.size(0) > self.B or tuple(t.shape[1:]) != tuple(slot["inp"].shape[1:]):
                    raise ValueError(f"batch of shape {tuple(t.shape)} does not fit the pipeline's {tuple(slot['inp'].shape)}")
            if not slot["primed"]:
                if pts is not None:
                    slot["inp"][:t.size(0)].copy_(t)
                    if t.size(0) < self.B:       # pad BEFORE priming: the primed pair dispatch reads the fills of this buffer, and zero
                        slot["inp"][t.size(0):] = slot["inp"][t.size(0) - 1]      # scenes (fill 1.0) are padding, not data
                self._prime(slot)
            if pts is not None and t.is_cuda:       # produced on the caller's stream: order the slot's copy behind it
                slot["stream"].wait_stream(torch.cuda.current_stream(self.device))
                t.record_stream(slot["stream"])
            with torch.cuda.stream(slot["stream"]):
                if pts is not None:
                    slot["inp"][:t.size(0)].copy_(t, non_blocking=True)
                    if t.size(0) < self.B:
                        slot["inp"][t.size(0):] = slot["inp"][t.size(0) - 1]
                if slot["graph"] is not None:
                    slot["graph"].replay()
                else:
                    from . import fastpath
                    with fastpath.compact_only_scales(self.compact_only or frozenset()):
                        slot["out"] = self.body(slot["inp"], slot["exchange"])
                if slot["exchange"] is not None:         # the step's one collective: resident buffers, the slot's stream, behind the replay
                    slot["out"]["gathered"] = slot["exchange"].gather()
                slot["done"].record(slot["stream"])
        self.submitted += 1
        return self.submitted - 1

    def result(self, ticket: int) -> dict:
        """wait for a ticket -> its output dict (device tensors owned by the slot)"""
        if not (self.submitted - self.depth <= ticket < self.submitted):
            raise ValueError(f"ticket {ticket} is not in flight (submitted {self.submitted}, depth {self.depth})")
        slot = self.slots[ticket % self.depth]
        slot["done"].synchronize()
        return slot["out"]

    def map(self, batches: Iterable) -> Iterator[Tuple[int, int, dict]]:
        """run every batch of an iterable, `depth` in flight -> (index of its first scene, number of valid
        scenes, output dict) in submission order"""
        pending = []          # (ticket, first scene, valid scenes)
        first = 0
        for pts in batches:
            n = int(pts.shape[0])
            pending.append((self.submit(pts), first, n))
            first += n
            if len(pending) == self.depth:
                ticket, f0, nv = pending.pop(0)
                yield f0, nv, self.result(ticket)
        for ticket, f0, nv in pending:
            yield f0, nv, self.result(ticket)
