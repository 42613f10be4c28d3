"""Channels-last inference path for the Stage-1 network (same weights, same operators, other
memory layout).

The reference keeps point features as (B, C, N) and runs every SharedMLP as a 1x1 convolution.
On MI355X the natural layout is (B, N, C) -- "one row per point":
  * a SharedMLP layer is ONE row-major GEMM (B*L, C) x (C, O) over all scenes with bias + ReLU
    fused into the GEMM epilogue (``torch._addmm_activation`` -> hipBLASLt), instead of a batched
    GEMM followed by a separate bias/ReLU pass over the output (measured 20-35 % faster per layer,
    bit-identical results);
  * grouping, interpolation and RoI pooling gather contiguous rows (``ws3d_query_and_group_nlc``,
    ``ws3d_three_interpolate_nlc``; ``ws3d_roipool3d`` takes this layout already);
  * the heads' outputs (B, N, 1) / (B, N, 40) come out in the layout the reference transposes to.
Sampling, ball query and 3-NN are layout-independent and shared with the channels-first path.

``rpn_forward(model, pts)`` reads the folded weights of an eval-mode ``Stage1Net`` and returns the
same dict as ``Stage1Net.rpn_forward``; ``backbone_features`` is a (B, C, N) *view* of the (B, N, C)
result, so ``.transpose(1, 2).contiguous()`` on it costs nothing.  ``Stage1Net`` uses this path in
eval mode on the GPU unless ``stage1.CHANNELS_LAST_FASTPATH`` is cleared.
"""
from __future__ import annotations

import contextlib
import contextvars

import torch
import torch.nn as nn

from . import compat as _C
from . import nn_blocks, pn2_ops

# The switches below are module attributes, not environment variables (round 4): every one of them selects between two exact
# forms of the same function; tests/test_gpu_parity.py::test_fast_path_switches_agree flips them, bench.py flips COMPACT_PAIRS
# for its all-rows block.  The defaults are what every measurement in profiles/ ran.
FUSED_SA_MLP = True   # ws3d_sa_mlp3_pool for the 4-channel SA level (clear to A/B against the GEMM chain)
SA1_FROM_LISTS = True  # first level: rows built from the neighbour lists inside the MLP kernel (0: grouped tensor)
FUSED_GEMM_POOL = True   # ws3d_gemm_pool: last layer of the other SA levels + pool on the matrix cores
FUSED_INTERP_GEMM = True  # ws3d_interp_gemm: three_interpolate + skip concat fused into the first FP layer's A operand
NESTED_FPS = True  # levels 2-4: verified-prefix sampling (pn2_ops.furthest_point_sample_gather_nested)
GEOMETRY_AHEAD = True  # sampling chain + searches on side streams beside the GEMMs
_AHEAD_OVERRIDE = contextvars.ContextVar("ws3d_geometry_ahead", default=None)


@contextlib.contextmanager
def geometry_ahead(on: bool):
    """the forward passes issued inside this block (by this thread / task only) use / do not use the side streams, whatever the
    module default says: Stage1Pipeline primes its slots on ONE stream, bench_c3's per-operator timers want one stream"""
    token = _AHEAD_OVERRIDE.set(bool(on))
    try:
        yield
    finally:
        _AHEAD_OVERRIDE.reset(token)


@contextlib.contextmanager
def compact_only_scales(keys):
    """inside this block (this thread / task only) the scales (level, scale index) named in `keys` launch ONLY their compact
    kernels -- no gated dense twin behind them (ten launches of the Stage-1 step that return in their prologue on sparse clouds,
    0.2 % of the c3 throughput each).  The compact kernels are complete for every batch (the dense form is a speed decision for
    lists fuller than COMPACT_MAX_FILL), so a denser batch than the caller expected runs slower, not wrong."""
    token = _COMPACT_ONLY.set(frozenset(keys))
    try:
        yield
    finally:
        _COMPACT_ONLY.reset(token)


def primed_compact_scales(net, pointcloud: torch.Tensor, margin: float = None):
    """the (level, scale) keys of ``compact_only_scales`` for a caller that primes on `pointcloud`: every scale whose fill on this
    batch is <= margin * COMPACT_MAX_FILL.  Runs ``list_fill`` (synchronises: set-up time only)."""
    margin = PRIMED_MARGIN if margin is None else margin
    keys, per_level = set(), {}
    for row in list_fill(net, pointcloud):
        si = per_level.get(row["level"], 0)
        per_level[row["level"]] = si + 1
        if row["fill"] <= margin * COMPACT_MAX_FILL:
            keys.add((row["level"] - 1, si))
    return frozenset(keys)


def _geometry_ahead_now() -> bool:
    v = _AHEAD_OVERRIDE.get()
    return GEOMETRY_AHEAD if v is None else v
GEOMETRY_IN_CAPTURE = False  # ... also while a hipGraph is captured (fork / join inside the graph)
PER_POINT_L1 = True  # SA2..SA4: layer 1 as feats @ W_f per point + gather (ws3d_pgather_*)
COMPACT_MAX_FILL = 0.55  # lists fuller than this (distinct rows / all rows) take the dense kernels
COMPACT_PAIRS = True  # the SharedMLPs over the distinct (centre, sample) pairs only (0: all m * nsample rows)
# How a scale chooses between the compact and the dense form of its SharedMLP (both exact):
#   device  (default) both forms are launched, every kernel reads the pair total of THIS batch in its prologue and the form on the
#           wrong side of COMPACT_MAX_FILL returns at once (include/ws3d_ops.h "launch gates") -- no host synchronisation, nothing
#           latched per process, a captured hipGraph adapts per batch;
#   compact always the compact kernels (A/B runs).
# COMPACT_PAIRS = False is "always dense".  A caller that launches the same step many times (Stage1Pipeline's graphs) can drop the
# dense twin of the scales its priming batch shows to be far from the threshold: ``compact_only_scales`` below.
PAIR_DISPATCH = "device"
PRIMED_MARGIN = 0.85       # Stage1Pipeline(pair_dispatch="primed"): scales whose priming fill <= PRIMED_MARGIN * COMPACT_MAX_FILL lose the dense twin
_COMPACT_ONLY = contextvars.ContextVar("ws3d_compact_only_scales", default=frozenset())
CHAIN_MLP = True       # SA2 (o1 = 64, o2 <= 96, o3 = 128): the whole SharedMLP of the level's two scales on the register-chained kernel (ws3d_chain_mlp3, round 6: activations in registers, weights resident in LDS, ticketed 32-row tiles) instead of ws3d_compact_mlp_pair(3); bit-identical
PAIRED_SCALES = True   # the compact SharedMLP kernels of a level's two scales in ONE launch each (ws3d_compact_mlp_pair) where neither scale carries a gated dense twin (Stage1Pipeline's primed graphs, PAIR_DISPATCH "compact")
MERGED_THREE_NN = True   # serial order: the 3-NN searches of the FP modules whose known set is binned in ONE launch behind that binning launch (ws3d_three_nn_jobs)
FUSED_PROLOGUE = True   # the coordinate / feature split of the input rows and the clear of the pass's zero arena in ONE launch (ws3d_split_points_clear) instead of two strided copies + a fill
MERGED_BINNING = True   # serial order (graph capture): the binned copies of levels 2.. (ball query's grid + three_nn's (x, z) grid) in ONE launch behind the sampling chain (ws3d_sort_points_jobs) instead of one per level and flavour
PER_POINT_FP = True  # FP modules: first layer as (known_feats @ W_a) interpolated + skip @ W_b (ws3d_qinterp_rows)
FUSED_MLP2_ROWS = True  # ws3d_mlp2_rows: the two layers of a head in one kernel
FUSED_GATHER_GEMM2 = True  # ws3d_gather_gemm2: layers 1 + 2 of SA2-SA4 in one kernel
BIN_INPUT_AHEAD = True     # eager pass with geometry ahead: bin the input cloud on the search stream beside the first level's sampling kernel
FUSED_QINTERP_GEMM_MIN_ROWS = 30000    # ws3d_qinterp_gemm (both layers of an FP module in one kernel) from this many rows on (batch 8: FP1, FP2; smaller modules lose, profiles/r04_qinterp_gemm_ab.txt); 1 << 60: never
FUSED_COMPACT3_MAX_LDS = 64 * 1024   # ws3d_pgather_gemm3_compact (the whole SharedMLP of a scale over compact rows in one kernel) where its two LDS tiles fit in this many bytes (SA2: 41 / 50 KB); 0: the two-kernel form everywhere
FUSED_GATHER_GEMM = True  # ws3d_gather_gemm: grouping fused into the first layer's A operand (no grouped tensor in HBM)
# levels with fewer points search by brute force (LDS-tiled scan) without a binned copy.  256: every level of the Stage-1 network takes
# the fine-grid kernel, which also emits the pair table (2048, the operators' own default: 8 launches more per batch, -0.6 % throughput)
GRID_MIN_N = 256
DUAL_SCALE_SEARCH = True  # both ball queries of a level in one launch (ws3d_ball_query_pairs2)
NESTED_CHAIN = True       # levels 2-4: ONE chain call (ws3d_furthest_point_sampling_nested_chain: four launches) instead of three launches per level
QUERY_CELL_ORDER = True   # 3-NN: queries taken in the cell order of the level's binned copy (ws3d_three_nn_wq; same rows, -15-30 % per search)
PARALLEL_SCALES = True  # eager side-stream mode: the second scale of a level beside the first
PARALLEL_HEADS = True  # ... and the regression head beside the classification head + top-k


def _row_weights(block):
    """(W^T (C,O) contiguous, bias (O,) or None, relu?) of a Conv+BN+ReLU block, cached on the block"""
    w, shift, act = block._folded()
    if act is not None and not isinstance(act, nn.ReLU):
        raise NotImplementedError("fast path supports ReLU / no activation only")
    cache = block.__dict__.get("_row_cache")
    if cache is None or cache[0] is not w:
        cache = (w, w.t().contiguous())
        if not torch.cuda.is_current_stream_capturing():
            block.__dict__["_row_cache"] = cache
    return cache[1], shift, act is not None


def _row_weights_xyz_last(block):
    """_row_weights with the three xyz rows moved behind the feature rows (the K order ws3d_gather_gemm reads), cached"""
    wt, bias, relu = _row_weights(block)
    cache = block.__dict__.get("_row_cache_xyz_last")
    if cache is None or cache[0] is not wt:
        cache = (wt, torch.cat((wt[3:], wt[:3]), dim=0).contiguous())
        if not torch.cuda.is_current_stream_capturing():
            block.__dict__["_row_cache_xyz_last"] = cache
    return cache[1], bias, relu


def _split_rows(block, wt: torch.Tensor, c2: int):
    """(wt[:c2], wt[c2:]) as contiguous matrices, cached on the block (the interpolated / the skip channels of an FP module's first layer)"""
    cache = block.__dict__.get("_split_cache")
    if cache is None or cache[0] is not wt or cache[1] != c2:
        cache = (wt, c2, wt[:c2].contiguous(), wt[c2:].contiguous())
        if not torch.cuda.is_current_stream_capturing():
            block.__dict__["_split_cache"] = cache
    return cache[2], cache[3]


def _layer(x2d: torch.Tensor, block) -> torch.Tensor:
    wt, bias, relu = _row_weights(block)
    if bias is None:
        y = torch.mm(x2d, wt)
        return torch.relu_(y) if relu else y
    if relu:
        return torch._addmm_activation(bias, x2d, wt, use_gelu=False)
    return torch.addmm(bias, x2d, wt)


def _blocks(seq):
    """the Conv blocks of a SharedMLP / head Sequential (Dropout is the identity in eval mode)"""
    return [m for m in seq if isinstance(m, nn_blocks._ConvBlock)]


def mlp_rows(x2d: torch.Tensor, seq, ticket=None) -> torch.Tensor:
    blocks = _blocks(seq)
    if FUSED_MLP2_ROWS and len(blocks) == 2 and x2d.size(1) == 128 and x2d.is_contiguous():
        # the heads (128 -> 128 -> 1 / 40): both layers in one kernel, the activation between them stays in registers
        (w1, b1, r1), (w2, b2, r2) = _row_weights(blocks[0]), _row_weights(blocks[1])
        y = _C.mlp2_rows(x2d, w1, b1, r1, w2, b2, r2, ticket)
        if y is not None:
            return y
    for blk in blocks:
        x2d = _layer(x2d, blk)
    return x2d


def supported(model) -> bool:
    try:
        for m in model.modules():
            if isinstance(m, nn_blocks._ConvBlock):
                if not m._pointwise or not isinstance(getattr(m, "activation", None), (nn.ReLU, type(None))):
                    return False
        for sa in model.rpn.backbone_net.SA_modules:
            if sa.pool_method != "max_pool" or sa.npoint is None:
                return False
            for g, mlp in zip(sa.groupers, sa.mlps):
                if not isinstance(g, pn2_ops.QueryAndGroup) or any(b.conv.out_channels % 4 for b in _blocks(mlp)):
                    return False
        return True
    except AttributeError:
        return False


def _gather_gemm_ok(sa, grouper, blocks, c_feat: int, B: int) -> bool:
    return (FUSED_GATHER_GEMM and c_feat >= 16 and c_feat % 4 == 0 and grouper.use_xyz and len(blocks) >= 2 and
            blocks[0].conv.out_channels % 64 == 0 and (B * sa.npoint * grouper.nsample) % 64 == 0 and
            (B * sa.npoint * grouper.nsample) // 64 <= 65535)        # (the gather-GEMM kernels' grid: one row tile of 64 per workgroup, 16-bit y)


def _pair_limit(rows: int, dense_available: bool, key=None) -> int:
    """the launch gate of a scale with `rows` = B * npoint * nsample list entries: the compact kernels run iff the distinct pairs
    of this batch number <= limit, the dense ones iff > limit (-1: no gate, always compact).  The compact path beats the dense
    kernels up to ~55-60 % distinct rows (profiles/r02_compact_vs_dense_fill.txt: 46 vs 133 us at 5 %, 111 vs 129 at 48 %, 180 vs
    122 at 96 %); both are exact, so this is a speed decision only -- taken on the device, per batch."""
    if PAIR_DISPATCH != "device" or not dense_available or (key is not None and key in _COMPACT_ONLY.get()):
        return -1
    return max(0, int(COMPACT_MAX_FILL * rows))          # 0: a batch always holds >= 1 pair, i.e. always dense


@torch.no_grad()
def list_fill(net, pointcloud: torch.Tensor):
    """distinct (centre, sample) pairs / all list entries of every ball-query scale of the backbone on this batch -- the statistic
    the launch gates act on -- as a list of dicts.  A diagnostic (bench.py's ``list_fill``): runs the sampling chain and the
    searches on their own and synchronises; the forward pass itself never reads these numbers on the host."""
    xyz = pointcloud[..., 0:3].contiguous()
    rows = []
    for level, sa in enumerate(net.SA_modules):
        _, new_xyz = (pn2_ops.furthest_point_sample_gather_nested if NESTED_FPS and level >= 1 else pn2_ops.furthest_point_sample_gather)(xyz, sa.npoint)
        srt = pn2_ops.sort_points_x(xyz)
        for g in sa.groupers:
            nbr = _C.ball_query_lists(g.radius, g.nsample, xyz, new_xyz, srt)
            total = _C.compact_pairs(nbr)[2]
            rows.append({"level": level + 1, "n": xyz.size(1), "npoint": sa.npoint, "radius": float(g.radius), "nsample": g.nsample,
                         "distinct_per_list": float(total.item()) / max(nbr.numel() // g.nsample, 1),
                         "fill": float(total.item()) / max(nbr.numel(), 1)})
        xyz = new_xyz
    return rows


class _ZeroArena:
    """ONE cleared buffer per forward pass for everything that must start at zero -- the pooled outputs the compact SharedMLPs
    reduce into with an atomic max, the pair totals, the heads' tickets -- instead of a fill launch each (13 per batch).  Cleared on
    the caller's stream before the first sampling kernel, so the side streams (which start behind it) see it cleared."""

    def __init__(self, numel: int, device, clear: bool = True):
        n = (max(int(numel), 4) + 3) & ~3                         # whole 16-byte granules (ws3d_split_points_clear clears those)
        self.buf = (torch.zeros if clear else torch.empty)(n, dtype=torch.float32, device=device)
        self.dirty = not clear                                    # True: whoever splits the input rows clears it in the same launch
        self.off = 0
        self.fallbacks = 0                                        # takes the arena was too small for (a fill launch each: tests assert 0)

    def ensure_clear(self):
        if self.dirty:
            self.buf.zero_()
            self.dirty = False

    def take(self, shape, dtype=torch.float32):
        n = 1
        for d in shape:
            n *= int(d)
        if self.off + n > self.buf.numel():                     # (not sized for this caller: a fill of its own)
            self.fallbacks += 1
            self.fallback_shapes = getattr(self, "fallback_shapes", []) + [(tuple(shape), self.off, self.buf.numel())]
            return torch.zeros(shape, dtype=dtype, device=self.buf.device)
        v = self.buf[self.off:self.off + n]
        self.off += (n + 3) & ~3                                 # 16-byte aligned pieces
        return (v if dtype == torch.float32 else v.view(dtype)).view(shape)


LAST_ARENA_FALLBACKS = 0      # of the last rpn_forward: takes its zero arena could not serve (diagnostic; 0 for the networks _arena_for sizes)


def _arena_for(net, B: int, device, extra: int = 0, clear: bool = True) -> _ZeroArena:
    n = extra + 64
    if COMPACT_PAIRS:
        for sa in net.SA_modules:
            n += B * sa.npoint * sum(_blocks(mlp)[-1].conv.out_channels for mlp in sa.mlps) + 4
            n += 4 * len(sa.groupers)
        if CHAIN_MLP:
            n += _C.chain_ticket_ints() + 4          # the ticket counters of ws3d_chain_mlp3 (SA2)
    return _ZeroArena(n, device, clear)


class _PairList:
    """a (B, M, ns) neighbour list together with its compact distinct pairs (rowc, rowsrc, total)"""
    __slots__ = ("nbr", "pairs")

    def __init__(self, nbr, pairs):
        self.nbr, self.pairs = nbr, pairs


def _per_point_l1(sa, feats: torch.Tensor, nbrs):
    """(P, column offsets, [W_x per scale]): P = feats (B*N, C) @ [W_f of every scale whose first layer gathers its own rows]"""
    B, N, C = feats.shape
    wts = [(_row_weights_xyz_last(_blocks(mlp)[0])[0] if nbr is not None else None) for mlp, nbr in zip(sa.mlps, nbrs)]
    cache = sa.__dict__.get("_pp_cache")
    if cache is None or len(cache[0]) != len(wts) or any(a is not b for a, b in zip(cache[0], wts)):
        used = [w for w in wts if w is not None]
        wcat = torch.cat([w[:C] for w in used], dim=1).contiguous()                      # (C, sum O1)
        offs, w1xs, o = [], [], 0
        for w in wts:
            offs.append(o if w is not None else -1)
            w1xs.append(w[C:C + 3].contiguous() if w is not None else None)
            o += w.size(1) if w is not None else 0
        cache = (wts, wcat, offs, w1xs)
        if not torch.cuda.is_current_stream_capturing():
            sa.__dict__["_pp_cache"] = cache
    return torch.mm(feats.view(B * N, C), cache[1]), cache[2], cache[3]


def _lists_and_pairs(radius, nsample, xyz, new_xyz, sorted_xyz, zeros=None):
    """(lists (B, M, nsample), (rowc, rowsrc, total)): one launch when the fine-grid kernel covers the call, else two"""
    total = zeros.take((1,), torch.int32) if zeros is not None else None
    both = _C.ball_query_pairs(radius, nsample, xyz, new_xyz, sorted_xyz, total)
    if both is not None:
        return both
    nbr = _C.ball_query_lists(radius, nsample, xyz, new_xyz, sorted_xyz)
    return nbr, _C.compact_pairs(nbr)


def _neighbour_lists(sa, xyz, new_xyz, sorted_xyz, c_feat: int, zeros=None):
    """per grouper: the (B, npoint, nsample) neighbour list when its first layer gathers its own rows, else None.
    (The two scales' searches of a level on two streams -- a FOURTH side stream -- was measured in round 4: the whole eager pass
    ran 0.9 ms slower, every segment of it, profiles/r04_latency_segments.txt: three side streams + the caller's is what this runtime
    drives without loss.)"""
    B = xyz.size(0)
    lists = []
    want_pairs = [(_gather_gemm_ok(sa, g, _blocks(mlp), c_feat, B) and COMPACT_PAIRS and PER_POINT_L1 and _blocks(mlp)[0].conv.out_channels <= 256
                   and len(_blocks(mlp)) == 3) for g, mlp in zip(sa.groupers, sa.mlps)]
    if DUAL_SCALE_SEARCH and len(want_pairs) == 2 and all(want_pairs) and sorted_xyz is not None:
        totals = [zeros.take((1,), torch.int32) for _ in range(2)] if zeros is not None else None
        both = _C.ball_query_pairs2([g.radius for g in sa.groupers], [g.nsample for g in sa.groupers], xyz, new_xyz, sorted_xyz, totals)
        if both is not None:
            return [_PairList(nbr, pairs) for nbr, pairs in both]
        # (declined: the arena's two counters stay zero and unused, the per-scale calls below take their own)
    for grouper, mlp in zip(sa.groupers, sa.mlps):
        if not _gather_gemm_ok(sa, grouper, _blocks(mlp), c_feat, B):
            lists.append(None)
            continue
        if COMPACT_PAIRS and PER_POINT_L1 and _blocks(mlp)[0].conv.out_channels <= 256 and len(_blocks(mlp)) == 3:
            # the lists and their distinct pairs (coordinate-only work: on the search stream)
            nbr, pairs = _lists_and_pairs(grouper.radius, grouper.nsample, xyz, new_xyz, sorted_xyz, zeros)
            lists.append(_PairList(nbr, pairs))
        else:
            lists.append(_C.ball_query_lists(grouper.radius, grouper.nsample, xyz, new_xyz, sorted_xyz))
    return lists


def _side_streams(main: torch.cuda.Stream, count: int = 3, fresh: bool = False):
    """the three side streams of a forward pass: the sampling chain, the searches, and one for the second scale of a level / the
    second head.  Taken from the per-device pool that Stage1Pipeline's slots use (ws3d_amd/streams.py) -- never the caller's own
    stream: each HIP stream may claim a hardware queue, and a process that drives more queues than the device has descriptors
    gets time-sliced (measured: a pair per Stage1Pipeline slot, 60 streams, ran every kernel of the process ~1.6x slower; the
    limit sits at 24 queues: 20 pipeline slots + the null stream + a pair of its own = 23 was fine, a third one was not).
    Sharing the pool costs nothing: a pass that runs eagerly beside a pipeline in flight merely queues behind its slots -- the side
    streams are taken from the far end of the pool and never one that is capturing a hipGraph (streams.side_streams)."""
    from .streams import pass_streams
    return pass_streams(main, count, fresh)


class _Geometry:
    """Everything of a forward pass that depends on coordinates only -- the sampling chain of levels 2.., the binned copies,
    the neighbour lists of the gather-GEMM groupers and the 3-NN indices / weights of all FP modules -- issued on two side
    streams right after the first level's sampling, so that it runs beside the feature path (SharedMLP GEMMs) instead of
    in front of it: at batch 8 the sampling kernels occupy 8 of 256 CUs.  The caller's stream waits on one event per product.
    All tensors stay referenced here until the forward pass returns (they are allocated on the side streams; the side streams'
    first action of the next pass is to wait for the caller's stream, so their reuse is ordered; passes issued from different
    caller streams share the side streams and are therefore ordered on them as well)."""

    def __init__(self, net, xyz: torch.Tensor, c0: int, zeros=None):
        sas = list(net.SA_modules)
        main = torch.cuda.current_stream(xyz.device)
        s_fps, s_search, s_aux = _side_streams(main, fresh=True)      # resolved once per pass: rpn_forward's heads reuse them
        self.aux = s_aux
        self.xyz = [xyz]
        # the binned copy of the input cloud needs the coordinates only: on the search stream BESIDE the first level's sampling kernel
        # (round 2 measured this as a loss -- 5.22-5.26 vs 5.19-5.20 ms per batch -- when the sampling kernel re-read the scene every
        # step and anything on the memory path slowed it; the round-4 kernel keeps the scene in registers)
        srt0 = ev0 = None
        if BIN_INPUT_AHEAD:
            pre = torch.cuda.Event()
            pre.record(main)
        _, nx = pn2_ops.furthest_point_sample_gather(xyz, sas[0].npoint)            # everything waits for this one: caller's stream
        if BIN_INPUT_AHEAD:                  # (issued BEHIND the sampling launch: the host's time in front of it is on the critical path)
            s_search.wait_event(pre)
            with torch.cuda.stream(s_search):
                srt0 = pn2_ops.sort_points_x(xyz)
                ev0 = torch.cuda.Event()
                ev0.record(s_search)
        self.xyz.append(nx)
        start = torch.cuda.Event()
        start.record(main)
        fps_done = [start]
        s_fps.wait_event(start)
        with torch.cuda.stream(s_fps):
            if NESTED_FPS and NESTED_CHAIN and len(sas) > 1:
                # every level after the first samples the previous level's centres, in the order they were picked: one chain call
                for _, nx in pn2_ops.furthest_point_sample_gather_nested_chain(self.xyz[-1], [sa.npoint for sa in sas[1:]]):
                    self.xyz.append(nx)
                ev = torch.cuda.Event()
                ev.record(s_fps)
                fps_done.extend([ev] * (len(sas) - 1))
            else:
                for sa in sas[1:]:
                    _, nx = (pn2_ops.furthest_point_sample_gather_nested if NESTED_FPS else pn2_ops.furthest_point_sample_gather)(self.xyz[-1], sa.npoint)
                    self.xyz.append(nx)
                    ev = torch.cuda.Event()
                    ev.record(s_fps)
                    fps_done.append(ev)
        self.sorted, self.nbr, self.sa_ready = [srt0], [None], [ev0]
        self.nn, self.nn_ready = [], []
        c_feat = [c0] + [sum(_blocks(mlp)[-1].conv.out_channels for mlp in sa.mlps) for sa in sas]
        with torch.cuda.stream(s_search):
            # first what the SA levels wait for (their lists, in the order they run), then the 3-NN of the FP modules, deepest
            # first as they run: with the SharedMLPs over the compact pairs the caller's stream reaches level 2 some 0.1 ms
            # after level 1's sampling, and the 3-NN of FP1 (131072 queries) in front of level 2's lists was on its path
            for i in range(1, len(sas)):
                s_search.wait_event(fps_done[i - 1])                                 # level i exists: its binned copy does not need level i + 1
                srt = pn2_ops.sort_points_x(self.xyz[i], GRID_MIN_N)
                self.sorted.append(srt)
                s_search.wait_event(fps_done[i])                                     # level i + 1 exists
                self.nbr.append(_neighbour_lists(sas[i], self.xyz[i], self.xyz[i + 1], srt, c_feat[i], zeros))
                ev = torch.cuda.Event()
                ev.record(s_search)
                self.sa_ready.append(ev)
            self.nn, self.nn_ready = [None] * len(sas), [None] * len(sas)
            for i in range(len(sas) - 1, -1, -1):
                self.nn[i] = _C.three_nn_with_weights(self.xyz[i], self.xyz[i + 1], pn2_ops.sort_points_xz(self.xyz[i + 1]),
                                                      self.sorted[i] if QUERY_CELL_ORDER else None)
                ev = torch.cuda.Event()
                ev.record(s_search)
                self.nn_ready[i] = ev
        self.main, self.side = main, (s_fps, s_search, s_aux)

    def release(self):
        """every consumer of the side streams' tensors has been issued on the caller's stream: later side-stream work (of
        any caller) is ordered behind them, so the allocator may hand the blocks out again"""
        if torch.cuda.is_current_stream_capturing():
            return          # inside a capture the side streams are joined already (every product was waited for): nothing may follow the join
        done = torch.cuda.Event()
        done.record(self.main)
        for s in self.side:
            s.wait_event(done)


def sa_forward(sa, xyz: torch.Tensor, feats: torch.Tensor, geo: _Geometry = None, level: int = 0, zeros: _ZeroArena = None, binned: list = None,
               new_xyz_pre: torch.Tensor = None, prebinned=None):
    """xyz (B,N,3), feats (B,N,C) or None -> new_xyz (B,M,3), new_feats (B,M,sum O); `binned` (a list) receives the level's binned
    copy of xyz (or None): the FP module of this level takes its 3-NN queries in that order; prebinned: (that copy,) when the caller
    binned this level already (backbone_forward's one launch for levels 2..)"""
    B = xyz.size(0)
    c_feat = 0 if feats is None else feats.size(2)
    if geo is not None:
        new_xyz = geo.xyz[level + 1]
        if level >= 1:
            geo.main.wait_event(geo.sa_ready[level])
            sorted_xyz, nbrs = geo.sorted[level], geo.nbr[level]
        else:
            if geo.sa_ready[0] is not None:
                geo.main.wait_event(geo.sa_ready[0])
            sorted_xyz = geo.sorted[0] if geo.sa_ready[0] is not None else pn2_ops.sort_points_x(xyz)
            nbrs = _neighbour_lists(sa, xyz, new_xyz, None, c_feat, zeros)
    else:
        if new_xyz_pre is not None:                  # the caller sampled this level already (the chain call of backbone_forward)
            new_xyz = new_xyz_pre
        else:
            _, new_xyz = (pn2_ops.furthest_point_sample_gather_nested if NESTED_FPS and level >= 1 else pn2_ops.furthest_point_sample_gather)(xyz, sa.npoint)
        sorted_xyz = prebinned[0] if prebinned is not None else pn2_ops.sort_points_x(xyz, GRID_MIN_N)
        nbrs = _neighbour_lists(sa, xyz, new_xyz, sorted_xyz, c_feat, zeros)
    if binned is not None:
        binned.append(sorted_xyz)
    widths = [_blocks(mlp)[-1].conv.out_channels for mlp in sa.mlps]
    if not COMPACT_PAIRS:
        out = torch.empty((B * sa.npoint, sum(widths)), dtype=torch.float32, device=xyz.device)
    elif zeros is not None:
        out = zeros.take((B * sa.npoint, sum(widths)))                               # zeros: the atomic max
    else:
        out = torch.zeros((B * sa.npoint, sum(widths)), dtype=torch.float32, device=xyz.device)
    cols = [sum(widths[:i]) for i in range(len(widths))]
    pp = [None]
    # first level: both scales' lists + pair tables in ONE launch when every scale takes the lists-only SharedMLP kernel
    sa1_pairs = {}
    if (DUAL_SCALE_SEARCH and COMPACT_PAIRS and FUSED_SA_MLP and SA1_FROM_LISTS and len(widths) == 2 and feats is not None and feats.size(2) == 1
            and sorted_xyz is not None and all(n_ is None for n_ in nbrs) and all(g_.use_xyz and len(_blocks(m_)) == 3 for g_, m_ in zip(sa.groupers, sa.mlps))):
        totals = [zeros.take((1,), torch.int32) for _ in range(2)] if zeros is not None else None
        both = _C.ball_query_pairs2([g_.radius for g_ in sa.groupers], [g_.nsample for g_ in sa.groupers], xyz, new_xyz, sorted_xyz, totals)
        if both is not None:
            sa1_pairs = dict(enumerate(both))

    def run_scale(si: int) -> None:
        """the SharedMLP + pool of scale si into its column slice of `out`"""
        grouper, mlp, width, nbr, col = sa.groupers[si], sa.mlps[si], widths[si], nbrs[si], cols[si]
        blocks = _blocks(mlp)
        pairs = None
        if isinstance(nbr, _PairList):
            nbr, pairs = nbr.nbr, nbr.pairs
        if nbr is not None:
            # neighbour lists only, then layer 1 gathers its own rows: the (rows, 3 + C) grouped tensor never exists
            wt1, b1, r1 = _row_weights_xyz_last(blocks[0])
            if PER_POINT_L1 and len(blocks) >= 3:
                # layer 1 is linear in the grouped row: its feature part is ONE product over the level's points (both scales in
                # one GEMM), the pairs only gather that row and add the xyz term
                if pp[0] is None:
                    pp[0] = _per_point_l1(sa, feats, nbrs)
                pmat, offs, w1xs = pp[0]
                o1 = blocks[0].conv.out_channels
                y = None
                wt3, b3, r3 = _row_weights(blocks[-1])
                if pairs is not None and o1 <= 256 and len(blocks) == 3 and r3:
                    # the whole SharedMLP over the DISTINCT (centre, sample) pairs: padded rows repeat row 0 and cannot change the
                    # maximum; `out` is zero-initialised for the atomic max of the last layer.  Device-side dispatch: the dense form
                    # (all rows, pooled in registers, stored) is launched behind it into the same buffers, and the pair total of
                    # this batch decides in every kernel's prologue which of the two runs
                    wt2, b2, r2 = _row_weights(blocks[1])
                    rows, ns, o3 = nbr.numel(), grouper.nsample, wt3.size(1)
                    dense_ok = (o1 in (64, 128) and (sa.npoint * ns) % 64 == 0 and rows // 64 <= 65535 and o3 % 64 == 0 and
                                wt2.size(1) % 4 == 0 and ns in (16, 32))
                    limit = _pair_limit(rows, dense_ok, (level, si))
                    fused = _C.pgather_gemm3_compact(pmat, offs[si], o1, xyz, new_xyz, pairs, w1xs[si], b1, r1, wt2, b2, r2, wt3, b3, out, col,
                                                     limit=limit, max_lds=FUSED_COMPACT3_MAX_LDS)
                    yc = None if fused else _C.pgather_gemm2_compact(pmat, offs[si], o1, xyz, new_xyz, pairs, w1xs[si], b1, r1, wt2, b2, r2, limit=limit)
                    if fused or (yc is not None and _C.gemm_pool_compact(yc, pairs, wt3, b3, out, col, limit=limit)):
                        if limit >= 0:
                            gate = (pairs[2], limit)
                            yd = _C.pgather_gemm2(pmat, offs[si], o1, xyz, new_xyz, nbr, w1xs[si], b1, r1, wt2, b2, r2, out=yc, gate=gate)
                            if yd is None or not _C.gemm_pool(yd, wt3, b3, r3, ns, out, col, gate=gate):
                                # the dense side declined a call dense_ok let through (an alignment or stride check the C side makes and
                                # this restatement does not): the compact kernels once more, ungated -- both are idempotent (plain row
                                # stores; an atomic max into the zeroed pool), so the batch is complete whichever side of the limit it is on
                                yc = _C.pgather_gemm2_compact(pmat, offs[si], o1, xyz, new_xyz, pairs, w1xs[si], b1, r1, wt2, b2, r2, limit=-1)
                                if yc is None or not _C.gemm_pool_compact(yc, pairs, wt3, b3, out, col, limit=-1):
                                    raise RuntimeError("neither the gated dense nor the ungated compact kernels took the shape")
                        return
                if o1 <= 128:
                    wt2, b2, r2 = _row_weights(blocks[1])
                    y = _C.pgather_gemm2(pmat, offs[si], o1, xyz, new_xyz, nbr, w1xs[si], b1, r1, wt2, b2, r2)
                    rest_pp = blocks[2:-1]
                if y is None:
                    y = _C.pgather_rows(pmat, offs[si], o1, xyz, new_xyz, nbr, w1xs[si], b1, r1)
                    rest_pp = blocks[1:-1]
                if y is not None:
                    for blk in rest_pp:
                        y = _layer(y, blk)
                    wt, bias, relu = _row_weights(blocks[-1])
                    if not (FUSED_GEMM_POOL and _C.gemm_pool(y, wt, bias, relu, grouper.nsample, out, col)):
                        _C.rowmax_rows(_layer(y, blocks[-1]), grouper.nsample, out, col)
                    return
            y = None
            rest = blocks[1:-1]
            if FUSED_GATHER_GEMM2 and len(blocks) >= 3 and blocks[0].conv.out_channels <= 128:
                # layers 1 and 2 in one kernel: the first activation never leaves the chip.  (Measured per scale at batch 8,
                # fused vs gather-GEMM + library GEMM: SA2 44 vs 59 and 101 vs 115 us, SA3 60 vs 63 and 107 vs 112 us; SA4, 256
                # channels wide with only 128 / 256 row tiles: 98 vs 52 and 120 vs 93 us -- one workgroup per CU, long serial K
                # loops -- so the widest level keeps the two-kernel form)
                wt2, b2, r2 = _row_weights(blocks[1])
                y = _C.gather_gemm2(feats, xyz, new_xyz, nbr, wt1, b1, r1, wt2, b2, r2)
                if y is not None:
                    rest = blocks[2:-1]
            if y is None:
                y = _C.gather_gemm(feats, xyz, new_xyz, nbr, wt1, b1, r1)
            if y is not None:
                for blk in rest:
                    y = _layer(y, blk)
                wt, bias, relu = _row_weights(blocks[-1])
                if not (FUSED_GEMM_POOL and _C.gemm_pool(y, wt, bias, relu, grouper.nsample, out, col)):
                    _C.rowmax_rows(_layer(y, blocks[-1]), grouper.nsample, out, col)
                return
            # (the kernel declined -- e.g. a non-contiguous feature tensor: the grouped path below covers every shape)
        if FUSED_SA_MLP and SA1_FROM_LISTS and feats is not None and feats.size(2) == 1 and grouper.use_xyz and len(blocks) == 3:
            # first level: lists only (no grouped tensor), the three layers chained in registers -- over the distinct pairs of the
            # lists (atomic max into the zeroed `out`) or over all their rows (pool in registers, stored), decided on the device
            layers = [_row_weights(b) for b in blocks]
            if not COMPACT_PAIRS:
                nbr1 = _C.ball_query_lists(grouper.radius, grouper.nsample, xyz, new_xyz, sorted_xyz)
                if _C.sa_mlp3_pool_lists(xyz, new_xyz, feats, nbr1, layers, out, col):
                    return
            else:
                nbr1, pairs1 = sa1_pairs[si] if si in sa1_pairs else _lists_and_pairs(grouper.radius, grouper.nsample, xyz, new_xyz, sorted_xyz, zeros)
                limit = _pair_limit(nbr1.numel(), layers[2][2] and nbr1.numel() % 32 == 0 and grouper.nsample in (16, 32), (level, si))
                if _C.sa_mlp3_pool_compact(xyz, new_xyz, feats, pairs1, layers, out, col, limit=limit):
                    if limit >= 0 and not _C.sa_mlp3_pool_lists(xyz, new_xyz, feats, nbr1, layers, out, col, gate=(pairs1[2], limit)):
                        if not _C.sa_mlp3_pool_compact(xyz, new_xyz, feats, pairs1, layers, out, col, limit=-1):      # (ungated, idempotent: see above)
                            raise RuntimeError("neither ws3d_sa_mlp3_pool_lists nor the ungated compact kernel took the shape")
                    return
        g = _C.query_and_group_nlc(grouper.radius, grouper.nsample, xyz, new_xyz, feats, grouper.use_xyz, sorted_xyz)
        rows = g.view(-1, g.size(3))
        # 4-channel level (dx,dy,dz,intensity): three layers + pool in one kernel, nothing but the
        # pooled rows leaves the chip; otherwise the GEMM chain + pool kernel
        if not (FUSED_SA_MLP and rows.size(1) == 4 and len(blocks) == 3 and
                _C.sa_mlp3_pool(rows, grouper.nsample, [_row_weights(b) for b in blocks], out, col)):
            y = rows
            for blk in blocks[:-1]:
                y = _layer(y, blk)                                     # (B*M*ns, C_k), bias + ReLU in the GEMM epilogue
            wt, bias, relu = _row_weights(blocks[-1])
            # last layer + pool in one matrix-core kernel (only the pooled rows reach HBM); shapes it does not
            # cover: GEMM with fused bias + ReLU, then the pool kernel into the column slice
            if not (FUSED_GEMM_POOL and _C.gemm_pool(y, wt, bias, relu, grouper.nsample, out, col)):
                _C.rowmax_rows(_layer(y, blocks[-1]), grouper.nsample, out, col)

    def run_paired() -> bool:
        """both scales' compact SharedMLP in one launch per kernel (three -> the fused kernel; else layers 1 + 2, then layer 3 + pool);
        only where neither scale launches a gated dense twin.  False: nothing launched, the per-scale path runs"""
        if not (PAIRED_SCALES and COMPACT_PAIRS and PER_POINT_L1 and len(widths) == 2 and all(isinstance(n_, _PairList) for n_ in nbrs)):
            return False
        blk = [_blocks(m_) for m_ in sa.mlps]
        if any(len(b_) != 3 or b_[0].conv.out_channels > 256 for b_ in blk) or blk[0][0].conv.out_channels != blk[1][0].conv.out_channels:
            return False
        if any(_pair_limit(n_.nbr.numel(), True, (level, si_)) != -1 for si_, n_ in enumerate(nbrs)):
            return False
        if pp[0] is None:
            pp[0] = _per_point_l1(sa, feats, nbrs)
        pmat, offs, w1xs = pp[0]
        scales = []
        for si_ in range(2):
            _w1, b1, r1 = _row_weights_xyz_last(blk[si_][0])
            wt2, b2, r2 = _row_weights(blk[si_][1])
            wt3, b3, r3 = _row_weights(blk[si_][2])
            if not r3:
                return False
            scales.append({"pmat": pmat, "col0": offs[si_], "o1": blk[si_][0].conv.out_channels, "xyz": xyz, "new_xyz": new_xyz, "pairs": nbrs[si_].pairs,
                           "w1x": w1xs[si_], "b1": b1, "relu1": r1, "w2t": wt2, "b2": b2, "relu2": r2, "w3t": wt3, "b3": b3, "out2d": out, "col_offset": cols[si_]})
        if (CHAIN_MLP and zeros is not None and all(sc_["o1"] == 64 and sc_["w2t"].size(1) <= 96 and sc_["w3t"].size(1) == 128 for sc_ in scales)      # (the widths ws3d_chain_mlp3 covers: checked before the tickets are taken out of the arena)
                and _C.chain_mlp3(scales, zeros.take((_C.chain_ticket_ints(),), torch.int32))):
            return True
        if _C.compact_mlp_pair(3, scales, max_lds=FUSED_COMPACT3_MAX_LDS):
            return True
        mids = _C.compact_mlp_pair(2, scales)
        if mids is None:
            return False
        if not _C.compact_mlp_pair(1, scales, mids=mids):
            for sc_, mid in zip(scales, mids):
                if not _C.gemm_pool_compact(mid, sc_["pairs"], sc_["w3t"], sc_["b3"], out, sc_["col_offset"], limit=-1):
                    raise RuntimeError("ws3d_gemm_pool_compact declined a shape ws3d_pgather_gemm2_compact took")
        return True

    if run_paired():
        pass
    elif geo is not None and geo.aux is not None and PARALLEL_SCALES and len(widths) == 2:
        # the two scales of a multi-scale level are independent and neither fills the chip at batch 8: the second one on a side
        # stream beside the first.  What both read is ready on the caller's stream at this point (the level's features, the
        # per-point product of layer 1 -- taken here, once, for both -- and, waited for above, the neighbour lists)
        if PER_POINT_L1 and any(n is not None and len(_blocks(m)) >= 3 for n, m in zip(nbrs, sa.mlps)):
            pp[0] = _per_point_l1(sa, feats, nbrs)
        fork = torch.cuda.Event()
        fork.record(geo.main)
        geo.aux.wait_event(fork)
        with torch.cuda.stream(geo.aux):
            run_scale(1)
            join = torch.cuda.Event()
            join.record(geo.aux)
        run_scale(0)
        geo.main.wait_event(join)
    else:
        for si in range(len(widths)):
            run_scale(si)
    return new_xyz, out.view(B, sa.npoint, -1)


def fp_forward(fp, unknown: torch.Tensor, known: torch.Tensor, unknown_feats, known_feats: torch.Tensor, nn3=None, sorted_unknown=None,
               known_xz=None):
    """unknown (B,n,3), known (B,m,3), unknown_feats (B,n,C1) or None, known_feats (B,m,C2) -> (B,n,O); sorted_unknown: a binned
    copy of `unknown` (the level's ball-query copy) for the 3-NN's query order; known_xz: (sort_points_xz(known),) when the caller
    has it already"""
    B, n = unknown.size(0), unknown.size(1)
    idx, weight = nn3 if nn3 is not None else _C.three_nn_with_weights(unknown, known, known_xz[0] if known_xz is not None else pn2_ops.sort_points_xz(known),
                                                                        sorted_unknown if QUERY_CELL_ORDER else None)
    c2 = known_feats.size(2)
    c1 = 0 if unknown_feats is None else unknown_feats.size(2)
    blocks = _blocks(fp.mlp)
    # (the fused kernel has no split-K: the deepest module -- 2048 rows x K = 1536 at batch 8, 256 workgroups of 96 k-tiles -- is
    # faster as interpolate + library GEMM, 55 vs 84 us)
    if PER_POINT_FP and blocks and blocks[0].conv.out_channels % 4 == 0:
        # interpolation is linear: the first layer's product over the interpolated channels is taken over the KNOWN points
        # (a quarter of the rows), the skip channels over the unknown ones, and one row kernel interpolates + adds + ReLU
        wt1, b1, r1 = _row_weights(blocks[0])
        wa, wb = _split_rows(blocks[0], wt1, c2)
        m = known_feats.size(1)
        q = torch.mm(known_feats.reshape(B * m, c2), wa).view(B, m, -1)
        fuse2 = len(blocks) == 2 and B * n >= FUSED_QINTERP_GEMM_MIN_ROWS
        if fuse2:
            wt2, b2, r2 = _row_weights(blocks[1])
        if c1 > 4:
            lin = torch.mm(unknown_feats.reshape(B * n, c1), wb) if b1 is None else torch.addmm(b1, unknown_feats.reshape(B * n, c1), wb)
            y = _C.qinterp_gemm(q, idx, weight, wt2, b2, r2, lin=lin, relu=r1) if fuse2 else None
            if y is not None:
                return y.view(B, n, -1)
            y = _C.qinterp_rows(q, idx, weight, lin=lin, relu=r1)
        else:
            sk = None if c1 == 0 else unknown_feats.contiguous()
            y = _C.qinterp_gemm(q, idx, weight, wt2, b2, r2, skip=sk, wb=wb if c1 else None, bias=b1, relu=r1) if fuse2 else None
            if y is not None:
                return y.view(B, n, -1)
            y = _C.qinterp_rows(q, idx, weight, skip=sk, wb=wb if c1 else None, bias=b1, relu=r1)
        if y is not None:
            for blk in blocks[1:]:
                y = _layer(y, blk)
            return y.view(B, n, -1)
    if FUSED_INTERP_GEMM and blocks and blocks[0].conv.out_channels % 64 == 0 and B * n >= 8192:
        wt1, b1, r1 = _row_weights(blocks[0])
        y = _C.interp_gemm(known_feats.contiguous(), None if unknown_feats is None else unknown_feats.contiguous(), idx, weight, wt1, b1, r1)
        if y is not None:
            for blk in blocks[1:]:
                y = _layer(y, blk)
            return y.view(B, n, -1)
    cat = torch.empty((B, n, c2 + c1), dtype=torch.float32, device=unknown.device)
    _C.three_interpolate_nlc(known_feats, idx, weight, cat)         # left columns, any row stride (4-byte-aligned 16-byte stores)
    if c1:
        cat[:, :, c2:] = unknown_feats
    return mlp_rows(cat.view(B * n, c2 + c1), fp.mlp).view(B, n, -1)


@torch.no_grad()
def backbone_forward(net, pointcloud: torch.Tensor, zeros: _ZeroArena = None):
    """Pointnet2MSG.forward on channels-last tensors -> xyz (B,N,3), features (B,N,C)"""
    if FUSED_PROLOGUE and pointcloud.is_cuda and pointcloud.dim() == 3 and pointcloud.is_contiguous() and pointcloud.size(-1) > 3:
        if zeros is None:
            zeros = _arena_for(net, pointcloud.size(0), pointcloud.device, clear=False)
        xyz, feats = _C.split_points_clear(pointcloud, zeros.buf if zeros.dirty else None)      # the pass's first launch: side streams start behind it
        zeros.dirty = False
    else:
        xyz = pointcloud[..., 0:3].contiguous()
        feats = pointcloud[..., 3:].contiguous() if pointcloud.size(-1) > 3 else None
        if zeros is None:
            zeros = _arena_for(net, xyz.size(0), xyz.device)
        zeros.ensure_clear()
    # (not while a hipGraph is being captured: a graph with such branches replays slower than one stream on this runtime --
    # measured 1,657 vs 3,813 scenes/s at 8 graphs in flight -- so Stage1Pipeline's graphs keep the serial order)
    ahead = _geometry_ahead_now() and (GEOMETRY_IN_CAPTURE or not torch.cuda.is_current_stream_capturing())
    geo = _Geometry(net, xyz, 0 if feats is None else feats.size(2), zeros) if ahead else None
    l_xyz, l_feats, binned = [xyz], [feats], []
    try:
        sas, chain, pre_grid, pre_xz, nn_pre = list(net.SA_modules), None, {}, {}, {}
        for level, sa in enumerate(sas):
            pre = chain[level - 1][1] if (chain is not None and level >= 1) else None
            nx, nf = sa_forward(sa, l_xyz[-1], l_feats[-1], geo, level, zeros, binned, pre, pre_grid.get(level))
            l_xyz.append(nx)
            l_feats.append(nf)
            if geo is None and level == 0 and NESTED_FPS and NESTED_CHAIN and len(sas) > 1:
                # serial order (graph capture): the levels below the first in one chain call, right behind the first level's centres
                chain = pn2_ops.furthest_point_sample_gather_nested_chain(nx, [s_.npoint for s_ in sas[1:]])
                if MERGED_BINNING:
                    # every level's coordinates exist now: the ball queries' grids of levels 2.. and the (x, z) grids of the FP modules'
                    # known sets in ONE launch (the decisions of sort_points_x(.., GRID_MIN_N) / sort_points_xz: which levels get one)
                    lv = [nx] + [c[1] for c in chain]                                  # coordinates of levels 1 .. len(sas)
                    jobs = [(x_, "grid") for x_ in lv[:-1] if x_.size(1) >= GRID_MIN_N] + [(x_, "xz") for x_ in lv if x_.size(1) >= 256]
                    bufs = iter(_C.sort_points_jobs(jobs) if _C.BQ_FINE_GRID else [])
                    if _C.BQ_FINE_GRID:
                        pre_grid = {k + 1: ((next(bufs),) if x_.size(1) >= GRID_MIN_N else (None,)) for k, x_ in enumerate(lv[:-1])}
                        pre_xz = {k + 1: (next(bufs) if x_.size(1) >= 256 else None) for k, x_ in enumerate(lv)}
                        if MERGED_THREE_NN:
                            # ... and behind it the 3-NN of every FP module whose known set got an (x, z) grid (queries of level k
                            # against the centres of level k + 1; query order: level k's ball-query grid, as fp_forward passes it)
                            allx = [l_xyz[0]] + lv
                            qgrid = {0: binned[0]}
                            qgrid.update({k_: v_[0] for k_, v_ in pre_grid.items()})
                            lvls = [k_ for k_ in range(len(sas)) if pre_xz.get(k_ + 1) is not None and allx[k_ + 1].size(1) <= 4096]
                            got = _C.three_nn_jobs([(allx[k_], allx[k_ + 1], pre_xz[k_ + 1], qgrid.get(k_) if QUERY_CELL_ORDER else None) for k_ in lvls])
                            if got is not None:
                                nn_pre = dict(zip(lvls, got))
        for i in range(-1, -(len(net.FP_modules) + 1), -1):
            lvl = len(l_xyz) + i - 1                                       # unknown level of this module
            nn3 = nn_pre.get(lvl)
            if geo is not None:
                geo.main.wait_event(geo.nn_ready[lvl])
                nn3 = geo.nn[lvl]
            known_level = len(l_xyz) + i
            l_feats[i - 1] = fp_forward(net.FP_modules[i], l_xyz[i - 1], l_xyz[i], l_feats[i - 1], l_feats[i], nn3, binned[len(l_xyz) + i - 1],
                                        (pre_xz[known_level],) if known_level in pre_xz else None)
    finally:
        if geo is not None:       # also when a layer raised: the side streams' tensors go back to their pools behind the caller's stream
            geo.release()
    return l_xyz[0], l_feats[0]


@torch.no_grad()
def rpn_forward(model, pts_input: torch.Tensor, defer_reg_join: bool = False) -> dict:
    """defer_reg_join (eager side-stream mode only): the regression head runs on a side stream beside the classification head;
    with the flag set the caller's stream is NOT made to wait for it here -- the dict then carries the event ``rpn_reg_ready`` that a
    consumer of ``rpn_reg`` must wait for (stage1.proposals_from_rpn does: the top-k over the scores runs beside the head)."""
    rpn = model.rpn
    zeros = _arena_for(rpn.backbone_net, pts_input.size(0), pts_input.device, extra=8, clear=not FUSED_PROLOGUE)      # ONE clear for the whole pass (with the prologue: inside its launch)
    xyz, feats = backbone_forward(rpn.backbone_net, pts_input, zeros)     # (B,N,3), (B,N,128)
    B, N, C = feats.shape
    rows = feats.view(B * N, C)
    tickets = zeros.take((2,), torch.int32) if FUSED_MLP2_ROWS else (None, None)
    reg_ready = None
    if PARALLEL_HEADS and _geometry_ahead_now() and not torch.cuda.is_current_stream_capturing():
        main = torch.cuda.current_stream(rows.device)
        aux = _side_streams(main)[2]
        fork = torch.cuda.Event()
        fork.record(main)
        aux.wait_event(fork)
        with torch.cuda.stream(aux):
            rpn_reg = mlp_rows(rows, rpn.rpn_reg_layer, tickets[1:2]).view(B, N, -1)
            reg_ready = torch.cuda.Event()
            reg_ready.record(aux)
        # rpn_reg's block belongs to the aux stream's pool but is read on the caller's stream (and whichever stream consumes the
        # dict): without this the caching allocator may hand the block to another pass's aux-stream allocation as soon as the
        # tensor is dropped, while a kernel of the caller's stream still reads it (several caller streams share the pooled aux)
        rpn_reg.record_stream(main)
        rpn_cls = mlp_rows(rows, rpn.rpn_cls_layer, tickets[0:1]).view(B, N, -1)
        if not defer_reg_join:
            main.wait_event(reg_ready)
            reg_ready = None
    else:
        rpn_cls = mlp_rows(rows, rpn.rpn_cls_layer, tickets[0:1]).view(B, N, -1)
        rpn_reg = mlp_rows(rows, rpn.rpn_reg_layer, tickets[1:2]).view(B, N, -1)
    global LAST_ARENA_FALLBACKS
    LAST_ARENA_FALLBACKS = zeros.fallbacks if not zeros.fallbacks else getattr(zeros, "fallback_shapes", zeros.fallbacks)
    out = {"rpn_cls": rpn_cls, "rpn_reg": rpn_reg, "backbone_xyz": xyz,
           "backbone_features": feats.transpose(1, 2),                   # (B,C,N) view of the (B,N,C) tensor
           "backbone_features_nlc": feats}
    if reg_ready is not None:
        out["rpn_reg_ready"] = reg_ready
    return out
