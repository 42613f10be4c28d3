"""Per-scene data parallelism for Stage-1 inference.

Scenes are independent (every kernel indexes the batch dimension and nothing crosses it), so
the path shards with NO data-path collective: one process per GPU, rank r owns the contiguous
scene range [r*B/W, (r+1)*B/W) of the global batch, full model replica, weights materialised
locally.  The ONLY exchange is one fixed-shape all-gather per batch of the per-scene
proposals ``(B/W, K, 8) f32 = [x, y, z, h, w, l, ry, score]`` plus the ``(B/W,)`` valid counts in the
same buffer (25.6 KB/rank at K=100: latency-bound on xGMI, no bucketing needed).  Backend "nccl" is RCCL on
ROCm; "gloo" is used by the CPU tests.  The reference's own multi-GPU story
(nn.DataParallel, tools/train_rpn.py:175-176) is not reproduced.
"""
from __future__ import annotations

import os
from typing import Callable, Tuple

import torch
import torch.distributed as dist


def _local_device(local: int, world: int, backend: str, local_world: int = None) -> int:
    """The device of local rank `local`.  Over RCCL every rank needs a device of its own ON ITS NODE: a LOCAL_RANK beyond the node's
    devices, or more ranks on this node (``LOCAL_WORLD_SIZE``, when the launcher exports it) than devices, raises instead of wrapping
    (two ranks on one device hang or fail late inside the collective library).  The global WORLD_SIZE is not part of the check -- a
    2-node x 8-GPU job has WORLD_SIZE 16 and 8 devices per node.  Only the explicit test mode (backend "gloo":
    control-flow runs of the multi-rank paths with the ranks sharing a device) wraps."""
    ndev = torch.cuda.device_count()
    if local_world is None:
        lw = os.environ.get("LOCAL_WORLD_SIZE")
        local_world = int(lw) if lw else None
    if backend == "nccl" and (local >= ndev or (local_world is not None and local_world > ndev)):
        raise RuntimeError("ws3d_amd.dist.init: backend 'nccl' (RCCL) needs one device per rank, but LOCAL_RANK=%d / LOCAL_WORLD_SIZE=%s "
                           "(WORLD_SIZE=%d) and this node shows %d device(s); WS3D_DIST_BACKEND=gloo is the test mode that lets ranks "
                           "share a device" % (local, local_world, world, ndev))
    return local % max(ndev, 1)


def init(backend: str = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the torchrun environment.  Returns (world, rank, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        # WS3D_DIST_BACKEND=gloo: control-flow runs of the multi-rank paths on a box with fewer GPUs than
        # ranks (ranks then share devices); production is nccl (= RCCL over xGMI)
        backend = backend or os.environ.get("WS3D_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if torch.cuda.is_available():
            local = _local_device(local, world, backend)
            torch.cuda.set_device(local)
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", local)
        elif backend == "nccl":
            raise RuntimeError("ws3d_amd.dist.init: backend 'nccl' (RCCL) without a HIP device")
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    elif torch.cuda.is_available():
        local = _local_device(local, world, dist.get_backend() if dist.is_initialized() else (backend or "nccl") if world > 1 else "single")
    return world, rank, local


def shard_range(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous scene range of `rank`; the remainder goes to the lowest ranks."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_proposals(boxes: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
    """(b,K,7),(b,K) -> (b,K,8)"""
    return torch.cat([boxes, scores.unsqueeze(-1)], dim=-1).contiguous()


def all_gather_proposals(packed: torch.Tensor, count: torch.Tensor, global_batch: int, force: bool = False):
    """Gather every rank's (b_r, K, 8) proposals + (b_r,) counts into (global_batch, K, 8) /
    (global_batch,) on every rank, in scene order.  Ranks may own different b_r (uneven
    division): shards are padded to the largest shard so the collective has a fixed shape.

    ONE collective per batch: the counts ride in the same fp32 buffer as the proposals (one extra
    column per scene; exact, counts are < 2**24), `all_gather_into_tensor` on the caller's current
    stream (RCCL).  gloo has no device all-gather, so with that backend device tensors are staged
    through host memory -- an explicit, synchronising test mode (``WS3D_DIST_BACKEND=gloo``: ranks
    sharing one GPU), never a fallback: any failure of the collective propagates.
    ``force`` runs the collective even at world size 1 (covers the RCCL call on a 1-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return packed, count
    world = dist.get_world_size()
    if world == 1 and not force:
        return packed, count
    b, K, F = packed.shape
    bmax = -(-global_batch // world)
    buf = packed.new_zeros((bmax, K * F + 1))
    buf[:b, :K * F] = packed.reshape(b, K * F)
    buf[:b, K * F] = count.to(packed.dtype)
    out = packed.new_empty((world * bmax, K * F + 1))
    if packed.is_cuda and dist.get_backend() == "gloo":
        host_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host_out, buf.cpu())
        out.copy_(host_out)
    else:
        dist.all_gather_into_tensor(out, buf)
    if global_batch != world * bmax:        # uneven shards: drop the padding rows
        rows = []
        for r in range(world):
            s, e = shard_range(global_batch, world, r)
            rows.append(torch.arange(r * bmax, r * bmax + (e - s), device=packed.device))
        out = out[torch.cat(rows)]
    return out[:, :K * F].reshape(-1, K, F), out[:, K * F].round().to(count.dtype)


class ProposalExchange:
    """The step's ONE collective with nothing allocated per step.

    ``send`` (bmax, K*8 + 1) and ``recv`` (world * bmax, K*8 + 1) are allocated once (per pipeline slot): the selection kernel writes a
    scene's K packed rows and, behind them, its count straight into ``send`` (``ws3d_select_proposals_send`` through
    ``stage1.proposals_from_rpn(with_packed=exchange.send)``), so ``gather()`` is exactly one ``all_gather_into_tensor`` on two resident
    buffers -- no pack launch, no slice writes, no allocator traffic -- issued on the caller's current stream (the slot's stream, behind
    the graph replay that filled ``send``).  Even shards return views of ``recv``; uneven shards (global_batch % world != 0) drop the
    padding rows with one pre-built index (``index_select``: the only allocation, and only in that case).  ``collectives`` counts the
    calls (tests/test_dist_gloo.py asserts one per step).  Rows of ``send`` beyond this rank's scenes stay zero from construction."""

    def __init__(self, local_batch: int, K: int, global_batch: int, device, world: int = None, rank: int = None, F: int = 8):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.K, self.F, self.b, self.global_batch = int(K), int(F), int(local_batch), int(global_batch)
        self.bmax = -(-self.global_batch // self.world)
        if self.b > self.bmax:
            raise ValueError("ProposalExchange: %d local scenes but the largest shard of %d over %d ranks holds %d" % (self.b, global_batch, self.world, self.bmax))
        cols = self.K * self.F + 1
        self.send = torch.zeros((self.bmax, cols), dtype=torch.float32, device=device)
        self.recv = torch.zeros((self.world * self.bmax, cols), dtype=torch.float32, device=device)
        self._host = torch.empty(self.recv.shape, dtype=torch.float32) if self.send.is_cuda else None      # gloo test mode only
        self.index = None
        if self.global_batch != self.world * self.bmax:
            rows = []
            for r in range(self.world):
                s_, e_ = shard_range(self.global_batch, self.world, r)
                rows.extend(range(r * self.bmax, r * self.bmax + (e_ - s_)))
            self.index = torch.tensor(rows, dtype=torch.long, device=device)
        self.collectives = 0

    def fill(self, packed: torch.Tensor, count: torch.Tensor) -> None:
        """for producers that do not write ``send`` themselves (the torch composition, tests): two copies into the resident buffer"""
        b = packed.size(0)
        self.send[:b, :self.K * self.F].copy_(packed.reshape(b, self.K * self.F))
        self.send[:b, self.K * self.F].copy_(count)

    def gather(self, force: bool = False):
        """-> (global_batch, K, F) proposals, (global_batch,) float counts (exact integers), in scene order, on every rank"""
        KF = self.K * self.F
        if not (dist.is_available() and dist.is_initialized()) or (self.world == 1 and not force):
            out = self.send[:self.b]
            return out[:, :KF].unflatten(1, (self.K, self.F)), out[:, KF]
        self.collectives += 1
        if self.send.is_cuda and dist.get_backend() == "gloo":      # explicit, synchronising test mode: ranks sharing one GPU
            dist.all_gather_into_tensor(self._host, self.send.cpu())
            self.recv.copy_(self._host)
        else:
            dist.all_gather_into_tensor(self.recv, self.send)
        out = self.recv if self.index is None else self.recv.index_select(0, self.index)
        return out[:, :KF].unflatten(1, (self.K, self.F)), out[:, KF]


def run_sharded(global_batch: int, compute: Callable[[int, int], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]):
    """compute(start, end) -> (boxes (b,K,7), scores (b,K), count (b,)) for the scenes this rank
    owns; returns the gathered (global_batch,K,8) proposals and (global_batch,) counts."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    s, e = shard_range(global_batch, world, rank)
    boxes, scores, count = compute(s, e)
    return all_gather_proposals(pack_proposals(boxes, scores), count, global_batch)
