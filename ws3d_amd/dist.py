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


def _local_device(local: int, world: int, backend: str) -> int:
    """The device of local rank `local`.  Over RCCL every rank needs a device of its own: more ranks than devices raises instead of
    wrapping (two ranks on one device hang or fail late inside the collective library).  Only the explicit test mode
    (backend "gloo": control-flow runs of the multi-rank paths with the ranks sharing a device) wraps."""
    ndev = torch.cuda.device_count()
    if backend == "nccl" and (world > ndev or local >= ndev):
        raise RuntimeError("ws3d_amd.dist.init: backend 'nccl' (RCCL) needs one device per rank, but WORLD_SIZE=%d / LOCAL_RANK=%d and "
                           "this node shows %d device(s); WS3D_DIST_BACKEND=gloo is the test mode that lets ranks share a device"
                           % (world, local, ndev))
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


def run_sharded(global_batch: int, compute: Callable[[int, int], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]):
    """compute(start, end) -> (boxes (b,K,7), scores (b,K), count (b,)) for the scenes this rank
    owns; returns the gathered (global_batch,K,8) proposals and (global_batch,) counts."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    s, e = shard_range(global_batch, world, rank)
    boxes, scores, count = compute(s, e)
    return all_gather_proposals(pack_proposals(boxes, scores), count, global_batch)
