"""Stage-1 (RPN) trainer: the counterpart of the reference's ``tools/train_rpn.py`` for this
repository's network and kernels.

    python -m ws3d_amd.train_rpn --synthetic 64 --batch_size 8 --total_iters 200 --output_dir out/
    python -m ws3d_amd.train_rpn --data_root /data/KITTI/object --batch_size 25 --total_iters 8000

Same flags as the reference (train_rpn.py:24-46): --batch_size, --total_iters, --ckpt_save_interval,
--workers, --output_dir, --mgpus, --ckpt, --pretrain_ckpt, --noise_kind, --weakly_num (+ --cfg_file,
accepted for command-line compatibility: the weaklyRPN.yaml values live in ``stage1.RPNConfig`` and
``TrainConfig``).  What it reproduces:
  * optimizer 'adam_onecycle' (train_rpn.py:84-101 + learning_schedules_fastai.py:54-77): Adam
    betas (mom, 0.99), decoupled weight decay p *= 1 - wd*lr on every parameter (fastai
    ``OptimWrapper(true_wd=True, bn_wd=True)``), one-cycle cosine lr / momentum schedule stepped
    every iteration; gradient-norm clipping at 1.0; BN-momentum step decay (train_rpn.py:113-118);
  * the loss (``ws3d_amd.losses.rpn_loss``) on Gaussian centre labels;
  * checkpoints ``ckpt/checkpoint_iter_%05d.pth`` = {'it', 'model_state', 'optimizer_state'}
    (train_utils.py:67-99) -- ``model_state`` interchanges with the reference (identical keys).
  * scene augmentation (rotation / scaling / flip, ``losses.scene_augmentation``) and the evaluation
    pass of ``Trainer.eval_epoch_rpn`` (``evaluate``: loss, point precision, centre recall).
Not reproduced: the reference's GT-sampling database (it needs the KITTI object crops) and
tensorboard.  ``--mgpus`` maps to one process per GPU
(``torchrun``, DistributedDataParallel over RCCL) instead of ``nn.DataParallel``.
"""
from __future__ import annotations

import argparse
import logging
import math
import os
import time
from dataclasses import dataclass
from typing import Iterator, Optional

import numpy as np
import torch
import torch.nn as nn
from torch.nn.utils import clip_grad_norm_

from . import kitti_io, loader, losses, pn2_ops, stage1


@dataclass(frozen=True)
class TrainConfig:
    """TRAIN block of tools/cfgs/weaklyRPN.yaml:69-96"""
    lr: float = 0.002
    weight_decay: float = 0.001
    moms: tuple = (0.95, 0.85)
    div_factor: float = 10.0
    pct_start: float = 0.4
    grad_norm_clip: float = 1.0
    bn_momentum: float = 0.1
    bn_decay: float = 0.5
    bnm_clip: float = 0.01
    bn_decay_step_list: tuple = (1000,)


# ----------------------------------------------------------------------------- schedules
def annealing_cos(start: float, end: float, pct: float) -> float:
    return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)


def one_cycle(step: int, total_step: int, lr_max: float, moms=(0.95, 0.85), div_factor: float = 10.0,
              pct_start: float = 0.4):
    """(lr, momentum) at iteration `step` (learning_schedules_fastai.py:40-77): cosine from
    lr_max/div_factor up to lr_max over the first pct_start of training, then down to 2e-6;
    momentum mirrors it between moms[0] and moms[1]"""
    a1 = int(pct_start * total_step)
    low = lr_max / div_factor
    if step >= a1:
        pct = (step - a1) / (total_step - a1)
        return annealing_cos(lr_max, 2e-6, pct), annealing_cos(moms[1], moms[0], pct)
    pct = step / a1
    return annealing_cos(low, lr_max, pct), annealing_cos(moms[0], moms[1], pct)


def bn_momentum_at(it: int, cfg: TrainConfig) -> float:
    decay = 1.0
    for s in cfg.bn_decay_step_list:
        if it >= s:
            decay *= cfg.bn_decay
    return max(cfg.bn_momentum * decay, cfg.bnm_clip)


def set_bn_momentum(model: nn.Module, momentum: float) -> None:
    """(the walk over ~400 modules is skipped while the value does not change: it changes at the
    few BN_DECAY steps only, and the step is launch-bound on the host)"""
    cache = model.__dict__.get("_ws3d_bn_momentum")
    if cache is not None and cache[0] == momentum and all(m.momentum == momentum for m in cache[1][:1]):
        return
    norms = [m for m in model.modules() if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d))]
    for m in norms:
        m.momentum = momentum
    model.__dict__["_ws3d_bn_momentum"] = (momentum, norms)


class AdamOneCycle:
    """Adam + decoupled weight decay + per-iteration one-cycle schedule (see module docstring)"""

    def __init__(self, params, total_step: int, cfg: TrainConfig = TrainConfig()):
        self.cfg, self.total_step = cfg, total_step
        self.opt = torch.optim.Adam([p for p in params if p.requires_grad], lr=cfg.lr / cfg.div_factor,
                                    betas=(cfg.moms[0], 0.99))
        self.lr, self.mom = cfg.lr / cfg.div_factor, cfg.moms[0]

    def schedule(self, it: int) -> None:
        self.lr, self.mom = one_cycle(it, self.total_step, self.cfg.lr, self.cfg.moms, self.cfg.div_factor, self.cfg.pct_start)
        for g in self.opt.param_groups:
            g["lr"] = self.lr
            g["betas"] = (self.mom, g["betas"][1])

    def zero_grad(self) -> None:
        self.opt.zero_grad(set_to_none=True)

    @torch.no_grad()
    def step(self) -> None:
        for g in self.opt.param_groups:     # decoupled weight decay, one multi-tensor launch per group
            torch._foreach_mul_([p for p in g["params"]], 1 - self.cfg.weight_decay * g["lr"])
        self.opt.step()

    def state_dict(self):
        return self.opt.state_dict()

    def load_state_dict(self, sd):
        self.opt.load_state_dict(sd)


# ----------------------------------------------------------------------------- checkpoints
def checkpoint_state(model: Optional[nn.Module] = None, optimizer=None, it=None) -> dict:
    if isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel)):
        model = model.module
    return {"it": it, "model_state": None if model is None else model.state_dict(),
            "optimizer_state": None if optimizer is None else optimizer.state_dict()}


def save_checkpoint(state: dict, filename: str = "checkpoint") -> str:
    path = "{}.pth".format(filename)
    torch.save(state, path)
    return path


def load_checkpoint(model: Optional[nn.Module] = None, optimizer=None, filename: str = "checkpoint", logger=None):
    """-> (it, epoch).  ``model_state`` interchanges with the reference's train_utils.save_checkpoint (same keys).
    ``optimizer_state`` interchanges only between runs of THIS trainer: the reference's comes from fastai's OptimWrapper
    (two parameter groups from split_bn_bias), AdamOneCycle has one -- such a state is skipped with a warning and the
    optimizer starts fresh at iteration ``it`` of the one-cycle schedule."""
    if not os.path.isfile(filename):
        raise FileNotFoundError(filename)
    ck = torch.load(filename, map_location="cpu")
    if model is not None and ck.get("model_state") is not None:
        model.load_state_dict(ck["model_state"])
    if optimizer is not None and ck.get("optimizer_state") is not None:
        state = ck["optimizer_state"]
        ours = optimizer.state_dict()
        same = (isinstance(state, dict) and "param_groups" in state and len(state["param_groups"]) == len(ours["param_groups"])
                and [len(g["params"]) for g in state["param_groups"]] == [len(g["params"]) for g in ours["param_groups"]])
        if same:
            optimizer.load_state_dict(state)
        else:
            import warnings
            warnings.warn("load_checkpoint: optimizer_state of '%s' has another parameter-group layout (e.g. written by the "
                          "reference's OptimWrapper); it is skipped, only model_state and `it` are restored" % filename)
    if logger:
        logger.info("==> loaded checkpoint '%s' (it %s)", filename, ck.get("it"))
    return ck.get("it", 0), ck.get("epoch", -1)


# ----------------------------------------------------------------------------- data
class SyntheticCenters:
    """`count` seeded KITTI-shaped scenes with car-centre annotations (the weak labels of WS3D are
    BEV centre clicks): {'pts_input' (N,4), 'gt_centers' (K,3), 'rpn_cls_label', 'rpn_reg_label'}"""

    def __init__(self, count: int, npoints: int = 16384, config_id: int = 8, augment: bool = False, rng=np.random):
        from . import synth
        self.synth, self.count, self.npoints, self.config_id = synth, count, npoints, config_id
        self.augment, self.rng = augment, rng
        self._scenes, self._items = {}, {}   # generated scans (and, without augmentation, their labels) are kept

    def __len__(self):
        return self.count

    def __getitem__(self, i: int) -> dict:
        if not self.augment and i in self._items:
            return self._items[i]
        if i not in self._scenes:
            seed = 1000 * self.config_id + i
            self._scenes[i] = (self.synth.lidar_cloud(self.npoints, seed),
                               self.synth.random_boxes3d(15, seed * 7919 + 13)[:, :3].astype(np.float32))
        pc, centres = self._scenes[i]
        if self.augment:   # AUG_DATA (rotation / scaling / flip), like the reference's TRAIN loader
            xyz, centres, _ = losses.scene_augmentation(pc[:, :3], centres, self.rng)
            pc = np.concatenate((xyz.astype(np.float32), pc[:, 3:]), axis=1)
            centres = centres.astype(np.float32)
        cls, reg = losses.gaussian_center_labels(pc[:, :3], centres)
        item = {"sample_id": i, "pts_input": pc, "gt_centers": centres, "rpn_cls_label": cls.astype(np.float32),
                "rpn_reg_label": reg}
        if not self.augment:
            self._items[i] = item
        return item


class KittiCenters:
    """KITTI directory: scans through ``kitti_io`` (no augmentation), centre annotations from
    label_2 (``noise_kind`` selects another label directory like the reference's --noise_kind), the
    first ``weakly_num`` scenes that contain a Car/Van"""

    def __init__(self, root: str, split: str = "train", npoints: int = 16384, noise_kind: Optional[str] = None,
                 weakly_num: int = 500, rng=np.random, augment: bool = False):
        self.augment = augment
        self.scenes = kitti_io.KittiScenes(root, split, npoints=npoints, rng=rng)
        self.label_sub = noise_kind or "label_2"
        keep = []
        for sid in self.scenes.sample_id_list:
            if self._objects(sid):
                keep.append(sid)
            if len(keep) >= weakly_num:
                break
        self.ids = keep

    def _objects(self, sid):
        """Car / Van objects inside PC_AREA_SCOPE: the reference's TRAIN-mode filtrate_objects (kitti_rcnn_dataset.py:
        115-135: class whitelist incl. Van, then check_pc_range on obj.pos) -- applied BEFORE the 'scene has an object'
        test and before the centres become labels, so the weakly_num subset and the labels match the reference's"""
        path = os.path.join(self.scenes.imageset_dir, self.label_sub, "%06d.txt" % sid)
        objs = kitti_io.read_label_file(path) if os.path.isfile(path) else []
        (x0, x1), (y0, y1), (z0, z1) = kitti_io.PC_AREA_SCOPE
        return [o for o in objs if o.cls_type in ("Car", "Van")
                and x0 <= o.pos[0] <= x1 and y0 <= o.pos[1] <= y1 and z0 <= o.pos[2] <= z1]

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i: int) -> dict:
        sid = self.ids[i]
        pts = kitti_io.rpn_input_from_scan(self.scenes.get_lidar(sid), self.scenes.get_calib(sid),
                                           self.scenes.get_image_shape(sid), self.scenes.npoints, True, self.scenes.rng)
        centres = np.array([o.pos for o in self._objects(sid)], dtype=np.float32).reshape(-1, 3)
        if self.augment:
            xyz, centres, _ = losses.scene_augmentation(pts[:, :3], centres, self.scenes.rng)
            pts = np.concatenate((xyz, pts[:, 3:]), axis=1)
            centres = centres.astype(np.float32)
        cls, reg = losses.gaussian_center_labels(pts[:, :3], centres)
        return {"sample_id": sid, "pts_input": pts.astype(np.float32), "gt_centers": centres,
                "rpn_cls_label": cls.astype(np.float32), "rpn_reg_label": reg}


def _collate(items) -> dict:
    return {"pts_input": np.stack([s["pts_input"] for s in items]).astype(np.float32),
            "rpn_cls_label": np.stack([s["rpn_cls_label"] for s in items]),
            "rpn_reg_label": np.stack([s["rpn_reg_label"] for s in items]),
            "sample_id": [s["sample_id"] for s in items]}


def batches(dataset, batch_size: int, rng: np.random.RandomState, rank: int = 0, world: int = 1, workers: int = 0,
            ahead: int = 3) -> Iterator[dict]:
    """endless shuffled mini-batches (drop_last, like the reference's DataLoader); with world > 1
    every rank takes its own slice of each global batch.  workers > 0 (the reference's --workers):
    the scenes of the next `ahead` batches are read / sampled / labelled by that many forked
    processes -- same batch composition and order as workers = 0; scenes whose preparation draws
    random numbers (the 16384-point sampler, augmentation) then use per-process streams."""
    n = len(dataset)
    if n < batch_size * world:
        raise ValueError(f"{n} scenes < global batch {batch_size * world}")

    def id_batches():
        while True:
            order = rng.permutation(n)
            for i0 in range(0, n - batch_size * world + 1, batch_size * world):
                yield [int(i) for i in order[i0 + rank * batch_size:i0 + (rank + 1) * batch_size]]

    worker_seed = int(rng.get_state()[1][0]) % (2 ** 31)      # derived without drawing: the batch order must not depend on `workers`
    for items in loader.item_batches(dataset, id_batches(), workers, ahead, seed=worker_seed):
        yield _collate(items)


class DevicePrefetcher:
    """Keeps one mini-batch ahead of the training step, on a side HIP stream.

    ``advance()`` takes the next host batch from `source`, uploads it on the side stream and runs the
    backbone's furthest-point-sampling chain there (``pointnet2_utils.sampling_plan``).  FPS is the one long serial kernel of the step -- 4095
    dependent rounds, one workgroup per scene, 5.5 ms at batch 8 with 8 of 256 CUs busy -- and it
    needs nothing but the input coordinates, so it runs beside the previous step's backward GEMM /
    BN kernels instead of heading the critical path.  ``train_step`` calls ``advance`` right after
    it has enqueued the backward pass (the host is otherwise idle until the gradient norm is read
    back), so building the next batch on the host overlaps device work too.  Everything happens on
    the calling thread: a worker thread was measured 30 % SLOWER than no prefetching at all (it
    competes with the kernel-launching thread for the interpreter lock).
    ``next()`` yields dicts of DEVICE tensors: ``pts_input``, ``rpn_cls_label`` / ``rpn_reg_label``
    (float32), ``sampling_plan``, plus ``sample_id``; the current stream is made to wait for the
    upload + sampling first."""

    KEYS = ("pts_input", "rpn_cls_label", "rpn_reg_label")

    def __init__(self, source: Iterator[dict], device, npoints):
        self.source, self.device, self.npoints = iter(source), torch.device(device), tuple(npoints)
        self.side = torch.cuda.Stream(device=self.device)
        self.pending = None                # (batch, ready event) | exception raised by the source
        self.timing = None                 # set to a list to collect (start, end) HIP events of the sampling chain

    def advance(self) -> None:
        """start the next batch (no-op if one is already pending)"""
        if self.pending is not None:
            return
        try:
            batch = next(self.source)
        except BaseException as exc:       # StopIteration included: re-raised by __next__
            self.pending = exc
            return
        with torch.no_grad(), torch.cuda.stream(self.side):
            out = {"sample_id": batch.get("sample_id")}
            for key in self.KEYS:
                # plain pageable upload: staging through pinned buffers + non_blocking copies was measured
                # slower here (36 vs 23 ms per iteration at batch 8)
                out[key] = torch.from_numpy(np.ascontiguousarray(batch[key], dtype=np.float32)).to(self.device)
            xyz = out["pts_input"][..., 0:3].contiguous()
            if self.timing is not None:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(self.side)
            out["sampling_plan"] = pn2_ops.sampling_plan(xyz, self.npoints)
            if self.timing is not None:
                t1.record(self.side)
                self.timing.append((t0, t1))
            ready = torch.cuda.Event()
            ready.record(self.side)
        self.pending = (out, ready)

    def __iter__(self):
        return self

    def __next__(self) -> dict:
        self.advance()
        pending, self.pending = self.pending, None
        if isinstance(pending, BaseException):
            raise pending
        out, ready = pending
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ready)
        for v in list(out.values()) + list(out["sampling_plan"]):
            if torch.is_tensor(v):
                v.record_stream(cur)          # allocated on the side stream, consumed on this one
        return out

    def close(self):
        self.pending = None


def _on_device(x, device) -> torch.Tensor:
    t = x if torch.is_tensor(x) else torch.from_numpy(x)
    return t.to(device).float()


# ----------------------------------------------------------------------------- training
def train_step(model: nn.Module, optimizer: AdamOneCycle, batch: dict, it: int, net_cfg: stage1.RPNConfig,
               train_cfg: TrainConfig, device, overlap=None) -> dict:
    """one iteration of Trainer._train_it (train_utils.py:137-147) with the per-iteration schedules.
    overlap: host work to do once the backward pass is enqueued (DevicePrefetcher.advance)"""
    set_bn_momentum(model, bn_momentum_at(it, train_cfg))
    optimizer.schedule(it)
    if not model.training:
        model.train()
    optimizer.zero_grad()
    inputs = {"pts_input": _on_device(batch["pts_input"], device)}
    if batch.get("sampling_plan") is not None:
        inputs["sampling_plan"] = batch["sampling_plan"]
    out = model(inputs)
    loss, tb = losses.rpn_loss(out["rpn_cls"], out["rpn_reg"], _on_device(batch["rpn_cls_label"], device),
                               _on_device(batch["rpn_reg_label"], device), net_cfg.loc_scope, net_cfg.loc_bin_size, lazy=True)
    loss.backward()
    if overlap is not None:
        overlap()
    tb["grad_norm"] = clip_grad_norm_(model.parameters(), train_cfg.grad_norm_clip)    # stays on the device ...
    optimizer.step()
    tb["loss"] = loss
    losses.resolve_scalars(tb)                       # ... the step's scalars are read back once, at its end
    tb["lr"] = optimizer.lr
    return tb


@torch.no_grad()
def evaluate(model: nn.Module, dataset, net_cfg: stage1.RPNConfig = stage1.DEFAULT_CFG, device="cuda:0",
             max_scenes: Optional[int] = None) -> dict:
    """Trainer.eval_epoch_rpn (train_utils.py:149-249), batch size 1 like the reference: mean loss,
    point precision = P(label > 0.3 | sigmoid(cls) > 0.3), centre recall = annotated centres with a
    decoded foreground centre within 1.4 m (BEV), and the mean BEV offset of matched predictions."""
    dev = torch.device(device)
    was_training = model.training
    model.eval()
    tot_loss, n, pt_hit, pt_fg, gt_hit, gt_cnt, offs = 0.0, 0, 0.0, 0.0, 0, 0, []
    for i in range(len(dataset) if max_scenes is None else min(max_scenes, len(dataset))):
        s = dataset[i]
        pts = torch.from_numpy(s["pts_input"][None].astype(np.float32)).to(dev)
        out = model({"pts_input": pts})
        label = torch.from_numpy(s["rpn_cls_label"][None]).to(dev).float()
        loss, _ = losses.rpn_loss(out["rpn_cls"], out["rpn_reg"], label, torch.from_numpy(s["rpn_reg_label"][None]).to(dev).float(),
                                  net_cfg.loc_scope, net_cfg.loc_bin_size)
        tot_loss += float(loss.item()); n += 1
        score = torch.sigmoid(out["rpn_cls"]).view(-1)
        fg = score > 0.3
        pt_hit += float((fg & (label.view(-1) > 0.3)).sum()); pt_fg += float(fg.sum())
        gt = torch.from_numpy(np.asarray(s["gt_centers"], dtype=np.float32).reshape(-1, 3)).to(dev)
        if int(fg.sum()) and gt.shape[0]:
            xyz = pts[0, :, :3][fg]
            pred = stage1.decode_center_target(torch.zeros_like(xyz), out["rpn_reg"][0][fg], net_cfg.loc_scope,
                                               net_cfg.loc_bin_size) + xyz
            d = torch.cdist(pred[:, [0, 2]], gt[:, [0, 2]])                  # (P, K)
            dmin, arg = d.min(dim=1)
            gt_hit += len(set(arg[dmin < 1.4].tolist())); gt_cnt += gt.shape[0]
            if bool((dmin < 2.0).any()):
                offs.append(dmin[dmin < 2.0])
    if was_training:
        model.train()
    return {"val_loss": tot_loss / max(n, 1), "point_precision": pt_hit / max(pt_fg, 1.0),
            "gt_recall": gt_hit / max(gt_cnt, 1), "mean_offset": float(torch.cat(offs).mean()) if offs else float("nan")}


def train(dataset, total_iters: int, batch_size: int, output_dir: Optional[str] = None, ckpt: Optional[str] = None,
          pretrain_ckpt: Optional[str] = None, ckpt_save_interval: int = 20, seed: int = 0, device: str = "cuda:0",
          net_cfg: stage1.RPNConfig = stage1.DEFAULT_CFG, train_cfg: TrainConfig = TrainConfig(), logger=None,
          distributed: bool = False, prefetch: bool = True, workers: int = 0) -> dict:
    """-> {'model', 'optimizer', 'it', 'history' (loss per iteration), 'checkpoints'}
    prefetch: build / upload the next batch and run its furthest point sampling on a side stream
    while the current step computes (DevicePrefetcher); same losses either way."""
    log = logger or logging.getLogger("ws3d_amd.train_rpn")
    rank, world = 0, 1
    if distributed:
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
    torch.manual_seed(seed)
    dev = torch.device(device)
    model = stage1.Stage1Net(mode="TRAIN", cfg=net_cfg).to(dev)
    optimizer = AdamOneCycle(model.parameters(), total_iters, train_cfg)
    it = 0
    if pretrain_ckpt:
        load_checkpoint(model, None, pretrain_ckpt, log)
        it = int(total_iters * 9 / 30)                   # train_rpn.py:186
    if ckpt:
        it, _ = load_checkpoint(model, optimizer, ckpt, log)
        it = int(it)
    net = nn.parallel.DistributedDataParallel(model, device_ids=[dev.index]) if distributed else model
    ckpt_dir = os.path.join(output_dir, "ckpt") if output_dir else None
    if ckpt_dir and rank == 0:
        os.makedirs(ckpt_dir, exist_ok=True)
    per_epoch = max(len(dataset) // (batch_size * world), 1)
    n_epochs = max(int(total_iters / per_epoch), 1)
    save_every = max(int(n_epochs / min(n_epochs, ckpt_save_interval)), 1) * per_epoch
    source = batches(dataset, batch_size, np.random.RandomState(seed), rank, world, workers=workers)
    stream = DevicePrefetcher(source, dev, net_cfg.npoints) if prefetch and dev.type == "cuda" else source
    history, saved = [], []
    try:
        return _train_loop(net, model, optimizer, stream, it, total_iters, net_cfg, train_cfg, dev, rank, log, ckpt_dir,
                           save_every, history, saved)
    finally:
        if isinstance(stream, DevicePrefetcher):
            stream.close()
        source.close()                      # stops the loader processes


def _train_loop(net, model, optimizer, stream, it, total_iters, net_cfg, train_cfg, dev, rank, log, ckpt_dir, save_every,
                history, saved) -> dict:
    t_mark, it_mark, step_ms = time.perf_counter(), it, []
    while it < total_iters:
        ahead = stream.advance if isinstance(stream, DevicePrefetcher) and it + 1 < total_iters else None
        tb = train_step(net, optimizer, next(stream), it, net_cfg, train_cfg, dev, overlap=ahead)
        it += 1
        history.append(tb["loss"])
        if rank == 0 and (it % 10 == 0 or it == total_iters):
            now = time.perf_counter()            # train_step ends on loss.item(): the device is idle here
            step_ms.append((now - t_mark) * 1e3 / max(it - it_mark, 1))
            t_mark, it_mark = now, it
            log.info("it %5d  loss %.4f  cls %.4f  reg %.4f  fg %d  lr %.2e  %.1f ms/it", it, tb["loss"], tb["rpn_loss_cls"],
                     tb["rpn_loss_reg"], tb["rpn_fg_sum"], tb["lr"], step_ms[-1])
        if ckpt_dir and rank == 0 and (it % save_every == 0 or it == total_iters):
            saved.append(save_checkpoint(checkpoint_state(model, optimizer, it),
                                         os.path.join(ckpt_dir, "checkpoint_iter_%05d" % it)))
    return {"model": model, "optimizer": optimizer, "it": it, "history": history, "checkpoints": saved,
            "step_ms": step_ms}


def main():
    ap = argparse.ArgumentParser(description="Stage-1 RPN trainer (flags of the reference's tools/train_rpn.py)")
    ap.add_argument("--cfg_file", type=str, default="cfgs/", help="accepted for compatibility; weaklyRPN values are built in")
    ap.add_argument("--batch_size", type=int, default=25)
    ap.add_argument("--total_iters", type=int, default=8000)
    ap.add_argument("--ckpt_save_interval", type=int, default=20)
    ap.add_argument("--workers", type=int, default=0, help="loader processes preparing the next batches (0: in the training process)")
    ap.add_argument("--output_dir", type=str, default=None)
    ap.add_argument("--mgpus", action="store_true", default=False, help="one process per GPU under torchrun (DDP over RCCL)")
    ap.add_argument("--ckpt", type=str, default=None)
    ap.add_argument("--pretrain_ckpt", type=str, default=None)
    ap.add_argument("--noise_kind", type=str, default="label_noise")
    ap.add_argument("--weakly_num", type=int, default=500)
    ap.add_argument("--data_root", type=str, default=None, help="KITTI object directory (ImageSets/, training/)")
    ap.add_argument("--no_prefetch", action="store_true", default=False,
                    help="build / upload / sample the next batch inside the step instead of one step ahead")
    ap.add_argument("--synthetic", type=int, default=0, help="train on this many seeded synthetic scenes instead")
    a = ap.parse_args()
    logging.basicConfig(level=logging.INFO, format="%(asctime)s  %(levelname)5s  %(message)s")
    out_dir = a.output_dir or os.path.join("output", "rpn", "weaklyRPN")
    distributed = a.mgpus and int(os.environ.get("WORLD_SIZE", "1")) > 1
    local = 0
    if distributed:
        from . import dist as wdist
        _, _, local = wdist.init()
    if a.synthetic:
        ds = SyntheticCenters(a.synthetic)
    elif a.data_root:
        label_dir = a.noise_kind if os.path.isdir(os.path.join(a.data_root, "training", a.noise_kind)) else None
        ds = KittiCenters(a.data_root, "train", noise_kind=label_dir, weakly_num=a.weakly_num)
    else:
        ap.error("give --data_root or --synthetic N")
    res = train(ds, a.total_iters, a.batch_size, out_dir, a.ckpt, a.pretrain_ckpt, a.ckpt_save_interval,
                device=f"cuda:{local}", distributed=distributed, prefetch=not a.no_prefetch, workers=a.workers)
    if int(os.environ.get("RANK", "0")) == 0:
        logging.getLogger("ws3d_amd.train_rpn").info("done: it %d, last checkpoint %s, median %.1f ms/it", res["it"],
                                                     res["checkpoints"][-1] if res["checkpoints"] else None,
                                                     float(np.median(res["step_ms"])) if res["step_ms"] else float("nan"))


if __name__ == "__main__":
    main()
