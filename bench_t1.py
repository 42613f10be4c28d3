"""(part of bench.py) Workload t1: one Stage-1 RPN TRAINING iteration on `batch`
synthetic KITTI-shaped scenes per GPU -- forward in train() mode, focal + bin regression loss on
Gaussian centre labels, deterministic backward, grad-norm clip, Adam one-cycle step -- exactly
``ws3d_amd.train_rpn.train_step`` fed by ``DevicePrefetcher`` (next batch uploaded and furthest-
point-sampled on a side HIP stream while the current step computes).  world > 1: the model is
wrapped in DistributedDataParallel (gradient all-reduce over RCCL), every rank trains on its own
scenes (weak scaling)."""
from __future__ import annotations

import itertools

import numpy as np
import torch
import torch.nn as nn

from bench_c3 import _fps_model_bytes
from ws3d_amd import stage1
from ws3d_amd.seeded import seeded_state_dict
from ws3d_amd.train_rpn import AdamOneCycle, DevicePrefetcher, SyntheticCenters, TrainConfig, batches, train_step


class T1:
    name = "t1_stage1_rpn_training_iteration"
    metric = "KITTI scenes/sec (16384 pts) Stage-1 RPN training iteration (fwd + loss + deterministic bwd + Adam)"

    def __init__(self, batch, rank, world, kind="lidar", prefetch=True):
        self.B, self.rank, self.world, self.cfg, self.tcfg = batch, rank, world, stage1.DEFAULT_CFG, TrainConfig()
        self.kind = "lidar"          # SyntheticCenters (train_rpn.py) draws its scenes from synth.lidar_cloud whatever --kind says
        self.dev = torch.device("cuda", torch.cuda.current_device())
        torch.manual_seed(1234)
        model = stage1.Stage1Net(mode="TRAIN", cfg=self.cfg)
        model.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 7))
        self.model = model.to(self.dev)
        self.net = (nn.parallel.DistributedDataParallel(self.model, device_ids=[self.dev.index]) if world > 1 else self.model)
        self.opt = AdamOneCycle(self.model.parameters(), 100000, self.tcfg)
        ds = SyntheticCenters(max(2 * batch, 16), config_id=30 + rank)          # scenes generated once, then cached
        host = list(itertools.islice(batches(ds, batch, np.random.RandomState(rank)), 2))
        self.host = host
        source = itertools.cycle(host)
        self.prefetch = prefetch
        self.stream = DevicePrefetcher(source, self.dev, self.cfg.npoints) if prefetch else source
        if prefetch:
            self.stream.timing = []
        self.it, self.ev, self.last = 0, [], None
        for _ in range(6):      # MIOpen's solver search and the allocator's growth happen in the first iterations
            self.step()
        torch.cuda.synchronize()

    def config(self):
        return {"optimizer": "adam_onecycle", "backward": "deterministic (sorted-segment scatter)",
                "norm": "ws3d_bn_relu_train (fused BatchNorm+ReLU)", "pool": "ws3d_pool_nsample",
                "weight_gradients": "ws3d_conv1x1_wgrad (fp32 matrix cores, fixed summation order)",
                "sampling": "one step ahead on a side HIP stream" if self.prefetch else "inside the step",
                "primed_iterations": 6,
                "data_parallel": "DistributedDataParallel (RCCL all-reduce)" if self.world > 1 else "single GPU"}

    def step(self, timed=False):
        if timed:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            e[0].record()
        ahead = self.stream.advance if self.prefetch else None
        self.last = train_step(self.net, self.opt, next(self.stream), self.it, self.cfg, self.tcfg, self.dev, overlap=ahead)
        self.it += 1
        if timed:
            e[1].record()
            self.ev.append(e)

    def scenes(self):
        return self.B

    def kernel_table(self):
        torch.cuda.synchronize()
        step_ms = float(np.mean([a[0].elapsed_time(a[1]) for a in self.ev]))
        rows = [{"name": "training iteration, device time first to last launch", "ms_per_step": step_ms, "launches_per_step": 0,
                 "alg_bytes_per_step": 0, "traffic_key": None, "comment": "library convolutions + our norm / pool / group / scatter kernels"}]
        tm = getattr(self.stream, "timing", None)
        if not tm:      # sampling inside the step: time the chain on its own
            from ws3d_amd import pn2_ops
            xyz = torch.from_numpy(self.host[0]["pts_input"][..., :3].copy()).to(self.dev)
            tm = []
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); pn2_ops.sampling_plan(xyz, self.cfg.npoints); b.record()
                tm.append((a, b))
            torch.cuda.synchronize()
        if tm:
            fps_ms = float(np.mean([a.elapsed_time(b) for a, b in tm[-max(len(self.ev), 1):]]))
            rows.append({"name": "fps_reg_kernel chain 16384->4096->1024->256->64 (side stream, beside the previous step)" if self.prefetch else
                         "fps_reg_kernel chain 16384->4096->1024->256->64",
                         "ms_per_step": fps_ms, "launches_per_step": 4, "alg_bytes_per_step": _fps_model_bytes(self.cfg) * self.B,
                         "traffic_key": None, "bound": "valu", "lane_instr_per_step": self._fps_lane_instr() * self.B,
                         "comment": "A_model (12 B/point re-read per sampled point); latency-bound: one workgroup per scene"})
        return rows

    def _fps_lane_instr(self):
        from bench import fps_lane_instr
        n, tot = self.cfg.num_points, 0
        for m in self.cfg.npoints:
            tot += fps_lane_instr(n, m)
            n = m
        return tot

    def path_gbps(self, scenes_per_s_per_gpu):
        return {"loss_last_step": None if self.last is None else self.last["loss"]}

    def cpu_baseline(self, min_seconds=6.0):
        return {"value": None, "unit": "scenes/s", "cores": 0, "kind": "port",
                "sample": "none: the oracle restates the operators (timed under --workload c2/c3), not the network's "
                          "library convolutions, so there is no CPU leg for a training iteration"}
