"""Does any kernel of the forward pass read memory it never wrote?  torch.empty() hands out recycled blocks of the caching allocator
with whatever the previous owner left there; a kernel that relies on such a block being zero (an atomic-max pool into an output that
"must be zero on entry", a padded tail) gives results that depend on the process's history -- the signature of the one-in-a-full-run
failure of test_sampling_plan_equals_sampling_inside_the_modules (round 3).  Here the allocator's free blocks are POISONED before
every pass (a large tensor filled with a pattern, then freed), and the passes are compared bit for bit across patterns.

    python scripts/poison_forward.py [train|eval|eval_nofast]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ws3d_amd import stage1, synth  # noqa: E402


def poison(value, gib=6):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    blocks = [torch.full((256 << 20,), value, dtype=torch.float32, device="cuda") for _ in range(gib)]     # 1 GiB each
    small = [torch.full((n,), value, dtype=torch.float32, device="cuda") for n in (1 << 8, 1 << 12, 1 << 16, 1 << 18) for _ in range(64)]
    torch.cuda.synchronize()
    del blocks, small                                  # back to the caching allocator, contents intact


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "train"
    n = 4096 if mode == "train" else 16384
    cfg = stage1.RPNConfig(num_points=4096, npoints=(1024, 256, 64, 16)) if mode == "train" else stage1.DEFAULT_CFG
    torch.manual_seed(0)
    net = stage1.Stage1Net(mode="TRAIN" if mode == "train" else "TEST", cfg=cfg).cuda()
    net = net.train() if mode == "train" else net.eval()
    if mode == "eval_nofast":
        stage1.CHANNELS_LAST_FASTPATH = False
    if mode == "train":
        pts = torch.from_numpy(np.stack([synth.velodyne_scan(4096, seed=s) for s in (1, 2)])).cuda()
    else:
        pts = torch.from_numpy(synth.make_batch("hdl64", 4, n, 3)).cuda()
    outs = {}
    for name, value in (("zero", 0.0), ("nan", float("nan")), ("big", 3.0e38), ("neg", -7.5), ("zero2", 0.0)):
        poison(value)
        torch.manual_seed(1)
        o = net({"pts_input": pts})
        if mode != "train":
            boxes, scores, count = stage1.proposals_from_rpn(o, cfg)
            o = dict(o, boxes=boxes, scores=scores, count=count)
        torch.cuda.synchronize()
        outs[name] = {k: v.detach().clone() for k, v in o.items() if isinstance(v, torch.Tensor)}
    bad = 0
    for name in outs:
        for k in outs["zero"]:
            a, b = outs["zero"][k], outs[name][k]
            if not torch.equal(a, b) and not (torch.isnan(a) & torch.isnan(b)).all():
                d = (a.float() - b.float()).abs()
                print("DIFF pattern=%s tensor=%s: %d of %d elements differ (max %g, nan %d)" % (name, k, int((a != b).sum()), a.numel(),
                                                                                             float(d[~torch.isnan(d)].max()) if (~torch.isnan(d)).any() else -1, int(torch.isnan(b).sum())))
                bad += 1
    print("RESULT mode=%s differing=%d" % (mode, bad))


if __name__ == "__main__":
    main()
