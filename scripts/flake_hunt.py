"""Hunt for the once-in-31-runs bit difference of tests/test_gpu_parity.py::test_sampling_plan_equals_sampling_inside_the_modules
(profiles/r04_sampling_plan_test_flake.txt).

    python scripts/flake_hunt.py train   [iterations] [churn|plain] [busy]     the test's own body: two TRAIN-mode passes (sampling inside
                                                                               the modules / a sampling plan), fresh net per iteration
    python scripts/flake_hunt.py eval    [iterations] [side|one|sync] [busy]   the EVAL fast path (the one with side streams): pass 0 of an
                                                                               iteration is the reference, passes 1.. are compared with it

train: the forward pass is ONE stream (ws3d_amd/stage1.py: the fast path with its side streams is eval-only), so the three stream modes
the verdict names cannot differ there; what can differ between two passes of one process is the allocator's state.  `churn` therefore
allocates and frees a random set of blocks (filled with NaN) between the passes and between the iterations -- the state "a full suite
after seven minutes of fuzzers" leaves behind -- and `busy` keeps a second stream hammering the chip.  Every module output is cloned at
its exit; on a mismatch the report says (a) the first module whose output differs between the passes, (b) whether a pass's final tensor
still equals the clone taken when its head returned (a buffer overwritten AFTER it was produced = somebody's out-of-bounds / stale write),
(c) where and by how much (ULPs) the tensors differ.

eval: `side` = fastpath.GEOMETRY_AHEAD (three side streams), `one` = one stream, `sync` = side streams with a device synchronisation
after every SA / FP module (forward hooks).  Prints one RESULT line per run: mode, iterations, mismatches."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ws3d_amd import fastpath, pn2_ops, stage1, synth  # noqa: E402
from ws3d_amd.seeded import seeded_state_dict  # noqa: E402

RNG = np.random.default_rng(12345)


def churn(n_blocks=40):
    """random allocations, filled with NaN, freed in random order: the next pass gets other blocks than the last one did"""
    sizes = (RNG.integers(1, 1 << 20, n_blocks) * RNG.choice([1, 4, 64, 257], n_blocks)).tolist()
    blocks = [torch.full((int(s),), float("nan"), device="cuda") for s in sizes]
    keep = RNG.random(n_blocks) < 0.3
    held = [b for b, k in zip(blocks, keep) if k]       # a third stays allocated across the next pass: fragments the pools
    del blocks
    return held


class Busy:
    """a second stream that keeps the chip busy (GEMMs + elementwise) while the passes run"""

    def __init__(self, on):
        self.on = on
        if on:
            self.st = torch.cuda.Stream()
            self.a = torch.randn(2048, 2048, device="cuda")
            self.b = torch.randn(2048, 2048, device="cuda")

    def kick(self, n=6):
        if self.on:
            with torch.cuda.stream(self.st):
                for _ in range(n):
                    self.b = torch.tanh(self.a @ self.b) * 0.5


def ulps(a, b):
    ia, ib = a.contiguous().view(torch.int32).long(), b.contiguous().view(torch.int32).long()
    return (ia - ib).abs()


def describe(x, y, what):
    bad = (x != y).flatten().nonzero().flatten()
    u = ulps(x.flatten()[bad], y.flatten()[bad])
    return "%s: %d of %d elements differ, flat positions %s .. %s, max |diff| %.3g, ULPs min/max %d/%d" % (
        what, bad.numel(), x.numel(), bad[:6].tolist(), bad[-3:].tolist(), float((x - y).abs().max()), int(u.min()), int(u.max()))


def hunt_train(iters, do_churn, busy):
    cfg = stage1.RPNConfig(num_points=4096, npoints=(1024, 256, 64, 16))
    pts = torch.from_numpy(np.stack([synth.velodyne_scan(4096, seed=s) for s in (1, 2)])).cuda()
    plan = pn2_ops.sampling_plan(pts[..., 0:3].contiguous(), cfg.npoints)
    bad = 0
    for it in range(iters):
        held = churn() if do_churn else None
        torch.manual_seed(0)
        net = stage1.Stage1Net(mode="TRAIN", cfg=cfg).cuda().train()
        rec, names = [], {m: n for n, m in net.named_modules()}
        hooks = [m.register_forward_hook(lambda m, i, o: rec[-1].append((names[m], o.detach().clone())) if isinstance(o, torch.Tensor) else None)
                 for m in net.modules()]
        rec.append([])
        busy.kick()
        torch.manual_seed(1)
        a = net({"pts_input": pts})
        held2 = churn(20) if do_churn else None
        rec.append([])
        busy.kick()
        torch.manual_seed(1)
        b = net({"pts_input": pts, "sampling_plan": plan})
        torch.cuda.synchronize()
        for h in hooks:
            h.remove()
        first = next(((n, describe(x, y, n)) for (n, x), (_, y) in zip(rec[0], rec[1]) if not torch.equal(x, y)), None)
        same = torch.equal(a["rpn_cls"], b["rpn_cls"]) and torch.equal(a["rpn_reg"], b["rpn_reg"])
        if first is not None or not same:
            bad += 1
            print("MISMATCH iteration %d: first differing module: %s" % (it, first), flush=True)
            for key, head in (("rpn_cls", "rpn_cls_layer"), ("rpn_reg", "rpn_reg_layer")):
                for p, t in ((0, a[key]), (1, b[key])):
                    then = [o for n_, o in rec[p] if n_ == "rpn." + head][-1].transpose(1, 2)
                    if not torch.equal(t, then):
                        print("   pass %d %s CHANGED after its head returned: %s" % (p, key, describe(t, then, "now vs then")), flush=True)
                if not torch.equal(a[key], b[key]):
                    print("   " + describe(a[key], b[key], key + " pass 0 vs pass 1"), flush=True)
        del held, held2, net, rec, a, b
    return bad


def hunt_eval(iters, stream_mode, busy, passes=4):
    cfg = stage1.RPNConfig(rpn_pre_nms_top_n=2000, rpn_post_nms_top_n=50)
    model = stage1.Stage1Net(mode="TEST", cfg=cfg).eval()
    model.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 7))
    model = model.cuda()
    pts = torch.from_numpy(np.stack([synth.cloud("hdl64", 16384, 5000 + j) for j in range(2)])).cuda()
    hooks = []
    if stream_mode == "sync":
        bb = model.rpn.backbone_net
        hooks = [m.register_forward_hook(lambda *_: torch.cuda.synchronize()) for m in list(bb.SA_modules) + list(bb.FP_modules)]

    @torch.no_grad()
    def body():
        out = model.rpn_forward({"pts_input": pts, "defer_reg_join": True})
        boxes, scores, count = stage1.proposals_from_rpn(out, cfg)
        return {"rpn_cls": out["rpn_cls"], "rpn_reg": out["rpn_reg"], "features": out["backbone_features_nlc"], "boxes": boxes,
                "scores": scores, "count": count}

    bad = 0
    with fastpath.geometry_ahead(stream_mode != "one"):
        ref = {k: v.clone() for k, v in body().items()}
        torch.cuda.synchronize()
        for it in range(iters):
            held = churn(20) if it % 2 else None
            for p in range(passes):
                busy.kick(3)
                got = body()
                # compared WITHOUT a synchronisation in between: torch.equal runs on the caller's stream behind the pass
                for k, v in got.items():
                    if not torch.equal(v, ref[k]):
                        bad += 1
                        print("MISMATCH iteration %d pass %d %s" % (it, p, describe(v.float(), ref[k].float(), k)), flush=True)
            del held
    for h in hooks:
        h.remove()
    return bad


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "train"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    mode = sys.argv[3] if len(sys.argv) > 3 else ("churn" if what == "train" else "side")
    busy = Busy(len(sys.argv) > 4 and sys.argv[4] == "busy")
    t0 = time.time()
    bad = hunt_train(iters, mode == "churn", busy) if what == "train" else hunt_eval(iters, mode, busy)
    torch.cuda.synchronize()
    print("RESULT %s mode=%s busy=%s iterations=%d mismatches=%d wall=%.1fs" % (what, mode, busy.on, iters, bad, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
