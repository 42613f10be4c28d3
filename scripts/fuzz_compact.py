"""Whole Stage-1 forward with the SharedMLPs over the distinct (centre, sample) pairs against the dense kernels, on random clouds of
different kinds and densities (sparse lidar wedges, uniform boxes of varying size -- from one neighbour per list to full lists --
quantised grids full of exact duplicates): the two forms must agree to fp32 round-off (SA4's dense form uses a library GEMM)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ws3d_amd import fastpath, stage1, synth
from ws3d_amd.seeded import seeded_state_dict
ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=60); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
cfg = stage1.RPNConfig(num_points=16384, rpn_pre_nms_top_n=1000, rpn_post_nms_top_n=20)
model = stage1.Stage1Net(mode="TEST", cfg=cfg)
model.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 11))
model = model.cuda().eval()
t0, rounds, worst, fills = time.time(), 0, 0.0, []
while time.time() - t0 < a.seconds:
    B = int(rng.choice([1, 2, 8]))
    kind = rng.choice(["lidar", "box", "grid"])
    if kind == "lidar":
        pts = np.stack([synth.velodyne_scan(16384, seed=int(rng.integers(1, 10 ** 6))) for _ in range(B)])
    elif kind == "box":      # uniform box: the edge length sets the density (0.5 m: every list full; 60 m: one neighbour)
        edge = float(rng.choice([0.5, 2.0, 6.0, 20.0, 60.0]))
        pts = np.concatenate([rng.uniform(-edge / 2, edge / 2, (B, 16384, 3)), rng.uniform(0, 1, (B, 16384, 1))], axis=2).astype(np.float32)
    else:                    # quantised coordinates: many exact duplicates
        pts = np.concatenate([rng.integers(0, 40, (B, 16384, 3)) * float(rng.choice([0.25, 1.0])), rng.uniform(0, 1, (B, 16384, 1))], axis=2).astype(np.float32)
    x = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
    outs = []
    for thr in (2.0, -1.0):      # always compact / always dense
        fastpath.COMPACT_MAX_FILL = thr
        with torch.no_grad():
            o = model.rpn_forward({"pts_input": x})
        outs.append((o["rpn_cls"].clone(), o["rpn_reg"].clone()))
    for u, v in zip(*outs):
        e = float((u - v).abs().max()) / max(float(v.abs().max()), 1.0)
        worst = max(worst, e)
        assert e <= 5e-5, (kind, B, e)
    # the serial order a primed Stage1Pipeline captures (one stream, no dense twins): its merged launches -- one binning launch, the FP
    # modules' 3-NN as jobs, both scales per compact SharedMLP kernel, the prologue -- against one launch each: the same bits
    fastpath.COMPACT_MAX_FILL = 0.55
    every = {(l_, s_) for l_ in range(4) for s_ in range(2)}
    sw = ("MERGED_BINNING", "MERGED_THREE_NN", "PAIRED_SCALES", "FUSED_PROLOGUE", "DUAL_SCALE_SEARCH", "NESTED_CHAIN")
    res = []
    for on in (True, False):
        saved = {n_: getattr(fastpath, n_) for n_ in sw}
        for n_ in sw:
            setattr(fastpath, n_, on)
        try:
            with torch.no_grad(), fastpath.geometry_ahead(False), fastpath.compact_only_scales(every):
                o = model.rpn_forward({"pts_input": x})
        finally:
            for n_, v_ in saved.items():
                setattr(fastpath, n_, v_)
        res.append((o["rpn_cls"].clone(), o["rpn_reg"].clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), ("merged launches", kind, B)
    rounds += 1
fastpath.COMPACT_MAX_FILL = 0.55
print(f"fuzz_compact: {rounds} forward passes (lidar wedges, uniform boxes 0.5-60 m, duplicate-laden grids), compact vs dense SharedMLPs: worst relative difference {worst:.1e}; merged launches of the serial order bit-identical to one launch each (seed {a.seed})")
