#!/usr/bin/env python
"""Fixture for BASELINE.json configs[0] ("C1"): ONE synthetic 16384 x 4 cloud through the REFERENCE's
own set-abstraction layer -- ``PointnetSAModuleMSG(npoint=4096, radii=[0.1, 0.5], nsamples=[16, 32],
mlps=[[1,16,16,32],[1,32,32,64]], use_xyz=True, bn=True)`` (pointnet2_modules.py:58-92, the network's
SA1, weaklyRPN.yaml:44-50) imported unmodified and run on CPU, leaf ops backed by the CPU oracle
(same recipe and shims as make_golden.py; build container only).

    python -B tests/golden/make_golden_c1.py   ->  tests/golden/c1_sa_layer.npz + c1_sa_layer.json

Data only: the seed/params, the 4096 FPS indices, sha256 of the two ball-query index tensors and of
new_xyz, and 256 sampled values of the (1, 96, 4096) output features."""
from __future__ import annotations

import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import make_golden as mg  # noqa: E402  (shims + helpers; also puts the repo root on sys.path)
from ws3d_amd import synth  # noqa: E402
from ws3d_amd.seeded import seeded_state_dict  # noqa: E402

CASE = {"config_id": 1, "n": 16384, "npoint": 4096, "radii": [0.1, 0.5], "nsamples": [16, 32],
        "mlps": [[1, 16, 16, 32], [1, 32, 32, 64]], "seed": 11}


def main():
    mg.install_reference_shims()
    from pointnet2_lib.pointnet2 import pointnet2_modules as ref_mod
    from pointnet2_lib.pointnet2 import pointnet2_utils as ref_utils
    pc = synth.make_batch("lidar", 1, CASE["n"], CASE["config_id"])
    xyz = torch.from_numpy(pc[:, :, :3].copy())
    feats = torch.from_numpy(np.ascontiguousarray(pc[:, :, 3:].transpose(0, 2, 1)))
    sa = ref_mod.PointnetSAModuleMSG(npoint=CASE["npoint"], radii=list(CASE["radii"]), nsamples=list(CASE["nsamples"]),
                                     mlps=[list(m) for m in CASE["mlps"]], use_xyz=True, bn=True).eval()
    sa.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in sa.state_dict().items()}, CASE["seed"]))
    taps = {"fps": [], "bq": []}
    orig_fps, orig_bq = ref_utils.furthest_point_sample, ref_utils.ball_query

    def fps(xyz_, npoint):
        r = orig_fps(xyz_, npoint)
        taps["fps"].append(mg._np(r).astype(np.int32))
        return r

    def bq(radius, nsample, xyz_, new_xyz_):
        r = orig_bq(radius, nsample, xyz_, new_xyz_)
        taps["bq"].append(mg.sha(mg._np(r).astype(np.int32)))
        return r

    ref_utils.furthest_point_sample, ref_utils.ball_query = fps, bq
    try:
        with torch.no_grad():
            new_xyz, new_feat = sa(xyz, feats)
    finally:
        ref_utils.furthest_point_sample, ref_utils.ball_query = orig_fps, orig_bq
    assert len(taps["fps"]) == 1 and len(taps["bq"]) == 2
    out = mg._np(new_feat)
    pos, val = mg.sample(out, 256, seed=1)
    np.savez_compressed(os.path.join(HERE, "c1_sa_layer.npz"), fps_idx=taps["fps"][0], feat_pos=pos, feat_val=val)
    meta = dict(CASE, generator="tests/golden/make_golden_c1.py", ball_query_sha256=taps["bq"],
                new_xyz_sha256=mg.sha(mg._np(new_xyz)), features_shape=list(out.shape), features_abs_mean=float(np.abs(out).mean()),
                keys={k: list(v.shape) for k, v in sa.state_dict().items()})
    json.dump(meta, open(os.path.join(HERE, "c1_sa_layer.json"), "w"), indent=1)
    for f in ("c1_sa_layer.npz", "c1_sa_layer.json"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
