#!/usr/bin/env python
"""Golden vectors for furthest point sampling from the ONE reference-held artefact that restates it on the CPU:
``lib/utils/greedFurthestPoint.getGreedyPerm`` (greedFurthestPoint.py:11-37), imported from /root/reference where it lies and run
here (build container only; the reference tree never travels): ``python -B tests/golden/make_golden_greedyperm.py``.

getGreedyPerm works on a float64 EUCLIDEAN distance matrix and takes numpy's argmax (first maximum); the CUDA kernel
(sampling_gpu.cu:93-209) works on float32 SQUARED distances with its own tie order.  The two orders agree wherever the running
maximum is separated from the runner-up by more than the float32 rounding of the distances, so next to every permutation the
fixture stores the relative gap between the best and the second-best candidate of each step (`margin`, computed here in float64
from the same distance matrix): tests compare the leading steps up to the first gap below 1e-6 and report the matched prefix.

Round 4: three FULL-SIZE cases (n = 16384, the size at which the headline's level-1 kernel `fps_rounds_kernel` engages,
csrc/fps.hip dispatch) on `hdl64`, `lidar` and the tie-free `uniform`; the reference's loop runs over the 16384 x 16384 float64
matrix (2.1 GB) in seconds, and the fixture keeps the first 4096 steps (= SA1's npoint) of each -> fps_greedyperm_16k.npz.

Fixture = data only: generator name / n / seed per case, the permutation, the insertion radii, the margins."""
from __future__ import annotations

import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_FILE = "/root/reference/lib/utils/greedFurthestPoint.py"

import numpy as np  # noqa: E402

from ws3d_amd import synth  # noqa: E402

CASES = [("uniform", 256, 11), ("uniform", 1024, 12), ("uniform", 2048, 13), ("lidar", 512, 14), ("lidar", 2048, 15),
         ("hdl64", 256, 16), ("hdl64", 1024, 17), ("hdl64", 2048, 18)]
FULL_CASES = [("hdl64", 16384, 21), ("lidar", 16384, 22), ("uniform", 16384, 23), ("hdl64", 16384, 24)]
FULL_KEEP = 4096          # SA1's npoint: the prefix the fixture keeps of a full-size permutation


def load_reference_module():
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:                       # the module imports pyplot at the top for its demo function only
        mpl = types.ModuleType("matplotlib")
        mpl.pyplot = types.ModuleType("matplotlib.pyplot")
        sys.modules["matplotlib"], sys.modules["matplotlib.pyplot"] = mpl, mpl.pyplot
    spec = importlib.util.spec_from_file_location("ref_greedFurthestPoint", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cloud(kind: str, n: int, seed: int) -> np.ndarray:
    """the first n points of a 16384-point scene of the generator (hdl64 needs the full sub-sampling; the prefix of a shuffled scene
    is a uniform sub-sample of it)"""
    return np.ascontiguousarray(synth.cloud(kind, 16384, seed)[:n, :3])


def run(ref, cases, keep, fname):
    from sklearn.metrics.pairwise import pairwise_distances          # what the reference's own caller feeds it (:84-87)
    out = {}
    meta = []
    for kind, n, seed in cases:
        xyz = cloud(kind, n, seed)
        D = pairwise_distances(xyz.astype(np.float64), metric="euclidean")
        perm, lambdas = ref.getGreedyPerm(D)
        # the gap between the winner and the runner-up of every step, from the same matrix (float64)
        k = min(n, keep or n)
        margin = np.zeros(k)
        ds = D[0, :].copy()
        for i in range(1, k):
            top2 = np.partition(ds, -2)[-2:]
            margin[i] = (top2[1] - top2[0]) / top2[1] if top2[1] > 0 else 0.0
            assert int(np.argmax(ds)) == int(perm[i])
            ds = np.minimum(ds, D[perm[i], :])
        key = "%s_%d_%d" % (kind, n, seed)
        out[key + "_perm"] = perm[:k].astype(np.int32)
        out[key + "_lambdas"] = lambdas[:k]
        out[key + "_margin"] = margin.astype(np.float32)
        meta.append(key)
        print(key, "first step with a gap below 1e-6:", int(np.argmax(margin[1:] < 1e-6)) + 1 if (margin[1:] < 1e-6).any() else None, flush=True)
        del D
    np.savez_compressed(os.path.join(HERE, fname), cases=np.array(meta), **out)


def main():
    ref = load_reference_module()
    if "--full-only" not in sys.argv:
        run(ref, CASES, None, "fps_greedyperm.npz")
    run(ref, FULL_CASES, FULL_KEEP, "fps_greedyperm_16k.npz")


if __name__ == "__main__":
    main()
