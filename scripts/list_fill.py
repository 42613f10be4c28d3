"""Distinct neighbours per ball-query list at every SA scale of the c3 network, on the CPU oracle (no GPU needed):
    python scripts/list_fill.py [--kind hdl64|lidar|uniform] [--scenes 4]
The SharedMLPs may run over the distinct (centre, sample) pairs only (fastpath.COMPACT_PAIRS); this is the statistic that decides."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from ws3d_amd import synth  # noqa: E402
from ws3d_amd.stage1 import DEFAULT_CFG as cfg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="hdl64")
    ap.add_argument("--scenes", type=int, default=4)
    a = ap.parse_args()
    oracle.set_threads(min(oracle.max_threads(), len(os.sched_getaffinity(0))))
    pc = synth.make_batch(a.kind, a.scenes, 16384, 3)
    lv = np.ascontiguousarray(pc[:, :, :3])
    for k, m in enumerate(cfg.npoints):
        idx = oracle.furthest_point_sample(lv, m)
        new = np.stack([lv[b][idx[b]] for b in range(a.scenes)])
        for r, ns in zip(cfg.radius[k], cfg.nsample[k]):
            nbr = oracle.ball_query(r, ns, lv, new)
            # a list is [hits in ascending index order, padded with the first hit]: distinct = position of the first repeat of entry 0
            pad = (nbr == nbr[:, :, :1])
            pad[:, :, 0] = False
            distinct = np.where(pad.any(2), pad.argmax(2), ns)
            print("SA%d n=%d m=%d r=%.1f ns=%d: mean distinct neighbours per centre %.1f of %d (%.0f %%), full lists %.0f %%"
                  % (k + 1, lv.shape[1], m, r, ns, distinct.mean(), ns, 100 * distinct.mean() / ns, 100 * (distinct == ns).mean()))
        lv = new


if __name__ == "__main__":
    main()
