"""The reference-layout leg of bench.py alone (stage1.CHANNELS_LAST_FASTPATH = False: the reference's module composition on (B, C, N)
tensors through the API-named operators, eager, one batch in flight), for `rocprofv3 --kernel-trace --stats`: where do the 5.8 ms of
`value_reference_layout` go?  python scripts/r06/reflayout_prof.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ws3d_amd import stage1           # noqa: E402
from bench_c3 import C3               # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
stage1.CHANNELS_LAST_FASTPATH = False
w = C3(8, 0, 1, "hdl64", depth=1)
for _ in range(3):
    w.step(eager=True)
torch.cuda.synchronize()
ts = []
for _ in range(steps):
    t0 = time.perf_counter()
    w.step(eager=True)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("reference layout, eager, one batch in flight: median %.3f ms per batch of 8 (%d batches) = %.1f scenes/s" %
      (float(np.median(ts)) * 1e3, steps, 8 / float(np.median(ts))), flush=True)
