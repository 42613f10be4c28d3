"""Are consecutive forward passes of a COLD process bit-equal?  (a TRAIN-mode equality test failed once in a
full run and passed alone.)  Fresh process: build the TRAIN-mode Stage-1 net of tests/test_gpu_parity.py
test_sampling_plan_equals_sampling_inside_the_modules, run the forward pass K times on the same input with the same dropout seed,
record every leaf module's output, and report the first module whose output differs between pass 0 and pass j.

    python scripts/cold_forward_bits.py [passes] [mode]      mode: train (default) | eval | eval_nofast
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ws3d_amd import stage1, synth  # noqa: E402


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    mode = sys.argv[2] if len(sys.argv) > 2 else "train"
    cfg = stage1.RPNConfig(num_points=4096, npoints=(1024, 256, 64, 16))
    torch.manual_seed(0)
    net = stage1.Stage1Net(mode="TRAIN" if mode == "train" else "TEST", cfg=cfg).cuda()
    net = net.train() if mode == "train" else net.eval()
    if mode == "eval_nofast":
        stage1.CHANNELS_LAST_FASTPATH = False
    pts = torch.from_numpy(np.stack([synth.velodyne_scan(4096, seed=s) for s in (1, 2)])).cuda()
    rec = []
    names = {m: n for n, m in net.named_modules()}

    def hook(m, inp, out):
        if isinstance(out, torch.Tensor):
            rec[-1].append((names[m], type(m).__name__, out.detach().clone()))
    for m in net.modules():
        if not list(m.children()):
            m.register_forward_hook(hook)
    outs = []
    for j in range(passes):
        rec.append([])
        torch.manual_seed(1)
        o = net({"pts_input": pts})
        torch.cuda.synchronize()
        outs.append({k: v.detach().clone() for k, v in o.items() if isinstance(v, torch.Tensor)})
    bad = 0
    for j in range(1, passes):
        first = None
        for (n0, t0, a), (n1, t1, b) in zip(rec[0], rec[j]):
            if a.shape != b.shape or not torch.equal(a, b):
                first = (n0, t0, tuple(a.shape), float((a - b).abs().max()) if a.shape == b.shape else None)
                break
        final = all(torch.equal(outs[0][k], outs[j][k]) for k in outs[0])
        if first or not final:
            bad += 1
        print("pass 0 vs %d: %s%s" % (j, "EQUAL" if not first and final else "DIFFER", "" if not first else "  first differing leaf: %s" % (first,)))
    print("RESULT mode=%s passes=%d differing=%d" % (mode, passes, bad))


if __name__ == "__main__":
    main()
