"""Random-shape check of the two matrix-core kernels against float64: ws3d_gemm_pool and ws3d_conv1x1_wgrad."""
import argparse, time
import numpy as np, torch
from ws3d_amd import compat as c
ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=60); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0, rounds, worst_g, worst_w = time.time(), 0, 0.0, 0.0
while time.time() - t0 < a.seconds:
    ns = int(rng.choice([16, 32])); rows = 64 * int(rng.integers(1, 40)); k = 4 * int(rng.integers(1, 150)); o = 64 * int(rng.integers(1, 9))
    x = torch.randn(rows, k, device="cuda"); wt = torch.randn(k, o, device="cuda") * 0.1
    bias = torch.randn(o, device="cuda") if rng.random() < 0.8 else None
    relu = bool(rng.random() < 0.7)
    y = x.double() @ wt.double()
    if bias is not None: y = y + bias.double()
    if relu: y = torch.relu(y)
    ref = y.view(rows // ns, ns, o).amax(1)
    out = torch.empty(rows // ns, o, device="cuda")
    assert c.gemm_pool(x, wt, bias, relu, ns, out, 0)
    e = float((out.double() - ref).abs().max() / (ref.abs().max() + 1e-9)); worst_g = max(worst_g, e)
    assert e < 5e-6, ("gemm_pool", rows, ns, k, o, e)
    B = int(rng.integers(1, 5)); C = int(rng.integers(1, 600)); O = int(rng.integers(1, 600)); L = int(rng.choice([1, 3, 16, 100, 1000, 4096, 5001, 20000]))
    if B * (C + O) * L > 4e7: L = max(1, int(4e7 // (B * (C + O))))
    xx = torch.randn(B, C, L, device="cuda"); gy = torch.randn(B, O, L, device="cuda")
    refw = torch.einsum("bol,bcl->oc", gy.double(), xx.double())
    got = c.conv1x1_wgrad(gy, xx)
    e = float((got.double() - refw).abs().max() / (refw.abs().max() + 1e-9)); worst_w = max(worst_w, e)
    assert e < 5e-6, ("conv1x1_wgrad", B, C, O, L, e)
    assert torch.equal(got, c.conv1x1_wgrad(gy, xx))
    rounds += 1
print(f"fuzz_mfma: {rounds} rounds, worst relative error gemm_pool {worst_g:.1e}, conv1x1_wgrad {worst_w:.1e} (seed {a.seed})")
