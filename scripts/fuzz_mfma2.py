"""Random-shape check of the matrix-core kernels against float64: ws3d_gemm_pool, ws3d_mlp2_rows (ticket counter), ws3d_interp_gemm
and the per-point first layers (ws3d_pgather_gemm2 / ws3d_pgather_rows / ws3d_qinterp_rows, every fourth round).  The output tile is
chosen by the library from the shape (the per-tile environment switches of round 2 are gone): every third round draws
a LARGE shape (>= 512 tiles of 128 x 128: rows >= 16384 at o = 512) so that the big-tile dispatch of gemm_pool / interp_gemm is hit
as well as the small-tile one.    python scripts/fuzz_mfma2.py --seconds 60 [--seed S]"""
import argparse, os, subprocess, sys, time
ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=60); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
a.child = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ws3d_amd import compat as c, synth
rng = np.random.default_rng(a.seed)
pc_all = torch.from_numpy(synth.make_batch("lidar", 8, 16384, 5)[:, :, :3].copy()).cuda()
t0, rounds, worst = time.time(), 0, {"gemm_pool": 0.0, "mlp2_rows": 0.0, "interp_gemm": 0.0}
while time.time() - t0 < a.seconds:
    # gemm_pool: rows a multiple of 64 (sometimes of 128 / 256), so both the forced tile and its fallback are hit
    big = rounds % 3 == 2
    ns = int(rng.choice([16, 32])); rows = 64 * int(rng.integers(256, 400) if big else rng.integers(1, 80)); k = 4 * int(rng.integers(1, 150)); o = 512 if big else 64 * int(rng.integers(1, 9))
    x = torch.randn(rows, k, device="cuda"); wt = torch.randn(k, o, device="cuda") * 0.1
    bias = torch.randn(o, device="cuda") if rng.random() < 0.8 else None
    relu = bool(rng.random() < 0.7)
    y = x.double() @ wt.double()
    if bias is not None: y = y + bias.double()
    if relu: y = torch.relu(y)
    ref = y.view(rows // ns, ns, o).amax(1)
    out = torch.empty(rows // ns, o, device="cuda")
    assert c.gemm_pool(x, wt, bias, relu, ns, out, 0)
    e = float((out.double() - ref).abs().max() / (ref.abs().max() + 1e-9)); worst["gemm_pool"] = max(worst["gemm_pool"], e)
    assert e < 5e-6, ("gemm_pool", a.child, rows, ns, k, o, e)
    # mlp2_rows
    rows = 32 * int(rng.integers(1, 700)); o2 = int(rng.integers(1, 65))
    x = torch.randn(rows, 128, device="cuda"); w1 = torch.randn(128, 128, device="cuda") / 11; w2 = torch.randn(128, o2, device="cuda") / 11
    b1 = torch.randn(128, device="cuda") if rng.random() < 0.8 else None; b2 = torch.randn(o2, device="cuda") if rng.random() < 0.8 else None
    r1, r2 = bool(rng.random() < 0.8), bool(rng.random() < 0.3)
    h = x.double() @ w1.double()
    if b1 is not None: h = h + b1.double()
    if r1: h = torch.relu(h)
    ref = h @ w2.double()
    if b2 is not None: ref = ref + b2.double()
    scale = float(ref.abs().max()) + 1e-9          # before the ReLU: a negative bias can leave almost nothing after it
    if r2: ref = torch.relu(ref)
    got = c.mlp2_rows(x, w1, b1, r1, w2, b2, r2)
    e = float((got.double() - ref).abs().max()) / scale; worst["mlp2_rows"] = max(worst["mlp2_rows"], e)
    assert e < 1e-5, ("mlp2_rows", rows, o2, e)
    # interp_gemm
    B = int(rng.choice([1, 2, 8])); N = 64 * int(rng.integers(1, 40)); M = max(3, N // int(rng.choice([2, 4, 8])))
    if big: B, N, M = 8, 64 * int(rng.integers(160, 256)), 2048
    C2 = 4 * int(rng.integers(1, 100)); C1 = int(rng.choice([0, 1, 5, 32, 96])); O = 512 if big else 64 * int(rng.integers(1, 7))
    unknown = pc_all[:B, :N].contiguous(); known = unknown[:, :M].contiguous()
    kf = torch.randn(B, M, C2, device="cuda"); uf = torch.randn(B, N, C1, device="cuda") if C1 else None
    idx, weight = c.three_nn_with_weights(unknown, known, None)
    wt = torch.randn(C2 + C1, O, device="cuda") / (C2 + C1) ** 0.5; bias = torch.randn(O, device="cuda")
    got = c.interp_gemm(kf, uf, idx, weight, wt, bias, True)
    interp = torch.empty((B, N, C2), device="cuda"); c.three_interpolate_nlc(kf, idx, weight, interp)
    xx = interp if uf is None else torch.cat((interp, uf), dim=2)
    want = torch.relu(xx.view(-1, C2 + C1).double() @ wt.double() + bias.double())
    e = float((got.double() - want).abs().max() / (want.abs().max() + 1e-9)); worst["interp_gemm"] = max(worst["interp_gemm"], e)
    assert e < 1e-5, ("interp_gemm", a.child, B, N, M, C2, C1, O, e)
    # per-point layer 1 (SA) and per-known-point first FP layer against the float64 products over the grouped / interpolated rows
    if rounds % 4 == 0:
        B = int(rng.choice([1, 2])); N = 64 * int(rng.integers(2, 40)); M = 64 * int(rng.integers(1, max(2, N // 128))); ns = int(rng.choice([16, 32]))
        C = 4 * int(rng.integers(1, 80)); O1 = int(rng.choice([64, 128, 256])); O2 = 4 * int(rng.integers(1, 80))
        xyz = pc_all[:B, :N].contiguous(); new_xyz = xyz[:, :M].contiguous()
        feats = torch.randn(B, N, C, device="cuda")
        nbr = torch.randint(0, N, (B, M, ns), device="cuda", dtype=torch.int32)
        w1 = torch.randn(C + 3, O1, device="cuda") / C ** 0.5; b1 = torch.randn(O1, device="cuda")
        w2 = torch.randn(O1, O2, device="cuda") / O1 ** 0.5; b2 = torch.randn(O2, device="cuda")
        pmat = feats.view(B * N, C) @ w1[:C]; w1x = w1[C:].contiguous()
        li = nbr.long()
        gx = torch.gather(xyz, 1, li.view(B, M * ns, 1).expand(B, M * ns, 3)).view(B, M, ns, 3) - new_xyz.unsqueeze(2)
        gf = torch.gather(feats, 1, li.view(B, M * ns, 1).expand(B, M * ns, C)).view(B, M, ns, C)
        x = torch.cat((gf, gx), dim=3).view(-1, C + 3).double()
        h = torch.relu(x @ w1.double() + b1.double())
        got = c.pgather_rows(pmat, 0, O1, xyz, new_xyz, nbr, w1x, b1, True)
        e = float((got.double() - h).abs().max() / (h.abs().max() + 1e-9)); worst["pgather_rows"] = max(worst.get("pgather_rows", 0.0), e)
        assert e < 2e-5, ("pgather_rows", B, N, M, ns, C, O1, e)
        if O1 <= 128:
            want = torch.relu(h @ w2.double() + b2.double())
            got = c.pgather_gemm2(pmat, 0, O1, xyz, new_xyz, nbr, w1x, b1, True, w2, b2, True)
            e = float((got.double() - want).abs().max() / (want.abs().max() + 1e-9)); worst["pgather_gemm2"] = max(worst.get("pgather_gemm2", 0.0), e)
            assert e < 2e-5, ("pgather_gemm2", B, N, M, ns, C, O1, O2, e)
        # qinterp_rows
        Bq = int(rng.choice([1, 2, 8])); Nq = int(rng.integers(8, 2000)); Mq = max(3, Nq // int(rng.choice([2, 4, 8])))
        C2 = int(rng.integers(1, 300)); C1 = int(rng.choice([0, 1, 3, 4, 5, 32, 96])); O = 4 * int(rng.integers(1, 130))
        unknown = pc_all[:Bq, :Nq].contiguous(); known = unknown[:, :Mq].contiguous()
        kf = torch.randn(Bq, Mq, C2, device="cuda"); uf = torch.randn(Bq, Nq, C1, device="cuda") if C1 else None
        idx, weight = c.three_nn_with_weights(unknown, known, None)
        wt = torch.randn(C2 + C1, O, device="cuda") / (C2 + C1) ** 0.5; bias = torch.randn(O, device="cuda")
        q = (kf.view(Bq * Mq, C2) @ wt[:C2]).view(Bq, Mq, O)
        if C1 > 4:
            got = c.qinterp_rows(q, idx, weight, lin=torch.addmm(bias, uf.view(Bq * Nq, C1), wt[C2:].contiguous()), relu=True)
        else:
            got = c.qinterp_rows(q, idx, weight, skip=uf, wb=wt[C2:].contiguous() if C1 else None, bias=bias, relu=True)
        i3 = idx.long(); g3 = torch.gather(kf.double().unsqueeze(1).expand(Bq, Nq, Mq, C2), 2, i3.unsqueeze(-1).expand(Bq, Nq, 3, C2))
        interp = (g3 * weight.double().unsqueeze(-1)).sum(2)
        xx = interp if uf is None else torch.cat((interp, uf.double()), dim=2)
        want = torch.relu(xx.view(-1, C2 + C1) @ wt.double() + bias.double())
        e = float((got.double() - want).abs().max() / (want.abs().max() + 1e-9)); worst["qinterp_rows"] = max(worst.get("qinterp_rows", 0.0), e)
        assert e < 2e-5, ("qinterp_rows", Bq, Nq, Mq, C2, C1, O, e)
    rounds += 1
print(f"fuzz_mfma2 tile {a.child}: {rounds} rounds, worst relative errors " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()) + f" (seed {a.seed})")
