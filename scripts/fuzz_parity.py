"""Randomised GPU-vs-oracle parity sweep (run on the GPU box): random shapes, radii, thresholds,
duplicate fractions and degenerate geometry for every operator; stops at the first mismatch.

    python scripts/fuzz_parity.py [--seconds 120] [--seed 0]
"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from ws3d_amd import compat as c, pn2_ops, synth, kitti_utils

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
host = lambda t: t.detach().cpu().numpy()


def cloud(rng, B, N):
    kind = rng.choice(["lidar", "uniform", "grid", "line", "hdl64", "hdl64", "clump"])
    if kind in ("hdl64", "clump"):     # KITTI's density (a prefix of a shuffled ray-cast scan is a uniform sub-sample of it); clump: thousands of points in a small cube
        pc = np.stack([synth.hdl64_cloud(16384, int(rng.integers(1, 10 ** 6)))[np.arange(N) % 16384, :3] for _ in range(B)])
        if kind == "clump" and N >= 64:
            k = int(rng.integers(N // 8, N // 2))
            for b_ in range(B):
                pc[b_, rng.choice(N, k, replace=False)] = (np.array([3.0, 1.0, 12.0]) + rng.uniform(-0.3, 0.3, (k, 3))).astype(np.float32)
    elif kind in ("lidar", "uniform"):
        pc = synth.make_batch(kind, B, N, int(rng.integers(1, 10 ** 6)), dup_frac=float(rng.choice([0, 0, 0.05, 0.5])))[:, :, :3].copy()
    elif kind == "grid":    # quantised coordinates: masses of exact distance ties
        pc = (rng.integers(0, 12, (B, N, 3)) * rng.choice([0.25, 0.5, 1.0])).astype(np.float32)
    else:
        pc = np.zeros((B, N, 3), np.float32); pc[:, :, 0] = rng.uniform(-40, 40, (B, N)).astype(np.float32)
    return np.ascontiguousarray(pc.astype(np.float32))


def one_round(rng):
    B = int(rng.integers(1, 4))
    N = int(rng.choice([7, 64, 100, 513, 1024, 2048, 3000, 4096, 9000, 16384, 20000]))
    M = int(rng.integers(1, min(N, 2048) + 1))
    pc = cloud(rng, B, N)
    # FPS
    ref = oracle.furthest_point_sample(pc, M)
    idx, new_xyz = pn2_ops.furthest_point_sample_gather(dev(pc), M)
    assert np.array_equal(host(idx), ref), ("fps", B, N, M)
    cen = host(new_xyz)
    # ball query (+ binned), fused group
    r = float(rng.choice([0.05, 0.1, 0.5, 1.0, 2.0, 4.0, 100.0])); ns = int(rng.choice([1, 8, 16, 32, 64]))
    C = int(rng.choice([0, 1, 3, 8, 96]))
    feat = rng.standard_normal((B, C, N)).astype(np.float32) if C else None
    ref_bq = oracle.ball_query(r, ns, pc, cen)
    srt = c.sort_points_x(dev(pc), min_n=int(rng.choice([64, 2048])))
    if srt is not None and ns <= 64:      # lists + compact pair table in one launch (fine-grid kernel): the lists, and the pairs as a multiset
        both = c.ball_query_pairs(r, ns, dev(pc), dev(cen), srt)
        assert both is not None and np.array_equal(host(both[0]), ref_bq), ("ball_query_pairs lists", B, N, M, r, ns)
        T = int(both[1][2].item())
        distinct = (np.diff(ref_bq.reshape(-1, ns), axis=1) > 0).sum(1) + 1
        want = np.concatenate([np.stack([np.full(int(k_), c_), ref_bq.reshape(-1, ns)[c_, :int(k_)]], 1) for c_, k_ in enumerate(distinct)]) if M else np.zeros((0, 2))
        got = np.stack([host(both[1][0])[:T], host(both[1][1])[:T]], 1)
        assert T == len(want) and np.array_equal(got[np.lexsort((got[:, 1], got[:, 0]))], want[np.lexsort((want[:, 1], want[:, 0]))]), ("ball_query_pairs table", B, N, M, r, ns)
    for s in ([None, srt] if srt is not None else [None]):
        out, nb = pn2_ops.query_and_group(r, ns, dev(pc), dev(cen), None if feat is None else dev(feat), True, return_idx=True, sorted_xyz=s)
        assert np.array_equal(host(nb), ref_bq), ("ball_query", B, N, M, r, ns, s is not None)
        gx = oracle.grouping_operation(np.ascontiguousarray(pc.transpose(0, 2, 1)), ref_bq) - cen.transpose(0, 2, 1)[..., None]
        assert np.array_equal(host(out[:, :3]), gx), ("group xyz", B, N, M, r, ns)
        if C:
            assert np.array_equal(host(out[:, 3:]), oracle.grouping_operation(feat, ref_bq)), ("group feat", B, N, M, C)
            nlc = c.query_and_group_nlc(r, ns, dev(pc), dev(cen), dev(np.ascontiguousarray(feat.transpose(0, 2, 1))), True, s)
            assert torch.equal(nlc.permute(0, 3, 1, 2), out), ("group nlc", B, N, M, C)
    # three_nn (+ binned) and interpolation
    if M >= 1:
        d2r, ir = oracle.three_nn_dist2(pc, cen)
        sk = c.sort_points_x(dev(cen), min_n=64)
        for s in ([None, sk, c.sort_points_xz(dev(cen), min_n=1)] if sk is not None else [None, c.sort_points_xz(dev(cen), min_n=1)]):
            d2 = torch.empty((B, N, 3), device="cuda"); i3 = torch.empty((B, N, 3), dtype=torch.int32, device="cuda")
            c.three_nn_wrapper(B, N, M, dev(pc), dev(cen), d2, i3, s)
            assert np.array_equal(host(i3), ir) and np.array_equal(host(d2), d2r), ("three_nn", B, N, M, s is not None)
            iw, ww_ = c.three_nn_with_weights(dev(pc), dev(cen), s)          # the one-launch form: same indices, weights of the same distances
            rcp = 1.0 / (torch.sqrt(d2) + 1e-8)
            assert torch.equal(iw, i3) and torch.equal(ww_, rcp / ((rcp[..., 0] + rcp[..., 1]) + rcp[..., 2]).unsqueeze(-1)), ("three_nn_w", B, N, M)
            if s is not None:        # queries taken in the cell order of a binned copy of the unknown set (ws3d_three_nn_wq): the same rows
                for q in (c.sort_points_x(dev(pc), min_n=1), c.sort_points_xz(dev(pc), min_n=1)):
                    if q is not None:
                        iq, wq = c.three_nn_with_weights(dev(pc), dev(cen), s, q)
                        assert torch.equal(iq, iw) and torch.equal(wq, ww_), ("three_nn_wq", B, N, M)
        Ck = int(rng.choice([4, 20, 128]))
        kf = rng.standard_normal((B, Ck, M)).astype(np.float32)
        w = rng.uniform(0, 1, (B, N, 3)).astype(np.float32)
        refi = oracle.three_interpolate(kf, ir, w)
        o = torch.empty((B, Ck, N), device="cuda")
        c.three_interpolate_wrapper(B, Ck, M, N, dev(kf), dev(ir), dev(w), o)
        assert np.array_equal(host(o), refi), ("interp", B, Ck, M, N)
        assert np.array_equal(host(c.three_interpolate_nlc(dev(np.ascontiguousarray(kf.transpose(0, 2, 1))), dev(ir), dev(w))),
                              refi.transpose(0, 2, 1)), ("interp nlc", B, Ck, M, N)
    # round 5's merged launches on the same cloud / centres: several binning jobs, several 3-NN searches, both radii of a ball query
    if 3 <= M and N <= 16384:
        half = np.ascontiguousarray(pc[:, :max(N // 2, 1)])
        bufs = c.sort_points_jobs([(dev(pc), "grid"), (dev(cen), "xz"), (dev(half), "grid"), (dev(pc), "xz")])
        if ns <= 64:
            r2 = float(rng.choice([0.05, 0.3, 1.0, 3.0])); ns2 = int(rng.choice([1, 16, 32, 64]))
            two = c.ball_query_pairs2([r, r2], [ns, ns2], dev(pc), dev(cen), bufs[0])
            assert two is not None
            for (lst, (rc_, rs_, tt_)), (rr_, nn_) in zip(two, ((r, ns), (r2, ns2))):
                want_l = ref_bq if rr_ == r and nn_ == ns else oracle.ball_query(rr_, nn_, pc, cen)
                assert np.array_equal(host(lst), want_l), ("ball_query_pairs2 lists", B, N, M, rr_, nn_)
                dl = (np.diff(want_l.reshape(-1, nn_), axis=1) > 0).sum(1) + 1
                assert int(tt_.item()) == int(dl.sum()), ("ball_query_pairs2 total", B, N, M, rr_, nn_)
        if M <= 4096:
            jobs = [(dev(pc), dev(cen), bufs[1], bufs[0] if rng.random() < 0.5 else None), (dev(half), dev(cen), bufs[1], bufs[2])]
            got = c.three_nn_jobs(jobs)
            assert got is not None
            for (un, kn, sk_, su_), (ij, wj) in zip(jobs, got):
                wi, ww2 = c.three_nn_with_weights(un, kn, sk_, su_)
                assert torch.equal(ij, wi) and torch.equal(wj, ww2), ("three_nn_jobs", B, N, M)
            assert np.array_equal(host(got[0][0]), ir), ("three_nn_jobs vs oracle", B, N, M)
    # roipool3d
    m = int(rng.integers(1, 80)); S = int(rng.choice([1, 16, 64, 512])); Cf = int(rng.choice([0, 3, 8, 128]))
    boxes = synth.proposal_boxes(B, m, int(rng.integers(1, 10 ** 6)))
    boxes[:, :, 3:6] *= rng.choice([0.2, 1.0, 3.0])
    f = rng.standard_normal((B, N, max(Cf, 1))).astype(np.float32)[:, :, :Cf] if Cf else np.zeros((B, N, 0), np.float32)
    if Cf:
        rp, re = oracle.roipool3d(pc, boxes, f, S)
        pooled = torch.zeros((B, m, S, 3 + Cf), device="cuda"); empty = torch.zeros((B, m), dtype=torch.int32, device="cuda")
        c.roipool3d_forward(dev(pc), dev(boxes), dev(f), pooled, empty)
        assert np.array_equal(host(empty), re) and np.array_equal(host(pooled), rp), ("roipool", B, N, m, S, Cf)
    # iou / nms
    n = int(rng.integers(1, 700)); thr = float(rng.choice([0.0, 0.1, 0.5, 0.8, 1.0, -0.5]))
    b3 = synth.proposal_boxes(1, n, int(rng.integers(1, 10 ** 6)))[0]
    b3[:, [0, 2]] *= rng.choice([0.05, 0.3, 1.0])
    bev = np.ascontiguousarray(synth.boxes3d_to_bev(b3))
    if rng.random() < 0.3:
        bev[:, 4] = 0.0
    for normal in (False, True):
        rk = oracle.nms_sorted(bev, thr, normal)
        k, num = c.nms_device(dev(bev), thr, normal)
        assert int(num.item()) == len(rk) and np.array_equal(host(k)[:len(rk)], rk), ("nms", n, thr, normal)
        mk = int(rng.integers(1, 60))
        kb, nb2 = c.nms_device_batched(dev(bev[None]), thr, normal, mk)
        assert int(nb2[0]) == min(mk, len(rk)) and np.array_equal(host(kb[0])[:min(mk, len(rk))], rk[:mk]), ("nms max_keep", n, thr, mk)
    # clusters of near-identical boxes (centimetres / degrees apart, same or slightly different sizes): thousands of pairs right at the
    # threshold -- the mask kernel's bound-based shortcut (iou3d.hip iou_surely_not_above) must never clear a bit the oracle sets
    nc = int(rng.integers(2, 400)); thr2 = float(rng.choice([0.2, 0.5, 0.7, 0.8, 0.9, 0.97]))
    cb = np.zeros((nc, 7), np.float32)
    ncl = int(rng.integers(1, 6))
    cc = rng.uniform(-10, 10, (ncl, 2)); ch = rng.uniform(-np.pi, np.pi, ncl); which = rng.integers(0, ncl, nc)
    sp, asp = float(rng.choice([0.01, 0.05, 0.2, 0.6])), float(rng.choice([0.005, 0.05, 0.3]))
    cb[:, 0] = cc[which, 0] + rng.normal(0, sp, nc); cb[:, 2] = 30 + cc[which, 1] + rng.normal(0, sp, nc)
    cb[:, 1], cb[:, 3] = 1.0, 1.5
    jit = 0.0 if rng.random() < 0.5 else float(rng.choice([0.02, 0.15]))
    cb[:, 4] = 1.6 * (1 + rng.uniform(-jit, jit, nc)); cb[:, 5] = 3.9 * (1 + rng.uniform(-jit, jit, nc))
    cb[:, 6] = ch[which] + rng.normal(0, asp, nc) + (np.pi / 2) * rng.integers(0, 4, nc) * (rng.random() < 0.3)
    cbev = np.ascontiguousarray(synth.boxes3d_to_bev(cb))
    assert np.array_equal(host(c.nms_mask(dev(cbev), thr2, False, full_grid=True)).view(np.uint64), oracle.nms_mask(cbev, thr2, False)), ("cluster mask", nc, thr2, sp, asp, jit)
    rk2 = oracle.nms_sorted(cbev, thr2, False)
    k2, num2 = c.nms_device(dev(cbev), thr2, False)
    assert int(num2.item()) == len(rk2) and np.array_equal(host(k2)[:len(rk2)], rk2), ("cluster nms", nc, thr2)
    na = min(n, 60)
    ov = torch.zeros((na, n), device="cuda")
    c.boxes_overlap_bev_gpu(dev(bev[:na]), dev(bev), ov)
    assert np.array_equal(host(ov), oracle.boxes_overlap_bev(bev[:na], bev)), ("overlap", n)
    # proposal-stage kernels against their torch compositions
    from ws3d_amd import stage1
    Bp, Np = int(rng.integers(1, 4)), int(rng.choice([1, 64, 1000, 4096, 16384]))
    sc = torch.from_numpy(rng.standard_normal((Bp, Np)).astype(np.float32)).cuda()
    if Np > 10:
        sc[0, 3:8] = sc[0, 1]
    kk = int(rng.integers(0, Np + 1))
    v, i = c.topk_sorted(sc, kk)
    rv, ri = torch.sort(sc, dim=1, descending=True, stable=True)
    assert torch.equal(v, rv[:, :kk]) and torch.equal(i, ri[:, :kk]), ("topk", Bp, Np, kk)
    xyz_t = torch.from_numpy(rng.uniform(-40, 40, (Bp, Np, 3)).astype(np.float32)).cuda()
    reg_t = torch.from_numpy(rng.standard_normal((Bp, Np, 40)).astype(np.float32)).cuda()
    if rng.random() < 0.3:
        reg_t[:, :, :20] = torch.round(reg_t[:, :, :20])           # ties in the bin argmax
    h, w, l = stage1.DEFAULT_CFG.cls_mean_size
    box = c.decode_center_boxes(xyz_t, reg_t, 4.0, 0.8, (h, w, l))
    ctr = stage1.decode_center_target(xyz_t.view(-1, 3), reg_t.view(-1, 40), 4.0, 0.8).view(Bp, Np, 3)
    ref_box = torch.stack((ctr[..., 0], xyz_t[..., 1] + h / 2, ctr[..., 2], torch.full_like(ctr[..., 0], h), torch.full_like(ctr[..., 0], w),
                           torch.full_like(ctr[..., 0], l), stage1.synthetic_orientation(Np, xyz_t.device).unsqueeze(0).expand(Bp, Np)), dim=2)
    assert torch.equal(box, ref_box), ("decode", Bp, Np)
    if Np <= 16384:
        lg = sc * float(rng.choice([1.0, 8.0, 40.0]))                 # logits up to saturation: equal sigmoids of distinct logits
        for spread in (False, True):
            vs, is_ = c.topk_sorted(lg, kk, spread=spread, sigmoid=True)
            wv, wi_ = c.topk_sorted(torch.sigmoid(lg), kk, spread=spread)
            assert torch.equal(vs.view(torch.int32), wv.view(torch.int32)) and torch.equal(is_, wi_), ("topk sigmoid", Bp, Np, kk, spread)
        if kk > 0:
            rows, bev_ = c.decode_gather_boxes_bev(xyz_t, reg_t, i, 4.0, 0.8, (h, w, l))
            wr, wb_ = c.gather_boxes_bev(box, i)
            assert torch.equal(rows.view(torch.int32), wr.view(torch.int32)) and torch.equal(bev_.view(torch.int32), wb_.view(torch.int32)), ("decode_gather", Bp, Np, kk)
    Cc = int(rng.choice([3, 4, 9]))
    pcl = torch.from_numpy(rng.standard_normal((Bp, Np, Cc)).astype(np.float32)).cuda()
    junk = torch.full((int(rng.integers(1, 5000)) * 4,), float("nan"), device="cuda")
    sx, sf = c.split_points_clear(pcl, junk)
    assert torch.equal(sx, pcl[..., :3].contiguous()) and (sf is None if Cc == 3 else torch.equal(sf, pcl[..., 3:].contiguous())) and bool((junk.view(torch.int32) == 0).all()), ("split_points_clear", Bp, Np, Cc)
    # training-step kernels: deterministic scatter (bit-equal to the sequential loop), pool (bit-equal to F.max_pool2d)
    Bg, Cg = int(rng.integers(1, 4)), int(rng.choice([1, 7, 31, 32, 33, 64, 96, 129, 256, 300, 513]))
    Ng, Mg, nsg = int(rng.integers(1, 600)), int(rng.integers(0, 80)), int(rng.choice([1, 3, 16, 32]))
    gi = rng.integers(0, max(Ng // int(rng.choice([1, 3, 50])), 1), (Bg, Mg, nsg)).astype(np.int32)
    gg = (rng.standard_normal((Bg, Cg, Mg, nsg)) * 10 ** rng.uniform(-2, 2, (Bg, Cg, Mg, nsg))).astype(np.float32)
    og = torch.full((Bg, Cg, Ng), float("nan"), device="cuda")
    c.group_points_grad_det(Bg, Cg, Ng, Mg, nsg, dev(gg), dev(gi), og)
    assert np.array_equal(host(og), oracle.grouping_operation_grad(gg, gi, Ng)), ("group_grad_det", Bg, Cg, Ng, Mg, nsg)
    nu, mk = int(rng.integers(1, 700)), int(rng.integers(1, 90))
    ii = rng.integers(0, mk, (Bg, nu, 3)).astype(np.int32)
    ww = rng.uniform(0, 1, (Bg, nu, 3)).astype(np.float32)
    gu = rng.standard_normal((Bg, Cg, nu)).astype(np.float32)
    oi = torch.full((Bg, Cg, mk), float("nan"), device="cuda")
    c.three_interpolate_grad_det(Bg, Cg, nu, mk, dev(gu), dev(ii), dev(ww), oi)
    assert np.array_equal(host(oi), oracle.three_interpolate_grad(gu, ii, ww, mk)), ("interp_grad_det", Bg, Cg, nu, mk)
    nsp = int(rng.choice([1, 2, 4, 7, 8, 16, 32, 33, 64, 128, 255]))
    xp = torch.from_numpy(np.round(rng.standard_normal((int(rng.integers(1, 4)), int(rng.integers(1, 9)), int(rng.integers(1, 200)), nsp)) * 2).astype(np.float32) / 2).cuda()
    if rng.random() < 0.3 and xp.numel() > 4:
        flat = xp.view(-1)
        flat[int(rng.integers(0, flat.numel()))] = float("nan"); flat[int(rng.integers(0, flat.numel()))] = float("-inf")
    pv, pa = c.pool_nsample(xp)
    rvp, rip = torch.nn.functional.max_pool2d(xp, kernel_size=[1, nsp], return_indices=True)
    same = torch.equal(torch.nan_to_num(pv, nan=7e30), torch.nan_to_num(rvp.squeeze(-1), nan=7e30))
    assert same and torch.equal(pa.long(), rip.squeeze(-1) % nsp), ("pool_nsample", tuple(xp.shape))
    cxz = np.ascontiguousarray(b3[None, :, [0, 2]])
    rr = float(rng.choice([0.05, 0.3, 1.0]))
    kr, nr = c.radius_nms_device_batched(dev(cxz), rr)
    refr = oracle.radius_nms_sorted(cxz[0], rr)
    assert int(nr[0]) == len(refr) and np.array_equal(host(kr[0])[:len(refr)], refr), ("radius nms", n, rr)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=120); ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    oracle.set_threads(max(1, min(oracle.max_threads(), 64)))
    rng = np.random.default_rng(a.seed)
    t0, rounds = time.time(), 0
    while time.time() - t0 < a.seconds:
        one_round(rng); rounds += 1
    print(f"fuzz_parity: {rounds} rounds, all operators bit-equal to the oracle (seed {a.seed})")
