"""Property tests for the CPU oracle (no reference vectors exist for these leaf
kernels -- SURVEY.md section 4 -- so they are pinned by independent re-derivations
in tests/helpers.py, derived from the kernel sources the oracle cites)."""
import numpy as np
import pytest

from tests import helpers as H
from ws3d_amd import synth


def test_sqdist_convention(oracle):
    rng = np.random.default_rng(1)
    a = rng.uniform(-40, 70, (20000, 3)).astype(np.float32)
    b = a + rng.normal(0, 0.5, a.shape).astype(np.float32)
    got = np.array([oracle.sqdist(a[i], b[i]) for i in range(2000)], dtype=np.float32)
    np.testing.assert_array_equal(got, H.sqdist_np(a[:2000], b[:2000]))
    # the fused form really differs from the un-fused one on some inputs (the convention matters)
    dx, dy, dz = (a - b).T
    unfused = (dx * dx + dy * dy) + dz * dz
    assert (unfused != H.sqdist_np(a, b)).any()
    assert oracle.dist_mode() == 0


def test_opt_n_threads(oracle):
    for n in list(range(1, 70)) + [127, 128, 129, 255, 256, 1000, 1023, 1024, 1025, 2048, 4096, 16384, 65536]:
        assert oracle.opt_n_threads(n) == H.opt_n_threads(n), n


@pytest.mark.parametrize("n,m,kind,seed", [
    (64, 64, "uniform", 1), (100, 37, "uniform", 2), (512, 128, "lidar", 3), (1000, 250, "lidar", 4),
    (1024, 256, "uniform", 5), (2048, 256, "lidar", 6), (4096, 64, "uniform", 7), (3000, 100, "lidar", 8),
])
def test_fps_matches_rank_formulation(oracle, n, m, kind, seed):
    pc = synth.uniform_cloud(n, seed) if kind == "uniform" else synth.lidar_cloud(n, seed)
    xyz = pc[:, :3]
    got = oracle.furthest_point_sample(xyz[None], m)[0]
    np.testing.assert_array_equal(got, H.fps_np(xyz, m))
    assert got[0] == 0
    assert len(np.unique(got)) == min(m, n)


@pytest.mark.parametrize("case", ["dups", "lattice", "all_same", "two_points"])
def test_fps_tie_breaks(oracle, case):
    rng = np.random.default_rng(5)
    if case == "dups":
        base = synth.lidar_cloud(300, 9)[:, :3]
        xyz = base[rng.integers(0, 300, 1500)]          # heavy exact duplication
        m = 400                                          # > number of distinct points
    elif case == "lattice":
        g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(9), indexing="ij"), -1)
        xyz = g.reshape(-1, 3).astype(np.float32)[rng.permutation(12 * 12 * 9)]
        m = 300
    elif case == "all_same":
        xyz = np.ones((130, 3), dtype=np.float32)
        m = 20
    else:
        xyz = np.array([[0, 0, 0], [1, 0, 0]], dtype=np.float32)
        m = 2
    got = oracle.furthest_point_sample(xyz[None], m)[0]
    np.testing.assert_array_equal(got, H.fps_np(xyz, m))


def test_fps_batch_and_temp(oracle):
    pcs = synth.make_batch("lidar", 3, 777, 42)[:, :, :3]
    idx, temp = oracle.furthest_point_sample(pcs, 50, return_temp=True)
    for b in range(3):
        np.testing.assert_array_equal(idx[b], H.fps_np(pcs[b], 50))
        # temp = running min squared distance to the first 49 selected points
        d = np.min(np.stack([H.sqdist_np(pcs[b], pcs[b][i][None]) for i in idx[b][:-1]]), 0)
        np.testing.assert_array_equal(temp[b], np.minimum(d, np.float32(1e10)))


def test_fps_loose_float64_crosscheck(oracle):
    """In the spirit of the reference's own getGreedyPerm (lib/utils/greedFurthestPoint.py:26-37):
    float64 argmax-first FPS agrees when there are no near-ties.  (The reference function ITSELF is run by
    tests/golden/make_golden_greedyperm.py; tests/test_golden.py compares oracle and HIP kernel with its permutations.)"""
    xyz = synth.uniform_cloud(256, 123)[:, :3]
    got = oracle.furthest_point_sample(xyz[None], 32)[0]
    x = xyz.astype(np.float64)
    temp = np.full(256, np.inf)
    sel = [0]
    for _ in range(31):
        temp = np.minimum(temp, ((x - x[sel[-1]]) ** 2).sum(1))
        sel.append(int(np.argmax(temp)))
    np.testing.assert_array_equal(got, np.asarray(sel, dtype=np.int32))


@pytest.mark.parametrize("n,m,r,ns,seed", [(2048, 256, 0.5, 16, 1), (4096, 300, 1.0, 32, 2),
                                          (1024, 128, 0.1, 64, 3), (500, 77, 4.0, 8, 4)])
def test_ball_query(oracle, n, m, r, ns, seed):
    xyz = synth.lidar_cloud(n, seed)[:, :3]
    cidx = oracle.furthest_point_sample(xyz[None], m)[0]
    new_xyz = xyz[cidx]
    got = oracle.ball_query(r, ns, xyz[None], new_xyz[None])[0]
    np.testing.assert_array_equal(got, H.ball_query_np(r, ns, xyz, new_xyz))
    # centres are members of the cloud => every row has a hit, and the first hit <= own index
    assert (got[:, 0] <= cidx).all()


def test_ball_query_no_hit_rows_stay_zero(oracle):
    xyz = synth.uniform_cloud(300, 5)[:, :3]
    far = xyz[:10] + np.float32(1000.0)
    got = oracle.ball_query(0.5, 8, xyz[None], far[None])[0]
    assert (got == 0).all()
    # strict '<' : a point at exactly distance r is NOT a neighbour
    p = np.array([[0, 0, 0], [0.5, 0, 0], [0.25, 0, 0]], dtype=np.float32)
    got = oracle.ball_query(0.5, 4, p[None], p[None, :1])[0]
    np.testing.assert_array_equal(got[0], [0, 2, 0, 0])


def test_group_and_gather(oracle):
    rng = np.random.default_rng(3)
    feat = rng.standard_normal((2, 5, 333)).astype(np.float32)
    idx = rng.integers(0, 333, (2, 40, 7)).astype(np.int32)
    out = oracle.grouping_operation(feat, idx)
    ref = np.stack([feat[b][:, idx[b]] for b in range(2)])
    np.testing.assert_array_equal(out, ref)
    gi = rng.integers(0, 333, (2, 50)).astype(np.int32)
    np.testing.assert_array_equal(oracle.gather_operation(feat, gi),
                                  np.stack([feat[b][:, gi[b]] for b in range(2)]))
    # backward = scatter-add (tolerance op)
    g = rng.standard_normal(out.shape).astype(np.float32)
    gp = oracle.grouping_operation_grad(g, idx, 333)
    ref_g = np.zeros((2, 5, 333))
    for b in range(2):
        for c in range(5):
            np.add.at(ref_g[b, c], idx[b].ravel(), g[b, c].ravel().astype(np.float64))
    np.testing.assert_allclose(gp, ref_g, rtol=1e-4, atol=1e-5)
    g2 = rng.standard_normal((2, 5, 50)).astype(np.float32)
    gg = oracle.gather_operation_grad(g2, gi, 333)
    ref_g = np.zeros((2, 5, 333))
    for b in range(2):
        for c in range(5):
            np.add.at(ref_g[b, c], gi[b], g2[b, c].astype(np.float64))
    np.testing.assert_allclose(gg, ref_g, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n,m,seed", [(700, 64, 1), (1024, 256, 2), (50, 3, 3), (40, 2, 4), (10, 1, 5)])
def test_three_nn(oracle, n, m, seed):
    unk = synth.lidar_cloud(n, seed)[:, :3]
    kn = unk[oracle.furthest_point_sample(unk[None], m)[0]]
    d2, idx = oracle.three_nn_dist2(unk[None], kn[None])
    rd2, ridx = H.three_nn_np(unk, kn)
    if m >= 3:
        np.testing.assert_array_equal(idx[0], ridx)
        np.testing.assert_array_equal(d2[0], rd2)
    else:  # fewer than 3 known points: missing slots keep (inf, 0)  (interpolate_gpu.cu:30-31)
        np.testing.assert_array_equal(idx[0][:, :m], ridx[:, :m])
        assert np.isinf(d2[0][:, m:]).all() and (idx[0][:, m:] == 0).all()
    dist, _ = oracle.three_nn(unk[None], kn[None])
    np.testing.assert_array_equal(dist, np.sqrt(d2))


def test_three_nn_equal_distance_ties(oracle):
    kn = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1]], dtype=np.float32)
    unk = np.zeros((1, 3), dtype=np.float32)
    _, idx = oracle.three_nn_dist2(unk[None], kn[None])
    np.testing.assert_array_equal(idx[0, 0], [0, 1, 2])  # strict '<' keeps the earlier index


def test_three_interpolate(oracle):
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((2, 6, 64)).astype(np.float32)
    idx = rng.integers(0, 64, (2, 200, 3)).astype(np.int32)
    w = rng.uniform(0, 1, (2, 200, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    out = oracle.three_interpolate(feat, idx, w)
    ref = np.zeros((2, 6, 200))
    for b in range(2):
        for j in range(3):
            ref[b] += feat[b][:, idx[b, :, j]].astype(np.float64) * w[b, :, j].astype(np.float64)
    np.testing.assert_allclose(out, ref, atol=1e-5)
    g = rng.standard_normal(out.shape).astype(np.float32)
    gp = oracle.three_interpolate_grad(g, idx, w, 64)
    refg = np.zeros((2, 6, 64))
    for b in range(2):
        for c in range(6):
            for j in range(3):
                np.add.at(refg[b, c], idx[b, :, j], (g[b, c] * w[b, :, j]).astype(np.float64))
    np.testing.assert_allclose(gp, refg, rtol=1e-4, atol=1e-5)


# ----------------------------------------------------------------------------- iou3d
def _bev_boxes(n, seed, spread=6.0):
    rng = np.random.default_rng(seed)
    b3 = synth.random_boxes3d(n, seed)
    b3[:, 0] = rng.uniform(-spread, spread, n)
    b3[:, 2] = 30 + rng.uniform(-spread, spread, n)
    return synth.boxes3d_to_bev(b3), b3


def test_box_overlap_vs_polygon_clipping(oracle):
    A, _ = _bev_boxes(40, 1)
    B, _ = _bev_boxes(50, 2)
    ov = oracle.boxes_overlap_bev(A, B)
    ref = np.array([[H.overlap_sh(a, b) for b in B] for a in A])
    assert (ref > 0.5).sum() > 50
    np.testing.assert_allclose(ov, ref, atol=2e-3)
    iou = oracle.boxes_iou_bev(A, B)
    sa = (A[:, 2] - A[:, 0]) * (A[:, 3] - A[:, 1])
    sb = (B[:, 2] - B[:, 0]) * (B[:, 3] - B[:, 1])
    np.testing.assert_allclose(iou, ref / np.maximum(sa[:, None] + sb[None] - ref, 1e-8), atol=1e-3)
    assert (iou <= 1.0 + 1e-4).all() and (iou >= 0).all()


def test_box_overlap_special_cases(oracle):
    a = np.array([0, 0, 4, 2, 0.0], dtype=np.float32)
    assert oracle.box_overlap_pair(a, a) == pytest.approx(8.0, abs=1e-4)           # identical
    far = np.array([100, 100, 104, 102, 0.3], dtype=np.float32)
    assert oracle.box_overlap_pair(a, far) == 0.0                                   # disjoint (cnt==0)
    inner = np.array([1, 0.5, 2, 1.5, 0.4], dtype=np.float32)
    assert oracle.box_overlap_pair(a, inner) == pytest.approx(1.0, abs=1e-4)       # containment
    assert oracle.box_overlap_pair(inner, a) == pytest.approx(1.0, abs=1e-4)
    cross = np.array([1, -3, 3, 5, 0.0], dtype=np.float32)
    assert oracle.box_overlap_pair(a, cross) == pytest.approx(4.0, abs=1e-4)       # plus-shape
    rot = np.array([0, 0, 4, 2, np.pi / 2], dtype=np.float32)
    assert oracle.box_overlap_pair(a, rot) == pytest.approx(4.0, abs=1e-3)         # rotated by 90 deg


@pytest.mark.parametrize("n,thresh,normal,seed", [(64, 0.5, False, 1), (65, 0.3, False, 2), (300, 0.7, False, 3),
                                                 (200, 0.5, True, 4), (1, 0.5, False, 5), (129, 0.1, True, 6)])
def test_nms_mask_and_sweep(oracle, n, thresh, normal, seed):
    boxes, _ = _bev_boxes(n, seed, spread=4.0 if n > 1 else 1.0)
    if normal:
        x1, y1, x2, y2 = boxes[:, 0:1], boxes[:, 1:2], boxes[:, 2:3], boxes[:, 3:4]
        w = np.maximum(np.minimum(x2, x2.T) - np.maximum(x1, x1.T), 0)
        h = np.maximum(np.minimum(y2, y2.T) - np.maximum(y1, y1.T), 0)
        inter = (w * h).astype(np.float32)
        s = ((x2 - x1) * (y2 - y1)).astype(np.float32)
        iou = inter / np.maximum(s + s.T - inter, np.float32(1e-8))
    else:
        iou = oracle.boxes_iou_bev(boxes, boxes)
    mask = oracle.nms_mask(boxes, thresh, normal)
    cb = (n + 63) // 64
    assert mask.shape == (n, cb)
    bits = ((mask[:, :, None] >> np.arange(64, dtype=np.uint64)[None, None, :]) & np.uint64(1)).astype(bool)
    bits = bits.reshape(n, cb * 64)[:, :n]
    expect = iou > np.float32(thresh)
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    same_block_lower = (ii // 64 == jj // 64) & (jj <= ii)
    expect = expect & ~same_block_lower
    np.testing.assert_array_equal(bits, expect)
    keep = oracle.nms_sweep(mask)
    np.testing.assert_array_equal(keep, H.greedy_nms_from_iou(iou, thresh))
    np.testing.assert_array_equal(oracle.nms_sorted(boxes, thresh, normal), keep)
    # the mask-free greedy sweep (bench_cpu.py's lazy CPU baseline): the same keep list, also when it stops early
    np.testing.assert_array_equal(oracle.nms_sorted_lazy(boxes, thresh, normal), keep)
    np.testing.assert_array_equal(oracle.nms_sorted_lazy(boxes, thresh, normal, max_keep=7), keep[:7])
    scores = synth.distinct_scores(n, seed)
    order = np.argsort(-scores, kind="stable")
    np.testing.assert_array_equal(oracle.nms(boxes, scores, thresh, normal),
                                  order[oracle.nms_sorted(boxes[order], thresh, normal)])


@pytest.mark.parametrize("n,r,seed", [(1, 0.3, 0), (2, 0.3, 1), (200, 0.3, 2), (700, 1.0, 3), (65, 0.05, 4)])
def test_radius_nms(oracle, n, r, seed):
    """greedy centre-distance NMS of the Stage-1 proposals (generate_box_dataset.py:127-140),
    re-derived with the reference's own formulation: dense distance matrix + Python loop"""
    rng = np.random.default_rng(seed)
    c = (rng.uniform(-3, 3, (n, 2)) + rng.integers(0, 3, (n, 1)) * 0.2).astype(np.float32)
    if n > 10:
        c[5] = c[2]                                    # exact duplicate: distance 0 <= r
        c[7] = c[3] + np.float32([r, 0])               # (almost) exactly at the radius
    d = np.sqrt(((c[None, :, :] - c[:, None, :]) ** 2).sum(2, dtype=np.float32)).astype(np.float32)
    keep = [0]
    for i in range(1, n):
        if d[keep, i].min() > np.float32(r):
            keep.append(i)
    np.testing.assert_array_equal(oracle.radius_nms_sorted(c, r), np.asarray(keep))


def _intersection_upper_bounds(A, B):
    """numpy (float64) restatement of the two upper bounds of ws3d_amd/csrc/iou3d.hip iou_surely_not_above on the intersection area of
    two rotated BEV rectangles (rows x1, y1, x2, y2, ry): (1) overlap of the projections on the centre line x the narrower extent
    across it, (2) the parallelogram of one strip of each box"""
    A = A.astype(np.float64)[:, None, :]
    B = B.astype(np.float64)[None, :, :]
    hxa, hya = (A[..., 2] - A[..., 0]) / 2, (A[..., 3] - A[..., 1]) / 2
    hxb, hyb = (B[..., 2] - B[..., 0]) / 2, (B[..., 3] - B[..., 1]) / 2
    ca, sa, cb, sb = np.cos(-A[..., 4]), np.sin(-A[..., 4]), np.cos(-B[..., 4]), np.sin(-B[..., 4])
    sn, cs = np.abs(ca * sb - sa * cb), np.abs(ca * cb + sa * sb)
    with np.errstate(divide="ignore", invalid="ignore"):
        b2 = np.minimum(np.minimum(4 * hya * hyb / sn, 4 * hxa * hxb / sn), np.minimum(4 * hya * hxb / cs, 4 * hxa * hyb / cs))
        vx = (B[..., 0] + B[..., 2]) / 2 - (A[..., 0] + A[..., 2]) / 2
        vy = (B[..., 1] + B[..., 3]) / 2 - (A[..., 1] + A[..., 3]) / 2
        d2 = vx * vx + vy * vy
        va1, va2 = np.abs(vx * ca + vy * sa), np.abs(vy * ca - vx * sa)
        vb1, vb2 = np.abs(vx * cb + vy * sb), np.abs(vy * cb - vx * sb)
        E = hxa * va1 + hya * va2 + hxb * vb1 + hyb * vb2 - d2
        P = np.minimum(hxa * va2 + hya * va1, hxb * vb2 + hyb * vb1)
        b1 = np.where(d2 > 0, 2 * np.maximum(E, 0) * P / d2, np.inf)
    return np.minimum(b1, b2)


@pytest.mark.parametrize("seed", range(10))
def test_overlap_never_exceeds_the_bounds_the_nms_kernel_prunes_with(oracle, seed):
    """the HIP mask kernel clears a bit without clipping polygons when bound < thresh * (SA + SB) / (1 + thresh) minus a margin
    (2e-3 relative + 1e-3): sound iff the overlap the reference algorithm computes never exceeds the bound by more than that"""
    rng = np.random.default_rng(seed)
    n = 100
    b3 = np.zeros((n, 7), np.float32)
    spread = [0.05, 0.2, 0.6, 2.0, 6.0][seed % 5]
    b3[:, 0] = rng.normal(0, spread, n)
    b3[:, 2] = 30 + rng.normal(0, spread, n)
    b3[:, 1], b3[:, 3] = 1.0, 1.5
    if seed % 2:
        b3[:, 4], b3[:, 5] = 1.6, 3.9
    else:
        b3[:, 4], b3[:, 5] = rng.uniform(0.3, 3, n), rng.uniform(0.3, 6, n)
    b3[:, 6] = rng.uniform(-np.pi, np.pi, n) if seed % 3 else rng.normal(0.3, 0.1, n)
    bev = synth.boxes3d_to_bev(b3)
    ov = oracle.boxes_overlap_bev(bev, bev).astype(np.float64)
    bound = _intersection_upper_bounds(bev, bev)
    np.fill_diagonal(bound, np.inf)
    assert float((ov - bound).max()) < 2e-4
