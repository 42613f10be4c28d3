"""GPU parity at BASELINE.json's FULL sizes.  The oracle handles single scenes of these sizes in
seconds (it is threaded for the occasion); whole batches are checked through size-independent
properties of the domain: batch independence / replica determinism, "ascending then padded" rows,
gather identities, NMS suppression invariants."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from ws3d_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from ws3d_amd import compat, iou3d_ops, kitti_utils, pn2_ops
    import types
    return types.SimpleNamespace(pn=pn2_ops, iou=iou3d_ops, c=compat, ku=kitti_utils)


@pytest.fixture(scope="module")
def oracle():
    import oracle as o
    o.set_threads(max(1, min(o.max_threads(), 64)))
    yield o
    o.set_threads(1)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def test_config2_batch256_properties(ops, oracle):
    """c2 at the benchmark batch: 256 scenes x 16384 points -> 4096 centres, r=0.1, nsample=64"""
    B, N, M, NS, R = 256, 16384, 4096, 64, 0.1
    distinct = 32
    base = np.stack([synth.lidar_cloud(N, 2000 + s) for s in range(distinct)])
    pc = np.ascontiguousarray(np.tile(base, (B // distinct, 1, 1)))             # scene b == scene b % 32
    xyz = dev(pc[:, :, :3].copy())
    feat = dev(np.ascontiguousarray(pc[:, :, 3:].transpose(0, 2, 1)))
    idx = torch.empty((B, M), dtype=torch.int32, device="cuda")
    new_xyz = torch.empty((B, M, 3), device="cuda")
    ops.c.furthest_point_sampling_gather(B, N, M, xyz, None, idx, new_xyz)
    nbr = torch.empty((B, M, NS), dtype=torch.int32, device="cuda")
    out = torch.empty((B, 4, M, NS), device="cuda")
    ops.c.query_and_group(B, N, M, 1, R, NS, True, xyz, new_xyz, feat, nbr, out, ops.c.sort_points_x(xyz))
    i64 = idx.long()
    # FPS: starts at 0, in range, no point twice, gathered coordinates are the points
    assert (i64[:, 0] == 0).all() and (i64 >= 0).all() and (i64 < N).all()
    assert (torch.sort(i64, dim=1)[0].diff(dim=1) > 0).all()
    assert torch.equal(new_xyz, torch.gather(xyz, 1, i64.unsqueeze(-1).expand(B, M, 3)))
    # replicas of a scene (other workgroups / CUs / XCDs) give the identical answer
    for t in (idx, nbr, out):
        v = t.view(B // distinct, distinct, *t.shape[1:])
        assert (v == v[0:1]).all()
    # the first scenes against the oracle
    k = 3
    ref_idx = oracle.furthest_point_sample(pc[:k, :, :3].copy(), M)
    np.testing.assert_array_equal(host(idx[:k]), ref_idx)
    ref_nbr = oracle.ball_query(R, NS, pc[:k, :, :3].copy(), host(new_xyz[:k]))
    np.testing.assert_array_equal(host(nbr[:k]), ref_nbr)
    # ball-query rows: ascending prefix, then the first hit repeated; all inside the radius
    n64 = nbr.long()
    asc = (n64.diff(dim=2) > 0).long()
    cnt = 1 + asc.cumprod(dim=2).sum(dim=2)                                  # length of the strictly ascending prefix
    pos = torch.arange(NS, device="cuda").view(1, 1, NS)
    tail = pos >= cnt.unsqueeze(-1)
    assert (n64[tail] == n64[:, :, :1].expand(B, M, NS)[tail]).all()          # the rest repeats the first hit
    assert (cnt >= 1).all() and (cnt < NS).any() and (cnt > 1).any()         # padded rows and multi-hit rows both occur
    g = torch.gather(xyz, 1, n64.view(B, M * NS, 1).expand(B, M * NS, 3)).view(B, M, NS, 3)
    rel = g - new_xyz.unsqueeze(2)
    assert ((rel * rel).sum(-1) < R * R * (1 + 1e-5)).all()
    # fused grouping == gather identities, exactly
    assert torch.equal(out[:, :3], rel.permute(0, 3, 1, 2))
    assert torch.equal(out[:, 3], torch.gather(feat[:, 0], 1, n64.view(B, M * NS)).view(B, M, NS))
    # batch independence: a scene processed alone equals its row in the batch
    i1 = torch.empty((1, M), dtype=torch.int32, device="cuda"); x1 = torch.empty((1, M, 3), device="cuda")
    ops.c.furthest_point_sampling_gather(1, N, M, xyz[7:8].contiguous(), None, i1, x1)
    assert torch.equal(i1[0], idx[7])


def test_config5_dense_scene_bit_exact(ops, oracle):
    """c5: N=65536 points, 512 proposals, S=512, C=128 -- roipool3d and rotated NMS against the oracle
    at full size, FPS through the streaming kernel"""
    N, M, C, S = 65536, 512, 128, 512
    B = 2
    pc = np.stack([synth.lidar_cloud(N, 5000 + s) for s in range(B)])
    boxes = synth.proposal_boxes(B, M, 5)
    for b in range(B):
        cars = synth.random_boxes3d(15, (5000 + b) * 7919 + 13)
        boxes[b, :M // 2] = cars[np.arange(M // 2) % 15]
        boxes[b, :M // 2, [0, 2]] += np.random.default_rng(b).normal(0, 0.3, (2, M // 2)).astype(np.float32)
        boxes[b, -10:, 1] += 30.0                                               # ten proposals far below the ground: empty RoIs
    feat = np.random.default_rng(1).standard_normal((B, N, C)).astype(np.float32)
    enl = ops.ku.enlarge_box3d(torch.from_numpy(boxes).view(-1, 7), 1.0).view(B, M, 7).contiguous()
    pooled = torch.zeros((B, M, S, 3 + C), device="cuda")
    empty = torch.zeros((B, M), dtype=torch.int32, device="cuda")
    ops.c.roipool3d_forward(dev(pc[:, :, :3].copy()), enl.cuda(), dev(feat), pooled, empty)
    ref_p, ref_e = oracle.roipool3d(pc[:, :, :3].copy(), enl.numpy(), feat, S)
    np.testing.assert_array_equal(host(empty), ref_e)
    assert 0 < int(ref_e.sum()) < B * M                                        # both kinds of RoI occur
    np.testing.assert_array_equal(host(pooled), ref_p)
    # rotated NMS, 512 boxes, thresh 0.7
    scores = synth.distinct_scores(M, 50)
    order = np.argsort(-scores, kind="stable")
    bev = np.ascontiguousarray(synth.boxes3d_to_bev(boxes[0])[order])
    keep, num = ops.c.nms_device(dev(bev), 0.7, False)
    ref_keep = oracle.nms_sorted(bev, 0.7, False)
    assert int(num.item()) == len(ref_keep)
    np.testing.assert_array_equal(host(keep)[:len(ref_keep)], ref_keep)
    # FPS beyond the register-resident limit
    ref_idx = oracle.furthest_point_sample(pc[:1, :, :3].copy(), 2048)
    np.testing.assert_array_equal(host(ops.pn.furthest_point_sample(dev(pc[:1, :, :3].copy()), 2048)), ref_idx)


def test_config3_nms_9000_boxes(ops, oracle):
    """the Stage-1 proposal NMS at its real size: 9000 score-sorted rotated boxes, thresh 0.8 --
    full keep list and the first-100 early stop against the oracle, plus the suppression invariants
    checked with the pairwise IoU kernel"""
    n, thr = 9000, 0.8
    rng = np.random.default_rng(9)
    boxes3d = synth.proposal_boxes(1, n, 3)[0]
    boxes3d[:, [0, 2]] += rng.normal(0, 0.15, (n, 2)).astype(np.float32)        # crowded around the cars
    bev = np.ascontiguousarray(synth.boxes3d_to_bev(boxes3d))
    ref = oracle.nms_sorted(bev, thr, False)
    keep, num = ops.c.nms_device(dev(bev), thr, False)
    k = int(num.item())
    assert k == len(ref)
    np.testing.assert_array_equal(host(keep)[:k], ref)
    keep100, num100 = ops.c.nms_device_batched(dev(bev[None]), thr, False, 100)
    assert int(num100[0]) == min(100, k)
    np.testing.assert_array_equal(host(keep100[0])[:min(100, k)], ref[:100])
    # invariants: kept boxes do not suppress each other; every dropped box is suppressed by an EARLIER kept one
    kept = keep[:k]
    iou = ops.iou.boxes_iou_bev(dev(bev)[kept], dev(bev))                        # (k, n)
    kk = iou[:, kept]
    assert (kk.triu(1) <= thr).all()
    earlier = kept.view(-1, 1) < torch.arange(n, device="cuda").view(1, -1)
    sup = ((iou > thr) & earlier).any(dim=0)
    is_kept = torch.zeros(n, dtype=torch.bool, device="cuda")
    is_kept[kept] = True
    assert torch.equal(sup, ~is_kept)


def test_dense_scan_fps_65536_to_16384_full_oracle_check(ops, oracle):
    """SURVEY a1's dense shape: furthest_point_sample 65536 -> 16384 (fps_big_kernel: min-distance in registers, xyz streamed)
    against the oracle over ALL 16384 indices, plus the min-distance buffer contract, on two scenes (one with duplicates)"""
    N, M = 65536, 16384
    pcs = np.stack([synth.lidar_cloud(N, 5100)[:, :3], synth.make_batch("lidar", 1, N, 5101, dup_frac=0.05)[0, :, :3]]).copy()
    ref, ref_temp = oracle.furthest_point_sample(pcs, M, return_temp=True)
    x = dev(pcs)
    temp = torch.full((2, N), 1e10, device="cuda")
    idx = torch.empty((2, M), dtype=torch.int32, device="cuda")
    new_xyz = torch.empty((2, M, 3), device="cuda")
    ops.c.furthest_point_sampling_gather(2, N, M, x, temp, idx, new_xyz)
    np.testing.assert_array_equal(host(idx), ref)
    np.testing.assert_array_equal(host(temp), ref_temp)
    np.testing.assert_array_equal(host(new_xyz), np.stack([pcs[b][ref[b]] for b in range(2)]))


@pytest.mark.parametrize("N,M", [(16385, 300), (20000, 500), (32768, 400), (33000, 200), (50001, 300)])
def test_large_n_fps_ragged_sizes(ops, oracle, N, M):
    """the register-resident-min-distance kernel at ragged sizes either side of its two instances (<= 32768, <= 65536),
    without a temp buffer (the round-1 streaming kernel required one)"""
    pcs = synth.make_batch("lidar", 2, N, 5200 + N % 97, dup_frac=0.02)[:, :, :3].copy()
    ref = oracle.furthest_point_sample(pcs, M)
    idx = torch.empty((2, M), dtype=torch.int32, device="cuda")
    ops.c.furthest_point_sampling_wrapper(2, N, M, dev(pcs), None, idx)
    np.testing.assert_array_equal(host(idx), ref)
