"""Pin the oracle's roipool3d restatement against the REAL reference.

The reference's roipool3d host file carries a complete CPU implementation
(lib/utils/roipool3d/src/roipool3d.cpp:82-195).  oracle/build_ref.py compiles it
from the reference tree into oracle/_ref/ (git-ignored, shipped to the GPU box).
These tests run wherever that .so exists and are skipped otherwise.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import build_ref  # noqa: E402
from ws3d_amd import synth  # noqa: E402

ref = build_ref.load()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/roipool3d_ref.so not built")


def _scene(n, m, c, seed, enlarge=1.0):
    pc, cars = synth.lidar_cloud(n, seed, return_boxes=True)
    boxes = synth.proposal_boxes(1, m, seed // 1000 if seed >= 1000 else 0)[0]
    # put the first boxes exactly on the synthetic cars so they are non-empty
    k = min(m, cars.shape[0])
    boxes[:k] = cars[:k]
    boxes[:, 3:6] += 2 * enlarge
    boxes[:, 1] += enlarge
    feat = np.random.default_rng(seed).standard_normal((n, c)).astype(np.float32)
    return pc[:, :3].copy(), boxes, feat


@pytest.mark.parametrize("n,m,seed", [(2048, 24, 5), (16384, 100, 3000), (4096, 64, 77)])
def test_pts_in_boxes3d_matches_reference(oracle, n, m, seed):
    pts, boxes, _ = _scene(n, m, 4, seed)
    flag_ref = torch.zeros((m, n), dtype=torch.int64)
    ref.pts_in_boxes3d_cpu(flag_ref, torch.from_numpy(pts), torch.from_numpy(boxes))
    flag = oracle.pts_in_boxes3d(pts, boxes)
    assert flag_ref.sum() > 0
    np.testing.assert_array_equal(flag, flag_ref.numpy())


@pytest.mark.parametrize("n,m,c,s,seed", [(2048, 24, 8, 64, 11), (16384, 100, 128, 512, 3001),
                                          (4096, 40, 16, 512, 9), (512, 8, 3, 16, 2)])
def test_roipool3d_cpu_matches_reference(oracle, n, m, c, s, seed):
    pts, boxes, feat = _scene(n, m, c, seed)
    pp = torch.zeros((m, s, 3))
    pf = torch.zeros((m, s, c))
    pe = torch.zeros((m,), dtype=torch.int64)
    ref.roipool3d_cpu(torch.from_numpy(pts), torch.from_numpy(boxes), torch.from_numpy(feat), pp, pf, pe)
    o_pts, o_feat, o_empty = oracle.roipool3d_cpu(pts, boxes, feat, s)
    np.testing.assert_array_equal(o_empty, pe.numpy())
    np.testing.assert_array_equal(o_pts, pp.numpy())
    np.testing.assert_array_equal(o_feat, pf.numpy())
    assert (o_empty == 0).any() and (o_empty == 1).any() or m < 16
    # the batched GPU-semantics entry point must agree with the CPU twin
    pooled, empty = oracle.roipool3d(pts[None], boxes[None], feat[None], s)
    np.testing.assert_array_equal(empty[0], pe.numpy().astype(np.int32))
    np.testing.assert_array_equal(pooled[0, :, :, :3], pp.numpy())
    np.testing.assert_array_equal(pooled[0, :, :, 3:], pf.numpy())


def test_boundary_and_degenerate_boxes(oracle):
    """points exactly on faces (closed interval), zero-size box, far prefilter (|dx|>10)."""
    box = np.array([[0, 1, 10, 2, 2, 4, 0.0]], dtype=np.float32)  # y in [-1,1], x in [-2,2], z in [9,11]
    pts = np.array([[2, 0, 10], [-2, 0, 10], [0, 1, 11], [0, -1, 9], [2.0000002, 0, 10],
                    [0, 1.0000001, 10], [0, 0, 10], [10.5, 0, 10]], dtype=np.float32)
    flag_ref = torch.zeros((1, len(pts)), dtype=torch.int64)
    ref.pts_in_boxes3d_cpu(flag_ref, torch.from_numpy(pts), torch.from_numpy(box))
    np.testing.assert_array_equal(oracle.pts_in_boxes3d(pts, box), flag_ref.numpy())
    assert flag_ref.numpy()[0, :4].all() and flag_ref.numpy()[0, 6] == 1
    big = np.array([[0, 1, 10, 2, 30, 30, 0.3]], dtype=np.float32)  # box wider than the 10 m prefilter
    rng = np.random.default_rng(0)
    p2 = rng.uniform([-16, -1, -6], [16, 1, 26], (4000, 3)).astype(np.float32)
    flag_ref = torch.zeros((1, len(p2)), dtype=torch.int64)
    ref.pts_in_boxes3d_cpu(flag_ref, torch.from_numpy(p2), torch.from_numpy(big))
    np.testing.assert_array_equal(oracle.pts_in_boxes3d(p2, big), flag_ref.numpy())
