"""KITTI ingest against the fixture produced by the reference's own data path
(tests/golden/make_golden_kitti.py -> tests/golden/kitti_ingest.json).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

from ws3d_amd import kitti_io, synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(HERE, "golden", "kitti_ingest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def tree(tmp_path_factory, fx):
    root = str(tmp_path_factory.mktemp("kitti"))
    synth.write_kitti_tree(root, [tuple(s) for s in fx["scenes"]])
    return root


def test_pts_input_matches_reference_loader_bit_for_bit(fx, tree):
    """same directory, same numpy seed -> the reference's KittiRCNNDataset(mode='TEST') and
    KittiScenes select the same 16384 points in the same order (subsample and tiling paths)"""
    ds = kitti_io.KittiScenes(tree, "val", npoints=16384)
    assert len(ds) == len(fx["samples"])
    np.random.seed(fx["np_seed"])
    for i, ref in enumerate(fx["samples"]):
        s = ds[i]
        p = np.ascontiguousarray(s["pts_input"])
        assert s["sample_id"] == ref["sample_id"] and list(p.shape) == ref["shape"] and str(p.dtype) == ref["dtype"]
        np.testing.assert_array_equal(p.reshape(-1)[np.array(ref["pos"])], np.array(ref["val"], dtype=p.dtype))
        assert hashlib.sha256(p.tobytes()).hexdigest() == ref["sha256"]
        assert (p[:, 3] >= -0.5).all() and (p[:, 3] <= 0.5).all()
        (x0, x1), (y0, y1), (z0, z1) = kitti_io.PC_AREA_SCOPE
        assert (p[:, 0] >= x0).all() and (p[:, 0] <= x1).all() and (p[:, 2] >= z0).all() and (p[:, 2] <= z1).all()


def test_sampler_with_private_stream_and_without_sampling(tree):
    ds = kitti_io.KittiScenes(tree, "val", npoints=12000, rng=np.random.RandomState(3))
    a = ds[0]["pts_input"]
    b = kitti_io.KittiScenes(tree, "val", npoints=12000, rng=np.random.RandomState(3))[0]["pts_input"]
    np.testing.assert_array_equal(a, b)
    assert a.shape == (12000, 4)
    full = kitti_io.KittiScenes(tree, "val", random_select=False)[0]["pts_input"]
    assert full.shape[0] > 16384                       # every valid point, unsampled
    # every sampled point is one of the valid points
    assert set(map(bytes, np.ascontiguousarray(a))) <= set(map(bytes, np.ascontiguousarray(full.astype(a.dtype))))
    with pytest.raises(ValueError):
        kitti_io.KittiScenes(tree, "val", npoints=4096)[0]
    batch = kitti_io.collate_scenes([kitti_io.KittiScenes(tree, "val")[i] for i in range(2)])
    assert batch["pts_input"].shape == (2, 16384, 4) and batch["pts_input"].dtype == np.float32


def test_calibration_and_labels(fx, tree):
    ds = kitti_io.KittiScenes(tree, "val")
    calib = ds.get_calib(7)
    pts = synth.velodyne_scan(50, 9)[:, :3]
    rect = calib.lidar_to_rect(pts)
    img, depth = calib.rect_to_img(rect)
    np.testing.assert_array_equal(rect, np.array(fx["calib"]["rect"], dtype=rect.dtype))
    np.testing.assert_array_equal(img, np.array(fx["calib"]["img"], dtype=img.dtype))
    np.testing.assert_array_equal(depth, np.array(fx["calib"]["depth"], dtype=depth.dtype))
    np.testing.assert_array_equal(calib.img_to_rect(img[:, 0], img[:, 1], depth),
                                  np.array(fx["calib"]["img_to_rect"]))
    assert float(calib.tx) == fx["calib"]["tx"] and float(calib.ty) == fx["calib"]["ty"]
    i2, d2 = calib.lidar_to_img(pts)
    np.testing.assert_array_equal(i2, img)
    objs = ds.get_label(7)
    assert len(objs) == len(fx["labels"])
    for o, r in zip(objs, fx["labels"]):
        assert (o.cls_type, o.level, o.ry, o.score) == (r["cls_type"], r["level"], r["ry"], r["score"])
        assert [float(v) for v in o.pos] == r["pos"] and [o.h, o.w, o.l] == r["hwl"]
        assert o.box3d().shape == (7,)
    assert ds.get_image_shape(7) == (375, 1242, 3)


def test_result_writer_text_equals_reference(fx, tree, tmp_path):
    calib = kitti_io.KittiScenes(tree, "val").get_calib(7)
    b, m, cfg_id = fx["result_file"]["boxes_config"]
    boxes = synth.proposal_boxes(b, m, cfg_id)[0].astype(np.float32)
    boxes[:, 2] = np.abs(boxes[:, 2]) + 6.0
    boxes[3, 2] = 1.5
    scores = np.random.default_rng(3).normal(0, 2, 12).astype(np.float32)
    path = kitti_io.save_kitti_format(7, calib, boxes, str(tmp_path), scores, (375, 1242, 3))
    text = open(path).read()
    assert text == fx["result_file"]["text"]
    assert len(text.splitlines()) < m                   # the oversized 2-D box was dropped
    # written files parse back as labels with scores
    back = kitti_io.read_label_file(path)
    assert all(o.cls_type == "Car" and o.score != -1.0 for o in back)
    assert kitti_io.format_kitti_result(boxes[:0], scores[:0], calib, (375, 1242, 3)) == ""


def test_corners_and_bad_files(tmp_path):
    box = np.array([[1.0, 2.0, 10.0, 1.5, 1.6, 4.0, 0.0], [0.0, 1.0, 5.0, 2.0, 1.0, 3.0, np.pi / 2]], dtype=np.float32)
    c = kitti_io.boxes3d_to_corners3d(box)
    assert c.shape == (2, 8, 3)
    np.testing.assert_allclose(c[0, :4, 1], 2.0)
    np.testing.assert_allclose(c[0, 4:, 1], 0.5)
    np.testing.assert_allclose(sorted(set(np.round(c[0, :, 0], 5))), [-1.0, 3.0])      # x = cx -+ l/2 at ry = 0
    np.testing.assert_allclose(sorted(set(np.round(c[1, :, 2], 4))), [3.5, 6.5], atol=1e-4)  # ry = 90 deg: l along z
    bad = tmp_path / "x.bin"
    np.arange(7, dtype=np.float32).tofile(bad)
    with pytest.raises(ValueError):
        kitti_io.read_velodyne_bin(str(bad))
    (tmp_path / "x.png").write_bytes(b"not a png at all, really not")
    with pytest.raises(ValueError):
        kitti_io.read_image_shape(str(tmp_path / "x.png"))
