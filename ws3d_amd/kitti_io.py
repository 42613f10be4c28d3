"""KITTI ingest for the Stage-1 path: velodyne ``.bin`` reader, calibration,
image-frustum / range filter, the 16384-point sampler that produces ``pts_input`` and the
KITTI-format result writer.  Host-side numpy, like the reference: this is the caller side of the
hot path, not a kernel.

Reference behaviour reproduced (file:line in /root/reference):
  * ``KittiDataset.get_lidar / get_calib / get_image_shape`` (lib/datasets/kitti_dataset.py:36-52)
  * ``Calibration`` (lib/utils/calibration.py:5-141): float32 matrices straight from the file,
    ``lidar_to_rect = [p 1] . (V2C^T . R0^T)``, ``rect_to_img`` divides by the rect depth
  * ``KittiRCNNDataset.get_valid_flag`` (lib/datasets/kitti_rcnn_dataset.py:139-160)
  * ``KittiRCNNDataset.get_rpn_sample`` in TEST/EVAL mode (:399-452): sort by descending lidar z,
    project, filter, near/far sampling with the *global* ``numpy.random`` stream, intensity - 0.5
  * ``save_kitti_format`` (tools/eval_auto.py:108-135)
  * ``Object3d`` label parsing (lib/utils/object3d.py:12-33)

The sampler draws from ``numpy.random`` in exactly the reference's call order, so with the same
seed it selects the same points as the reference data loader (tests/golden/kitti_ingest.json).
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from typing import Sequence, Tuple

import numpy as np

PC_AREA_SCOPE = ((-40.0, 40.0), (-3.0, 3.0), (0.0, 70.4))   # tools/cfgs/weaklyRPN.yaml:18


# ----------------------------------------------------------------------------- files
def read_velodyne_bin(path: str) -> np.ndarray:
    """(N,4) float32 [x, y, z, reflectance] in the velodyne frame (kitti_dataset.py:43-46)"""
    pts = np.fromfile(path, dtype=np.float32)
    if pts.size % 4:
        raise ValueError(f"{path}: {pts.size} float32 values is not a multiple of 4")
    return pts.reshape(-1, 4)


def read_image_shape(path: str) -> Tuple[int, int, int]:
    """(height, width, 3) from the PNG header alone (the reference opens the image with PIL only
    to read its size, kitti_dataset.py:36-41)"""
    with open(path, "rb") as f:
        head = f.read(24)
    if len(head) < 24 or head[:8] != b"\x89PNG\r\n\x1a\n" or head[12:16] != b"IHDR":
        raise ValueError(f"{path}: not a PNG file")
    width, height = struct.unpack(">II", head[16:24])
    return int(height), int(width), 3


def read_calib_file(path: str) -> dict:
    """lines 2..5 of a KITTI object calib file: P2, P3, R0_rect, Tr_velo_to_cam (calibration.py:5-21)"""
    with open(path) as f:
        lines = f.readlines()

    def row(i, shape):
        return np.array(lines[i].strip().split(" ")[1:], dtype=np.float32).reshape(shape)

    return {"P2": row(2, (3, 4)), "P3": row(3, (3, 4)), "R0": row(4, (3, 3)), "Tr_velo2cam": row(5, (3, 4))}


class Calibration:
    """Same attribute and method names as lib/utils/calibration.py:24-141"""

    def __init__(self, calib):
        if isinstance(calib, (str, os.PathLike)):
            calib = read_calib_file(os.fspath(calib))
        self.P2 = calib["P2"]
        self.R0 = calib["R0"]
        self.V2C = calib["Tr_velo2cam"]
        self.cu, self.cv = self.P2[0, 2], self.P2[1, 2]
        self.fu, self.fv = self.P2[0, 0], self.P2[1, 1]
        self.tx = self.P2[0, 3] / (-self.fu)
        self.ty = self.P2[1, 3] / (-self.fv)

    @staticmethod
    def cart_to_hom(pts: np.ndarray) -> np.ndarray:
        return np.hstack((pts, np.ones((pts.shape[0], 1), dtype=np.float32)))

    def lidar_to_rect(self, pts_lidar: np.ndarray) -> np.ndarray:
        return np.dot(self.cart_to_hom(pts_lidar), np.dot(self.V2C.T, self.R0.T))

    def rect_to_img(self, pts_rect: np.ndarray):
        hom = self.cart_to_hom(pts_rect)
        img_hom = np.dot(hom, self.P2.T)
        pts_img = (img_hom[:, 0:2].T / hom[:, 2]).T
        depth = img_hom[:, 2] - self.P2.T[3, 2]
        return pts_img, depth

    def lidar_to_img(self, pts_lidar: np.ndarray):
        return self.rect_to_img(self.lidar_to_rect(pts_lidar))

    def img_to_rect(self, u, v, depth_rect):
        x = ((u - self.cu) * depth_rect) / self.fu + self.tx
        y = ((v - self.cv) * depth_rect) / self.fv + self.ty
        return np.concatenate((x.reshape(-1, 1), y.reshape(-1, 1), depth_rect.reshape(-1, 1)), axis=1)

    def corners3d_to_img_boxes(self, corners3d: np.ndarray):
        n = corners3d.shape[0]
        hom = np.concatenate((corners3d, np.ones((n, 8, 1))), axis=2)
        img = np.matmul(hom, self.P2.T)
        x, y = img[:, :, 0] / img[:, :, 2], img[:, :, 1] / img[:, :, 2]
        boxes = np.stack((x.min(axis=1), y.min(axis=1), x.max(axis=1), y.max(axis=1)), axis=1)
        return boxes, np.stack((x, y), axis=2)


# ----------------------------------------------------------------------------- labels
@dataclass
class Object3d:
    """one line of a KITTI label / result file (lib/utils/object3d.py:12-33)"""
    cls_type: str
    truncation: float
    occlusion: float
    alpha: float
    box2d: np.ndarray
    h: float
    w: float
    l: float
    pos: np.ndarray
    ry: float
    score: float = -1.0

    @classmethod
    def from_line(cls, line: str) -> "Object3d":
        t = line.strip().split(" ")
        return cls(t[0], float(t[1]), float(t[2]), float(t[3]),
                   np.array([float(v) for v in t[4:8]], dtype=np.float32), float(t[8]), float(t[9]), float(t[10]),
                   np.array([float(v) for v in t[11:14]], dtype=np.float32), float(t[14]),
                   float(t[15]) if len(t) == 16 else -1.0)

    @property
    def level(self) -> int:
        """1 easy / 2 moderate / 3 hard / 4 unknown (object3d.py:35-49)"""
        height = float(self.box2d[3]) - float(self.box2d[1]) + 1
        if height >= 40 and self.truncation <= 0.15 and self.occlusion <= 0:
            return 1
        if height >= 25 and self.truncation <= 0.3 and self.occlusion <= 1:
            return 2
        if height >= 25 and self.truncation <= 0.5 and self.occlusion <= 2:
            return 3
        return 4

    def box3d(self) -> np.ndarray:
        """(7,) [x, y, z, h, w, l, ry], the layout every op of the path takes (kitti_utils.py objs_to_boxes3d)"""
        return np.array([self.pos[0], self.pos[1], self.pos[2], self.h, self.w, self.l, self.ry], dtype=np.float32)


def read_label_file(path: str):
    with open(path) as f:
        return [Object3d.from_line(line) for line in f.readlines() if line.strip()]


# ----------------------------------------------------------------------------- filter + sampler
def valid_point_mask(pts_rect, pts_img, pts_rect_depth, img_shape, area_scope=PC_AREA_SCOPE):
    """inside the image, in front of the camera and (PC_REDUCE_BY_RANGE) inside PC_AREA_SCOPE
    (kitti_rcnn_dataset.py:139-160); area_scope=None disables the range part"""
    flag = (pts_img[:, 0] >= 0) & (pts_img[:, 0] < img_shape[1]) & (pts_img[:, 1] >= 0) & (pts_img[:, 1] < img_shape[0])
    flag &= pts_rect_depth >= 0
    if area_scope is not None:
        (x0, x1), (y0, y1), (z0, z1) = area_scope
        x, y, z = pts_rect[:, 0], pts_rect[:, 1], pts_rect[:, 2]
        flag &= (x >= x0) & (x <= x1) & (y >= y0) & (y <= y1) & (z >= z0) & (z <= z1)
    return flag


def sample_point_choice(pts_depth: np.ndarray, npoints: int, rng=np.random) -> np.ndarray:
    """indices of the npoints sampled points (kitti_rcnn_dataset.py:425-443): all points at
    depth >= 40 m are kept and the near ones subsampled when the scan is larger than npoints,
    otherwise the scan is tiled and drawn without replacement.  `rng` needs choice() and shuffle()
    (``numpy.random`` itself, like the reference, or a ``RandomState``)."""
    n = len(pts_depth)
    if npoints < n:
        near = pts_depth < 40.0
        far_idx = np.where(near == 0)[0]
        near_idx = np.where(near == 1)[0]
        if len(far_idx) > npoints:  # the reference dies inside numpy.random.choice here (negative size)
            raise ValueError(f"{len(far_idx)} points beyond 40 m do not fit in npoints={npoints}")
        choice = rng.choice(near_idx, npoints - len(far_idx), replace=False)
        if len(far_idx) > 0:
            choice = np.concatenate((choice, far_idx), axis=0)
        rng.shuffle(choice)
        return choice
    base = np.arange(0, n, dtype=np.int32)
    choice = base
    while npoints > len(choice):
        choice = np.concatenate((choice, base), axis=0)
    choice = rng.choice(choice, npoints, replace=False)
    rng.shuffle(choice)
    return choice


def rpn_input_from_scan(pts_lidar: np.ndarray, calib: Calibration, img_shape, npoints: int = 16384,
                        random_select: bool = True, rng=np.random, area_scope=PC_AREA_SCOPE) -> np.ndarray:
    """``pts_input`` (npoints, 4) = [x, y, z (rect camera frame), intensity - 0.5] of one scan, as
    ``get_rpn_sample`` builds it for inference (kitti_rcnn_dataset.py:399-452)"""
    pts_lidar = pts_lidar[np.argsort(-pts_lidar[:, 2]), :]
    pts_rect = calib.lidar_to_rect(pts_lidar[:, 0:3])
    intensity = pts_lidar[:, 3]
    pts_img, depth = calib.rect_to_img(pts_rect)
    keep = valid_point_mask(pts_rect, pts_img, depth, img_shape, area_scope)
    pts_rect, intensity, depth = pts_rect[keep][:, 0:3], intensity[keep], depth[keep]
    if random_select:
        choice = sample_point_choice(depth, npoints, rng)
        pts_rect, intensity = pts_rect[choice, :], intensity[choice]
    return np.concatenate((pts_rect, (intensity - 0.5).reshape(-1, 1)), axis=1)


# ----------------------------------------------------------------------------- result writer
_HALF_L = np.array([1, 1, -1, -1, 1, 1, -1, -1], dtype=np.float32) * np.float32(0.5)   # corner signs along l
_HALF_W = np.array([1, -1, -1, 1, 1, -1, -1, 1], dtype=np.float32) * np.float32(0.5)   # ... along w
_TOP = np.array([0, 0, 0, 0, 1, 1, 1, 1], dtype=np.float32)                           # corners 4..7 = top face


def boxes3d_to_corners3d(boxes3d: np.ndarray) -> np.ndarray:
    """(N,7) [x,y,z,h,w,l,ry] (y = bottom centre, camera y points down) -> (N,8,3) corners, bottom
    face first, in the corner order of lib/utils/kitti_utils.py:66-101"""
    b = np.asarray(boxes3d, dtype=np.float32).reshape(-1, 7)
    along_l, along_w = b[:, 5:6] * _HALF_L, b[:, 4:5] * _HALF_W
    c, s = np.cos(b[:, 6:7]), np.sin(b[:, 6:7])
    x = b[:, 0:1] + (along_l * c + along_w * s)
    y = b[:, 1:2] - b[:, 3:4] * _TOP
    z = b[:, 2:3] + (along_w * c - along_l * s)
    return np.stack((x, y, z), axis=2).astype(np.float32)


def format_kitti_result(boxes3d: np.ndarray, scores: Sequence[float], calib: Calibration, img_shape,
                        cls_name: str = "Car") -> str:
    """the text of one KITTI result file (tools/eval_auto.py:108-135): 2-D boxes from the projected
    corners clipped to the image, boxes wider/taller than 80 % of the image dropped, observation
    angle alpha from (x, z, ry)"""
    if boxes3d.shape[0] == 0:
        return ""
    img_boxes, _ = calib.corners3d_to_img_boxes(boxes3d_to_corners3d(boxes3d))
    img_boxes[:, 0] = np.clip(img_boxes[:, 0], 0, img_shape[1] - 1)
    img_boxes[:, 1] = np.clip(img_boxes[:, 1], 0, img_shape[0] - 1)
    img_boxes[:, 2] = np.clip(img_boxes[:, 2], 0, img_shape[1] - 1)
    img_boxes[:, 3] = np.clip(img_boxes[:, 3], 0, img_shape[0] - 1)
    ok = ((img_boxes[:, 2] - img_boxes[:, 0]) < img_shape[1] * 0.8) & ((img_boxes[:, 3] - img_boxes[:, 1]) < img_shape[0] * 0.8)
    lines = []
    for k in range(boxes3d.shape[0]):
        if not ok[k]:
            continue
        x, z, ry = boxes3d[k, 0], boxes3d[k, 2], boxes3d[k, 6]
        beta = np.arctan2(z, x)
        alpha = -np.sign(beta) * np.pi / 2 + beta + ry
        lines.append("%s -1 -1 %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f" %
                     (cls_name, alpha, img_boxes[k, 0], img_boxes[k, 1], img_boxes[k, 2], img_boxes[k, 3],
                      boxes3d[k, 3], boxes3d[k, 4], boxes3d[k, 5], boxes3d[k, 0], boxes3d[k, 1], boxes3d[k, 2],
                      boxes3d[k, 6], scores[k]))
    return "".join(line + "\n" for line in lines)


def save_kitti_format(sample_id: int, calib: Calibration, bbox3d: np.ndarray, kitti_output_dir: str, scores,
                      img_shape, cls_name: str = "Car") -> str:
    """reference-named wrapper (eval_auto.py:108): writes '<dir>/%06d.txt' and returns its path"""
    path = os.path.join(kitti_output_dir, "%06d.txt" % sample_id)
    with open(path, "w") as f:
        f.write(format_kitti_result(bbox3d, scores, calib, img_shape, cls_name))
    return path


# ----------------------------------------------------------------------------- directory reader
class KittiScenes:
    """KITTI object-detection directory (kitti_dataset.py:10-26): ``root/ImageSets/<split>.txt``
    lists sample ids, data lives in ``root/{training|testing}/{velodyne,calib,image_2,label_2}``.
    Indexing yields the inference input of one scene."""

    def __init__(self, root_dir: str, split: str = "val", npoints: int = 16384, random_select: bool = True,
                 rng=np.random):
        self.split = split
        self.imageset_dir = os.path.join(root_dir, "testing" if split == "test" else "training")
        with open(os.path.join(root_dir, "ImageSets", split + ".txt")) as f:
            self.sample_id_list = [int(x.strip()) for x in f.readlines() if x.strip()]
        self.npoints, self.random_select, self.rng = npoints, random_select, rng

    def _path(self, sub: str, idx: int, ext: str) -> str:
        return os.path.join(self.imageset_dir, sub, "%06d.%s" % (idx, ext))

    def get_lidar(self, idx: int) -> np.ndarray:
        return read_velodyne_bin(self._path("velodyne", idx, "bin"))

    def get_calib(self, idx: int) -> Calibration:
        return Calibration(self._path("calib", idx, "txt"))

    def get_image_shape(self, idx: int):
        return read_image_shape(self._path("image_2", idx, "png"))

    def get_label(self, idx: int):
        return read_label_file(self._path("label_2", idx, "txt"))

    def __len__(self) -> int:
        return len(self.sample_id_list)

    def __getitem__(self, index: int) -> dict:
        sample_id = self.sample_id_list[index]
        pts_input = rpn_input_from_scan(self.get_lidar(sample_id), self.get_calib(sample_id),
                                        self.get_image_shape(sample_id), self.npoints, self.random_select, self.rng)
        return {"sample_id": sample_id, "random_select": self.random_select, "pts_input": pts_input}


def collate_scenes(samples: Sequence[dict]) -> dict:
    """batch of equal-size scenes -> {'sample_id': (B,), 'pts_input': (B,N,4) float32}"""
    return {"sample_id": np.array([s["sample_id"] for s in samples], dtype=np.int64),
            "pts_input": np.stack([s["pts_input"] for s in samples]).astype(np.float32)}
