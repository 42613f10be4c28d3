"""ctypes binding of libws3d_hip.so (C ABI: include/ws3d_ops.h).

There is NO fallback: if the HIP library is missing or fails to load, every op
raises -- the product path never routes through a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# WS3D_DIST_MODE=1|2: the library built under another squared-distance convention (csrc/common.h; built by
# ``python -m ws3d_amd.build --dist-mode N``) -- for users whose CUDA reference outputs match that convention
DIST_MODE = int(os.environ.get("WS3D_DIST_MODE", "0") or 0)
LIB_PATH = os.environ.get("WS3D_HIP_LIB") or os.path.join(  # WS3D_HIP_LIB: A/B builds
    _HERE, "libws3d_hip.so" if DIST_MODE == 0 else "libws3d_hip_dm%d.so" % DIST_MODE)

E_INVALID, E_LAUNCH, E_WORKSPACE, E_UNSUPPORTED = -1, -2, -3, -4      # WS3D_E_* of include/ws3d_ops.h
ABI_VERSION = 6     # = WS3D_ABI_VERSION of include/ws3d_ops.h, the header SIGNATURES below restates

_vp = C.c_void_p
_i = C.c_int
_f = C.c_float
_sz = C.c_size_t

class CompactMlpArgs(C.Structure):
    """ws3d_compact_mlp_args of include/ws3d_ops.h (ws3d_compact_mlp_pair)"""
    _fields_ = [("b", _i), ("n", _i), ("m", _i), ("max_rows", C.c_long), ("o1", _i), ("o2", _i), ("o3", _i), ("pmat", _vp), ("p_stride", _i),
                ("xyz", _vp), ("new_xyz", _vp), ("rowc", _vp), ("rowsrc", _vp), ("total", _vp), ("w1x", _vp), ("b1", _vp), ("relu1", _i),
                ("w2t", _vp), ("b2", _vp), ("relu2", _i), ("w3t", _vp), ("b3", _vp), ("mid", _vp), ("out", _vp), ("out_stride", _i), ("limit", C.c_long)]


# name -> (restype, argtypes); kept in the order of include/ws3d_ops.h
SIGNATURES = {
    "ws3d_abi_version": (_i, []),
    "ws3d_tune": (_i, [_i, _i]),
    "ws3d_dist_mode": (_i, []),
    "ws3d_last_error": (C.c_char_p, []),
    "ws3d_device_info": (_i, [C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i)]),
    "ws3d_furthest_point_sampling": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "ws3d_furthest_point_sampling_gather": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_furthest_point_sampling_nested": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "ws3d_furthest_point_sampling_nested_chain": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_gather_points": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ws3d_gather_points_grad": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ws3d_ball_query": (_i, [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_ball_query_fill": (_i, [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_ball_query_pairs": (_i, [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_ball_query_pairs2": (_i, [_i, _i, _i, _vp, _vp, _vp, C.c_float, _i, _vp, _vp, _vp, _vp, C.c_float, _i, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_sorted_points_bytes": (_sz, [_i, _i]),
    "ws3d_sort_points_x": (_i, [_i, _i, _vp, _vp, _vp]),
    "ws3d_sort_points_xz": (_i, [_i, _i, _vp, _vp, _vp]),
    "ws3d_sort_points_grid": (_i, [_i, _i, _vp, _vp, _vp]),
    "ws3d_sort_points_jobs": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_group_points": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ws3d_group_points_grad": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ws3d_query_and_group": (_i, [_i, _i, _i, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_three_nn": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_three_interpolate": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_three_interpolate_grad": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_bn_workspace_bytes": (C.c_size_t, [_i, _i, C.c_long]),
    "ws3d_bn_relu_train_fwd": (_i, [_i, _i, C.c_long, _vp, _vp, _vp, C.c_float, C.c_float, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _vp, C.c_size_t, _vp]),
    "ws3d_bn_relu_train_bwd": (_i, [_i, _i, C.c_long, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ws3d_conv1x1_wgrad_workspace_bytes": (C.c_size_t, [_i, _i, _i, C.c_long]),
    "ws3d_conv1x1_wgrad": (_i, [_i, _i, _i, C.c_long, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ws3d_gemm_pool": (_i, [C.c_long, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, C.c_long, _vp]),
    "ws3d_pool_nsample": (_i, [C.c_long, _i, _vp, _vp, _vp, _vp]),
    "ws3d_pool_nsample_grad": (_i, [C.c_long, _i, _vp, _vp, _vp, _vp]),
    "ws3d_scatter_workspace_bytes": (C.c_size_t, [_i, _i, _i, C.c_long]),
    "ws3d_group_points_grad_det": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ws3d_three_interpolate_grad_det": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ws3d_query_and_group_nlc": (_i, [_i, _i, _i, _i, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_three_interpolate_nlc": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "ws3d_rowmax_rows": (_i, [C.c_long, _i, _i, _vp, _vp, _i, _vp]),
    "ws3d_sa_mlp3_pool": (_i, [C.c_long, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "ws3d_qinterp_gemm": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "ws3d_mlp2_rows": (_i, [C.c_long, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "ws3d_decode_center_boxes": (_i, [_i, _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "ws3d_topk_sorted": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "ws3d_topk_workspace_bytes": (C.c_size_t, [_i, _i]),
    "ws3d_topk_sorted_ws": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ws3d_topk_sorted_sigmoid": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "ws3d_topk_sorted_sigmoid_ws": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ws3d_three_nn_weights": (_i, [C.c_long, _vp, _vp, _vp]),
    "ws3d_three_nn_w": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_three_nn_wq": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_three_nn_jobs": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_bias_act_inplace": (_i, [_i, _i, C.c_long, _i, _vp, _vp, _vp]),
    "ws3d_rowmax_bias_act": (_i, [_i, _i, C.c_long, _i, _i, _vp, _vp, _vp, _vp]),
    "ws3d_boxes_overlap_bev": (_i, [_i, _vp, _i, _vp, _vp, _vp]),
    "ws3d_boxes_iou_bev": (_i, [_i, _vp, _i, _vp, _vp, _vp]),
    "ws3d_nms_mask": (_i, [_i, _vp, _f, _i, _i, _vp, _vp]),
    "ws3d_nms_workspace_bytes": (_sz, [_i]),
    "ws3d_nms": (_i, [_i, _vp, _f, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "ws3d_nms_batched": (_i, [_i, _i, _vp, _f, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "ws3d_radius_nms_batched": (_i, [_i, _i, _vp, _f, _i, _vp, _sz, _vp, _vp, _vp]),
    "ws3d_gather_gemm": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "ws3d_gather_gemm2": (_i, [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "ws3d_pgather_gemm2": (_i, [_i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, C.c_long, _vp]),
    "ws3d_pgather_rows": (_i, [_i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "ws3d_qinterp_rows": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    "ws3d_compact_pairs": (_i, [C.c_long, _i, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_compact_pairs_count": (_i, [C.c_long, _i, _vp, _vp, _vp]),
    "ws3d_compact_pairs_rows": (_i, [C.c_long, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_pgather_gemm2_compact": (_i, [_i, _i, _i, C.c_long, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, C.c_long, _vp]),
    "ws3d_pgather_gemm3_compact": (_i, [_i, _i, _i, C.c_long, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, C.c_long, _vp]),
    "ws3d_gemm_pool_compact": (_i, [C.c_long, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, C.c_long, _vp]),
    "ws3d_compact_mlp_pair": (_i, [_i, C.POINTER(CompactMlpArgs), C.POINTER(CompactMlpArgs), _vp]),
    "ws3d_chain_mlp3_ticket_ints": (_i, []),
    "ws3d_chain_mlp3_blob_floats": (_sz, [_i]),
    "ws3d_chain_mlp3_pack": (_i, [C.POINTER(CompactMlpArgs), _vp, _vp]),
    "ws3d_chain_mlp3": (_i, [C.POINTER(CompactMlpArgs), C.POINTER(CompactMlpArgs), _vp, _vp, _vp, _i, _vp]),
    "ws3d_sa_mlp3_pool_compact": (_i, [_i, _i, _i, C.c_long, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, C.c_long, _vp]),
    "ws3d_sa_mlp3_pool_lists": (_i, [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, C.c_long, _vp]),
    "ws3d_interp_gemm": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "ws3d_gather_boxes_bev": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_select_proposals": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_decode_gather_boxes_bev": (_i, [_i, _i, _i, _i, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_select_proposals_packed": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_select_proposals_send": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, C.c_long, _vp]),
    "ws3d_split_points_clear": (_i, [C.c_long, _i, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ws3d_roipool3d": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_roipool3d_fill": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ws3d_roipool3d_workspace_bytes": (C.c_size_t, [_i, _i]),
    "ws3d_roipool3d_ws": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, C.c_size_t, _vp]),
    "ws3d_pts_in_boxes3d": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
}


class Ws3dError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    """dlopen the HIP library (once).  Raises Ws3dError when it is absent -- build it with
    ``python -m ws3d_amd.build`` (or ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Ws3dError(
            f"{LIB_PATH} is missing: the MI355X HIP library has not been built "
            "(run `python -m ws3d_amd.build`).  ws3d_amd has no CPU fallback.")
    # torch bundles its own libamdhip64.so.7; it must be in the process BEFORE this library
    # is dlopen'ed so that both share ONE HIP runtime (same SONAME => the loader reuses it).
    # Loading ours first would bind torch to /opt/rocm's runtime and break device discovery.
    import torch  # noqa: F401
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64.so not found
        raise Ws3dError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.ws3d_abi_version() != ABI_VERSION:
        raise Ws3dError(f"{LIB_PATH} exports ABI version {lib.ws3d_abi_version()}, this package binds version {ABI_VERSION}: "
                        "rebuild it (`python -m ws3d_amd.build`)")
    if not os.environ.get("WS3D_HIP_LIB") and lib.ws3d_dist_mode() != DIST_MODE:
        raise Ws3dError(f"{LIB_PATH} was built with WS3D_DIST_MODE={lib.ws3d_dist_mode()}, expected {DIST_MODE}: rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ws3d_last_error()
        raise Ws3dError(f"{what or 'ws3d'} failed (rc={rc}): {msg.decode() if msg else ''}")
