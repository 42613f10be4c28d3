"""ws3d_amd -- MI355X-native (gfx950) PointNet++ / roipool3d / iou3d ops for WS3D Stage-1.

The package holds only what the hot path needs: ``csrc/`` (hand-written HIP kernels
behind the C ABI declared in ``include/ws3d_ops.h``) and the Python mirror of the
reference's operator interface.  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"


def __getattr__(name):
    """Lazy aliases under the reference's module names (``ws3d_amd.pointnet2_utils`` ...)."""
    import importlib
    alias = {"pointnet2_utils": "pn2_ops", "pointnet2_modules": "pn2_modules", "pytorch_utils": "nn_blocks",
             "iou3d_utils": "iou3d_ops", "roipool3d_utils": "roipool3d_ops", "calibration": "kitti_io",
             "loss_utils": "losses"}
    if name in alias:
        return importlib.import_module("." + alias[name], __name__)
    raise AttributeError(name)
