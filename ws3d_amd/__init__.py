"""ws3d_amd -- MI355X-native (gfx950) PointNet++ / roipool3d / iou3d ops for WS3D Stage-1.

The package holds only what the hot path needs: ``csrc/`` (hand-written HIP kernels
behind the C ABI declared in ``include/ws3d_ops.h``) and the Python mirror of the
reference's operator interface.  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"
