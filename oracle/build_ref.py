"""Build the ONE natively-buildable piece of the reference into oracle/_ref/.

TEST INFRASTRUCTURE ONLY.  The reference's roipool3d host file
(``lib/utils/roipool3d/src/roipool3d.cpp``) contains a complete CPU
implementation of the RoI-pooling selection semantics (``pt_in_box3d_cpu``
:82-95, ``pts_in_boxes3d_cpu`` :97-124, ``roipool3d_cpu`` :127-195).  It is
compiled here FROM WHERE IT LIES under /root/reference (nothing is copied into
this repository) with g++ through ``torch.utils.cpp_extension`` and the outputs
go only to ``oracle/_ref/`` (git-ignored, but shipped to the GPU box).

No stand-in source is written for the two CUDA launchers the file declares
(``roipool3dLauncher``/``roipool3dLauncher_slow``, defined in the un-buildable
``roipool3d_kernel.cu``): a shared object may carry undefined function symbols,
and the loader below opens it with RTLD_LAZY so they are never resolved because
the GPU entry points (``forward``/``forward_slow``) are never called.
The only build flag is ``-DAT_CHECK=TORCH_CHECK`` (the macro was renamed in
PyTorch >= 1.5; same semantics).

Everything else in the reference's native code is CUDA (needs nvcc + an NVIDIA
device) and is treated as unbuildable here -- see DESIGN.md.
"""
from __future__ import annotations

import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")
REF_SRC = "/root/reference/lib/utils/roipool3d/src/roipool3d.cpp"
MOD_NAME = "roipool3d_ref"
SO_PATH = os.path.join(REF_DIR, MOD_NAME + ".so")


def build(verbose: bool = False) -> str | None:
    """Compile the reference roipool3d.cpp if the reference tree is present.
    Returns the .so path, or None when neither source nor a prebuilt .so exists."""
    if not os.path.exists(REF_SRC):
        return SO_PATH if os.path.exists(SO_PATH) else None
    if os.path.exists(SO_PATH) and os.path.getmtime(SO_PATH) >= os.path.getmtime(REF_SRC):
        return SO_PATH
    os.makedirs(REF_DIR, exist_ok=True)
    from torch.utils import cpp_extension
    try:
      cpp_extension.load(
        name=MOD_NAME,
        sources=[REF_SRC],
        extra_cflags=["-O2", "-DAT_CHECK=TORCH_CHECK", "-w"],
        build_directory=REF_DIR,
        with_cuda=False,
        is_python_module=False,  # build only; we import it ourselves with RTLD_LAZY
        verbose=verbose,
      )
    except OSError:
        # expected: torch's post-build dlopen is RTLD_NOW and trips over the two
        # (never-called) CUDA launcher symbols; load() below uses RTLD_LAZY.
        if not os.path.exists(SO_PATH):
            raise
    return SO_PATH


_mod = None


def load():
    """Import the compiled reference module (or return None if unavailable)."""
    global _mod
    if _mod is not None:
        return _mod
    path = build()
    if path is None or not os.path.exists(path):
        return None
    import torch  # noqa: F401  (libtorch must be loaded before the extension)
    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
    try:
        spec = importlib.util.spec_from_file_location(MOD_NAME, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(old)
    _mod = mod
    return mod


if __name__ == "__main__":
    p = build(verbose=True)
    print("reference roipool3d CPU module:", p)
    m = load()
    print("exports:", [n for n in dir(m) if not n.startswith("_")] if m else None)
