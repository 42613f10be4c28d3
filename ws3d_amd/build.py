"""Build libws3d_hip.so (the C-ABI HIP library, include/ws3d_ops.h) for gfx950.

Plain ``hipcc`` per translation unit (parallel), then one link -- no hipify, no
CUDAExtension, no CMake.  The shared object is written IN-TREE
(``ws3d_amd/libws3d_hip.so``) so it travels to the GPU box with the snapshot.

    python -m ws3d_amd.build [--force] [--verbose]

Flags that are part of the numerical contract:
  -ffp-contract=off   the compiler never fuses a*b+c; every contractual FMA is spelled
                      with __builtin_fmaf in the sources
  -fno-fast-math      IEEE semantics for min/max/compare/division
(hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt keeps fp32 '/' and sqrtf
correctly rounded; fp32 denormals are preserved by default on gfx9.)
"""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libws3d_hip.so")
DIST_MODES = (0, 1, 2)   # squared-distance conventions (csrc/common.h); 0 is the product default


def lib_path(dist_mode: int = 0) -> str:
    return LIB if dist_mode == 0 else os.path.join(HERE, "libws3d_hip_dm%d.so" % dist_mode)
ARCH = "gfx950"
SOURCES = ["core.hip", "fps.hip", "fps_v3.hip", "fps_bucket.hip", "fps_nested.hip", "ballquery_group.hip", "interpolate.hip", "roipool3d.hip", "iou3d.hip", "proposals.hip", "scatter_det.hip", "sa_mlp.hip", "bn_relu.hip", "gemm_pool.hip", "conv_wgrad.hip", "chain_mlp.hip"]
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
            "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin)")


def _deps(src: str):
    return [src, os.path.join(CSRC, "common.h"), os.path.join(CSRC, "binning.h"), os.path.join(CSRC, "compact_pool.h"), os.path.join(HERE, "..", "include", "ws3d_ops.h"),
            os.path.join(CSRC, "exports.map"), os.path.abspath(__file__)]


def _compile(src: str, force: bool, verbose: bool, dist_mode: int = 0) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ if dist_mode == 0 else OBJ + "_dm%d" % dist_mode, os.path.splitext(src)[0] + ".o")
    if not force and os.path.exists(obj) and all(
            os.path.getmtime(obj) >= os.path.getmtime(d) for d in _deps(path)):
        return obj
    # WS3D_EXTRA_DEFS: extra -D switches for A/B builds of the kernels' tuning macros (scripts/r06/*.sh; pair with --force)
    cmd = [hipcc(), f"--offload-arch={ARCH}", *CXXFLAGS, f"-DWS3D_DIST_MODE={dist_mode}", *os.environ.get("WS3D_EXTRA_DEFS", "").split(), "-c", path, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False, dist_mode: int = 0) -> str:
    """dist_mode != 0 builds the same library under another squared-distance convention (csrc/common.h) into
    libws3d_hip_dm<N>.so; ``WS3D_DIST_MODE=N`` in the environment makes ws3d_amd load it."""
    LIB = lib_path(dist_mode)
    os.makedirs(OBJ if dist_mode == 0 else OBJ + "_dm%d" % dist_mode, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, verbose, dist_mode), SOURCES))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        # export only the extern "C" ws3d_* symbols (-fvisibility=hidden + default below)
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


def rebuild_one(src: str, verbose: bool = False) -> str:
    """recompile ONE translation unit (forced; with WS3D_EXTRA_DEFS if set) and relink the default library from the objects that
    are there -- the A/B loops of scripts/r06/*.sh, where a full mtime-driven rebuild per variant would cost minutes"""
    _compile(src, True, verbose)
    objs = [os.path.join(OBJ, os.path.splitext(s_)[0] + ".o") for s_ in SOURCES]
    missing = [o for o in objs if not os.path.exists(o)]
    if missing:
        return build(False, verbose)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--dist-mode", type=int, default=0, choices=DIST_MODES)
    ap.add_argument("--all-dist-modes", action="store_true")
    ap.add_argument("--only", default=None, help="recompile this one source (forced) and relink")
    a = ap.parse_args()
    if a.only:
        print(rebuild_one(a.only, a.verbose))
        sys.exit(0)
    for dm in (DIST_MODES if a.all_dist_modes else (a.dist_mode,)):
        print(build(a.force, a.verbose, dm))
    sys.exit(0)
