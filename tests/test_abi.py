"""CPU-side checks of the drop-in boundary: libws3d_hip.so builds for gfx950, loads
without a GPU, and exports every symbol include/ws3d_ops.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ws3d_ops.h")).read()
    return sorted(set(re.findall(r"WS3D_API\s+[\w\s\*]+?\b(ws3d_\w+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from ws3d_amd import build
    path = build.build()
    assert os.path.exists(path)
    return ctypes.CDLL(path)


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ws3d_ops.h but not exported"


def test_python_binding_covers_header():
    from ws3d_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    handle = _lib.load()
    assert handle.ws3d_abi_version() == _lib.ABI_VERSION == 6
    assert handle.ws3d_nms_workspace_bytes(9000) >= 9000 * 141 * 8
    assert handle.ws3d_nms_workspace_bytes(0) > 0


def test_only_ws3d_symbols_are_public():
    import subprocess
    from ws3d_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    public = [l.split()[-1] for l in out.splitlines() if len(l.split()) >= 3]          # every defined dynamic symbol, functions and objects
    stray = [s for s in public if not s.startswith("ws3d_")]
    assert public and not stray, stray      # (csrc/exports.map: the compiler's per-TU __hip_cuid_* markers stay local)


def test_invalid_arguments_return_codes(lib):
    """argument validation happens before any HIP call, so it is testable without a GPU"""
    lib.ws3d_last_error.restype = ctypes.c_char_p
    rc = lib.ws3d_furthest_point_sampling(1, 0, 4, None, None, None, None)
    assert rc == -1 and b"invalid" in lib.ws3d_last_error()
    assert lib.ws3d_ball_query(1, 16, 4, ctypes.c_float(0.5), 0, None, None, None, None, None) == -1
    assert lib.ws3d_roipool3d(1, 16, 4, 3, 0, None, None, None, None, None, None, None) == -1
    assert lib.ws3d_nms(8, None, ctypes.c_float(0.5), 0, 0, None, ctypes.c_size_t(0), None, None, None) == -1
    # zero-sized problems are no-ops that succeed (sampling_gpu.cu:101 "if (m <= 0) return")
    buf = (ctypes.c_float * 48)()
    assert lib.ws3d_furthest_point_sampling(0, 16, 4, buf, None, buf, None) == 0


def test_product_path_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under ws3d_amd/ may reference it"""
    pkg = os.path.join(ROOT, "ws3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f
                assert "libws3d_oracle" not in txt, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ws3d_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.Ws3dError, match="no CPU fallback"):
        _lib.load()
