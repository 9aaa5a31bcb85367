"""CPU tests of the drop-in boundary: libtpspp_hip.so loads without a GPU, exports every symbol that
include/tpspp.h declares, the ctypes binding agrees with the header on arity, and argument errors
come back as codes + messages (no compute is launched here)."""
import ctypes
import os
import re

import pytest

from tps_pp_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tpspp.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(tpspp_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


@pytest.fixture(scope="module")
def lib():
    build.build()           # hipcc cross-compiles for gfx950 without a GPU
    return _lib.lib()


def test_header_declares_what_the_binding_binds(lib):
    decl = header_functions()
    assert decl, "no declarations parsed from include/tpspp.h"
    assert set(decl) == set(_lib.exported_symbols())
    for name, nargs in decl.items():
        fn = getattr(lib, name)                     # raises AttributeError if not exported
        assert len(fn.argtypes) == nargs, f"{name}: header has {nargs} parameters, binding {len(fn.argtypes)}"


def test_every_symbol_is_exported_by_the_shared_object():
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(raw, name), f"{name} missing from libtpspp_hip.so"


def test_abi_version_and_error_reporting(lib):
    assert lib.tpspp_abi_version() == _lib.ABI_VERSION
    rc = lib.tpspp_solve_T(None, None, 1, 20, None, None)
    assert rc == -22 and b"null pointer" in lib.tpspp_last_error()
    rc = lib.tpspp_warp_fwd(1, 3, 32, 100, None, 0, 0, 0, 1, None, 1, 1, 23, None, None, 0, 4, 70, 32,
                            100, 1, None, None, None, None)
    assert rc == -22 and b"F" in lib.tpspp_last_error()          # F + 3 > 64
    rc = lib.tpspp_warp_set_tuning(0, 100, 0, 0)
    assert rc == -22
    assert lib.tpspp_warp_set_tuning(0, 0, 0, 0) == 0
    with pytest.raises(_lib.TpsppError):
        _lib.check(-22, "demo")


def test_mirror_symmetry_helper_runs_on_host_memory(lib):
    import numpy as np
    from tps_pp_amd import constants, ops
    k = constants.classic(20, (32, 100))
    assert ops.table_mirror_symmetry(k["P_hat"], (32, 100), 20) == 1
    assert ops.table_mirror_symmetry(k["P_hat"][:, ::-1].copy(), (32, 100), 20) == 0
    assert ops.table_mirror_symmetry(constants.classic(20, (31, 100))["P_hat"], (31, 100), 20) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.TpsppError, match="no CPU or PyTorch fallback"):
        _lib.lib()
