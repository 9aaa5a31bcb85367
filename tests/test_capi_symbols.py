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
    for m in re.finditer(r"\b(?:int|size_t|void|const char\*)\s+(tpspp_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
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


def test_prepared_table_geometry_planner(lib):
    """Host-side planning of the packed table (no GPU): which output geometries have a prepared form and how large it is.
    F + 3 transposed rows of Ho * Wo floats, then the packed copy: ceil(threads / 64) wavefronts x QP x 6 pieces x 64 lanes x
    4 floats, with QP = the smallest divisor of the quadrant's row groups that leaves <= 13 wavefronts, else the largest
    <= 4 (tpspp_warp_geo.hip: geo_qp); geometries with QP >= 3 carry a third section, the same packing with QP = 1 (every
    quadrant pixel: the span-staging kernel's copy, tpspp_warp_span.h).  (Round 5: W = 160 is tiled by 16 x 2 pixel blocks,
    5 column groups -- 32 x 1 blocks would be 3 groups = 96 columns for a half-row of 80.)"""
    K = 23
    want = {(32, 100): (1, 13 * 2 * 32), (32, 128): (2, 2 * 8 * 32), (48, 160): (3, 5 * 4 * 32), (32, 64): (1, 1 * 16 * 32),
            (32, 160): (2, 5 * 4 * 32), (64, 256): (4, 4 * 8 * 32), (64, 200): (4, 13 * 2 * 32), (16, 64): (1, 1 * 8 * 32)}
    for (Ho, Wo), (qp, nthr) in want.items():
        nw = (nthr + 63) // 64
        span = ((nthr * qp + 63) // 64) * 6 * 64 * 4 if qp >= 3 else 0
        assert lib.tpspp_prepared_table_floats(Ho, Wo, 20) == K * Ho * Wo + nw * qp * 6 * 64 * 4 + span, (Ho, Wo)
    for Ho, Wo in ((31, 100), (32, 99), (32, 102), (24, 100), (0, 100)):
        assert lib.tpspp_prepared_table_floats(Ho, Wo, 20) == 0, (Ho, Wo)   # Ho % 16 != 0 or Wo % 4 != 0: no packed form
    assert lib.tpspp_prepared_table_floats(32, 100, 62) == 0                 # F + 3 > 64
    assert lib.tpspp_warp_set_tuning(0, 0, 7, 9) == 0 and lib.tpspp_warp_set_tuning(0, 0, 10, 0) == -22
    assert lib.tpspp_warp_set_tuning(0, 0, 0, 9) == -22 and lib.tpspp_warp_set_tuning(0, 0, 2, 9) == -22   # 9..15 only with 7 / 9
    assert lib.tpspp_warp_set_tuning(0, 0, 8, 4 | 64 | (100 << 8)) == 0 and lib.tpspp_warp_set_tuning(0, 0, 9, 15) == 0
    assert lib.tpspp_warp_set_tuning(0, 0, 0, 0) == 0
    assert lib.tpspp_warp_bwd_set_accumulator(1) == 0 and lib.tpspp_warp_bwd_set_accumulator(0) == 0
    assert lib.tpspp_warp_bwd_workspace_floats(2, 32, 100) > 2 * 3200 * 2
    assert lib.tpspp_warp_bwd_workspace_floats(0, 32, 100) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.TpsppError, match="no CPU or PyTorch fallback"):
        _lib.lib()


def test_prepared_warp_plan_entry_points(lib):
    """tpspp_warp_plan_create / _run / _run_on / _destroy (ABI 5) without a GPU: an empty batch runs through the stored
    arguments' checks and launches nothing; argument errors come back as codes."""
    vp, ci = ctypes.c_void_p, ctypes.c_int
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, vp).value
    args = (vp(p), ci(3), ci(32), ci(100), vp(0), ci(0), ci(0), ci(0), vp(p), vp(0), vp(p), vp(p), ci(23), vp(0), vp(0), ci(0),
            ci(0), ci(20), ci(32), ci(100), vp(p), vp(0), vp(0), vp(0), vp(0))
    h = vp()
    assert lib.tpspp_warp_plan_create(*args, ctypes.byref(h)) == 0 and h.value
    assert lib.tpspp_warp_plan_run(h) == 0                       # N = 0: the same early return as tpspp_warp_fwd
    assert lib.tpspp_warp_plan_run_on(h, vp(0)) == 0
    lib.tpspp_warp_plan_destroy(h)
    lib.tpspp_warp_plan_destroy(vp(0))                           # no-op
    assert lib.tpspp_warp_plan_run(vp(0)) == -22 and b"NULL plan" in lib.tpspp_last_error()
    assert lib.tpspp_warp_plan_create(*args, None) == -22
    bad = list(args); bad[0] = vp(0)                             # in0 = NULL
    h2 = vp()
    assert lib.tpspp_warp_plan_create(*bad, ctypes.byref(h2)) == -22 and not h2.value
    # a plan whose stored arguments are wrong fails at run time with tpspp_warp_fwd's message
    bad = list(args); bad[17] = ci(70)                           # F + 3 > 64
    assert lib.tpspp_warp_plan_create(*bad, ctypes.byref(h2)) == 0
    assert lib.tpspp_warp_plan_run(h2) == -22 and b"F" in lib.tpspp_last_error()
    lib.tpspp_warp_plan_destroy(h2)
