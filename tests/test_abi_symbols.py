"""CPU: the C-ABI shared library loads and exports every symbol include/mpcgpu.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from helpers import ROOT, abi

HEADER = os.path.join(ROOT, "include", "mpcgpu.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mpc_[a-z_]+)\s*\(", txt)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ("mpc_create", "mpc_destroy", "mpc_set_bounds", "mpc_solve_batch", "mpc_solve_batch_dev", "mpc_last_error"):
        assert s in syms
    assert sorted(abi.EXPORTS) == syms


def test_library_exports_every_declared_symbol():
    assert os.path.exists(abi.LIB_PATH), "run `python __graft_entry__.py build` first"
    L = C.CDLL(abi.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(L, s), f"{s} missing from libmpcgpu.so"
    L.mpc_abi_version.restype = C.c_int
    assert L.mpc_abi_version() == 1


def test_desc_struct_matches_header_defaults():
    L = abi.load_library()
    d = abi.MpcProblemDesc()
    L.mpc_default_desc(C.byref(d), 30, 6)
    assert (d.N, d.nx, d.nu, d.max_iter, d.obst_mult) == (30, 6, 2, 100, 3)
    assert d.wheelbase == 2.5789128 and d.friction_div == 2.578 and d.ego_offset == 0.75 and d.tol == 1e-8
    assert list(d.Q)[:6] == [2.3, 2.3, 500.0, 0.1, 10.0, 0.0] and list(d.R) == [2.0, 0.2]
    assert list(d.obstacle) == [-100.0, 0.0] * 3


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(abi.MpcLibraryError):
        abi.load_library(str(tmp_path / "nope.so"))
