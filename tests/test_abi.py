"""CPU: the C-ABI library loads without a GPU, exports every symbol that
include/machisplin_hip.h declares, and refuses to compute without a device."""
import ctypes
import os
import re

import pytest

import machisplin_amd
from machisplin_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "machisplin_hip.h")).read()
    return sorted(set(re.findall(r"MHS_API[^;(]*?\b(mhs_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype"
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_string():
    lib = _lib.load()
    assert b"gfx950" in lib.mhs_version()
    assert isinstance(lib.mhs_last_error(), bytes)


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(machisplin_amd.MhsError) as ei:
        machisplin_amd.init(0)
    assert ei.value.code == _lib.ERR_NODEVICE
    # compute entry points refuse to run before a successful mhs_init
    h = ctypes.c_void_p()
    import numpy as np
    a = np.zeros(8)
    rc = _lib.load().mhs_tps_from_coef(a.ctypes.data, a.ctypes.data, a.ctypes.data, 4, 0.0,
                                       a.ctypes.data, a.ctypes.data, ctypes.byref(h))
    assert rc == _lib.ERR_NODEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "machisplin_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "oracle/" not in src, f
