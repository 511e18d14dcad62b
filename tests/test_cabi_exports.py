"""CPU: the C-ABI library loads (dlopen needs no GPU) and exports every symbol include/wmd.h declares;
argument validation paths that never launch a kernel behave as documented."""
import ctypes as C
import os
import re

import pytest

from wavelet_monodepth_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "wmd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wmd_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from wavelet_monodepth_amd import build
        build.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libwmd_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "%s has no ctypes signature in _lib.py" % n
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and wmd.h disagree"


def test_version_and_status_strings(lib):
    assert lib.wmd_version() >= 100
    assert lib.wmd_status_string(0) == b"ok"
    assert lib.wmd_status_string(-2) == b"bad shape"


def test_argument_validation_without_gpu(lib):
    assert lib.wmd_idwt_haar_fwd(None, None, None, None, 1, 4, 4, 1.0, 0, None) == -1
    assert b"null" in lib.wmd_last_error()
    a = _lib.ConvArgs(B=1, H=1, W=5, C1=3, up1=1, C2=0, Cout=4, ksize=3, pad_mode=1, act=0, slope=0.0,
                      x1=1, x2=None, wp=1, bias=None, y=1, workspace=None, workspace_floats=0)
    assert lib.wmd_conv_fwd(C.byref(a), None) == -2          # reflect pad needs H >= 2, like torch
    a.ksize = 5
    a.H = 4
    assert lib.wmd_conv_fwd(C.byref(a), None) == -3          # unsupported kernel size
    assert lib.wmd_conv_packed_weight_floats(32, 96, 3) == 2 * 24 * 9 * 64


def test_ops_refuse_cpu_tensors():
    import torch
    from wavelet_monodepth_amd import ops
    with pytest.raises(_lib.WmdError):
        ops.idwt_haar(torch.zeros(1, 1, 2, 2), torch.zeros(1, 1, 3, 2, 2))
