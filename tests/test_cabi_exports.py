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


def test_ctypes_structs_mirror_the_header_layout(tmp_path):
    """Every argument struct of include/wmd.h against its ctypes mirror: size and the offset of every field, as laid out
    by the C compiler (gcc on the header itself) -- a silent layout drift would corrupt arguments, not raise."""
    import re
    import subprocess
    import sys

    pairs = {"wmd_conv_args": _lib.ConvArgs, "wmd_conv_dgrad_args": _lib.ConvDgradArgs, "wmd_conv_wgrad_args": _lib.ConvWgradArgs,
             "wmd_dwconv_args": _lib.DwConvArgs, "wmd_head_args": _lib.HeadArgs, "wmd_head_fused_args": _lib.HeadFusedArgs,
             "wmd_head_shiftsum_args": _lib.HeadShiftsumArgs, "wmd_head_level_args": _lib.HeadLevelArgs,
             "wmd_head_bwd_head": _lib.HeadBwdHead, "wmd_head3x3_bwd_args": _lib.Head3x3BwdArgs, "wmd_head1x1_bwd_args": _lib.Head1x1BwdArgs,
             "wmd_pack_item": _lib.PackItem, "wmd_dilate_spec": _lib.DilateSpec, "wmd_level_spec": _lib.LevelSpec, "wmd_mask_level_args": _lib.MaskLevelArgs, "wmd_compact_spec": _lib.CompactSpec, "wmd_sparse_conv_args": _lib.SparseConvArgs,
             "wmd_eval_kitti_args": _lib.EvalKittiArgs, "wmd_warp_args": _lib.WarpArgs}
    header = os.path.join(ROOT, "include", "wmd.h")
    declared = set(re.findall(r"^\} (wmd_\w+);", open(header).read(), flags=re.M)) - {"wmd_status"}
    assert declared == set(pairs), declared ^ set(pairs)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "wmd.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append('printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", exe, str(src)])
    got = {}
    for ln in subprocess.check_output([exe]).decode().split("\n"):
        if ln:
            c, f, v = ln.split()
            got[(c, f)] = int(v)
    for cname, cls in pairs.items():
        assert got[(cname, "sizeof")] == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, "%s.%s" % (cname, fname)


def test_ctypes_signatures_match_the_header_prototypes():
    """Return type and every parameter of every prototype in include/wmd.h against _lib.SIGNATURES: count and kind
    (pointer / int / float / size_t).  A float passed where the C side reads an int would not raise anywhere."""
    src = open(os.path.join(ROOT, "include", "wmd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+;", "", src, flags=re.S)
    protos = re.findall(r"^\s*([\w ]+?[\s\*]+)(wmd_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.M | re.S)
    assert len(protos) == len(_lib.SIGNATURES), (len(protos), len(_lib.SIGNATURES))

    def kind_c(decl):
        decl = decl.strip()
        if decl == "void" or not decl:
            return None
        if "*" in decl:
            return "ptr"
        base = decl.replace("const", " ").split()
        ty = " ".join(base[:-1]) if len(base) > 1 else base[0]
        return {"int": "int", "float": "float", "double": "double", "size_t": "size_t", "unsigned": "int"}[ty]

    def kind_py(t):
        if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
            return "ptr"
        return {C.c_int: "int", C.c_long: "long", C.c_float: "float", C.c_double: "double", C.c_size_t: "size_t"}[t]

    for ret, name, params in protos:
        restype, argtypes = _lib.SIGNATURES[name]
        want = [k for k in (kind_c(p) for p in params.split(",")) if k]
        got = [kind_py(t) for t in argtypes]
        assert got == want, "%s: header %s, ctypes %s" % (name, want, got)
        r = ret.strip()
        want_r = "ptr" if "*" in r else {"int": "int", "long": "long", "size_t": "size_t", "double": "double", "float": "float", "void": None}[r.replace("const", "").strip()]
        got_r = None if restype is None else kind_py(restype)
        assert got_r == want_r, "%s returns %s, ctypes says %s" % (name, want_r, got_r)
