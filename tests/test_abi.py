"""CPU: the C-ABI library loads, exports every symbol include/nvq.h declares, and the ctypes
signatures in nerve_cl/_nvq.py agree with the header prototypes (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from nerve_cl import _nvq

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "nvq.h")


def _prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"typedef struct.*?\}\s*\w+;", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    src = src.replace('extern "C" {', "")
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(nvq_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip().split("\n")[-1].strip(), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (ret, params)
    return protos


def _ctype_of(decl: str):
    decl = decl.strip()
    if "nvq_conv_desc" in decl:
        return C.POINTER(_nvq.ConvDesc)
    if "nvq_wgrad_desc" in decl:
        return C.POINTER(_nvq.WgradDesc)
    if "nvq_wgrad_reduce_job" in decl:
        return C.POINTER(_nvq.WgradReduceJob)
    if "*" in decl:
        if re.match(r"const int\s*\*\s*\w*_host$", decl):
            return C.POINTER(C.c_int)
        if decl.startswith("const char"):
            return C.c_char_p
        return C.c_void_p
    base = re.sub(r"\s+\w+$", "", decl) if " " in decl else decl
    return {"int": C.c_int, "long": C.c_long, "float": C.c_float, "size_t": C.c_size_t}[base.strip()]


def test_header_parsed():
    protos = _prototypes()
    assert len(protos) >= 38
    assert "nvq_conv_forward" in protos and "nvq_ewc_penalty_grad" in protos


def test_signatures_match_header():
    protos = _prototypes()
    assert set(protos) == set(_nvq.SIGNATURES), set(protos) ^ set(_nvq.SIGNATURES)
    for name, (ret, params) in protos.items():
        res, args = _nvq.SIGNATURES[name]
        want = [_ctype_of(p) for p in params]
        assert len(want) == len(args), (name, len(want), len(args))
        for i, (a, b) in enumerate(zip(want, args)):
            assert a == b, f"{name} arg {i} ({params[i]}): header {a} vs binding {b}"
        want_res = _ctype_of(ret + " x") if "*" not in ret else C.c_char_p
        assert res == want_res, (name, res, want_res)


def test_library_loads_and_exports_everything():
    if not os.path.exists(_nvq.LIB_PATH):
        pytest.fail(f"{_nvq.LIB_PATH} missing: run __graft_entry__.build()")
    lib = _nvq.lib()   # also asserts the struct sizes
    for name in _prototypes():
        assert hasattr(lib, name), name
    assert lib.nvq_version() >= 100
    assert lib.nvq_wgrad_workspace_bytes() > 0
    assert lib.nvq_conv_pack_floats(32, 64, 3, 0) == 1 * 4 * 9 * 4 * 32 * 4
    assert lib.nvq_conv_pack_floats(32, 64, 3, 1) == 1 * 2 * 10240 // 2   # bf16 slabs are padded to 2048 halfs
    assert lib.nvq_tsum_blocks(540, 960) >= 1


def test_cpu_tensor_is_refused_loudly():
    import torch
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _nvq.ptr(torch.zeros(4))
