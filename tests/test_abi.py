"""The C-ABI shared library loads without a GPU and exports exactly the symbols include/ctrlv_hip.h declares."""
import os
import re

import __graft_entry__ as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ctrlv_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(ctrlv_\w+)\s*\(", src)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    g.build()
    from ctrlv_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 16
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and header out of sync"
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.ctrlv_abi_version() == _lib.ABI_VERSION
    assert _lib.build_id(lib) == _lib.source_build_id() == g.source_build_id()


def test_host_side_argument_errors_do_not_need_a_gpu():
    """Bad descriptors are rejected by host validation with the reference's ValueError convention."""
    import ctypes

    import pytest
    from ctrlv_amd import _lib
    lib = _lib.load()
    d = _lib.GemmDesc()
    with pytest.raises(ValueError, match="non-null"):
        _lib.check(lib.ctrlv_gemm(ctypes.byref(d), None), "ctrlv_gemm")
    assert lib.ctrlv_groupnorm_chunks(50, 9216, 320, 1) > 0
    with pytest.raises(ValueError):
        _lib.check(lib.ctrlv_groupnorm_chunks(50, 9216, 330, 1), "ctrlv_groupnorm_chunks")
    with pytest.raises(ValueError, match="multiple of 64"):
        _lib.check(lib.ctrlv_attention_spatial(ctypes.c_void_p(8), ctypes.c_void_p(8), 1, 16, 96, None), "attention")
    with pytest.raises(ValueError, match="frames"):
        _lib.check(lib.ctrlv_attention_temporal(ctypes.c_void_p(8), ctypes.c_void_p(8), 1, 33, 16, 64, None), "attn")


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "ctrlv_amd")):
        for f in fs:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(import|from)\s+(oracle|ctrlv_ref)", s, flags=re.M):
                    bad.append(f)
    assert not bad, bad
