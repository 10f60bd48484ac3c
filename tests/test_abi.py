"""The C-ABI shared library loads without a GPU and exports exactly the symbols include/ctrlv_hip.h declares."""
import os
import re

import __graft_entry__ as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ctrlv_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|size_t)\s+(ctrlv_\w+)\s*\(", src)))


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


def test_plan_graph_construction_needs_no_gpu():
    """ctrlv_plan_create builds the module graph on the host: residual bookkeeping and config validation."""
    import ctypes

    import pytest
    from ctrlv_amd import _lib
    from ctrlv_amd.plan import config_struct
    lib = _lib.load()
    svd = dict(in_channels=8, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
               down_block_types=("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",),
               up_block_types=("UpBlockSpatioTemporal",) + ("CrossAttnUpBlockSpatioTemporal",) * 3,
               num_attention_heads=(5, 10, 20, 20), cross_attention_dim=1024, addition_time_embed_dim=256,
               projection_class_embeddings_input_dim=768, num_frames=25)
    for kind in ("unet", "controlnet"):
        h = ctypes.c_void_p()
        c = config_struct(kind, svd)
        assert lib.ctrlv_plan_create(ctypes.byref(c), 0, ctypes.byref(h)) == 0
        assert lib.ctrlv_plan_num_down_residuals(h) == 12
        shapes = []
        for i in range(13):
            m, ch = ctypes.c_int64(), ctypes.c_int32()
            assert lib.ctrlv_plan_residual_shape(h, i, 2, 25, 72, 128, ctypes.byref(m), ctypes.byref(ch)) == 0
            shapes.append((m.value // 50, ch.value))
        # 3x(320, 72x128), (320, 36x64), 2x(640, 36x64), (640, 18x32), 2x(1280, 18x32), 3x(1280, 9x16) + mid
        assert shapes == [(9216, 320)] * 3 + [(2304, 320)] + [(2304, 640)] * 2 + [(576, 640)] + [(576, 1280)] * 2 + \
            [(144, 1280)] * 4
        assert lib.ctrlv_plan_workspace_bytes(h, 2, 25, 72, 128) == 0 and "not loaded" in _lib.last_error()
        assert lib.ctrlv_plan_destroy(h) == 0
    bad = config_struct("unet", dict(svd, num_attention_heads=(5, 10, 20, 16)))
    h = ctypes.c_void_p()
    with pytest.raises(ValueError, match="head_dim 64"):
        _lib.check(lib.ctrlv_plan_create(ctypes.byref(bad), 0, ctypes.byref(h)), "ctrlv_plan_create")


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "ctrlv_amd")):
        for f in fs:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(import|from)\s+(oracle|ctrlv_ref)", s, flags=re.M):
                    bad.append(f)
    assert not bad, bad
