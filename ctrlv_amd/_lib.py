"""ctypes binding of libctrlv_hip.so (include/ctrlv_hip.h).  No torch types cross this boundary."""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libctrlv_hip.so")            # element type bf16
LIB_PATH_F16 = os.path.join(_HERE, "lib", "libctrlv_hip_f16.so")     # element type fp16 (same sources, -DCTRLV_ELEM_F16)

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_size_t


class GemmDesc(ctypes.Structure):
    """Mirror of `ctrlv_gemm_desc`."""
    _fields_ = [
        ("A", c_void_p), ("A2", c_void_p), ("W", c_void_p), ("out", c_void_p),
        ("bias", c_void_p), ("R1", c_void_p), ("R2", c_void_p), ("V", c_void_p),
        ("M", c_int), ("N", c_int), ("Cin", c_int), ("taps", c_int),
        ("lda", c_int), ("lda2", c_int), ("c_split", c_int),
        ("mode", c_int),
        ("H", c_int), ("Wd", c_int), ("Ho", c_int), ("Wo", c_int), ("stride", c_int), ("up", c_int),
        ("F", c_int), ("S", c_int),
        ("ldo", c_int), ("n_store", c_int),
        ("ldr1", c_int), ("ldr2", c_int),
        ("s_acc", c_float), ("s1", c_float), ("s2", c_float),
        ("vmode", c_int), ("vdiv", c_int), ("vmod", c_int), ("vS", c_int), ("ldv", c_int),
        ("act", c_int), ("geglu", c_int), ("out_f32", c_int), ("tile", c_int),
        ("ld_raw", c_int), ("raw_out", c_void_p),
        ("n_scale2", c_int), ("s_acc2", c_float),
        ("gn_partials", c_void_p), ("splitk_ws", c_void_p), ("ksplit", c_int), ("w_cin", c_int),
        ("R1_lo", c_void_p), ("R2_lo", c_void_p), ("out_lo", c_void_p),      # split trunk planes (fp16 library)
    ]


CTRLV_MAX_BLOCKS = 8


class ModelConfig(ctypes.Structure):
    """Mirror of `ctrlv_model_config`."""
    _fields_ = [
        ("kind", c_int), ("in_channels", c_int), ("out_channels", c_int), ("n_blocks", c_int),
        ("block_out_channels", c_int * CTRLV_MAX_BLOCKS), ("down_cross_attn", c_int * CTRLV_MAX_BLOCKS),
        ("up_cross_attn", c_int * CTRLV_MAX_BLOCKS), ("layers_per_block", c_int * CTRLV_MAX_BLOCKS),
        ("num_attention_heads", c_int * CTRLV_MAX_BLOCKS),
        ("cross_attention_dim", c_int), ("addition_time_embed_dim", c_int),
        ("projection_class_embeddings_input_dim", c_int), ("num_frames", c_int), ("time_context_order", c_int),
    ]


class ProfileRecord(ctypes.Structure):
    """Mirror of `ctrlv_profile_record`."""
    _fields_ = [("family", c_int), ("M", c_int), ("N", c_int), ("K", c_int), ("flags", c_int), ("ms", c_float),
                ("flops", ctypes.c_double), ("bytes", ctypes.c_double)]


FAMILIES = ("gemm_linear", "gemm_conv3x3", "gemm_conv_temporal", "attention_spatial", "attention_temporal", "groupnorm",
            "layernorm", "residual_add", "gemm_temporal_block")


class TemporalFusedDesc(ctypes.Structure):
    """Mirror of `ctrlv_temporal_fused_desc`."""
    _fields_ = [
        ("x", c_void_p), ("ldx", c_int), ("wf", c_void_p), ("bias", c_void_p),
        ("R1", c_void_p), ("R1_lo", c_void_p), ("ldr1", c_int),
        ("V", c_void_p), ("vmode", c_int), ("vdiv", c_int), ("vmod", c_int), ("vS", c_int), ("ldv", c_int),
        ("out", c_void_p), ("out_lo", c_void_p), ("ldo", c_int),
        ("B", c_int), ("F", c_int), ("S", c_int), ("C", c_int),
        ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_eps", c_float),
    ]


class TensorDesc(ctypes.Structure):
    """Mirror of `ctrlv_tensor_desc`."""
    _fields_ = [("name", ctypes.c_char_p), ("data", c_void_p), ("dtype", c_int), ("on_device", c_int),
                ("numel", ctypes.c_int64)]


# name -> (restype, argtypes); lists every symbol include/ctrlv_hip.h declares (tests/test_abi.py checks this)
SIGNATURES = {
    "ctrlv_abi_version": (c_int, []),
    "ctrlv_elem_dtype": (c_int, []),
    "ctrlv_build_id": (c_int, [ctypes.c_char_p, c_size_t]),
    "ctrlv_last_error": (c_int, [ctypes.c_char_p, c_size_t]),
    "ctrlv_gemm": (c_int, [ctypes.POINTER(GemmDesc), c_void_p]),
    "ctrlv_gemm_gn_partials_serves": (c_int, [ctypes.POINTER(GemmDesc)]),
    "ctrlv_gemm_splitk_ws_bytes": (c_size_t, [ctypes.POINTER(GemmDesc)]),
    "ctrlv_groupnorm_from_partials": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                              c_int, c_void_p, c_void_p]),
    "ctrlv_groupnorm_from_partials_split": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                              c_int, c_void_p, c_void_p]),
    "ctrlv_groupnorm_chunks": (c_int, [c_int, c_int, c_int, c_int]),
    "ctrlv_groupnorm_stats_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                            c_void_p, c_void_p]),
    "ctrlv_groupnorm_apply_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                            c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "ctrlv_layernorm_split": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int,
                                      c_int, c_void_p, c_void_p]),
    "ctrlv_axpby_split": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ctrlv_plan_set_trunk_mode": (c_int, [c_void_p, c_int]),
    "ctrlv_groupnorm_stats": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                      c_void_p]),
    "ctrlv_groupnorm_apply": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                      c_void_p, c_int, c_void_p, c_void_p]),
    "ctrlv_layernorm": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int,
                                c_void_p, c_void_p]),
    "ctrlv_softmax_rows": (c_int, [c_void_p, c_int, c_int, ctypes.c_long, c_void_p, ctypes.c_long, c_void_p]),
    "ctrlv_ff_fused_w1f_bytes": (c_int, []),
    "ctrlv_ff_fused_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ctrlv_ff_fused_serves": (c_int, [ctypes.POINTER(GemmDesc), c_int]),
    "ctrlv_ff_fused": (c_int, [c_void_p, c_int, c_void_p, c_void_p, ctypes.POINTER(GemmDesc), c_void_p]),
    "ctrlv_ff_fused_ln": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p,
                                  c_void_p, ctypes.POINTER(GemmDesc), c_void_p]),
    "ctrlv_temporal_fused_weight_bytes": (c_size_t, []),
    "ctrlv_temporal_fused_pack": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "ctrlv_temporal_fused": (c_int, [ctypes.POINTER(TemporalFusedDesc), c_void_p]),
    "ctrlv_temporal_fused_serves": (c_int, [ctypes.POINTER(TemporalFusedDesc)]),
    "ctrlv_attention_spatial": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ctrlv_attention_spatial_prescaled": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ctrlv_attention_temporal": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ctrlv_attention_spatial_lse": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ctrlv_attention_bwd_scratch_floats": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "ctrlv_attention_spatial_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                            c_void_p]),
    "ctrlv_attention_temporal_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ctrlv_nchw_to_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "ctrlv_rows_to_nchw": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "ctrlv_time_conv_rows_to_nchw": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                             c_void_p]),
    "ctrlv_im2col3x3": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "ctrlv_axpby": (c_int, [c_void_p, c_void_p, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "ctrlv_timestep_embedding": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ctrlv_silu": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "ctrlv_cfg_euler_step": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float,
                                     c_float, c_void_p, c_void_p]),
    "ctrlv_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "ctrlv_gemm_wgrad_scratch_bytes": (c_size_t, [ctypes.POINTER(GemmDesc)]),
    "ctrlv_gemm_wgrad": (c_int, [ctypes.POINTER(GemmDesc), c_void_p, c_int, c_void_p, c_void_p, c_float, c_int, c_void_p, c_size_t,
                                 c_void_p]),
    "ctrlv_colsum_scratch_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ctrlv_colsum": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p]),
    "ctrlv_dot_diff": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_void_p, c_void_p, c_void_p]),
    "ctrlv_groupnorm_bwd_scratch_floats": (c_int, [c_int, c_int, c_int, c_int]),
    "ctrlv_groupnorm_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ctrlv_groupnorm_bwd_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ctrlv_layernorm_bwd_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_void_p, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ctrlv_layernorm_bwd_scratch_floats": (ctypes.c_size_t, [c_int, c_int]),
    "ctrlv_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_void_p, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ctrlv_geglu_bwd": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_void_p]),
    "ctrlv_plan_create": (c_int, [ctypes.POINTER(ModelConfig), c_int, ctypes.POINTER(c_void_p)]),
    "ctrlv_plan_load_weights": (c_int, [c_void_p, ctypes.POINTER(TensorDesc), c_size_t]),
    "ctrlv_plan_set_time_context_order": (c_int, [c_void_p, c_int]),
    "ctrlv_plan_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int, c_int]),
    "ctrlv_plan_num_down_residuals": (c_int, [c_void_p]),
    "ctrlv_plan_residual_shape": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_int64),
                                          ctypes.POINTER(c_int)]),
    "ctrlv_unet_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                   ctypes.POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                   c_void_p, c_size_t, c_void_p]),
    "ctrlv_unet_encoder_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                           ctypes.POINTER(c_void_p), c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                           c_size_t, c_void_p]),
    "ctrlv_controlnet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                         c_float, ctypes.POINTER(c_void_p), c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                         c_size_t, c_void_p]),
    "ctrlv_plan_destroy": (c_int, [c_void_p]),
    "ctrlv_plan_profile": (c_int, [c_void_p, c_int]),
    "ctrlv_plan_profile_read": (c_int, [c_void_p, ctypes.POINTER(ProfileRecord), c_int]),
}

_libs = {}                # element dtype code (2 bf16 / 1 fp16) -> CDLL
_tls = threading.local()  # .failed = the library whose call returned a negative status last (this thread)
# c_int-returning entry points whose value is NOT a status code
_NO_STATUS = {"ctrlv_abi_version", "ctrlv_elem_dtype", "ctrlv_build_id", "ctrlv_last_error", "ctrlv_ff_fused_w1f_bytes",
              "ctrlv_gemm_gn_partials_serves", "ctrlv_ff_fused_serves", "ctrlv_plan_num_down_residuals",
              "ctrlv_temporal_fused_serves"}


def _status_recorder(lib, fn):
    def call(*args):
        rc = fn(*args)
        _tls.failed = lib if rc < 0 else None      # (a later success clears the mark: it never names an OLD failure)
        return rc
    call.__name__ = getattr(fn, "__name__", "ctrlv_fn")
    return call
ABI_VERSION = 20


class CtrlvHipError(RuntimeError):
    pass


def source_build_id():
    """sha256 prefix over ctrlv_amd/csrc/*.{hip,h} and include/*.h (sorted by name): the id `__graft_entry__.build()`
    stamps into the library.  None when the sources are not next to the package (an installed binary-only copy)."""
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    inc = os.path.join(os.path.dirname(_HERE), "include")
    if not os.path.isdir(csrc) or not os.path.isdir(inc):
        return None
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h"))]
    files += [os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(".h")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build_id(lib=None):
    buf = ctypes.create_string_buffer(128)
    (lib or load()).ctrlv_build_id(buf, 128)
    return buf.value.decode()


def elem_code(dtype=None):
    """Element-type code of the library that serves activations of `dtype` (a torch dtype, a code, or None = bf16)."""
    if dtype is None or dtype == 2:
        return 2
    if dtype == 1:
        return 1
    name = str(dtype)
    if name == "torch.float16":
        return 1
    if name == "torch.bfloat16":
        return 2
    raise CtrlvHipError(f"ctrlv_amd: activations are stored as bf16 or fp16, not {dtype}")


def load(dtype=None):
    """Load the library whose ELEMENT type is `dtype` (torch.bfloat16 -- the default -- or torch.float16), once each.
    Raises if it has not been built -- there is no fallback path."""
    code = elem_code(dtype)
    lib = _libs.get(code)
    if lib is not None:
        return lib
    # developer override: A/B builds of the same ABI (tools/); CTRLV_HIP_LIB replaces the bf16 library, _F16 the fp16 one
    env = "CTRLV_HIP_LIB" if code == 2 else "CTRLV_HIP_LIB_F16"
    path = os.environ.get(env, LIB_PATH if code == 2 else LIB_PATH_F16)
    if not os.path.exists(path):
        raise CtrlvHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  ctrlv_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here = header / library out of sync
        fn.restype, fn.argtypes = res, args
        if res is c_int and name not in _NO_STATUS:
            # status-returning entry points remember WHICH library failed last on this thread: check() then reads that
            # library's message, not a stale one of the other library (two libraries may be loaded)
            setattr(lib, name, _status_recorder(lib, fn))
    if lib.ctrlv_abi_version() != ABI_VERSION:
        raise CtrlvHipError(f"{path}: ABI version {lib.ctrlv_abi_version()}, this host layer needs {ABI_VERSION}")
    if lib.ctrlv_elem_dtype() != code:
        raise CtrlvHipError(f"{path}: element dtype code {lib.ctrlv_elem_dtype()}, expected {code}")
    # a shipped .so must come from the sources it sits next to (it is git-ignored and travels prebuilt); variant
    # libraries selected through CTRLV_HIP_LIB (tools/ab_build.py A/B builds) are exempt
    want = source_build_id()
    if want is not None and env not in os.environ and build_id(lib) != want:
        raise CtrlvHipError(f"{path} is stale: built from sources {build_id(lib)}, tree is {want}; rebuild with "
                            "`python -c 'import __graft_entry__ as g; g.build()'`")
    _libs[code] = lib
    return lib


def last_error(lib=None):
    """Message of the last failing call of this thread (the libraries keep separate texts: the newest non-empty one of
    the loaded libraries is returned when `lib` is not given -- a failing call always sets its own)."""
    buf = ctypes.create_string_buffer(512)
    if lib is not None:
        lib.ctrlv_last_error(buf, 512)
        return buf.value.decode("utf-8", "replace")
    msgs = []
    for l in (_libs.values() or [load()]):
        l.ctrlv_last_error(buf, 512)
        if buf.value:
            msgs.append(buf.value.decode("utf-8", "replace"))
    return " | ".join(msgs)


def check(rc, what, lib=None):
    """Map the C status convention (include/ctrlv_hip.h) onto the reference's exception types: bad shapes /
    arguments -> ValueError (as controlnet.py:80-98, pipeline_video_control.py:51-68), HIP failures -> RuntimeError.
    The message is the one of the library whose call failed: `lib` when given (call sites whose entry point does not
    return a status -- size queries -- pass their own library), else the library whose status-returning call failed LAST on
    this thread (the mark is consumed here and cleared by any later successful call), else every loaded library's text."""
    if rc == 0:
        return
    failed, _tls.failed = getattr(_tls, "failed", None), None
    msg = f"{what}: {last_error(lib or failed)} (status {rc})"
    if rc in (-1, -2, -4, -5):     # bad argument / shape / dtype / workspace size
        raise ValueError(msg)
    raise CtrlvHipError(msg)
