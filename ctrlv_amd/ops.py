"""Tensor-level wrappers over the C ABI (include/ctrlv_hip.h).  Inputs are torch CUDA tensors used purely as
device-memory handles; every call enqueues on torch's current HIP stream.  No wrapper has a fallback."""
import ctypes

import torch

from . import _lib, profiler as _prof
from ._lib import GemmDesc, check

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
_NO_PACK_KERNEL = __import__("os").environ.get("CTRLV_PACK_KERNEL", "1") == "0"      # A/B handle: torch packers only


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _L(*tensors):
    """The library that serves these element tensors: libctrlv_hip_f16.so when they are fp16, else the bf16 library."""
    els = {t.dtype for t in tensors if t is not None and t.dtype in (torch.float16, torch.bfloat16)}
    if len(els) > 1:
        raise ValueError("ctrlv_amd: one call mixes bf16 and fp16 element tensors (a library serves ONE element type)")
    return _lib.load(torch.float16 if torch.float16 in els else None)


def _need_gpu(t, name="tensor"):
    if not t.is_cuda:
        raise _lib.CtrlvHipError(f"ctrlv_amd: {name} must live on a HIP device (got {t.device}); there is no CPU path")


# Parameter-gradient reductions of the training step (wgrad, bias / row-vector column sums, the mixing-weight dot product):
# True = ordered sums through scratch buffers (bit-reproducible gradients), False = fp32 atomics.
DETERMINISTIC = __import__("os").environ.get("CTRLV_DETERMINISTIC", "1") != "0"
_SCRATCH = {}


def _scratch(device, nbytes, tag):
    """Reduction scratch, one buffer per (device, stream, purpose), grown on demand (launches of a stream are ordered)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _SCRATCH[key] = buf
    return buf


def gemm_wgrad(A, dY, dW, *, N, cin, taps=1, mode=0, A2=None, c_split=0, conv=None, temporal=None, dbias=None, scale=1.0,
               torch_layout=False, assign=False):
    """dW[N, taps*cin] (fp32, packed tap-major K order) += scale * dY^T . gather(A) for the forward GEMM of the same
    geometry; dbias[N] (fp32, optional) += scale * column sums of dY.  torch_layout: dW is [N, cin, taps] (the conv
    parameter's own layout) instead of the packed [N, taps*cin].  assign (deterministic form only): dW / dbias are
    written, not accumulated into -- they need no zero fill."""
    _need_gpu(A, "A")
    d = GemmDesc()
    d.A, d.A2 = _p(A), _p(A2)
    d.M, d.N, d.Cin, d.taps, d.mode = dY.shape[0], N, cin, taps, mode
    d.lda = A.stride(0)
    d.lda2 = A2.stride(0) if A2 is not None else 0
    d.c_split = c_split
    if conv is not None:
        d.H, d.Wd, d.Ho, d.Wo, d.stride, d.up = conv
    if temporal is not None:
        d.F, d.S = temporal
    lib = _L(A, dY)
    # deterministic by default (DETERMINISTIC = False: fp32 atomics, arrival order): slab partials + an ordered sum
    scratch, nbytes = None, 0
    if DETERMINISTIC:
        nbytes = lib.ctrlv_gemm_wgrad_scratch_bytes(ctypes.byref(d))
        scratch = _scratch(A.device, nbytes, "wgrad")
    check(lib.ctrlv_gemm_wgrad(ctypes.byref(d), _p(dY), dY.stride(0), _p(dW), _p(dbias), float(scale),
                               (1 if torch_layout else 0) | (2 if assign else 0), _p(scratch), nbytes, _stream()),
          "ctrlv_gemm_wgrad")
    return dW


def pack_weight(weight, form=0, geglu=False, dtype=torch.bfloat16):
    """Element-type (`dtype`: bf16 / fp16) GEMM layout of a PyTorch-layout parameter in ONE kernel (ctrlv_pack_weight): form 0 = forward
    [N32, taps*C], form 1 = role-swapped dgrad [C32, taps*N64].  Returns None when the shape needs the torch packer
    (K not a multiple of 64 for Linear, odd channel counts)."""
    w = weight.detach()
    if _NO_PACK_KERNEL or not (w.is_cuda and w.is_contiguous() and w.dtype in _DT):
        return None
    N, C = w.shape[0], w.shape[1]
    taps = w.numel() // (N * C)
    if form == 0:
        if (taps == 1 and C % 64) or C % 8:
            return None
        rows, ld = (N + 31) // 32 * 32, taps * C
    else:
        if C % 32:
            return None
        rows, ld = C, taps * ((N + 63) // 64 * 64)
    dst = (torch.zeros if rows != (N if form == 0 else C) else torch.empty)(rows, ld, dtype=dtype, device=w.device)
    check(_L(dst).ctrlv_pack_weight(_p(w), _DT[w.dtype], N, C, taps, form, 1 if geglu else 0, _p(dst), ld, _stream()),
          "ctrlv_pack_weight")
    return dst


def colsum(x, out, vmode=0, vdiv=1, vmod=1, scale=1.0):
    """out[idx(m), :N] += scale * sum_m x[m, :]  (fp32, atomics): bias gradients (vmode 0) / per-clip row-vector gradients."""
    _need_gpu(x, "x")
    lib = _L(x)
    scratch = None
    if DETERMINISTIC:
        nfl = lib.ctrlv_colsum_scratch_floats(x.shape[0], x.shape[1], vmode, vdiv)
        if nfl:
            scratch = _scratch(x.device, nfl * 4, "colsum")
    check(lib.ctrlv_colsum(_p(x), x.shape[0], x.shape[1], x.stride(0), vmode, vdiv, vmod, float(scale), _p(out),
                           out.stride(0) if out.dim() > 1 else x.shape[1], _p(scratch), _stream()), "ctrlv_colsum")
    return out


def dot_diff(dy, p, q, out, scale=1.0):
    _need_gpu(dy, "dy")
    scratch = _scratch(dy.device, 4096, "dot_diff") if DETERMINISTIC else None
    check(_L(dy).ctrlv_dot_diff(_p(dy), _p(p), _p(q), dy.numel(), float(scale), _p(out), _p(scratch), _stream()),
          "ctrlv_dot_diff")
    return out


def groupnorm_bwd(x, dy, n_img, S, C, imgs_per_stat, fwd_partials, gamma, beta, silu, dx, dgamma, dbeta, add=None):
    """add (rows like dy, optional): dx = (norm backward) + add -- the gradient of the skip connection around the branch the norm
    opens, summed in the same pass (ctrlv_groupnorm_bwd_add)."""
    _need_gpu(x, "x")
    lib = _L(x)
    n = lib.ctrlv_groupnorm_bwd_scratch_floats(n_img, S, C, imgs_per_stat)
    if n < 0:
        check(n, "ctrlv_groupnorm_bwd_scratch_floats")
    scratch = torch.empty(n, dtype=torch.float32, device=x.device)
    check(lib.ctrlv_groupnorm_bwd_add(_p(x), _p(dy), _p(add), n_img, S, C, imgs_per_stat, _p(fwd_partials), _p(gamma), _p(beta),
                                      1 if silu else 0, _p(dx), _p(dgamma), _p(dbeta), _p(scratch), _stream()),
          "ctrlv_groupnorm_bwd_add")
    return dx


def layernorm_bwd(x, dy, gamma, eps, dx, dgamma, dbeta, V=None, vdiv=1, vmod=1 << 30, add=None):
    _need_gpu(x, "x")
    M, C = x.shape
    lib = _L(x)
    scratch = torch.empty(lib.ctrlv_layernorm_bwd_scratch_floats(M, C), dtype=torch.float32, device=x.device)
    check(lib.ctrlv_layernorm_bwd_add(_p(x), _p(dy), _p(add), M, C, _p(gamma), eps, _p(V), vdiv, vmod,
                                      V.stride(0) if V is not None else 0, _p(dx), _p(dgamma), _p(dbeta), _p(scratch),
                                      _stream()), "ctrlv_layernorm_bwd_add")
    return dx


def geglu_bwd(raw, du, draw):
    _need_gpu(raw, "raw")
    check(_L(raw).ctrlv_geglu_bwd(_p(raw), _p(du), du.shape[0], du.shape[1], _p(draw), _stream()), "ctrlv_geglu_bwd")
    return draw


def _gemm_desc(A, W, out, *, N, cin, taps=1, mode=0, bias=None, A2=None, c_split=0, conv=None, temporal=None,
               R1=None, s1=1.0, R2=None, s2=1.0, s_acc=1.0, V=None, vmode=0, vdiv=1, vmod=1 << 30, vS=1,
               act=0, geglu=0, out_f32=False, n_store=None, M=None, tile=0, raw_out=None, n_scale2=0, s_acc2=1.0, _dbg=0,
               gn_partials=None, splitk=True, rows_per_image=0, R1_lo=None, R2_lo=None, out_lo=None):
    d = GemmDesc()
    d.A, d.A2, d.W, d.out = _p(A), _p(A2), _p(W), _p(out)
    d.bias, d.R1, d.R2, d.V = _p(bias), _p(R1), _p(R2), _p(V)
    d.M = out.shape[0] if M is None else M
    d.N, d.Cin, d.taps = N, cin, taps
    d.lda = A.stride(0)
    d.lda2 = A2.stride(0) if A2 is not None else 0
    d.c_split = c_split
    d.mode = mode
    if conv is not None:
        d.H, d.Wd, d.Ho, d.Wo, d.stride, d.up = conv
    if temporal is not None:
        d.F, d.S = temporal
    elif mode == 0 and rows_per_image:
        d.S = rows_per_image          # mode 0: the split plan's shape key (rows per image)
    d.ldo = out.stride(0)
    d.n_store = out.shape[1] if n_store is None else n_store
    d.ldr1 = R1.stride(0) if R1 is not None else 0
    d.ldr2 = R2.stride(0) if R2 is not None else 0
    d.s_acc, d.s1, d.s2 = s_acc, s1, s2
    d.vmode = vmode if V is not None else 0
    d.vdiv, d.vmod, d.vS = vdiv, vmod, vS
    d.ldv = V.stride(0) if V is not None else 0
    d.act, d.geglu, d.out_f32, d.tile = act, geglu, (_dbg if _dbg else (1 if out_f32 else 0)), tile
    if raw_out is not None:
        d.raw_out, d.ld_raw = _p(raw_out), raw_out.stride(0)
    d.n_scale2, d.s_acc2 = n_scale2, s_acc2
    d.gn_partials = _p(gn_partials)
    # split residual-trunk planes (fp16 library; include/ctrlv_hip.h): same shapes / pitches as R1 / R2 / out
    d.R1_lo, d.R2_lo, d.out_lo = _p(R1_lo), _p(R2_lo), _p(out_lo)
    return d


_SPLITK_WS = {}


def _splitk_scratch(device, nbytes):
    """fp32 scratch of the split contractions, one per (device, stream), grown on demand: the launches of one stream are
    ordered, two streams (ControlNet beside the UNet encoder) must not share it."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _SPLITK_WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _SPLITK_WS[key] = buf
    return buf


def gemm_splitk_slices(A, W, out, **kw):
    """K slices `gemm(A, W, out, **kw)` runs this launch in (1 = not split): scratch bytes / (M * N * 4)."""
    d = _gemm_desc(A, W, out, **kw)
    return max(1, _L(A, W).ctrlv_gemm_splitk_ws_bytes(ctypes.byref(d)) // (d.M * d.N * 4))


def gemm_gn_partials_serves(A, W, out, **kw):
    """Whether `gemm(A, W, out, **kw, gn_partials=...)` writes GroupNorm chunk partials of `out` (the launcher's own
    predicate: a function of the layer's shape, never of the batch size)."""
    return bool(_L(A, W).ctrlv_gemm_gn_partials_serves(ctypes.byref(_gemm_desc(A, W, out, **kw))))


def gemm(A, W, out, **kw):
    """out = epilogue(gather-GEMM(A[, A2], W)).  `conv` = (H, W, Ho, Wo, stride, up); `temporal` = (F, S).
    Keywords: see `_gemm_desc` (the fields of ctrlv_gemm_desc)."""
    _need_gpu(A, "A")
    d = _gemm_desc(A, W, out, **kw)
    lib = _L(A, W)
    if kw.get("splitk", True):
        # split contraction of the small-image long-K convs (csrc/gemm.hip splitk_plan): the scratch is what enables it, and
        # it is offered to EVERY launch -- the plan is a function of the layer's shape, so a clip takes the same path alone
        # and in a batch
        need = lib.ctrlv_gemm_splitk_ws_bytes(ctypes.byref(d))
        if need:
            d.splitk_ws = _p(_splitk_scratch(A.device, need))
    cin, taps, mode, geglu = d.Cin, d.taps, d.mode, d.geglu
    R1, R2, raw_out, act = kw.get("R1"), kw.get("R2"), kw.get("raw_out"), kw.get("act", 0)
    out_f32 = kw.get("out_f32", False)
    ev = _prof.begin()
    check(lib.ctrlv_gemm(ctypes.byref(d), _stream()), "ctrlv_gemm")
    if ev is not None:
        n_alg = d.N if geglu else min(d.N, d.n_store)
        fam = "gemm_conv3x3" if mode == 1 else ("gemm_conv_temporal" if mode == 2 else "gemm_linear")
        esz = 4 if out_f32 else 2
        n_out = d.n_store if not geglu else min(d.n_store, d.N // 2)
        nbytes = (d.M * cin * 2 * (1 if mode == 0 else 1) + d.M * n_out * esz + d.N * taps * cin * 2
                  + (d.M * n_out * 2 if R1 is not None else 0) + (d.M * n_out * 2 if R2 is not None else 0)
                  + (d.M * n_out * 2 if raw_out is not None else 0))
        _prof.end(ev, fam, 2.0 * d.M * n_alg * taps * cin, float(nbytes),
                  detail=(fam, d.M, d.N, taps * cin, int(geglu), int(R1 is not None) + int(R2 is not None), d.vmode, act))
    return out


def ff_fused_pack(w1_packed, b1, w2_packed):
    """Fragment-major copies (w1f, w2f) of a C = 320 feed-forward's packed weights for `ff_fused`: w1_packed = the GEGLU
    projection in the interleaved packed form [2560, 320], b1 = its bias (fp32 [2560], same row order; carried inside w1f),
    w2_packed = the output projection [320, 1280]."""
    _need_gpu(w1_packed, "w1_packed")
    assert tuple(w1_packed.shape) == (2560, 320) and tuple(w2_packed.shape) == (320, 1280) and b1.numel() == 2560
    assert w1_packed.dtype in (torch.bfloat16, torch.float16) and w2_packed.dtype == w1_packed.dtype and b1.dtype == torch.float32
    assert w1_packed.is_contiguous() and w2_packed.is_contiguous() and b1.is_contiguous()
    lib = _L(w1_packed)
    w1f = torch.empty(lib.ctrlv_ff_fused_w1f_bytes() // 2, dtype=w1_packed.dtype, device=w1_packed.device)
    w2f = torch.empty_like(w2_packed)
    check(lib.ctrlv_ff_fused_pack(_p(w1_packed), _p(b1), _p(w2_packed), _p(w1f), _p(w2f), _stream()), "ctrlv_ff_fused_pack")
    return w1f, w2f


def _ff_out_desc(out, bias=None, R1=None, s1=1.0, R2=None, s2=1.0, s_acc=1.0, V=None, vmode=0, vdiv=1, vmod=1 << 30, vS=1,
                 R1_lo=None, R2_lo=None, out_lo=None):
    """The second projection's descriptor of a C = 320 feed-forward (what ctrlv_ff_fused / ctrlv_ff_fused_serves take)."""
    d = GemmDesc()
    d.out, d.bias, d.R1, d.R2, d.V = _p(out), _p(bias), _p(R1), _p(R2), _p(V)
    d.R1_lo, d.R2_lo, d.out_lo = _p(R1_lo), _p(R2_lo), _p(out_lo)
    d.M, d.N, d.Cin, d.taps, d.mode = out.shape[0], out.shape[1], 4 * out.shape[1], 1, 0
    d.ldo, d.n_store = out.stride(0), out.shape[1]
    d.ldr1 = R1.stride(0) if R1 is not None else 0
    d.ldr2 = R2.stride(0) if R2 is not None else 0
    d.s_acc, d.s1, d.s2 = s_acc, s1, s2
    d.vmode = vmode if V is not None else 0
    d.vdiv, d.vmod, d.vS = vdiv, vmod, vS
    d.ldv = V.stride(0) if V is not None else 0
    return d


def ff_fused_serves(x, out, **epi):
    """Whether `ff_fused(x, ..., out, **epi)` is served by the fused kernel (else: the two `gemm` launches) -- the
    launcher's own conditions (csrc/ff_fused.hip ff_check)."""
    d = _ff_out_desc(out, **{k: v for k, v in epi.items() if k != "bias"})
    return bool(_L(x).ctrlv_ff_fused_serves(ctypes.byref(d), x.stride(0)))


def ff_fused(x, w1f, w2f, out, *, bias=None, R1=None, s1=1.0, R2=None, s2=1.0, s_acc=1.0, V=None, vmode=0, vdiv=1,
             vmod=1 << 30, vS=1, ln=None, ln_V=None, ln_vdiv=1, ln_vmod=1 << 30, R1_lo=None, R2_lo=None, out_lo=None):
    """out = s_acc * (GEGLU(x W1^T + b1) W2^T + bias) + s1 R1 + s2 R2 + V[idx(m)] for a C = 320 feed-forward, the 4C-wide
    intermediate kept on chip (csrc/ff_fused.hip).  The epilogue operands are those of `gemm`."""
    _need_gpu(x, "x")
    d = _ff_out_desc(out, bias, R1, s1, R2, s2, s_acc, V, vmode, vdiv, vmod, vS, R1_lo, R2_lo, out_lo)
    ev = _prof.begin()
    if ln is None:        # ln = (gamma, beta, eps): the LayerNorm in front of the feed-forward, folded into the kernel
        check(_L(x).ctrlv_ff_fused(_p(x), x.stride(0), _p(w1f), _p(w2f), ctypes.byref(d), _stream()),
              "ctrlv_ff_fused")
    else:
        check(_L(x).ctrlv_ff_fused_ln(_p(x), x.stride(0), _p(ln[0]), _p(ln[1]), float(ln[2]), _p(ln_V), ln_vdiv,
                                      ln_vmod, ln_V.stride(0) if ln_V is not None else 0, _p(w1f), _p(w2f),
                                      ctypes.byref(d), _stream()), "ctrlv_ff_fused_ln")
    if ev is not None:
        M = out.shape[0]
        nbytes = M * 320 * 2 * (2 + (R1 is not None) + (R2 is not None))
        _prof.end(ev, "gemm_linear", 2.0 * M * 320 * (2560 + 1280), float(nbytes),
                  detail=("ff_fused", M, 320, 320, 1, int(R1 is not None) + int(R2 is not None), d.vmode, 0))
    return out


def groupnorm_chunks(n_img, S, C, imgs_per_stat):
    rc = _lib.load().ctrlv_groupnorm_chunks(n_img, S, C, imgs_per_stat)
    if rc < 0:
        check(rc, "ctrlv_groupnorm_chunks")
    return rc


def groupnorm_scratch_floats(n_img, S, C, imgs_per_stat):
    """fp32 elements `groupnorm` needs in `partials`: chunk partials + the (mean, rstd) table behind them."""
    return (n_img * groupnorm_chunks(n_img, S, C, imgs_per_stat) + n_img // imgs_per_stat) * 64


def groupnorm_fused_scratch_floats(n_img, S, imgs_per_stat):
    """fp32 elements of `partials` for `gemm(..., gn_partials=)` + `groupnorm_from_partials` (64-row chunks)."""
    return (n_img * (S // 64) + n_img // imgs_per_stat) * 64


def groupnorm_from_partials(x, n_img, S, C, imgs_per_stat, gamma, beta, eps, silu, y, partials, x_lo=None):
    """GroupNorm(+SiLU) of a tensor whose producing `gemm` wrote the chunk partials: finalize + apply.  x_lo: the lo plane of
    a SPLIT tensor (the launch wrote out_lo and gn_partials together; the normalised value is x + x_lo)."""
    _need_gpu(x, "x")
    assert partials.numel() >= groupnorm_fused_scratch_floats(n_img, S, imgs_per_stat)
    ev = _prof.begin()
    check(_L(x).ctrlv_groupnorm_from_partials_split(_p(x), _p(x_lo), n_img, S, C, imgs_per_stat, eps, _p(partials), _p(gamma),
                                                    _p(beta), 1 if silu else 0, _p(y), _stream()),
          "ctrlv_groupnorm_from_partials")
    _prof.end(ev, "groupnorm", 0.0, 2.0 * 2 * n_img * S * C)
    return y


def groupnorm(x, x2, n_img, S, C, imgs_per_stat, gamma, beta, eps, silu, y, partials, x_lo=None, x2_lo=None):
    """Two-pass GroupNorm(32)(+SiLU) over channels-last rows; (x | x2) is a channel concat when x2 is given.  x_lo / x2_lo:
    the lo planes of a SPLIT input (the normalised value is x + x_lo)."""
    _need_gpu(x, "x")
    lib = _L(x)
    c_split = x.shape[1] if x2 is not None else 0
    st = _stream()
    ev = _prof.begin()
    check(lib.ctrlv_groupnorm_stats_split(_p(x), _p(x_lo), _p(x2), _p(x2_lo), c_split, n_img, S, C, imgs_per_stat, eps,
                                          _p(partials), st), "ctrlv_groupnorm_stats")
    check(lib.ctrlv_groupnorm_apply_split(_p(x), _p(x_lo), _p(x2), _p(x2_lo), c_split, n_img, S, C, imgs_per_stat,
                                          _p(partials), _p(gamma), _p(beta), 1 if silu else 0, _p(y), st),
          "ctrlv_groupnorm_apply")
    _prof.end(ev, "groupnorm", 0.0, 2.0 * 2 * n_img * S * C)       # algorithmic: 1 read + 1 write, bf16
    return y


def layernorm(x, gamma, beta, eps, y, V=None, vdiv=1, vmod=1 << 30, x_lo=None):
    _need_gpu(x, "x")
    M, C = x.shape
    ev = _prof.begin()
    check(_L(x).ctrlv_layernorm_split(_p(x), _p(x_lo), M, C, _p(gamma), _p(beta), eps, _p(V), vdiv, vmod,
                                      V.stride(0) if V is not None else 0, _p(y), _stream()), "ctrlv_layernorm")
    _prof.end(ev, "layernorm", 0.0, 2.0 * 2 * M * C)
    return y


Q_PRESCALE = 0.125 * 1.44269504088896340736      # (1/sqrt(64)) * log2(e): the q block scale of the prescaled core


def attention_spatial(qkv, out, n_img, S, C, prescaled=False):
    """prescaled: the q columns already carry Q_PRESCALE (gemm(..., n_scale2=C, s_acc2=Q_PRESCALE))."""
    _need_gpu(qkv, "qkv")
    ev = _prof.begin()
    fn = _L(qkv).ctrlv_attention_spatial_prescaled if prescaled else _L(qkv).ctrlv_attention_spatial
    check(fn(_p(qkv), _p(out), n_img, S, C, _stream()), "ctrlv_attention_spatial")
    _prof.end(ev, "attention_spatial", 4.0 * n_img * (C // 64) * S * S * 64, 2.0 * 4 * n_img * S * C)
    return out


def attention_spatial_lse(qkv, out, lse, n_img, S, C):
    """Training forward: also writes lse [n_img, C/64, S] fp32 (log2-domain log-sum-exp of the scaled scores)."""
    _need_gpu(qkv, "qkv")
    ev = _prof.begin()
    check(_L(qkv).ctrlv_attention_spatial_lse(_p(qkv), _p(out), _p(lse), n_img, S, C, _stream()),
          "ctrlv_attention_spatial_lse")
    _prof.end(ev, "attention_spatial", 4.0 * n_img * (C // 64) * S * S * 64, 2.0 * 4 * n_img * S * C)
    return out


def attention_spatial_bwd(qkv, out, dout, lse, dqkv, n_img, S, C):
    _need_gpu(qkv, "qkv")
    delta = torch.empty(_L(qkv).ctrlv_attention_bwd_scratch_floats(n_img, S, C), dtype=torch.float32, device=qkv.device)
    ev = _prof.begin()
    check(_L(qkv).ctrlv_attention_spatial_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), _p(delta), n_img, S, C,
                                                  _stream()), "ctrlv_attention_spatial_bwd")
    _prof.end(ev, "attention_spatial_bwd", 14.0 * n_img * (C // 64) * S * S * 64, 2.0 * 8 * n_img * S * C)
    return dqkv


def attention_temporal_bwd(qkv, out, dout, dqkv, B, F, S, C):
    _need_gpu(qkv, "qkv")
    ev = _prof.begin()
    check(_L(qkv).ctrlv_attention_temporal_bwd(_p(qkv), _p(out), _p(dout), _p(dqkv), B, F, S, C, _stream()),
          "ctrlv_attention_temporal_bwd")
    _prof.end(ev, "attention_temporal_bwd", 14.0 * B * S * (C // 64) * F * F * 64, 2.0 * 8 * B * F * S * C)
    return dqkv


def attention_temporal(qkv, out, B, F, S, C):
    _need_gpu(qkv, "qkv")
    ev = _prof.begin()
    check(_L(qkv).ctrlv_attention_temporal(_p(qkv), _p(out), B, F, S, C, _stream()), "ctrlv_attention_temporal")
    _prof.end(ev, "attention_temporal", 4.0 * B * S * (C // 64) * F * F * 64, 2.0 * 4 * B * F * S * C)
    return out


def temporal_fused_pack(wqkv_packed, wo_packed):
    """Fragment-major weights of `temporal_fused` from the packed [960, >= 320] q|k|v projection and [320, >= 320] to_out."""
    _need_gpu(wqkv_packed, "wqkv_packed")
    assert wqkv_packed.shape[0] == 960 and wo_packed.shape[0] == 320 and wqkv_packed.dtype == wo_packed.dtype
    lib = _L(wqkv_packed)
    wf = torch.empty(lib.ctrlv_temporal_fused_weight_bytes() // 2, dtype=wqkv_packed.dtype, device=wqkv_packed.device)
    check(lib.ctrlv_temporal_fused_pack(_p(wqkv_packed), wqkv_packed.stride(0), _p(wo_packed), wo_packed.stride(0), _p(wf),
                                        _stream()), "ctrlv_temporal_fused_pack")
    return wf


def _temporal_fused_desc(x, wf, out, B, F, S, bias=None, R1=None, V=None, vmode=0, vdiv=1, vmod=1 << 30, vS=1, R1_lo=None,
                         out_lo=None, ln=None):
    from ._lib import TemporalFusedDesc
    d = TemporalFusedDesc()
    d.x, d.ldx, d.wf, d.bias = _p(x), x.stride(0), _p(wf), _p(bias)
    d.R1, d.R1_lo, d.ldr1 = _p(R1), _p(R1_lo), (R1.stride(0) if R1 is not None else 0)
    d.V, d.vmode, d.vdiv, d.vmod, d.vS = _p(V), (vmode if V is not None else 0), vdiv, vmod, vS
    d.ldv = V.stride(0) if V is not None else 0
    d.out, d.out_lo, d.ldo = _p(out), _p(out_lo), out.stride(0)
    d.B, d.F, d.S, d.C = B, F, S, x.shape[1]
    if ln is not None:                       # (gamma, beta, eps): x holds the raw rows, the kernel normalises them
        d.ln_gamma, d.ln_beta, d.ln_eps = _p(ln[0]), _p(ln[1]), float(ln[2])
    return d


def temporal_fused_serves(x, wf, out, B, F, S, **kw):
    return wf is not None and bool(_L(x).ctrlv_temporal_fused_serves(ctypes.byref(_temporal_fused_desc(x, wf, out, B, F, S, **kw))))


def temporal_fused(x, wf, out, B, F, S, **kw):
    """out = R1 + to_out(softmax_f(q k^T / 8) v) + bias + V[clip], (q|k|v) = x W_qkv^T, over the F frames of every pixel:
    the temporal self-attention block at C = 320 in one launch (csrc/temporal_fused.hip)."""
    _need_gpu(x, "x")
    d = _temporal_fused_desc(x, wf, out, B, F, S, **kw)
    ev = _prof.begin()
    check(_L(x).ctrlv_temporal_fused(ctypes.byref(d), _stream()), "ctrlv_temporal_fused")
    M, C = B * F * S, x.shape[1]
    _prof.end(ev, "gemm_temporal_block", 2.0 * M * C * 4 * C + 4.0 * B * S * (C // 64) * F * F * 64,
              2.0 * (2 if kw.get("ln") is not None else 3) * M * C)
    return out


def nchw_to_rows(src, dst, c_off=0):
    """src: contiguous (n_img, C, H, W) fp32/fp16/bf16 -> dst rows [n_img*H*W, ldc] bf16, channels [c_off, c_off+C)."""
    _need_gpu(src, "src")
    n_img, C = src.shape[0], src.shape[1]
    HW = src.shape[2] * src.shape[3]
    check(_L(dst).ctrlv_nchw_to_rows(_p(src), _DT[src.dtype], n_img, C, HW, _p(dst), dst.stride(0), c_off,
                                         _stream()), "ctrlv_nchw_to_rows")
    return dst


def rows_to_nchw(src, dst, C=None):
    """src rows [n_img*H*W, ldc] bf16 -> dst contiguous (n_img, C, H, W) of dst.dtype."""
    _need_gpu(src, "src")
    n_img, Cd = dst.shape[0], dst.shape[1]
    HW = dst.shape[2] * dst.shape[3]
    check(_L(src).ctrlv_rows_to_nchw(_p(src), src.stride(0), n_img, Cd if C is None else C, HW, _p(dst),
                                         _DT[dst.dtype], _stream()), "ctrlv_rows_to_nchw")
    return dst


def im2col3x3(x, n_img, H, W, col):
    _need_gpu(x, "x")
    check(_L(x).ctrlv_im2col3x3(_p(x), n_img, H, W, x.shape[1], _p(col), col.shape[1], _stream()),
          "ctrlv_im2col3x3")
    return col


def axpby(x, r, a, b, y):
    _need_gpu(x, "x")
    ev = _prof.begin()
    check(_L(x).ctrlv_axpby(_p(x), _p(r), a, b, _p(y), x.numel(), _stream()), "ctrlv_axpby")
    _prof.end(ev, "residual_add", 0.0, 2.0 * 3 * x.numel())
    return y


def axpby_split(x, x_lo, r, a, b, y, y_lo):
    """(y, y_lo) = split(a * (x + x_lo) + b * r): the residual add on a SPLIT trunk tensor (x_lo may be None)."""
    _need_gpu(x, "x")
    check(_L(x).ctrlv_axpby_split(_p(x), _p(x_lo), _p(r), a, b, _p(y), _p(y_lo), x.numel(), _stream()), "ctrlv_axpby_split")
    return y, y_lo


def silu(x, y):
    _need_gpu(x, "x")
    check(_L(x).ctrlv_silu(_p(x), _p(y), x.numel(), _stream()), "ctrlv_silu")
    return y


def timestep_embedding(t, dim, out):
    """t: fp32 [n] -> out [n, dim] bf16 = [cos | sin] (Timesteps(dim, flip_sin_to_cos=True, shift 0))."""
    _need_gpu(t, "t")
    check(_L(out).ctrlv_timestep_embedding(_p(t), t.numel(), dim, _p(out), _stream()),
          "ctrlv_timestep_embedding")
    return out


def cfg_euler_step(latents, noise_pred, guidance, sigma, sigma_next, scaled_next=None):
    """In-place fused CFG combine + Euler v-prediction update (pipeline_video_control.py:327-332)."""
    _need_gpu(latents, "latents")
    B, F = latents.shape[0], latents.shape[1]
    chw = latents[0, 0].numel()
    cfg = 1 if noise_pred.shape[0] == 2 * B else 0
    check(_L(scaled_next).ctrlv_cfg_euler_step(_p(latents), _p(noise_pred), _DT[noise_pred.dtype], cfg, _p(guidance), B,
                                           F, chw, float(sigma), float(sigma_next), _p(scaled_next), _stream()),
          "ctrlv_cfg_euler_step")
    return latents


def time_conv_rows_to_nchw(rows, n_frames, C, HW, weight, bias, out):
    """Conv3d(C, C, (3,1,1)) over the frames of one clip on channels-last rows, written as NCHW `out` (the VAE decoder's
    time_conv_out)."""
    _need_gpu(rows, "rows")
    check(_L(rows).ctrlv_time_conv_rows_to_nchw(_p(rows), rows.stride(0), n_frames, C, HW, _p(weight), _p(bias), _p(out),
                                                   _DT[out.dtype], _stream()), "ctrlv_time_conv_rows_to_nchw")
    return out


def softmax_rows(scores, probs):
    """probs (bf16) = row softmax of fp32 `scores` (both 2-D, unit inner stride)."""
    _need_gpu(scores, "scores")
    check(_L(probs).ctrlv_softmax_rows(_p(scores), scores.shape[0], scores.shape[1], scores.stride(0), _p(probs),
                                         probs.stride(0), _stream()), "ctrlv_softmax_rows")
    return probs
