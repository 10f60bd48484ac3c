"""Weight packing: diffusers state-dict layouts -> the K-contiguous 16-bit layouts the gather-GEMM consumes (element type
bf16 by default; `with element_dtype(torch.float16):` packs for libctrlv_hip_f16.so).

Pure host-side tensor shuffling (runs on any device, unit-tested on CPU).  Layouts (include/ctrlv_hip.h):
  W[n, tap*Cin + c]; N padded to a multiple of 32 rows, K padded to a multiple of 64 columns (zeros).
"""
import contextlib

import torch

_EL = [torch.bfloat16]          # element type the packers round to (see element_dtype)


@contextlib.contextmanager
def element_dtype(dtype):
    """Packers called inside round to `dtype` (torch.bfloat16 / torch.float16): the model's element type."""
    prev = _EL[0]
    _EL[0] = dtype
    try:
        yield
    finally:
        _EL[0] = prev


def _pad_rows(w, mult=32):
    n = w.shape[0]
    n_pad = (n + mult - 1) // mult * mult
    if n_pad != n:
        w = torch.cat([w, w.new_zeros(n_pad - n, *w.shape[1:])], 0)
    return w


def _pad_cols(w, mult=64):
    k = w.shape[1]
    k_pad = (k + mult - 1) // mult * mult
    if k_pad != k:
        w = torch.cat([w, w.new_zeros(w.shape[0], k_pad - k)], 1)
    return w


def pack_linear(weight):
    """nn.Linear / 1x1 Conv2d weight [N, K(,1,1)] -> bf16 [N32, K64]."""
    w = weight.detach().reshape(weight.shape[0], -1)
    return _pad_cols(_pad_rows(w)).to(_EL[0]).contiguous()


def pack_conv3x3(weight):
    """Conv2d weight [N, C, 3, 3] -> bf16 [N32, 9*C] with k = (ky*3+kx)*C + c."""
    n, c = weight.shape[:2]
    w = weight.detach().permute(0, 2, 3, 1).reshape(n, 9 * c)
    return _pad_rows(w).to(_EL[0]).contiguous()


def pack_conv_temporal(weight):
    """Conv3d weight [N, C, 3, 1, 1] -> bf16 [N32, 3*C] with k = t*C + c."""
    n, c = weight.shape[:2]
    w = weight.detach()[:, :, :, 0, 0].permute(0, 2, 1).reshape(n, 3 * c)
    return _pad_rows(w).to(_EL[0]).contiguous()


def pack_conv_in(weights, cp=16, kp=192):
    """The tiny-channel input convs (conv_in [N,8,3,3] and optionally control_conv_in [N,4,3,3]) share one im2col GEMM:
    channel slot layout [conv_in channels | control channels | zero pad] of width cp per tap, K padded to kp."""
    n = weights[0].shape[0]
    w = weights[0].new_zeros(n, 9, cp)
    off = 0
    for wt in weights:
        c = wt.shape[1]
        w[:, :, off:off + c] = wt.detach().permute(0, 2, 3, 1).reshape(n, 9, c)
        off += c
    assert off <= cp
    w = w.reshape(n, 9 * cp)
    return _pad_cols(_pad_rows(w), kp).to(_EL[0]).contiguous()


GEGLU_BLOCK = 16


def geglu_interleave(t):
    """Rows [a_0..a_{I-1} | g_0..g_{I-1}] -> alternating 16-row blocks (value block, gate block): every 32-column MFMA
    sub-tile of the GEMM then holds 16 values (accumulator quads 0,1) and their 16 gates (quads 2,3) in the same lane,
    so the GEGLU product needs no cross-lane or cross-sub-tile traffic and works for any tile width."""
    inner = t.shape[0] // 2
    assert inner % GEGLU_BLOCK == 0, "GEGLU inner dim must be a multiple of 16"
    a, g = t[:inner], t[inner:]
    rest = t.shape[1:]
    st = torch.stack([a.reshape(inner // GEGLU_BLOCK, GEGLU_BLOCK, *rest),
                      g.reshape(inner // GEGLU_BLOCK, GEGLU_BLOCK, *rest)], 1)
    return st.reshape(2 * inner, *rest)


def pack_geglu(weight, bias):
    w = _pad_cols(geglu_interleave(weight.detach())).to(_EL[0]).contiguous()
    b = geglu_interleave(bias.detach()).float().contiguous()
    return w, b


def pack_qkv(wq, wk, wv):
    return pack_linear(torch.cat([wq.detach(), wk.detach(), wv.detach()], 0))


def pad_bias(bias, mult=32):
    b = bias.detach().float()
    n = b.shape[0]
    n_pad = (n + mult - 1) // mult * mult
    if n_pad != n:
        b = torch.cat([b, b.new_zeros(n_pad - n)])
    return b.contiguous()
