"""`Encoder.forward` of the SVD VAE on the HIP kernels: the once-per-clip encode of the conditioning
image and of the 25 bounding-box frames (`vae.encode(x).latent_dist.mode()`, pipeline_video_control.py:71-101, 235).

Same approach as vae_decoder_hip.py: the module's parameters (diffusers state-dict layout) are executed through the
gather-GEMM / GroupNorm kernels -- GroupNorm(+SiLU) -> 3x3 conv with the residual in the epilogue, `conv_in` as an im2col
GEMM.  Two things are not native: the mid block's single-head attention core (head dim 512: torch SDPA between HIP
projections) and the down-samplers' ASYMMETRIC padding (`F.pad(x, (0, 1, 0, 1))` + stride-2 conv without padding, i.e.
out[yo, xo] = sum x[2 yo + ky, 2 xo + kx]): the gather-GEMM's stride-2 mode is the symmetric form, so the three
down-samplers run as a stride-1 convolution whose odd rows / columns are kept (out1[2 yo + 1, 2 xo + 1] is exactly that
sum, the zero padding included) -- 4x the arithmetic on 3 of 25 convolutions.
Limit: 32-bit byte offsets, n * H * W * 128 channels * 2 B < 4 GiB (28 frames at 576x1024; the pipeline encodes 25)."""
import torch
import torch.nn.functional as F

from .. import ops, packing
from .vae_decoder_hip import _Scratch, _f32, _rows

_CACHE = {}


def _pack_res(r):
    pk = dict(cin=r.conv1.weight.shape[1], cout=r.conv1.weight.shape[0], eps=r.norm1.eps,
              g1=_f32(r.norm1.weight), b1=_f32(r.norm1.bias), w1=packing.pack_conv3x3(r.conv1.weight), cb1=_f32(r.conv1.bias),
              g2=_f32(r.norm2.weight), b2=_f32(r.norm2.bias), w2=packing.pack_conv3x3(r.conv2.weight), cb2=_f32(r.conv2.bias))
    if r.conv_shortcut is not None:
        pk["wsc"], pk["bsc"] = packing.pack_linear(r.conv_shortcut.weight), _f32(r.conv_shortcut.bias)
    return pk


def _pack(enc):
    import weakref
    key = id(enc)
    ver = tuple(p._version for p in enc.parameters()) + (next(enc.parameters()).data_ptr(),)
    hit = _CACHE.get(key)
    if hit is not None and hit[0] == ver and hit[1]() is enc:
        return hit[2]
    pk = {}
    cin = enc.conv_in.weight.shape[1]
    cp = (cin + 7) // 8 * 8
    kp = (9 * cp + 63) // 64 * 64
    pk["cin"] = (cp, kp, packing.pack_conv_in([enc.conv_in.weight], cp, kp), packing.pad_bias(enc.conv_in.bias))
    pk["down"] = []
    for blk in enc.down_blocks:
        ds = None
        if blk.downsamplers is not None:
            ds = (packing.pack_conv3x3(blk.downsamplers[0].conv.weight), _f32(blk.downsamplers[0].conv.bias))
        pk["down"].append(([_pack_res(r) for r in blk.resnets], ds))
    pk["mid"] = [_pack_res(r) for r in enc.mid_block.resnets]
    a = enc.mid_block.attentions[0]
    from .vae_decoder_hip import pack_attn
    pk["attn"] = pack_attn(a)
    pk["gno"] = (_f32(enc.conv_norm_out.weight), _f32(enc.conv_norm_out.bias), enc.conv_norm_out.eps)
    pk["cout"] = (packing.pack_conv3x3(enc.conv_out.weight), packing.pad_bias(enc.conv_out.bias), enc.conv_out.weight.shape[0])
    _CACHE[key] = (ver, weakref.ref(enc), pk)
    return pk


def _res(pk, x, n, H, W, sc):
    S, M = H * W, n * H * W
    cin, cout, dev = pk["cin"], pk["cout"], x.device
    geo = (H, W, H, W, 1, 0)
    xn = _rows(M, cin, dev)
    ops.groupnorm(x, None, n, S, cin, 1, pk["g1"], pk["b1"], pk["eps"], True, xn, sc.get(n, S, cin, 1))
    h = _rows(M, cout, dev)
    ops.gemm(xn, pk["w1"], h, N=cout, cin=cin, taps=9, mode=1, conv=geo, bias=pk["cb1"])
    del xn
    hn = _rows(M, cout, dev)
    ops.groupnorm(h, None, n, S, cout, 1, pk["g2"], pk["b2"], pk["eps"], True, hn, sc.get(n, S, cout, 1))
    if "wsc" in pk:
        res = _rows(M, cout, dev)
        ops.gemm(x, pk["wsc"], res, N=cout, cin=cin, bias=pk["bsc"])
    else:
        res = x
    out = h
    ops.gemm(hn, pk["w2"], out, N=cout, cin=cout, taps=9, mode=1, conv=geo, bias=pk["cb2"], R1=res)
    return out


def supports(x, enc=None):
    if not x.is_cuda or x.dim() != 4 or x.shape[2] % 8 or x.shape[3] % 8:
        return False
    c0 = 128
    if enc is not None:
        chans = [m.weight.shape[0] for m in enc.modules() if isinstance(m, torch.nn.Conv2d) and m is not enc.conv_out]
        if any(c % 64 for c in chans) or len(enc.down_blocks) != 4:
            return False
        c0 = chans[0]
    s_lat = (x.shape[2] // 8) * (x.shape[3] // 8)             # mid-block attention (see vae_decoder_hip._attn)
    if s_lat % 64 or s_lat > 16384 or (enc is not None and enc.mid_block.attentions[0].heads != 1):
        return False
    return x.shape[0] * x.shape[2] * x.shape[3] * c0 * 2 < (1 << 32) - (1 << 24)


@torch.no_grad()
def encode(enc, x):
    """enc: `Encoder` module; x: (n, 3, H, W) images in [-1, 1].  Returns the moments (n, 2 * latent_channels, H/8, W/8)
    BEFORE quant_conv, in x.dtype."""
    from .vae_decoder_hip import _attn
    if not supports(x, enc):
        raise ValueError("HIP VAE encode: unsupported shape (CUDA (n, 3, H, W) with H, W multiples of 8, < 4 GiB per tensor)")
    pk = _pack(enc)
    n, _, H, W = x.shape
    dev = x.device
    sc = _Scratch(dev)
    M = n * H * W
    cp, kp, wci, bci = pk["cin"]
    x16 = torch.zeros(M, cp, dtype=torch.bfloat16, device=dev)
    ops.nchw_to_rows(x.contiguous(), x16, 0)
    col = _rows(M, kp, dev)
    ops.im2col3x3(x16, n, H, W, col)
    h = _rows(M, wci.shape[0], dev)
    ops.gemm(col, wci, h, N=wci.shape[0], cin=kp, bias=bci)
    del col, x16
    for res_list, ds in pk["down"]:
        for rp in res_list:
            h = _res(rp, h, n, H, W, sc)
        if ds is not None:
            C = h.shape[1]
            full = _rows(n * H * W, C, dev)
            ops.gemm(h, ds[0], full, N=C, cin=C, taps=9, mode=1, conv=(H, W, H, W, 1, 0), bias=ds[1])
            h = full.view(n, H, W, C)[:, 1::2, 1::2].contiguous().view(n * (H // 2) * (W // 2), C)
            del full
            H, W = H // 2, W // 2
    h = _res(pk["mid"][0], h, n, H, W, sc)
    h = _attn(pk["attn"], h, n, H, W, sc)
    h = _res(pk["mid"][1], h, n, H, W, sc)
    M, C = n * H * W, h.shape[1]
    g, b, eps = pk["gno"]
    hn = _rows(M, C, dev)
    ops.groupnorm(h, None, n, H * W, C, 1, g, b, eps, True, hn, sc.get(n, H * W, C, 1))
    wco, bco, co = pk["cout"]
    co_p = (co + 3) // 4 * 4
    y = _rows(M, co_p, dev)
    ops.gemm(hn, wco, y, N=wco.shape[0], cin=C, taps=9, mode=1, conv=(H, W, H, W, 1, 0), bias=bco, n_store=co_p)
    out = torch.empty(n, co, H, W, dtype=x.dtype, device=dev)
    ops.rows_to_nchw(y, out)
    return out
