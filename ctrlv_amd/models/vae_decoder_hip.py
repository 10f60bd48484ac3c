"""`TemporalDecoder.forward` of the SVD VAE on the HIP kernels (the once-per-clip decode that follows the
denoising loop: `self.vae.decode(latents[i:i+chunk], num_frames=n).sample`, pipeline_video_control.py:346 ->
diffusers `decode_latents`).

Why it exists: on this PyTorch-ROCm stack the 25-frame 576x1024 decode through MIOpen takes 3.4 s per clip (and minutes
on first use while MIOpen searches), against 6.1 s for the whole 25-step denoising loop (tools/vae_decode_bench.py).  The
decoder is the same op vocabulary as the UNet's res blocks -- GroupNorm(+SiLU), 3x3 convs, (3,1,1) convs, a learned
AlphaBlender, nearest-x2 upsampling fused into the next conv -- so it runs on the gather-GEMM / GroupNorm kernels of
libctrlv_hip.so with the same fusions (blend and residuals in GEMM epilogues, upsample in the conv's gather).  The one
exception is the mid block's single-head attention (head dim 512; the HIP attention cores are specialised for 64): its
projections are HIP GEMMs, the core is torch SDPA.  `time_conv_out` (3 -> 3 channels) is three tiny channel mixes in torch.

The parameters stay in the `AutoencoderKLTemporalDecoder` module (diffusers state-dict layout); this file only executes
them.  Packed weights are cached per decoder object (the VAE is frozen).  Activations are channels-last bf16 rows
[n_frames * H * W, C]; a chunk of n frames is ONE clip of n frames for the temporal blocks, exactly as the reference's
chunked decode treats it.  Limit: tensors are addressed with 32-bit byte offsets.  Per-frame ops run on frame ranges
(pointer-offset views) when a tensor is larger; the whole-clip tensors of the temporal halves (128 channels at
576x1024) must fit: at most 28 frames per chunk there (the reference scripts use 8, the pipeline default is all 25).
"""
import math

import torch
import torch.nn.functional as F

from .. import ops, packing

_CACHE = {}


def _f32(t):
    return t.detach().float().contiguous()


def _sig(p):
    return 1.0 / (1.0 + math.exp(-float(p.detach().float().cpu())))


def _pack_res(blk):
    s, t = blk.spatial_res_block, blk.temporal_res_block
    pk = dict(cin=s.conv1.weight.shape[1], cout=s.conv1.weight.shape[0],
              g1=_f32(s.norm1.weight), b1=_f32(s.norm1.bias), w1=packing.pack_conv3x3(s.conv1.weight), cb1=_f32(s.conv1.bias),
              g2=_f32(s.norm2.weight), b2=_f32(s.norm2.bias), w2=packing.pack_conv3x3(s.conv2.weight), cb2=_f32(s.conv2.bias),
              tg1=_f32(t.norm1.weight), tb1=_f32(t.norm1.bias), tw1=packing.pack_conv_temporal(t.conv1.weight),
              tcb1=_f32(t.conv1.bias),
              tg2=_f32(t.norm2.weight), tb2=_f32(t.norm2.bias), tw2=packing.pack_conv_temporal(t.conv2.weight),
              tcb2=_f32(t.conv2.bias),
              # switch_spatial_to_temporal_mix: alpha = 1 - sigmoid(mix);  alpha*xs + (1-alpha)*(xs + h) = xs + sigmoid(mix)*h
              mix=_sig(blk.time_mixer.mix_factor), eps_s=s.norm1.eps, eps_t=t.norm1.eps)
    if s.conv_shortcut is not None:
        pk["wsc"], pk["bsc"] = packing.pack_linear(s.conv_shortcut.weight), _f32(s.conv_shortcut.bias)
    return pk


def _pack(dec):
    key = id(dec)
    ver = tuple(p._version for p in dec.parameters()) + (next(dec.parameters()).data_ptr(),)
    hit = _CACHE.get(key)
    if hit is not None and hit[0] == ver and hit[1]() is dec:
        return hit[2]
    import weakref
    pk = {}
    cin = dec.conv_in.weight.shape[1]
    cp = (cin + 7) // 8 * 8
    kp = (9 * cp + 63) // 64 * 64
    pk["cin"] = (cp, kp, packing.pack_conv_in([dec.conv_in.weight], cp, kp), packing.pad_bias(dec.conv_in.bias))
    pk["mid_res"] = [_pack_res(r) for r in dec.mid_block.resnets]
    a = dec.mid_block.attentions[0]
    pk["attn"] = pack_attn(a)
    pk["up"] = []
    for blk in dec.up_blocks:
        up = None
        if blk.upsamplers is not None:
            up = (packing.pack_conv3x3(blk.upsamplers[0].conv.weight), _f32(blk.upsamplers[0].conv.bias))
        pk["up"].append(([_pack_res(r) for r in blk.resnets], up))
    pk["gno"] = (_f32(dec.conv_norm_out.weight), _f32(dec.conv_norm_out.bias), dec.conv_norm_out.eps)
    pk["cout"] = (packing.pack_conv3x3(dec.conv_out.weight), packing.pad_bias(dec.conv_out.bias), dec.conv_out.weight.shape[0])
    _CACHE[key] = (ver, weakref.ref(dec), pk)
    return pk


def _rows(M, C, dev):
    return torch.empty(M, C, dtype=torch.bfloat16, device=dev)


class _Scratch:
    """GroupNorm partial-sum scratch, grown on demand (one buffer per decode call)."""

    def __init__(self, dev):
        self.buf, self.dev = None, dev

    def get(self, n_img, S, C, ips):
        need = ops.groupnorm_scratch_floats(n_img, S, C, ips)
        if self.buf is None or self.buf.numel() < need:
            self.buf = torch.empty(max(need, 1 << 18), dtype=torch.float32, device=self.dev)
        return self.buf


_LIMIT = (1 << 32) - (1 << 24)        # the kernels address a tensor with 32-bit byte offsets


def _frame_batches(n, S, C):
    """Frame ranges whose [frames * S, C] bf16 slice stays below the 32-bit offset limit (per-frame ops only)."""
    per = max(1, min(n, _LIMIT // (S * C * 2)))
    return [(f0, min(n, f0 + per)) for f0 in range(0, n, per)]


def _res(pk, x, n, H, W, sc):
    """SpatioTemporalResBlock of the VAE (no time embedding): rows [n*H*W, cin] -> [.., cout]; n frames = one clip.
    The spatial half is per frame, so a tensor above 4 GiB (256 channels at 576x1024 with more than 14 frames) is
    processed in frame ranges through pointer-offset views; the temporal half needs the whole clip (its tensors have
    `cout` channels: 128 there)."""
    S, M = H * W, n * H * W
    cin, cout, dev = pk["cin"], pk["cout"], x.device
    geo = (H, W, H, W, 1, 0)
    xn = _rows(M, cin, dev)
    h = _rows(M, cout, dev)
    hn = _rows(M, cout, dev)
    res = _rows(M, cout, dev) if "wsc" in pk else x
    xs = _rows(M, cout, dev)
    for f0, f1 in _frame_batches(n, S, max(cin, cout)):
        r0, r1, k = f0 * S, f1 * S, f1 - f0
        ops.groupnorm(x[r0:r1], None, k, S, cin, 1, pk["g1"], pk["b1"], pk["eps_s"], True, xn[r0:r1], sc.get(k, S, cin, 1))
        ops.gemm(xn[r0:r1], pk["w1"], h[r0:r1], N=cout, cin=cin, taps=9, mode=1, conv=geo, bias=pk["cb1"])
        ops.groupnorm(h[r0:r1], None, k, S, cout, 1, pk["g2"], pk["b2"], pk["eps_s"], True, hn[r0:r1], sc.get(k, S, cout, 1))
        if "wsc" in pk:
            ops.gemm(x[r0:r1], pk["wsc"], res[r0:r1], N=cout, cin=cin, bias=pk["bsc"])
        ops.gemm(hn[r0:r1], pk["w2"], xs[r0:r1], N=cout, cin=cout, taps=9, mode=1, conv=geo, bias=pk["cb2"], R1=res[r0:r1])
    del xn, res
    # temporal res block on (1, C, n, H, W): GroupNorm statistics over (C/32, n, H, W), conv along the frames
    ops.groupnorm(xs, None, n, S, cout, n, pk["tg1"], pk["tb1"], pk["eps_t"], True, hn, sc.get(n, S, cout, n))
    ops.gemm(hn, pk["tw1"], h, N=cout, cin=cout, taps=3, mode=2, temporal=(n, S), bias=pk["tcb1"])
    ops.groupnorm(h, None, n, S, cout, n, pk["tg2"], pk["tb2"], pk["eps_t"], True, hn, sc.get(n, S, cout, n))
    out = h
    ops.gemm(hn, pk["tw2"], out, N=cout, cin=cout, taps=3, mode=2, temporal=(n, S), bias=pk["tcb2"], s_acc=pk["mix"], R1=xs)
    return out


def pack_attn(a):
    """Mid-block attention of the VAE (one head of dim C = 512).  q and k projections as separate GEMMs (the scores GEMM
    takes the k rows as its weight operand: contiguous [S, C]); the v projection is produced TRANSPOSED by a GEMM with
    swapped roles (V^T = W_v . t^T), and its bias joins the output projection's: rows of P sum to one, so
    P (V + 1 b_v^T) W_o^T = P V W_o^T + W_o b_v."""
    if a.heads != 1:
        raise ValueError("the HIP VAE attention is specialised for the SVD VAE's single head")
    wo = a.to_out[0].weight.detach().float()
    bo = a.to_out[0].bias.detach().float() + wo @ a.to_v.bias.detach().float()
    return dict(g=_f32(a.group_norm.weight), b=_f32(a.group_norm.bias), eps=a.group_norm.eps,
                wq=packing.pack_linear(a.to_q.weight), bq=_f32(a.to_q.bias),
                wk=packing.pack_linear(a.to_k.weight), bk=_f32(a.to_k.bias),
                wv_rows=a.to_v.weight.detach().to(torch.bfloat16).contiguous(),            # [C, C]: the A operand of V^T
                wo=packing.pack_linear(a.to_out[0].weight), bo=bo.contiguous())


def _attn(pk, x, n, H, W, sc):
    """VAE mid-block attention: GroupNorm -> q, k (bias), V^T -> per frame softmax(q k^T / sqrt(C)) v -> to_out -> + x, all on
    the HIP kernels: the scores are a GEMM with fp32 output (scale folded in), the softmax a row kernel (bf16 P), P.V a
    GEMM with K = S.  S must be a multiple of 64 (GEMM K granularity)."""
    S, M, C, dev = H * W, n * H * W, x.shape[1], x.device
    if S % 64 or S > 16384:
        raise ValueError(f"HIP VAE attention: {H}x{W} latent pixels per frame must be a multiple of 64 and <= 16384")
    t = _rows(M, C, dev)
    ops.groupnorm(x, None, n, S, C, 1, pk["g"], pk["b"], pk["eps"], False, t, sc.get(n, S, C, 1))
    q, k = _rows(M, C, dev), _rows(M, C, dev)
    ops.gemm(t, pk["wq"], q, N=C, cin=C, bias=pk["bq"])
    ops.gemm(t, pk["wk"], k, N=C, cin=C, bias=pk["bk"])
    o = _rows(M, C, dev)
    scores = torch.empty(S, S, dtype=torch.float32, device=dev)
    probs = torch.empty(S, S, dtype=torch.bfloat16, device=dev)
    vt = torch.empty(C, S, dtype=torch.bfloat16, device=dev)
    for f in range(n):
        r0, r1 = f * S, (f + 1) * S
        ops.gemm(q[r0:r1], k[r0:r1], scores, N=S, cin=C, s_acc=C ** -0.5, out_f32=True)      # q k^T / sqrt(C), fp32
        ops.softmax_rows(scores, probs)
        ops.gemm(pk["wv_rows"], t[r0:r1], vt, N=S, cin=C)                                     # V^T = W_v t^T  [C, S]
        ops.gemm(probs, vt, o[r0:r1], N=C, cin=S)                                             # P V
    del scores, probs, vt, q, k
    out = _rows(M, C, dev)
    ops.gemm(o, pk["wo"], out, N=C, cin=C, bias=pk["bo"], R1=x)
    return out


def supports(z, num_frames, dec=None):
    """Shapes the HIP decoder takes: bf16/fp16/fp32 CUDA latents, every intermediate tensor below 4 GiB, channel counts
    that are multiples of 64 (GEMM K granularity; the SVD VAE has 128 / 256 / 512)."""
    if not z.is_cuda or z.dim() != 4 or z.shape[0] % num_frames:
        return False
    # whole-clip tensors (the temporal halves): `cout` channels of every up block at its resolution
    levels = [(512, 1), (512, 2), (256, 4), (128, 8)]
    if dec is not None:
        chans = [m.weight.shape[0] for m in dec.modules() if isinstance(m, torch.nn.Conv2d) and m is not dec.conv_out]
        if any(c % 64 for c in chans) or len(dec.mid_block.attentions) != 1:
            return False
        levels = [(blk.resnets[0].spatial_res_block.conv1.weight.shape[0], 2 ** i) for i, blk in enumerate(dec.up_blocks)]
    h, w = z.shape[2], z.shape[3]
    if (h * w) % 64 or h * w > 16384:          # mid-block attention: scores GEMM with K = h * w, row softmax <= 16384 keys
        return False
    if dec is not None and dec.mid_block.attentions[0].heads != 1:
        return False
    return all(num_frames * h * w * f * f * c * 2 < _LIMIT for c, f in levels)


@torch.no_grad()
def decode(dec, z, num_frames):
    """dec: `TemporalDecoder` module (parameters on the device); z: (n, 4, h, w) latents already divided by the scaling
    factor; num_frames = frames per clip in z (the reference passes the chunk length).  Returns (n, 3, 8h, 8w) in z.dtype."""
    n_tot, _, h, w = z.shape
    if n_tot % num_frames:
        raise ValueError(f"decode: {n_tot} latent frames are not a multiple of num_frames={num_frames}")
    if not supports(z, num_frames, dec):
        raise ValueError(f"HIP VAE decode: a clip of {num_frames} frames at {8 * h}x{8 * w} exceeds the 4 GiB tensor limit "
                         "(at most 28 frames per chunk at 576x1024)")
    outs = []
    for c0 in range(0, n_tot, num_frames):                     # independent clips
        outs.append(_decode_clip(dec, z[c0:c0 + num_frames]))
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


def _decode_clip(dec, z):
    pk = _pack(dec)
    n, cz, H, W = z.shape
    dev = z.device
    sc = _Scratch(dev)
    M = n * H * W
    cp, kp, wci, bci = pk["cin"]
    x16 = torch.zeros(M, cp, dtype=torch.bfloat16, device=dev)
    ops.nchw_to_rows(z.contiguous(), x16, 0)
    col = _rows(M, kp, dev)
    ops.im2col3x3(x16, n, H, W, col)
    x = _rows(M, wci.shape[0], dev)
    ops.gemm(col, wci, x, N=wci.shape[0], cin=kp, bias=bci)
    del col, x16
    x = _res(pk["mid_res"][0], x, n, H, W, sc)
    for rp in pk["mid_res"][1:]:
        x = _attn(pk["attn"], x, n, H, W, sc)
        x = _res(rp, x, n, H, W, sc)
    for res_list, up in pk["up"]:
        for rp in res_list:
            x = _res(rp, x, n, H, W, sc)
        if up is not None:
            C = x.shape[1]
            y = _rows(n * 4 * H * W, C, dev)
            for f0, f1 in _frame_batches(n, 4 * H * W, C):                 # per-frame op: ranges below 4 GiB
                ops.gemm(x[f0 * H * W:f1 * H * W], up[0], y[f0 * 4 * H * W:f1 * 4 * H * W], N=C, cin=C, taps=9, mode=1,
                         conv=(H, W, 2 * H, 2 * W, 1, 1), bias=up[1])
            x, H, W = y, 2 * H, 2 * W
    M, C = n * H * W, x.shape[1]
    g, b, eps = pk["gno"]
    xn = _rows(M, C, dev)
    ops.groupnorm(x, None, n, H * W, C, 1, g, b, eps, True, xn, sc.get(n, H * W, C, 1))
    del x
    wco, bco, co = pk["cout"]
    co_p = (co + 3) // 4 * 4
    y = _rows(M, co_p, dev)
    ops.gemm(xn, wco, y, N=wco.shape[0], cin=C, taps=9, mode=1, conv=(H, W, H, W, 1, 0), bias=bco, n_store=co_p)
    del xn
    # time_conv_out: Conv3d(co, co, (3, 1, 1), padding (1, 0, 0)) over the clip's frames, fused with rows -> NCHW
    wt = dec.time_conv_out.weight.detach().float()[:, :, :, 0, 0].contiguous()          # [o, c, t]
    out = torch.empty(n, co, H, W, dtype=z.dtype if z.dtype in (torch.float32, torch.float16, torch.bfloat16)
                      else torch.float32, device=dev)
    ops.time_conv_rows_to_nchw(y, n, co, H * W, wt, dec.time_conv_out.bias.detach().float().contiguous(), out)
    return out.to(z.dtype)
