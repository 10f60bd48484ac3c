"""Autograd over the HIP kernels -- first slice of the training path (BASELINE config 5; the reference's training step is
tools/train_video_controlnet.py:451-488: ControlNet forward + backward with fp32 master parameters under bf16 compute).

Scope of this slice ("smallest verifiable slice"): the gather-GEMM family (nn.Linear / 1x1 conv
incl. the ControlNet zero-convs, 3x3 Conv2d, (3,1,1) Conv3d) with its fused epilogue operands {bias, residual R1,
per-clip row vector V, s_acc}, GroupNorm(+SiLU) in its 4-D and 5-D forms, and the folded AlphaBlender -- i.e. a complete
`SpatioTemporalResBlock` and the zero-convs -- checked against torch.autograd on the oracle (tests/test_backward_gpu.py).

  dgrad  = the forward kernel on role-swapped weights (Linear: W^T; convs: taps reversed, channels transposed)
  wgrad  = ctrlv_gemm_wgrad (transposed-LDS-read MFMA kernel), bias / row-vector gradients = ctrlv_colsum
  norms  = ctrlv_groupnorm_bwd on the statistics the forward saved
Activations and activation gradients are bf16 rows; parameter gradients are fp32 in the PyTorch layouts.  Attention,
LayerNorm and GEGLU backward, strided / upsampling conv dgrad and DDP bucketing are the next steps.
"""
import math

import torch

from . import ops, packing


def _rows(M, C, like):
    return torch.empty(M, C, dtype=torch.bfloat16, device=like.device)


class GatherGemm(torch.autograd.Function):
    """out = s_acc * (gather-GEMM(A, weight) + bias) + R1 + V[(m // vdiv)]   (what the res block's convs fuse).

    weight / bias: parameters in the PyTorch layout ([N, C], [N, C, 3, 3] or [N, C, 3, 1, 1]), any float dtype.
    geom: dict(mode, conv=(H, W, Ho, Wo, 1, 0) | None, temporal=(F, S) | None, vdiv)."""

    @staticmethod
    def _pack(weight, mode):
        if mode == 0:
            return packing.pack_linear(weight)
        return packing.pack_conv3x3(weight) if mode == 1 else packing.pack_conv_temporal(weight)

    @staticmethod
    def forward(ctx, A, weight, bias, R1, V, s_acc, geom):
        mode = geom["mode"]
        N, cin = weight.shape[0], weight.shape[1]
        taps = {0: 1, 1: 9, 2: 3}[mode]
        out = _rows(A.shape[0], N, A)
        ops.gemm(A, GatherGemm._pack(weight, mode), out, N=(N + 31) // 32 * 32, cin=cin, taps=taps, mode=mode,
                 conv=geom.get("conv"), temporal=geom.get("temporal"),
                 bias=None if bias is None else packing.pad_bias(bias), R1=R1, s_acc=float(s_acc),
                 V=V, vmode=1 if V is not None else 0, vdiv=geom.get("vdiv", 1))
        ctx.save_for_backward(A, weight)
        ctx.geom, ctx.s_acc, ctx.has = geom, float(s_acc), (bias is not None, R1 is not None, V is not None)
        ctx.vshape = None if V is None else tuple(V.shape)
        return out

    @staticmethod
    def backward(ctx, dY):
        A, weight = ctx.saved_tensors
        has_bias, has_r1, has_v = ctx.has
        need = ctx.needs_input_grad
        dY = dY.contiguous()
        dA, dW, db = gemm_grads(A, weight, dY, ctx.geom, ctx.s_acc, need[0], need[1], has_bias and need[2])
        dR1 = dY if (has_r1 and need[3]) else None
        dV = None
        if has_v and need[4]:
            dV = torch.zeros(ctx.vshape, dtype=torch.float32, device=A.device)
            ops.colsum(dY, dV, vmode=1, vdiv=ctx.geom.get("vdiv", 1), vmod=ctx.vshape[0])
        return dA, dW, db, dR1, dV, None, None


def gemm_grads(A, weight, dY, geom, s_acc, need_dA=True, need_dW=True, need_db=True):
    """(dA, dW, dbias) of out = s_acc * (gather-GEMM(A, weight) + bias) for the upstream gradient dY (bf16 rows)."""
    mode = geom["mode"]
    N, cin = weight.shape[0], weight.shape[1]
    taps = {0: 1, 1: 9, 2: 3}[mode]
    dA = dW = db = None
    if need_dA:
        # dgrad: the forward kernel with the weight's roles swapped
        if mode == 0:
            wt = packing.pack_linear(weight.detach().reshape(N, cin).t())
        elif mode == 1:
            wt = packing.pack_conv3x3(weight.detach().flip(2, 3).transpose(0, 1))
        else:
            wt = packing.pack_conv_temporal(weight.detach().flip(2).transpose(0, 1))
        dA = _rows(A.shape[0], cin, A)
        ops.gemm(dY, wt, dA, N=(cin + 31) // 32 * 32, cin=N, taps=taps, mode=mode, conv=geom.get("conv"),
                 temporal=geom.get("temporal"), s_acc=s_acc)
    if need_dW:
        dWp = torch.zeros(N, taps * cin, dtype=torch.float32, device=A.device)
        ops.gemm_wgrad(A, dY, dWp, N=N, cin=cin, taps=taps, mode=mode, conv=geom.get("conv"),
                       temporal=geom.get("temporal"))
        if mode == 0:
            dW = dWp.reshape(weight.shape)
        elif mode == 1:
            dW = dWp.reshape(N, 3, 3, cin).permute(0, 3, 1, 2)
        else:
            dW = dWp.reshape(N, 3, cin).permute(0, 2, 1).reshape(N, cin, 3, 1, 1)
        dW = (dW * s_acc).to(weight.dtype)
    if need_db:
        db = torch.zeros(N, dtype=torch.float32, device=A.device)
        ops.colsum(dY, db, scale=s_acc)
        db = db.to(weight.dtype)
    return dA, dW, db


class GroupNormSiLU(torch.autograd.Function):
    """y = [silu](GroupNorm32(x)) on channels-last rows; imgs_per_stat = 1 (4-D) or F (5-D statistics)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, n_img, S, imgs_per_stat, eps, silu):
        C = x.shape[1]
        part = torch.empty(ops.groupnorm_scratch_floats(n_img, S, C, imgs_per_stat), dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        ops.groupnorm(x, None, n_img, S, C, imgs_per_stat, g32, b32, eps, silu, y, part)
        ctx.save_for_backward(x, g32, b32, part)
        ctx.cfg = (n_img, S, C, imgs_per_stat, silu, gamma.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g32, b32, part = ctx.saved_tensors
        n_img, S, C, ips, silu, pdt = ctx.cfg
        dx = torch.empty_like(x)
        dg = torch.zeros(C, dtype=torch.float32, device=x.device)
        db = torch.zeros(C, dtype=torch.float32, device=x.device)
        ops.groupnorm_bwd(x, dy.contiguous(), n_img, S, C, ips, part, g32, b32, silu, dx, dg, db)
        return dx, dg.to(pdt), db.to(pdt), None, None, None, None, None


class BlendGemm(torch.autograd.Function):
    """AlphaBlender folded into the last temporal conv: out = xs + (1 - a) * (conv(hn) + bias), a = sigmoid(mix_factor)
    (a * xs + (1 - a) * (xs + conv) of SURVEY A.3).  Gradients for hn, the conv parameters, xs AND mix_factor."""

    @staticmethod
    def forward(ctx, hn, weight, bias, xs, mix_factor, geom):
        a = 1.0 / (1.0 + math.exp(-float(mix_factor.detach().float().cpu())))
        out = GatherGemm.apply(hn, weight, bias, xs, None, 1.0 - a, geom)        # (no graph: forward runs under no_grad)
        ctx.save_for_backward(hn, weight, xs, out, mix_factor)
        ctx.geom, ctx.a = geom, a
        return out

    @staticmethod
    def backward(ctx, dY):
        hn, weight, xs, out, mix = ctx.saved_tensors
        a, geom = ctx.a, ctx.geom
        dY = dY.contiguous()
        need = ctx.needs_input_grad
        dhn, dw, db = gemm_grads(hn, weight, dY, geom, 1.0 - a, need[0], need[1], need[2])
        dmix = None
        if need[4]:
            # dL/da = -sum dY * (conv + bias) = -sum dY * (out - xs) / (1 - a);  da/dmix = a (1 - a)
            acc = torch.zeros(1, dtype=torch.float32, device=dY.device)
            ops.dot_diff(dY, out, xs, acc, scale=-a)
            dmix = acc.to(mix.dtype).reshape(mix.shape)
        return dhn, dw, db, (dY if need[3] else None), dmix, None


class LayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x [+ V[(m // vdiv) % vmod]]) over the channel axis (eps 1e-5); V = the frame positional embedding
    table of TransformerSpatioTemporalModel (fp32 [F, C]) or None."""

    @staticmethod
    def forward(ctx, x, gamma, beta, V, vdiv, vmod):
        y = torch.empty_like(x)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        ops.layernorm(x, g32, b32, 1e-5, y, V=V, vdiv=vdiv, vmod=vmod)
        ctx.save_for_backward(x, g32, V if V is not None else torch.empty(0, device=x.device))
        ctx.cfg = (V is not None, vdiv, vmod, gamma.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g32, V = ctx.saved_tensors
        has_v, vdiv, vmod, pdt = ctx.cfg
        C = x.shape[1]
        dx = torch.empty_like(x)
        dg = torch.zeros(C, dtype=torch.float32, device=x.device)
        db = torch.zeros(C, dtype=torch.float32, device=x.device)
        ops.layernorm_bwd(x, dy.contiguous(), g32, 1e-5, dx, dg, db, V=V if has_v else None, vdiv=vdiv, vmod=vmod)
        dV = None
        if has_v and ctx.needs_input_grad[3]:
            dV = torch.zeros_like(V)
            ops.colsum(dx, dV, vmode=1, vdiv=vdiv, vmod=vmod)
        return dx, dg.to(pdt), db.to(pdt), dV, None, None


class GegluProj(torch.autograd.Function):
    """u = a * gelu_erf(g), (a | g) = x @ W^T + b   (diffusers GEGLU: `proj` Linear(C -> 2 I), chunk, exact gelu) -- the
    forward is ONE GEMM with the GEGLU epilogue; the backward recomputes the raw projection (activation recompute instead
    of storing an [M, 2I] tensor), applies ctrlv_geglu_bwd and reuses the GEMM family for dgrad / wgrad."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        two_i, cin = weight.shape
        wp, bp = packing.pack_geglu(weight, bias)
        u = _rows(x.shape[0], two_i // 2, x)
        ops.gemm(x, wp, u, N=two_i, cin=wp.shape[1], bias=bp, geglu=1)
        ctx.save_for_backward(x, weight, bias)
        return u

    @staticmethod
    def backward(ctx, du):
        x, weight, bias = ctx.saved_tensors
        two_i, cin = weight.shape
        inner = two_i // 2
        wi = packing.geglu_interleave(weight.detach())                 # rows in the packed (value, gate) block order
        wp, bp = packing.pack_geglu(weight, bias)
        raw = _rows(x.shape[0], two_i, x)
        ops.gemm(x, wp, raw, N=two_i, cin=wp.shape[1], bias=bp)        # activation recompute (no GEGLU epilogue)
        draw = torch.empty_like(raw)
        ops.geglu_bwd(raw, du.contiguous(), draw)
        dx, dwi, dbi = gemm_grads(x, wi, draw, dict(mode=0), 1.0, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                  ctx.needs_input_grad[2])
        # un-interleave the rows: interleaved row ((r >> 4) * 2 + is_gate) * 16 + (r & 15)  <-  original row r (+ inner)
        idx = torch.arange(two_i, device=x.device)
        isg, r = (idx >= inner).long(), idx % inner
        src = ((r >> 4) * 2 + isg) * 16 + (r & 15)
        dw = dwi[src] if dwi is not None else None
        db = dbi[src] if dbi is not None else None
        return dx, dw, db


def res_block_train_forward(block, x, temb_tables, B, F, H, W):
    """Training-mode forward of a `ctrlv_amd.models.blocks.SpatioTemporalResBlock` (no skip-concat input) built from the
    autograd functions above; same kernels, same fusion as `block.run` (bit-identical output), but every op records
    what its backward needs.  x: bf16 rows [B*F*H*W, cin] (requires_grad for dgrad); temb_tables: (fp32 [B, cout],
    fp32 [B, cout]) = time_emb_proj(silu(emb)) of the spatial / temporal half (per-clip vectors: computed by the caller,
    with torch autograd if their gradients are wanted)."""
    s, t = block.spatial_res_block, block.temporal_res_block
    N, S = B * F, H * W
    g2d = dict(mode=1, conv=(H, W, H, W, 1, 0), vdiv=F * S)
    g3d = dict(mode=2, temporal=(F, S), vdiv=F * S)
    xn = GroupNormSiLU.apply(x, s.norm1.weight, s.norm1.bias, N, S, 1, block.eps, True)
    h = GatherGemm.apply(xn, s.conv1.weight, s.conv1.bias, None, temb_tables[0], 1.0, g2d)
    hn = GroupNormSiLU.apply(h, s.norm2.weight, s.norm2.bias, N, S, 1, block.eps, True)
    res = x
    if s.conv_shortcut is not None:
        res = GatherGemm.apply(x, s.conv_shortcut.weight.reshape(block.cout, block.cin), s.conv_shortcut.bias, None, None,
                               1.0, dict(mode=0))
    xs = GatherGemm.apply(hn, s.conv2.weight, s.conv2.bias, res, None, 1.0, g2d)
    hn = GroupNormSiLU.apply(xs, t.norm1.weight, t.norm1.bias, N, S, F, block.eps, True)
    h = GatherGemm.apply(hn, t.conv1.weight, t.conv1.bias, None, temb_tables[1], 1.0, g3d)
    hn = GroupNormSiLU.apply(h, t.norm2.weight, t.norm2.bias, N, S, F, block.eps, True)
    return BlendGemm.apply(hn, t.conv2.weight, t.conv2.bias, xs, block.time_mixer.mix_factor, g3d)


def zero_conv_train_forward(conv, x, scale=1.0):
    """A ControlNet zero-conv (1x1) times conditioning_scale (controlnet.py:331-344) with gradients."""
    C = conv.weight.shape[0]
    return GatherGemm.apply(x, conv.weight.reshape(C, -1), conv.bias, None, None, float(scale), dict(mode=0))
