"""Spatio-temporal UNet blocks executed by libctrlv_hip.so.

The module tree mirrors the diffusers==0.27.2 state-dict layout so real SVD / Ctrl-V checkpoints
load by name; torch.nn layers are used ONLY as parameter containers -- their forward is never called.  Each block's
`run()` issues HIP kernels over channels-last bf16 rows [n_img*H*W, C]:

  SpatioTemporalResBlock  (reference call sites: controlnet.py:157 get_down_block, :186 UNetMidBlockSpatioTemporal)
      GN+SiLU -> conv3x3(+temb) -> GN+SiLU -> conv3x3(+shortcut) -> GN5D+SiLU -> conv(3,1,1)(+temb) -> GN5D+SiLU
      -> conv(3,1,1) with the AlphaBlender folded into its epilogue: out = xs + (1-a)*(conv2(h)+b)
  TransformerSpatioTemporalModel
      20 kernels, see `run`; LayerNorm-2 / to_q / softmax of both 1-key CLIP cross-attentions are dead compute
      (softmax over one key == 1), so they reduce to a per-clip row vector added in the to_out epilogue.

Dimension padding rules of the gather-GEMM (Cin % 64, N % 32) hold for every SVD width (320/640/1280 and their
concats); only the tiny test config needs the K zero-padding of the time-embedding inputs.
"""
import math
import os

import torch
from torch import nn

from .. import ops, packing


class FwdCtx:
    """Per-forward state shared by all blocks."""

    def __init__(self, ws, B, F, temb, xattn, time_context_order):
        self.ws, self.B, self.F = ws, B, F
        self.temb = temb                  # fp32 [B, sum Cout]; blocks hold their column offset
        self.xattn = xattn                # fp32 [B, sum C];   attentions hold their column offset
        self.quirk = time_context_order == "sb"
        self.gn_part = None               # shared fp32 scratch for GroupNorm partial sums
        self.gn_cross = None              # chunk partials a res block's last GEMM wrote for the transformer behind it
        self.gn_cross_valid = False       # (plan.hip Ctx::gn_cross)
        self.trace = None                 # debugging aid: list collecting (block, output rows clone, H, W)


def _sigmoid(x):
    return 1.0 / (1.0 + math.exp(-x))


# ------------------------------------------------------------------------------------------- parameter containers
class _TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)


class _ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(32, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None


class _TemporalResnetBlock(nn.Module):
    def __init__(self, c, temb, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, c, eps=eps)
        self.conv1 = nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb, c)
        self.norm2 = nn.GroupNorm(32, c, eps=eps)
        self.conv2 = nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0))


class _AlphaBlender(nn.Module):
    def __init__(self, alpha=0.5):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.Tensor([alpha]))


class _Attention(nn.Module):
    def __init__(self, query_dim, heads, dim_head, cross_attention_dim=None):
        super().__init__()
        inner = heads * dim_head
        kv = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv, inner, bias=False)
        self.to_v = nn.Linear(kv, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])


class _GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class _FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None):
        super().__init__()
        self.net = nn.ModuleList([_GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim_out or dim)])


class _BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = _Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = _Attention(dim, heads, dim_head, cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = _FeedForward(dim)


class _TemporalBasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim):
        super().__init__()
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = _FeedForward(dim, dim)
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = _Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = _Attention(dim, heads, dim_head, cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = _FeedForward(dim)


def _f32(t):
    return t.detach().float().contiguous()


def _lo(t):
    """lo plane of a split trunk tensor (Workspace.trunk), None for plain tensors"""
    return getattr(t, "lo", None) if t is not None else None


def _trk_epi(epi, out):
    """the lo planes of an epilogue's trunk operands, next to them (ops.gemm / ops.ff_fused / ops.temporal_fused keywords)"""
    e = dict(epi)
    for k in ("R1", "R2"):
        if _lo(e.get(k)) is not None:
            e[k + "_lo"] = _lo(e[k])
    if _lo(out) is not None:
        e["out_lo"] = _lo(out)
    return e


_FF_FUSED = os.environ.get("CTRLV_FF_FUSED", "1") != "0"      # the plan's switch (csrc/plan.hip ff_pair)
_TEMPORAL_FUSED = os.environ.get("CTRLV_TEMPORAL_FUSED", "1") != "0"     # the plan's switch (csrc/plan.hip run_tr)


def _ff_pair(ws, x, ffp, ubox, out, C, rows_per_image=0, **epi):
    """u = GEGLU(x); out = epilogue(u @ wout^T).  ffp = (wproj, bproj, wout, bout[, w1f, w2f]): at C = 320 one fused launch
    that keeps u on chip (ops.ff_fused, when it serves the epilogue: csrc/ff_fused.hip), else the two GEMMs.  `ubox` = [u or None]: the 4C-wide intermediate of the two-launch path is
    allocated from the arena on first need (csrc/plan.hip ff_pair)."""
    wproj, bproj, wout, bout = ffp[:4]
    epi = _trk_epi(epi, out)
    if _FF_FUSED and len(ffp) == 6 and ops.ff_fused_serves(x, out, **epi):
        ops.ff_fused(x, ffp[4], ffp[5], out, bias=bout, **epi)
        return
    M = x.shape[0]
    if ubox[0] is None:
        ubox[0] = ws.alloc((M, 4 * C))
    u = ubox[0]
    ops.gemm(x, wproj, u, N=8 * C, cin=C, bias=bproj, geglu=1)
    ops.gemm(u, wout, out, N=C, cin=4 * C, bias=bout, rows_per_image=rows_per_image, **epi)


_FF_LN = os.environ.get("CTRLV_FF_LN", "0") not in ("", "0")   # opt-in, the plan's switch (csrc/plan.hip ln_ff)


def _ln_ff(ws, xraw, ln, t, ffp, ubox, out, C, ln_V=None, ln_vdiv=1, ln_vmod=1 << 30, rows_per_image=0, **epi):
    """out = epilogue(FF(LayerNorm(xraw + ln_V))): with the fused kernel the norm is folded into its prologue, else
    ops.layernorm into `t` followed by _ff_pair (csrc/plan.hip ln_ff)."""
    if _FF_FUSED and _FF_LN and len(ffp) == 6 and _lo(xraw) is None and ops.ff_fused_serves(xraw, out, **_trk_epi(epi, out)):
        ops.ff_fused(xraw, ffp[4], ffp[5], out, bias=ffp[3], ln=(ln[0], ln[1], 1e-5), ln_V=ln_V, ln_vdiv=ln_vdiv,
                     ln_vmod=ln_vmod, **_trk_epi(epi, out))
        return
    if ln_V is not None:
        ops.layernorm(xraw, ln[0], ln[1], 1e-5, t, V=ln_V, vdiv=ln_vdiv, vmod=ln_vmod, x_lo=_lo(xraw))
    else:
        ops.layernorm(xraw, ln[0], ln[1], 1e-5, t, x_lo=_lo(xraw))
    _ff_pair(ws, t, ffp, ubox, out, C, rows_per_image=rows_per_image, **epi)


def _gn_scratch(ctx, n_img, S, C, ips):
    need = ops.groupnorm_scratch_floats(n_img, S, C, ips)
    if ctx.gn_part is None or ctx.gn_part.numel() < need:
        ctx.gn_part = torch.empty(max(need, 1 << 18), dtype=torch.float32, device=ctx.ws.device)
    return ctx.gn_part


def _gemm_groupnorm(ctx, A, W, out, gkw, n_img, S, C, ips, gamma, beta, eps, y):
    """`gemm` whose output goes straight into GroupNorm + SiLU (plan.hip gemm_groupnorm): where the launch serves it, its
    epilogue writes the norm's chunk partials and the norm is finalize + apply."""
    gkw = _trk_epi(gkw, out)
    if S % 64 == 0 and ops.gemm_gn_partials_serves(A, W, out, **gkw):
        need = ops.groupnorm_fused_scratch_floats(n_img, S, ips)
        if ctx.gn_part is None or ctx.gn_part.numel() < need:
            ctx.gn_part = torch.empty(max(need, 1 << 18), dtype=torch.float32, device=ctx.ws.device)
        ops.gemm(A, W, out, gn_partials=ctx.gn_part, **gkw)
        ops.groupnorm_from_partials(out, n_img, S, C, ips, gamma, beta, eps, True, y, ctx.gn_part, x_lo=_lo(out))
    else:
        ops.gemm(A, W, out, **gkw)
        ops.groupnorm(out, None, n_img, S, C, ips, gamma, beta, eps, True, y, _gn_scratch(ctx, n_img, S, C, ips),
                      x_lo=_lo(out))


# ------------------------------------------------------------------------------------------- res block
class SpatioTemporalResBlock(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, eps=1e-6):
        super().__init__()
        self.cin, self.cout, self.eps = in_channels, out_channels, eps
        self.spatial_res_block = _ResnetBlock2D(in_channels, out_channels, temb_channels, eps)
        self.temporal_res_block = _TemporalResnetBlock(out_channels, temb_channels, eps)
        self.time_mixer = _AlphaBlender(0.5)
        self.temb_off = None      # column offsets into ctx.temb (assigned by the model's pack())
        self._pk = None

    def temb_projections(self):
        return [self.spatial_res_block.time_emb_proj, self.temporal_res_block.time_emb_proj]

    def pack(self):
        s, t = self.spatial_res_block, self.temporal_res_block
        pk = dict(
            g1=_f32(s.norm1.weight), b1=_f32(s.norm1.bias), w1=packing.pack_conv3x3(s.conv1.weight),
            cb1=_f32(s.conv1.bias),
            g2=_f32(s.norm2.weight), b2=_f32(s.norm2.bias), w2=packing.pack_conv3x3(s.conv2.weight),
            cb2=_f32(s.conv2.bias),
            tg1=_f32(t.norm1.weight), tb1=_f32(t.norm1.bias), tw1=packing.pack_conv_temporal(t.conv1.weight),
            tcb1=_f32(t.conv1.bias),
            tg2=_f32(t.norm2.weight), tb2=_f32(t.norm2.bias), tw2=packing.pack_conv_temporal(t.conv2.weight),
            tcb2=_f32(t.conv2.bias),
            alpha=_sigmoid(float(self.time_mixer.mix_factor.detach().float().cpu())),
        )
        if s.conv_shortcut is not None:
            pk["wsc"] = packing.pack_linear(s.conv_shortcut.weight)
            pk["bsc"] = _f32(s.conv_shortcut.bias)
        self._pk = pk

    def run(self, ctx, x, H, W, x2=None, feeds_norm=False):
        """x (| x2 on channels): rows [B*F*H*W, cin] -> rows [.., cout].  feeds_norm: the next op is a transformer, whose
        opening GroupNorm takes its statistics from this block's last GEMM where that launch serves them."""
        pk, ws = self._pk, ctx.ws
        B, F = ctx.B, ctx.F
        N, S = B * F, H * W
        M = N * S
        cin, cout = self.cin, self.cout
        out = ws.trunk((M, cout))
        mk = ws.mark()
        part = _gn_scratch(ctx, N, S, max(cin, cout), 1)
        xn = ws.alloc((M, cin))
        ops.groupnorm(x, x2, N, S, cin, 1, pk["g1"], pk["b1"], self.eps, True, xn, part, x_lo=_lo(x), x2_lo=_lo(x2))
        h = ws.alloc((M, cout))
        vs = ctx.temb[:, self.temb_off[0]:]
        hn = ws.alloc((M, cout))
        _gemm_groupnorm(ctx, xn, pk["w1"], h, dict(N=cout, cin=cin, taps=9, mode=1, conv=(H, W, H, W, 1, 0), bias=pk["cb1"],
                                                   V=vs, vmode=1, vdiv=F * S),
                        N, S, cout, 1, pk["g2"], pk["b2"], self.eps, hn)
        if "wsc" in pk:
            res = ws.trunk((M, cout))
            ops.gemm(x, pk["wsc"], res, N=cout, cin=cin, A2=x2, c_split=x.shape[1] if x2 is not None else 0,
                     bias=pk["bsc"], out_lo=_lo(res))
        else:
            res = x
        xs = ws.trunk((M, cout))
        # temporal res block on (B, C, F, H, W): GroupNorm statistics over (C/32, F, H, W), conv along F
        _gemm_groupnorm(ctx, hn, pk["w2"], xs, dict(N=cout, cin=cout, taps=9, mode=1, conv=(H, W, H, W, 1, 0), bias=pk["cb2"],
                                                    R1=res),
                        N, S, cout, F, pk["tg1"], pk["tb1"], self.eps, hn)
        vt = ctx.temb[:, self.temb_off[1]:]
        _gemm_groupnorm(ctx, hn, pk["tw1"], h, dict(N=cout, cin=cout, taps=3, mode=2, temporal=(F, S), bias=pk["tcb1"],
                                                    V=vt, vmode=1, vdiv=F * S),
                        N, S, cout, F, pk["tg2"], pk["tb2"], self.eps, hn)
        # AlphaBlender: a*xs + (1-a)*(xs + conv2) = xs + (1-a)*conv2
        kw2 = _trk_epi(dict(N=cout, cin=cout, taps=3, mode=2, temporal=(F, S), bias=pk["tcb2"], s_acc=1.0 - pk["alpha"], R1=xs),
                       out)
        ctx.gn_cross_valid = False
        if feeds_norm and S % 64 == 0 and ops.gemm_gn_partials_serves(hn, pk["tw2"], out, **kw2):
            need = ops.groupnorm_fused_scratch_floats(N, S, 1)
            if ctx.gn_cross is None or ctx.gn_cross.numel() < need:
                ctx.gn_cross = torch.empty(need, dtype=torch.float32, device=ws.device)
            kw2["gn_partials"] = ctx.gn_cross
            ctx.gn_cross_valid = True
        ops.gemm(hn, pk["tw2"], out, **kw2)
        ws.release(mk)
        if ctx.trace is not None:
            ctx.trace.append((self, out.clone(), H, W))
        return out


# ------------------------------------------------------------------------------------------- transformer
class TransformerSpatioTemporalModel(nn.Module):
    def __init__(self, num_attention_heads, attention_head_dim, in_channels, cross_attention_dim):
        super().__init__()
        if attention_head_dim != 64:
            raise ValueError(f"ctrlv_amd attention kernels are specialised for head_dim 64 (got {attention_head_dim})")
        inner = num_attention_heads * attention_head_dim
        if inner != in_channels:
            raise ValueError("TransformerSpatioTemporalModel: inner_dim must equal in_channels (SVD configuration)")
        self.C, self.heads, self.cross_dim = in_channels, num_attention_heads, cross_attention_dim
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [_BasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim)])
        self.temporal_transformer_blocks = nn.ModuleList(
            [_TemporalBasicTransformerBlock(inner, num_attention_heads, attention_head_dim, cross_attention_dim)])
        self.time_pos_embed = _TimestepEmbedding(in_channels, in_channels * 4, out_dim=in_channels)
        self.time_mixer = _AlphaBlender(0.5)
        self.proj_out = nn.Linear(inner, in_channels)
        self.xattn_off = None     # column offsets (spatial, temporal) into ctx.xattn
        self._pk = None
        self._frame_emb = {}      # F -> fp32 [F, C] (input-independent constant of the weights)

    def cross_attentions(self):
        return [self.transformer_blocks[0].attn2, self.temporal_transformer_blocks[0].attn2]

    def pack(self):
        sb, tb = self.transformer_blocks[0], self.temporal_transformer_blocks[0]

        def ff(f):
            w, b = packing.pack_geglu(f.net[0].proj.weight, f.net[0].proj.bias)
            w2 = packing.pack_linear(f.net[2].weight)
            if tuple(w.shape) == (2560, 320) and tuple(w2.shape) == (320, 1280) and w.is_cuda:
                return (w, b, w2, _f32(f.net[2].bias)) + ops.ff_fused_pack(w.contiguous(), b.float().contiguous(), w2.contiguous())
            return w, b, w2, _f32(f.net[2].bias)

        def ln(n):
            return _f32(n.weight), _f32(n.bias)

        pk = dict(
            gn=(_f32(self.norm.weight), _f32(self.norm.bias)),
            pin=(packing.pack_linear(self.proj_in.weight), _f32(self.proj_in.bias)),
            pout=(packing.pack_linear(self.proj_out.weight), _f32(self.proj_out.bias)),
            s_ln1=ln(sb.norm1), s_ln3=ln(sb.norm3),
            s_qkv=packing.pack_qkv(sb.attn1.to_q.weight, sb.attn1.to_k.weight, sb.attn1.to_v.weight),
            s_o=(packing.pack_linear(sb.attn1.to_out[0].weight), _f32(sb.attn1.to_out[0].bias)),
            s_ff=ff(sb.ff),
            t_lnin=ln(tb.norm_in), t_ln1=ln(tb.norm1), t_ln3=ln(tb.norm3),
            t_ffin=ff(tb.ff_in),
            t_qkv=packing.pack_qkv(tb.attn1.to_q.weight, tb.attn1.to_k.weight, tb.attn1.to_v.weight),
            t_o=(packing.pack_linear(tb.attn1.to_out[0].weight), _f32(tb.attn1.to_out[0].bias)),
            t_ff=ff(tb.ff),
            t_wf=None,
            tpe=(packing.pack_linear(self.time_pos_embed.linear_1.weight), _f32(self.time_pos_embed.linear_1.bias),
                 packing.pack_linear(self.time_pos_embed.linear_2.weight), _f32(self.time_pos_embed.linear_2.bias)),
            alpha=_sigmoid(float(self.time_mixer.mix_factor.detach().float().cpu())),
        )
        if tuple(pk["t_qkv"].shape) == (960, 320) and tuple(pk["t_o"][0].shape) == (320, 320) and pk["t_qkv"].is_cuda:
            pk["t_wf"] = ops.temporal_fused_pack(pk["t_qkv"].contiguous(), pk["t_o"][0].contiguous())   # (csrc/plan.hip load_tr)
        self._pk = pk
        self._frame_emb = {}

    def frame_embedding(self, F, device):
        """time_pos_embed(time_proj(arange(F))) -> fp32 [F, C]; depends on the weights and F only."""
        if F not in self._frame_emb:
            C, pk = self.C, self._pk
            el = pk["tpe"][0].dtype
            t = torch.arange(F, dtype=torch.float32, device=device)
            kp = pk["tpe"][0].shape[1]
            te = torch.zeros(F, kp, dtype=el, device=device)
            if kp == C:
                ops.timestep_embedding(t, C, te)
            else:
                tmp = torch.empty(F, C, dtype=el, device=device)
                ops.timestep_embedding(t, C, tmp)
                te[:, :C] = tmp
            h = torch.empty(F, 4 * C, dtype=el, device=device)
            ops.gemm(te, pk["tpe"][0], h, N=4 * C, cin=kp, bias=pk["tpe"][1], act=1)
            e = torch.empty(F, C, dtype=torch.float32, device=device)
            ops.gemm(h, pk["tpe"][2], e, N=C, cin=4 * C, bias=pk["tpe"][3], out_f32=True)
            self._frame_emb[F] = e
        return self._frame_emb[F]

    def run(self, ctx, x, H, W):
        pk, ws = self._pk, ctx.ws
        B, F, C = ctx.B, ctx.F, self.C
        N, S = B * F, H * W
        M = N * S
        emb = self.frame_embedding(F, x.device)
        out = ws.trunk((M, C))
        mk = ws.mark()
        part = _gn_scratch(ctx, N, S, C, 1)
        t = ws.alloc((M, C))
        if ctx.gn_cross_valid:       # statistics from the res block's last GEMM (SpatioTemporalResBlock.run, feeds_norm)
            ctx.gn_cross_valid = False
            ops.groupnorm_from_partials(x, N, S, C, 1, pk["gn"][0], pk["gn"][1], 1e-6, False, t, ctx.gn_cross, x_lo=_lo(x))
        else:
            ops.groupnorm(x, None, N, S, C, 1, pk["gn"][0], pk["gn"][1], 1e-6, False, t, part, x_lo=_lo(x))
        h0 = ws.trunk((M, C))
        ops.gemm(t, pk["pin"][0], h0, N=C, cin=C, bias=pk["pin"][1], out_lo=_lo(h0))
        # ---- spatial BasicTransformerBlock
        ops.layernorm(h0, pk["s_ln1"][0], pk["s_ln1"][1], 1e-5, t, x_lo=_lo(h0))
        qkv = ws.alloc((M, 3 * C))
        ops.gemm(t, pk["s_qkv"], qkv, N=3 * C, cin=C, n_scale2=C, s_acc2=ops.Q_PRESCALE)   # q block pre-scaled
        a = ws.alloc((M, C))
        ops.attention_spatial(qkv, a, N, S, C, prescaled=True)
        h1 = ws.trunk((M, C))
        xs_vec = ctx.xattn[:, self.xattn_off[0]:]      # attn2 with one key == to_out(to_v(ehs[b])) for every query
        ops.gemm(a, pk["s_o"][0], h1, N=C, cin=C, bias=pk["s_o"][1], **_trk_epi(dict(R1=h0, V=xs_vec, vmode=1, vdiv=F * S), h1))
        u = [None]                                      # 4C-wide GEGLU output: allocated by _ff_pair on first need
        h2 = h0                                         # h0 is dead from here on
        _ln_ff(ws, h1, pk["s_ln3"], t, pk["s_ff"], u, h2, C, rows_per_image=S, R1=h1)
        # ---- temporal block on tokens (b, s) x frames; rows stay ordered (b, f, s)
        g0 = h1                                         # h1 is dead
        _ln_ff(ws, h2, pk["t_lnin"], t, pk["t_ffin"], u, g0, C, ln_V=emb, ln_vdiv=S, ln_vmod=F, rows_per_image=S, R1=h2, V=emb, vmode=1, vdiv=S,
               vmod=F)
        g1 = ws.trunk((M, C))
        xt_vec = ctx.xattn[:, self.xattn_off[1]:]
        # diffusers 0.27.2: time_context rows ordered (s, b), tokens ordered (b, s)
        vkw = dict(vmode=2, vdiv=F * S, vS=S, vmod=B) if (ctx.quirk and B > 1) else dict(vmode=1, vdiv=F * S)
        # (split trunk: the in-kernel LayerNorm normalises the hi plane of g0, as in the plan)
        fkw = _trk_epi(dict(bias=pk["t_o"][1], R1=g0, V=xt_vec, ln=(pk["t_ln1"][0], pk["t_ln1"][1], 1e-5), **vkw), g1)
        if _TEMPORAL_FUSED and ops.temporal_fused_serves(g0, pk["t_wf"], g1, B, F, S, **fkw):
            # norm1 + attn1 over the frames + residual + the cross-attention vector in ONE launch (csrc/plan.hip run_tr)
            ops.temporal_fused(g0, pk["t_wf"], g1, B, F, S, **fkw)
        else:
            ops.layernorm(g0, pk["t_ln1"][0], pk["t_ln1"][1], 1e-5, t, x_lo=_lo(g0))
            ops.gemm(t, pk["t_qkv"], qkv, N=3 * C, cin=C)
            ops.attention_temporal(qkv, a, B, F, S, C)
            ops.gemm(a, pk["t_o"][0], g1, N=C, cin=C, bias=pk["t_o"][1], **_trk_epi(dict(R1=g0, V=xt_vec, **vkw), g1))
        # AlphaBlender folded: h3 = a*h2 + (1-a)*(g1 + ff)
        al = pk["alpha"]
        h3 = g0
        _ln_ff(ws, g1, pk["t_ln3"], t, pk["t_ff"], u, h3, C, rows_per_image=S, s_acc=1.0 - al, R1=g1, s1=1.0 - al, R2=h2, s2=al)
        ops.gemm(h3, pk["pout"][0], out, N=C, cin=C, bias=pk["pout"][1], **_trk_epi(dict(R1=x), out))
        ws.release(mk)
        if ctx.trace is not None:
            ctx.trace.append((self, out.clone(), H, W))
        return out


# ------------------------------------------------------------------------------------------- resampling
class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.C = channels
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)
        self._pk = None

    def pack(self):
        self._pk = (packing.pack_conv3x3(self.conv.weight), _f32(self.conv.bias))

    def run(self, ctx, x, H, W):
        N = ctx.B * ctx.F
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        out = ctx.ws.trunk((N * Ho * Wo, self.C))
        ops.gemm(x, self._pk[0], out, N=self.C, cin=self.C, taps=9, mode=1, conv=(H, W, Ho, Wo, 2, 0),
                 bias=self._pk[1], out_lo=_lo(out))
        return out, Ho, Wo


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.C = channels
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)
        self._pk = None

    def pack(self):
        self._pk = (packing.pack_conv3x3(self.conv.weight), _f32(self.conv.bias))

    def run(self, ctx, x, H, W):
        """nearest x2 upsample fused into the conv's gather (source pixel = (y>>1, x>>1))."""
        N = ctx.B * ctx.F
        out = ctx.ws.trunk((N * 4 * H * W, self.C))
        ops.gemm(x, self._pk[0], out, N=self.C, cin=self.C, taps=9, mode=1, conv=(H, W, 2 * H, 2 * W, 1, 1),
                 bias=self._pk[1], out_lo=_lo(out))
        return out, 2 * H, 2 * W


# ------------------------------------------------------------------------------------------- block wiring (A.5)
class CrossAttnDownBlockSpatioTemporal(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, num_attention_heads,
                 cross_attention_dim, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels,
                                                             temb_channels, eps=1e-6) for i in range(num_layers)])
        self.attentions = nn.ModuleList([TransformerSpatioTemporalModel(
            num_attention_heads, out_channels // num_attention_heads, out_channels, cross_attention_dim)
            for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def run(self, ctx, x, H, W):
        taps = []
        for resnet, attn in zip(self.resnets, self.attentions):
            x = resnet.run(ctx, x, H, W, feeds_norm=True)
            x = attn.run(ctx, x, H, W)
            taps.append((x, H, W))
        if self.downsamplers is not None:
            x, H, W = self.downsamplers[0].run(ctx, x, H, W)
            taps.append((x, H, W))
        return x, H, W, taps


class DownBlockSpatioTemporal(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels,
                                                             temb_channels, eps=1e-5) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def run(self, ctx, x, H, W):
        taps = []
        for resnet in self.resnets:
            x = resnet.run(ctx, x, H, W)
            taps.append((x, H, W))
        if self.downsamplers is not None:
            x, H, W = self.downsamplers[0].run(ctx, x, H, W)
            taps.append((x, H, W))
        return x, H, W, taps


class UNetMidBlockSpatioTemporal(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, num_attention_heads, cross_attention_dim, num_layers=1):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(in_channels, in_channels, temb_channels, eps=1e-5)])
        self.attentions = nn.ModuleList()
        for _ in range(num_layers):
            self.attentions.append(TransformerSpatioTemporalModel(
                num_attention_heads, in_channels // num_attention_heads, in_channels, cross_attention_dim))
            self.resnets.append(SpatioTemporalResBlock(in_channels, in_channels, temb_channels, eps=1e-5))

    def run(self, ctx, x, H, W):
        x = self.resnets[0].run(ctx, x, H, W, feeds_norm=len(self.attentions) > 0)
        for i, (attn, resnet) in enumerate(zip(self.attentions, self.resnets[1:])):
            x = attn.run(ctx, x, H, W)
            x = resnet.run(ctx, x, H, W, feeds_norm=i + 1 < len(self.attentions))
        return x


class _UpBase(nn.Module):
    def _build(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers, add_upsample):
        self.resnets = nn.ModuleList()
        for i in range(num_layers):
            res_skip_channels = in_channels if (i == num_layers - 1) else out_channels
            resnet_in_channels = prev_output_channel if i == 0 else out_channels
            self.resnets.append(SpatioTemporalResBlock(resnet_in_channels + res_skip_channels, out_channels,
                                                       temb_channels, eps=1e-6))
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None


class UpBlockSpatioTemporal(_UpBase):
    has_cross_attention = False

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers, add_upsample):
        super().__init__()
        self._build(in_channels, prev_output_channel, out_channels, temb_channels, num_layers, add_upsample)

    def run(self, ctx, x, H, W, skips):
        for resnet in self.resnets:
            skip = skips.pop()                       # torch.cat([hidden, skip], dim=1) is read in place by GN / GEMM
            x = resnet.run(ctx, x, H, W, x2=skip)
        if self.upsamplers is not None:
            x, H, W = self.upsamplers[0].run(ctx, x, H, W)
        return x, H, W


class CrossAttnUpBlockSpatioTemporal(_UpBase):
    has_cross_attention = True

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers,
                 num_attention_heads, cross_attention_dim, add_upsample):
        super().__init__()
        self._build(in_channels, prev_output_channel, out_channels, temb_channels, num_layers, add_upsample)
        self.attentions = nn.ModuleList([TransformerSpatioTemporalModel(
            num_attention_heads, out_channels // num_attention_heads, out_channels, cross_attention_dim)
            for _ in range(num_layers)])

    def run(self, ctx, x, H, W, skips):
        for resnet, attn in zip(self.resnets, self.attentions):
            skip = skips.pop()
            x = resnet.run(ctx, x, H, W, x2=skip, feeds_norm=True)
            x = attn.run(ctx, x, H, W)
        if self.upsamplers is not None:
            x, H, W = self.upsamplers[0].run(ctx, x, H, W)
        return x, H, W
