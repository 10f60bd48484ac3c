"""ORACLE (test infrastructure only -- never imported by the product path).

Plain-PyTorch CPU restatement of the diffusers==0.27.2 blocks that carry all of the
arithmetic of Ctrl-V's denoising hot path.  diffusers is a pinned third-party dependency of
the reference (/root/reference/requirements.txt:3) that is NOT vendored in /root/reference
and NOT installable here, so this file restates its published algorithm from SURVEY.md
Appendix A and anchors on the reference's own call sites:

  * get_down_block / UNetMidBlockSpatioTemporal ....... src/ctrlv/models/controlnet.py:9,157-170,186-192
  * Timesteps / TimestepEmbedding ..................... src/ctrlv/models/controlnet.py:11,111-117
  * parent UNet class (down/mid/up wiring) ............ src/ctrlv/models/unet_spatio_temporal_condition.py:4,13
  * BasicTransformerBlock ctor args (vendored copy) ... src/ctrlv/models/attention.py:220-236

PARITY UNPINNED: the reference ships no tests / golden vectors for this path and cannot be
imported in this container (no diffusers), so this restatement is pinned only by structural
known-answers (exact parameter counts 1 524 623 082 / 680 946 897, state-dict key layout)
and by the training-side scheduler formulas in tools/train_video_controlnet.py:405-410,468-471.

Everything here is deliberately dumb and literal: NCHW tensors, torch.nn layers,
F.scaled_dot_product_attention, explicit permutes exactly where diffusers does them.
Module / parameter names follow the diffusers state-dict layout.
"""
import contextlib
import math

import torch
import torch.nn.functional as F
from torch import nn

# --------------------------------------------------------------------------- storage-rounding emulation
# The HIP path keeps every activation it writes to HBM in bf16 (fp32 accumulation and statistics in between).  To
# separate "bf16 storage rounding" from "wrong arithmetic" the tests can run this fp32 restatement with a rounding at
# exactly those storage points: `store(x)` marks them and is the identity unless `storage_rounding()` is active.
# The marks sit where ctrlv_amd/models/blocks.py writes a kernel output; fused epilogues (residual adds, AlphaBlender,
# the temb / frame-embedding / cross-attention row vectors) are therefore rounded once, after the fused sum.
_STORE_DTYPE = [None]
_TRUNK_DTYPE = ["same"]
_STORE_ABSMAX = [None]      # [max |x| over every store() seen] while `store_absmax()` is active (fp16 overflow headroom)


def store(x, trunk=False):
    """Storage point of the HIP path.  trunk=True marks the RESIDUAL STREAM (block inputs / outputs, the tensors every
    branch is added back into); `storage_rounding(dtype, trunk_dtype=...)` can give it its own storage precision
    (tests/trunk_precision_study.py: what an fp32 residual trunk under bf16 branches would buy)."""
    dt = _TRUNK_DTYPE[0] if (trunk and _TRUNK_DTYPE[0] != "same") else _STORE_DTYPE[0]
    if _STORE_ABSMAX[0] is not None:
        _STORE_ABSMAX[0].append(float(x.detach().abs().max()))
    if callable(dt):           # a rounding function (tests/trunk_precision_study.py: split-plane trunk formats)
        return dt(x)
    return x if dt is None else x.to(dt).to(x.dtype)


@contextlib.contextmanager
def store_absmax():
    """Collects max |x| of every tensor that passes a store() mark (the values the HIP path writes to HBM): the headroom
    of an fp16 activation-storage build against 65 504 (tests/trunk_precision_study.py)."""
    prev = _STORE_ABSMAX[0]
    _STORE_ABSMAX[0] = log = []
    try:
        yield log
    finally:
        _STORE_ABSMAX[0] = prev


@contextlib.contextmanager
def storage_rounding(dtype=torch.bfloat16, trunk_dtype="same"):
    """trunk_dtype: "same" (default: like every other store), None (trunk kept in fp32), a torch dtype, or a function
    x -> rounded x (a storage format torch has no dtype for: hi + lo planes)."""
    prev = _STORE_DTYPE[0], _TRUNK_DTYPE[0]
    _STORE_DTYPE[0], _TRUNK_DTYPE[0] = dtype, trunk_dtype
    try:
        yield
    finally:
        _STORE_DTYPE[0], _TRUNK_DTYPE[0] = prev


# --------------------------------------------------------------------------- embeddings (A.2)
def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=True, downscale_freq_shift=0.0,
                           scale=1.0, max_period=10000):
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half_dim, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos=True, downscale_freq_shift=0.0):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)

    def forward(self, sample):
        return self.linear_2(store(self.act(self.linear_1(sample))))


# --------------------------------------------------------------------------- res blocks (A.3)
class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, stride=1, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1, stride=1, padding=0)

    def forward(self, input_tensor, temb):
        h = self.conv1(store(self.nonlinearity(self.norm1(input_tensor))))
        temb = self.time_emb_proj(store(self.nonlinearity(temb)))[:, :, None, None]
        h = store(h + temb)
        h = self.conv2(store(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            input_tensor = store(self.conv_shortcut(input_tensor), trunk=True)
        return store(input_tensor + h, trunk=True)


class TemporalResnetBlock(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv3d(in_channels, out_channels, (3, 1, 1), stride=1, padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv3d(out_channels, out_channels, (3, 1, 1), stride=1, padding=(1, 0, 0))
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = nn.Conv3d(in_channels, out_channels, 1, stride=1, padding=0)

    def forward(self, input_tensor, temb):
        h = self.conv1(store(self.nonlinearity(self.norm1(input_tensor))))  # GroupNorm on 5-D: stats over (C/32,F,H,W)
        temb = self.time_emb_proj(store(self.nonlinearity(temb)))[:, :, :, None, None]   # (B,F,C,1,1)
        temb = temb.permute(0, 2, 1, 3, 4)                                         # (B,C,F,1,1)
        h = store(h + temb)
        h = self.conv2(store(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return input_tensor + h


class AlphaBlender(nn.Module):
    """merge_strategy="learned_with_images", switch_spatial_to_temporal_mix=False."""

    def __init__(self, alpha):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.Tensor([alpha]))

    def get_alpha(self, image_only_indicator, ndims):
        alpha = torch.where(image_only_indicator.bool(),
                            torch.ones(1, 1, device=image_only_indicator.device),
                            torch.sigmoid(self.mix_factor)[..., None])
        if ndims == 5:      # (batch, channel, frames, height, width)
            alpha = alpha[:, None, :, None, None]
        elif ndims == 3:    # (batch*frames, height*width, channels)
            alpha = alpha.reshape(-1)[:, None, None]
        else:
            raise ValueError(f"Unexpected ndims {ndims}. Dimensions should be 3 or 5")
        return alpha

    def forward(self, x_spatial, x_temporal, image_only_indicator):
        alpha = self.get_alpha(image_only_indicator, x_spatial.ndim).to(x_spatial.dtype)
        return alpha * x_spatial + (1.0 - alpha) * x_temporal


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, eps=1e-6, temporal_eps=None, merge_factor=0.5):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(in_channels, out_channels, temb_channels, eps)
        self.temporal_res_block = TemporalResnetBlock(out_channels, out_channels, temb_channels,
                                                      temporal_eps if temporal_eps is not None else eps)
        self.time_mixer = AlphaBlender(alpha=merge_factor)

    def forward(self, hidden_states, temb, image_only_indicator):
        num_frames = image_only_indicator.shape[-1]
        hidden_states = self.spatial_res_block(hidden_states, temb)
        batch_frames, channels, height, width = hidden_states.shape
        batch_size = batch_frames // num_frames
        hidden_states_mix = (hidden_states[None, :].reshape(batch_size, num_frames, channels, height, width)
                             .permute(0, 2, 1, 3, 4))
        hidden_states = (hidden_states[None, :].reshape(batch_size, num_frames, channels, height, width)
                         .permute(0, 2, 1, 3, 4))
        temb = temb.reshape(batch_size, num_frames, -1)
        hidden_states = self.temporal_res_block(hidden_states, temb)
        hidden_states = store(self.time_mixer(x_spatial=hidden_states_mix, x_temporal=hidden_states,
                                              image_only_indicator=image_only_indicator), trunk=True)
        return hidden_states.permute(0, 2, 1, 3, 4).reshape(batch_frames, channels, height, width)


# --------------------------------------------------------------------------- attention (A.4)
class Attention(nn.Module):
    """diffusers Attention with AttnProcessor2_0 (F.scaled_dot_product_attention), no bias on q/k/v."""

    def __init__(self, query_dim, heads, dim_head, cross_attention_dim=None):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv_dim, inner, bias=False)
        self.to_v = nn.Linear(kv_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])

    def forward(self, hidden_states, encoder_hidden_states=None):
        batch = hidden_states.shape[0]
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q, k, v = store(self.to_q(hidden_states)), store(self.to_k(ctx)), store(self.to_v(ctx))
        hd = q.shape[-1] // self.heads
        q = q.view(batch, -1, self.heads, hd).transpose(1, 2)
        k = k.view(batch, -1, self.heads, hd).transpose(1, 2)
        v = v.view(batch, -1, self.heads, hd).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
        o = store(o.transpose(1, 2).reshape(batch, -1, self.heads * hd).to(q.dtype))
        return self.to_out[1](self.to_out[0](o))


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return store(h * F.gelu(gate))   # exact (erf) gelu


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4):
        super().__init__()
        inner = int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim_out)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5, elementwise_affine=True)
        self.attn1 = Attention(dim, heads, dim_head, cross_attention_dim=None)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5, elementwise_affine=True)
        self.attn2 = Attention(dim, heads, dim_head, cross_attention_dim=cross_attention_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5, elementwise_affine=True)
        self.ff = FeedForward(dim)

    def forward(self, hidden_states, encoder_hidden_states):
        hidden_states = self.attn1(store(self.norm1(hidden_states))) + hidden_states
        hidden_states = store(self.attn2(self.norm2(hidden_states), encoder_hidden_states) + hidden_states, trunk=True)
        hidden_states = store(self.ff(store(self.norm3(hidden_states))) + hidden_states, trunk=True)
        return hidden_states


class TemporalBasicTransformerBlock(nn.Module):
    def __init__(self, dim, time_mix_inner_dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.is_res = dim == time_mix_inner_dim
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = FeedForward(dim, dim_out=time_mix_inner_dim)
        self.norm1 = nn.LayerNorm(time_mix_inner_dim)
        self.attn1 = Attention(time_mix_inner_dim, heads, dim_head, cross_attention_dim=None)
        self.norm2 = nn.LayerNorm(time_mix_inner_dim)
        self.attn2 = Attention(time_mix_inner_dim, heads, dim_head, cross_attention_dim=cross_attention_dim)
        self.norm3 = nn.LayerNorm(time_mix_inner_dim)
        self.ff = FeedForward(time_mix_inner_dim)

    def forward(self, hidden_states, num_frames, encoder_hidden_states):
        batch_frames, seq_length, channels = hidden_states.shape
        batch_size = batch_frames // num_frames
        hidden_states = hidden_states[None, :].reshape(batch_size, num_frames, seq_length, channels)
        hidden_states = hidden_states.permute(0, 2, 1, 3)
        hidden_states = hidden_states.reshape(batch_size * seq_length, num_frames, channels)

        residual = hidden_states
        hidden_states = self.ff_in(store(self.norm_in(hidden_states)))
        if self.is_res:
            hidden_states = hidden_states + residual
        hidden_states = store(hidden_states, trunk=True)
        hidden_states = self.attn1(store(self.norm1(hidden_states)), encoder_hidden_states=None) + hidden_states
        hidden_states = store(self.attn2(self.norm2(hidden_states), encoder_hidden_states=encoder_hidden_states)
                              + hidden_states, trunk=True)
        ff_output = self.ff(store(self.norm3(hidden_states)))
        hidden_states = ff_output + hidden_states if self.is_res else ff_output

        hidden_states = hidden_states[None, :].reshape(batch_size, seq_length, num_frames, channels)
        hidden_states = hidden_states.permute(0, 2, 1, 3)
        return hidden_states.reshape(batch_size * num_frames, seq_length, channels)


class TransformerSpatioTemporalModel(nn.Module):
    """time_context_order: "sb" = diffusers 0.27.2 (context rows ordered (h*w, batch) while the temporal
    tokens are ordered (batch, h*w) -- SURVEY.md hard part H1); "bs" = the later upstream fix."""

    def __init__(self, num_attention_heads, attention_head_dim, in_channels, cross_attention_dim,
                 time_context_order="sb"):
        super().__init__()
        inner_dim = num_attention_heads * attention_head_dim
        self.in_channels = in_channels
        self.time_context_order = time_context_order
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner_dim)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner_dim, num_attention_heads, attention_head_dim, cross_attention_dim)])
        self.temporal_transformer_blocks = nn.ModuleList(
            [TemporalBasicTransformerBlock(inner_dim, inner_dim, num_attention_heads, attention_head_dim,
                                           cross_attention_dim)])
        self.time_pos_embed = TimestepEmbedding(in_channels, in_channels * 4, out_dim=in_channels)
        self.time_proj = Timesteps(in_channels, True, 0)
        self.time_mixer = AlphaBlender(alpha=0.5)
        self.proj_out = nn.Linear(inner_dim, in_channels)

    def forward(self, hidden_states, encoder_hidden_states, image_only_indicator):
        batch_frames, _, height, width = hidden_states.shape
        num_frames = image_only_indicator.shape[-1]
        batch_size = batch_frames // num_frames

        time_context = encoder_hidden_states
        time_context_first_timestep = time_context[None, :].reshape(
            batch_size, num_frames, -1, time_context.shape[-1])[:, 0]
        if self.time_context_order == "sb":
            time_context = time_context_first_timestep[None, :].broadcast_to(
                height * width, batch_size, 1, time_context.shape[-1])
            time_context = time_context.reshape(height * width * batch_size, 1, time_context.shape[-1])
        else:
            time_context = time_context_first_timestep[:, None].broadcast_to(
                batch_size, height * width, 1, time_context.shape[-1])
            time_context = time_context.reshape(batch_size * height * width, 1, time_context.shape[-1])

        residual = hidden_states
        hidden_states = store(self.norm(hidden_states))
        inner_dim = hidden_states.shape[1]
        hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(batch_frames, height * width, inner_dim)
        hidden_states = store(self.proj_in(hidden_states), trunk=True)

        num_frames_emb = torch.arange(num_frames, device=hidden_states.device)
        num_frames_emb = num_frames_emb.repeat(batch_size, 1).reshape(-1)
        t_emb = store(self.time_proj(num_frames_emb).to(dtype=hidden_states.dtype))
        emb = self.time_pos_embed(t_emb)[:, None, :]

        for block, temporal_block in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            hidden_states = block(hidden_states, encoder_hidden_states=encoder_hidden_states)
            hidden_states_mix = hidden_states + emb
            hidden_states_mix = temporal_block(hidden_states_mix, num_frames=num_frames,
                                               encoder_hidden_states=time_context)
            hidden_states = store(self.time_mixer(x_spatial=hidden_states, x_temporal=hidden_states_mix,
                                                  image_only_indicator=image_only_indicator), trunk=True)

        hidden_states = self.proj_out(hidden_states)
        hidden_states = hidden_states.reshape(batch_frames, height, width, inner_dim).permute(0, 3, 1, 2).contiguous()
        return store(hidden_states + residual, trunk=True)


# --------------------------------------------------------------------------- resampling (A.5, a8)
class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return store(self.conv(x), trunk=True)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        dtype = x.dtype
        if dtype == torch.bfloat16:       # diffusers up-casts bf16 around F.interpolate
            x = x.to(torch.float32)
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        if dtype == torch.bfloat16:
            x = x.to(dtype)
        return store(self.conv(x), trunk=True)


# --------------------------------------------------------------------------- block wiring (A.5)
class CrossAttnDownBlockSpatioTemporal(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, num_attention_heads,
                 cross_attention_dim, add_downsample, time_context_order="sb"):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.attentions = nn.ModuleList()
        for i in range(num_layers):
            self.resnets.append(SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels,
                                                       temb_channels, eps=1e-6))
            self.attentions.append(TransformerSpatioTemporalModel(
                num_attention_heads, out_channels // num_attention_heads, out_channels, cross_attention_dim,
                time_context_order))
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, hidden_states, temb, encoder_hidden_states, image_only_indicator):
        output_states = ()
        for resnet, attn in zip(self.resnets, self.attentions):
            hidden_states = resnet(hidden_states, temb, image_only_indicator)
            hidden_states = attn(hidden_states, encoder_hidden_states, image_only_indicator)
            output_states = output_states + (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states = output_states + (hidden_states,)
        return hidden_states, output_states


class DownBlockSpatioTemporal(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels, temb_channels, eps=1e-5)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, hidden_states, temb, image_only_indicator):
        output_states = ()
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb, image_only_indicator)
            output_states = output_states + (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states = output_states + (hidden_states,)
        return hidden_states, output_states


class UNetMidBlockSpatioTemporal(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, num_attention_heads, cross_attention_dim, num_layers=1,
                 time_context_order="sb"):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(in_channels, in_channels, temb_channels, eps=1e-5)])
        self.attentions = nn.ModuleList()
        for _ in range(num_layers):
            self.attentions.append(TransformerSpatioTemporalModel(
                num_attention_heads, in_channels // num_attention_heads, in_channels, cross_attention_dim,
                time_context_order))
            self.resnets.append(SpatioTemporalResBlock(in_channels, in_channels, temb_channels, eps=1e-5))

    def forward(self, hidden_states, temb, encoder_hidden_states, image_only_indicator):
        hidden_states = self.resnets[0](hidden_states, temb, image_only_indicator)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            hidden_states = attn(hidden_states, encoder_hidden_states, image_only_indicator)
            hidden_states = resnet(hidden_states, temb, image_only_indicator)
        return hidden_states


class UpBlockSpatioTemporal(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList()
        for i in range(num_layers):
            res_skip_channels = in_channels if (i == num_layers - 1) else out_channels
            resnet_in_channels = prev_output_channel if i == 0 else out_channels
            self.resnets.append(SpatioTemporalResBlock(resnet_in_channels + res_skip_channels, out_channels,
                                                       temb_channels, eps=1e-6))
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb, image_only_indicator):
        for resnet in self.resnets:
            res_hidden_states = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res_hidden_states], dim=1)
            hidden_states = resnet(hidden_states, temb, image_only_indicator)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


class CrossAttnUpBlockSpatioTemporal(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers,
                 num_attention_heads, cross_attention_dim, add_upsample, time_context_order="sb"):
        super().__init__()
        self.resnets = nn.ModuleList()
        self.attentions = nn.ModuleList()
        for i in range(num_layers):
            res_skip_channels = in_channels if (i == num_layers - 1) else out_channels
            resnet_in_channels = prev_output_channel if i == 0 else out_channels
            self.resnets.append(SpatioTemporalResBlock(resnet_in_channels + res_skip_channels, out_channels,
                                                       temb_channels, eps=1e-6))
            self.attentions.append(TransformerSpatioTemporalModel(
                num_attention_heads, out_channels // num_attention_heads, out_channels, cross_attention_dim,
                time_context_order))
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb, encoder_hidden_states, image_only_indicator):
        for resnet, attn in zip(self.resnets, self.attentions):
            res_hidden_states = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res_hidden_states], dim=1)
            hidden_states = resnet(hidden_states, temb, image_only_indicator)
            hidden_states = attn(hidden_states, encoder_hidden_states, image_only_indicator)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states
