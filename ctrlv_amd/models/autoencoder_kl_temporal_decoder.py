"""AutoencoderKLTemporalDecoder (the SVD VAE) as a plain PyTorch-ROCm module.

The reference takes this model from diffusers==0.27.2 (`tools/train_video_controlnet.py:94-97`,
`AutoencoderKLTemporalDecoder.from_pretrained(..., subfolder="vae")`) and calls it OUTSIDE the per-step hot path:
once per clip to encode the conditioning image / the bbox frames (`.encode(x).latent_dist.mode()`,
`pipeline_video_control.py:71-101,235`) and once to decode the final latents in chunks
(`.decode(z, num_frames=n).sample`, `:346`).  north_star keeps the VAE on PyTorch-ROCm, so this is ordinary `torch.nn`
code (MIOpen / rocBLAS / SDPA) -- except that `decode` on a HIP device executes the TemporalDecoder's parameters through
the gather-GEMM / GroupNorm kernels of libctrlv_hip.so (vae_decoder_hip.py; the torch forward stays as the reference and
the CPU path).  It exists so that `StableVideoControlPipeline.from_pretrained` is
self-sufficient on a machine without diffusers.  Module and parameter names follow the diffusers state-dict layout, so
`vae/diffusion_pytorch_model.safetensors` of SVD-XT loads by name.

PARITY UNPINNED: restated from the published diffusers architecture (Encoder / UNetMidBlock2D / TemporalDecoder /
MidBlockTemporalDecoder / UpBlockTemporalDecoder / SpatioTemporalResBlock with merge_strategy="learned",
switch_spatial_to_temporal_mix=True); diffusers is not installable here, so the only pins are structural
(tests/test_host_logic.py: state-dict key layout, the encoder's 34 163 592 parameters -- the Stable Diffusion VAE
encoder it is identical to --, shapes, determinism).
"""
import types

import torch
import torch.nn.functional as F
from torch import nn

from .modeling_utils import HipModelMixin


class ResnetBlock2D(nn.Module):
    """temb-free ResnetBlock2D (groups 32, SiLU, eps 1e-6)."""

    def __init__(self, in_channels, out_channels, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class TemporalResnetBlock(nn.Module):
    """Conv3d (3,1,1) residual block over (B, C, F, H, W), no time embedding."""

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, channels, eps=eps)
        self.conv1 = nn.Conv3d(channels, channels, (3, 1, 1), padding=(1, 0, 0))
        self.norm2 = nn.GroupNorm(32, channels, eps=eps)
        self.conv2 = nn.Conv3d(channels, channels, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return x + h


class AlphaBlender(nn.Module):
    """merge_strategy="learned", switch_spatial_to_temporal_mix=True: alpha = 1 - sigmoid(mix_factor)."""

    def __init__(self, alpha=0.0):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.Tensor([alpha]))

    def forward(self, x_spatial, x_temporal):
        alpha = 1.0 - torch.sigmoid(self.mix_factor).to(x_spatial.dtype)
        return alpha * x_spatial + (1.0 - alpha) * x_temporal


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(in_channels, out_channels, eps=1e-6)
        self.temporal_res_block = TemporalResnetBlock(out_channels, eps=1e-5)
        self.time_mixer = AlphaBlender(0.0)

    def forward(self, x, num_frames):
        x = self.spatial_res_block(x)
        bf, c, h, w = x.shape
        x5 = x.reshape(bf // num_frames, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        y5 = self.time_mixer(x5, self.temporal_res_block(x5))
        return y5.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class Attention(nn.Module):
    """Single-head spatial self-attention of the VAE mid blocks (GroupNorm in, residual out, biases on)."""

    def __init__(self, channels, head_dim):
        super().__init__()
        self.heads = channels // head_dim
        self.group_norm = nn.GroupNorm(32, channels, eps=1e-6)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x.reshape(b, c, h * w)).transpose(1, 2)
        q, k, v = (f(t).reshape(b, h * w, self.heads, c // self.heads).transpose(1, 2)
                   for f in (self.to_q, self.to_k, self.to_v))
        o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, h * w, c)
        o = self.to_out[0](o).transpose(1, 2).reshape(b, c, h, w)
        return o + x


class Downsample2D(nn.Module):
    """use_conv=True, padding=0: asymmetric (0,1,0,1) zero pad, then 3x3 stride 2."""

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        dt = x.dtype
        if dt == torch.bfloat16:
            x = x.float()
        x = F.interpolate(x, scale_factor=2.0, mode="nearest").to(dt)
        return self.conv(x)


class _Container(nn.Module):
    pass


class Encoder(nn.Module):
    def __init__(self, in_channels, latent_channels, block_out_channels, layers_per_block):
        super().__init__()
        boc = tuple(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, ch in enumerate(boc):
            inp, out = out, ch
            blk = _Container()
            blk.resnets = nn.ModuleList([ResnetBlock2D(inp if j == 0 else out, out) for j in range(layers_per_block)])
            blk.downsamplers = nn.ModuleList([Downsample2D(out)]) if i != len(boc) - 1 else None
            self.down_blocks.append(blk)
        self.mid_block = _Container()
        self.mid_block.attentions = nn.ModuleList([Attention(boc[-1], boc[-1])])
        self.mid_block.resnets = nn.ModuleList([ResnetBlock2D(boc[-1], boc[-1]), ResnetBlock2D(boc[-1], boc[-1])])
        self.conv_norm_out = nn.GroupNorm(32, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for blk in self.down_blocks:
            for r in blk.resnets:
                x = r(x)
            if blk.downsamplers is not None:
                x = blk.downsamplers[0](x)
        x = self.mid_block.resnets[0](x)
        x = self.mid_block.attentions[0](x)
        x = self.mid_block.resnets[1](x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class TemporalDecoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block):
        super().__init__()
        boc = tuple(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, boc[-1], 3, padding=1)
        self.mid_block = _Container()
        self.mid_block.resnets = nn.ModuleList([SpatioTemporalResBlock(boc[-1], boc[-1]) for _ in range(layers_per_block)])
        self.mid_block.attentions = nn.ModuleList([Attention(boc[-1], boc[-1])])
        self.up_blocks = nn.ModuleList()
        rev = boc[::-1]
        out = rev[0]
        for i, ch in enumerate(rev):
            prev, out = out, ch
            blk = _Container()
            blk.resnets = nn.ModuleList([SpatioTemporalResBlock(prev if j == 0 else out, out)
                                         for j in range(layers_per_block + 1)])
            blk.upsamplers = nn.ModuleList([Upsample2D(out)]) if i != len(rev) - 1 else None
            self.up_blocks.append(blk)
        self.conv_norm_out = nn.GroupNorm(32, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)
        self.time_conv_out = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, z, num_frames):
        x = self.conv_in(z)
        x = self.mid_block.resnets[0](x, num_frames)
        for resnet, attn in zip(self.mid_block.resnets[1:], self.mid_block.attentions):
            x = attn(x)
            x = resnet(x, num_frames)
        for blk in self.up_blocks:
            for r in blk.resnets:
                x = r(x, num_frames)
            if blk.upsamplers is not None:
                x = blk.upsamplers[0](x)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        bf, c, h, w = x.shape
        x = x.reshape(bf // num_frames, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        x = self.time_conv_out(x)
        return x.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class DiagonalGaussianDistribution:
    def __init__(self, parameters):
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise


class AutoencoderKLTemporalDecoder(HipModelMixin):
    _class_name = "AutoencoderKLTemporalDecoder"

    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                 block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, sample_size=768,
                 scaling_factor=0.18215, force_upcast=True):
        super().__init__()
        if any(t != "DownEncoderBlock2D" for t in down_block_types) or len(down_block_types) != len(block_out_channels):
            raise ValueError("AutoencoderKLTemporalDecoder: down_block_types must be DownEncoderBlock2D, one per "
                             "entry of block_out_channels")
        self.register_to_config(in_channels=in_channels, out_channels=out_channels,
                                down_block_types=tuple(down_block_types), block_out_channels=tuple(block_out_channels),
                                layers_per_block=layers_per_block, latent_channels=latent_channels,
                                sample_size=sample_size, scaling_factor=scaling_factor, force_upcast=force_upcast)
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block)
        self.decoder = TemporalDecoder(latent_channels, out_channels, block_out_channels, layers_per_block)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)

    def encode(self, x, return_dict=True):
        import os
        from . import vae_encoder_hip as ve
        if (os.environ.get("CTRLV_VAE_HIP", "1") != "0" and x.is_cuda and not torch.is_grad_enabled()
                and x.dtype in (torch.bfloat16, torch.float16, torch.float32) and ve.supports(x, self.encoder)):
            moments = ve.encode(self.encoder, x)           # HIP kernels (vae_encoder_hip.py)
        else:
            moments = self.encoder(x)
        dist = DiagonalGaussianDistribution(self.quant_conv(moments))
        return types.SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    def decode(self, z, num_frames, return_dict=True):
        if z.shape[0] % num_frames:
            raise ValueError(f"decode: {z.shape[0]} latent frames are not a multiple of num_frames={num_frames}")
        # the decoder runs on the HIP kernels when it can (vae_decoder_hip.py: ~30x the MIOpen path at 576x1024);
        # CTRLV_VAE_HIP=0, CPU tensors, autograd or > 4 GiB intermediates use the torch modules below
        from . import vae_decoder_hip as vh
        import os
        if (os.environ.get("CTRLV_VAE_HIP", "1") != "0" and z.is_cuda and not torch.is_grad_enabled()
                and vh.supports(z, num_frames, self.decoder)):
            sample = vh.decode(self.decoder, z, num_frames)
        else:
            sample = self.decoder(z, num_frames)
        return types.SimpleNamespace(sample=sample) if return_dict else (sample,)

    def forward(self, sample, sample_posterior=False, generator=None, num_frames=1):
        dist = self.encode(sample).latent_dist
        z = dist.sample(generator) if sample_posterior else dist.mode()
        return self.decode(z, num_frames=num_frames)
