"""CPU oracle of the SVD VAE (AutoencoderKLTemporalDecoder of diffusers==0.27.2: autoencoder_kl_temporal_decoder.py,
vae.py `Encoder`, unet_3d_blocks.py `MidBlockTemporalDecoder` / `UpBlockTemporalDecoder`, resnet.py
`SpatioTemporalResBlock` / `TemporalResnetBlock`, attention_processor.py `Attention` with `AttnProcessor2_0`).

TEST INFRASTRUCTURE ONLY (oracle/ctrlv_ref/__init__.py): the checker of ctrlv_amd's HIP VAE paths (tests/test_vae_gpu.py).
PARITY UNPINNED: diffusers is absent from the image and the reference repository holds no VAE vectors; this is a
restatement of the published architecture, written independently of ctrlv_amd/models/autoencoder_kl_temporal_decoder.py
(that one is an nn.Module tree for checkpoint loading; this one is a set of pure functions over a diffusers-layout state
dict) -- tests/test_oracle.py checks that the two agree on the same weights.

Call sites in the reference: vae.encode(x).latent_dist.mode() (pipeline_video_control.py:71-101,235) and
vae.decode(z, num_frames=n).sample (pipeline_video_control.py:346).  All arithmetic in the dtype of the given tensors
(fp32 for the oracle's use).
"""
import math

import torch
import torch.nn.functional as F


def _gn(sd, key, x, eps):
    return F.group_norm(x, 32, sd[key + ".weight"], sd[key + ".bias"], eps)


def _conv(sd, key, x, **kw):
    return F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], **kw)


def resnet2d(sd, key, x):
    """ResnetBlock2D without time embedding: GN(eps 1e-6)+SiLU, 3x3, GN+SiLU, 3x3, (1x1 shortcut), residual, scale 1."""
    h = _conv(sd, key + ".conv1", F.silu(_gn(sd, key + ".norm1", x, 1e-6)), padding=1)
    h = _conv(sd, key + ".conv2", F.silu(_gn(sd, key + ".norm2", h, 1e-6)), padding=1)
    if key + ".conv_shortcut.weight" in sd:
        x = _conv(sd, key + ".conv_shortcut", x)
    return x + h


def temporal_resnet(sd, key, x5):
    """TemporalResnetBlock on (B, C, F, H, W): GN(eps 1e-5)+SiLU, (3,1,1) conv, GN+SiLU, (3,1,1) conv, residual."""
    def c3(k, t):
        return F.conv3d(t, sd[k + ".weight"], sd[k + ".bias"], padding=(1, 0, 0))
    h = c3(key + ".conv1", F.silu(_gn(sd, key + ".norm1", x5, 1e-5)))
    h = c3(key + ".conv2", F.silu(_gn(sd, key + ".norm2", h, 1e-5)))
    return x5 + h


def spatio_temporal_resblock(sd, key, x, num_frames):
    """SpatioTemporalResBlock of the temporal decoder: merge_strategy "learned", switch_spatial_to_temporal_mix=True, i.e.
    alpha = 1 - sigmoid(mix_factor); out = alpha * spatial + (1 - alpha) * temporal."""
    x = resnet2d(sd, key + ".spatial_res_block", x)
    bf, c, h, w = x.shape
    x5 = x.reshape(bf // num_frames, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
    t5 = temporal_resnet(sd, key + ".temporal_res_block", x5)
    alpha = 1.0 - torch.sigmoid(sd[key + ".time_mixer.mix_factor"]).to(x.dtype)
    y5 = alpha * x5 + (1.0 - alpha) * t5
    return y5.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


def attention(sd, key, x):
    """Single-head self-attention over the h*w tokens of every frame (head dim = channels), GroupNorm(eps 1e-6) on the
    way in, residual on the way out, scale 1/sqrt(head_dim); written out explicitly (no fused SDPA kernel)."""
    b, c, h, w = x.shape
    t = _gn(sd, key + ".group_norm", x.reshape(b, c, h * w), 1e-6).transpose(1, 2)          # (b, hw, c)
    q = F.linear(t, sd[key + ".to_q.weight"], sd[key + ".to_q.bias"])
    k = F.linear(t, sd[key + ".to_k.weight"], sd[key + ".to_k.bias"])
    v = F.linear(t, sd[key + ".to_v.weight"], sd[key + ".to_v.bias"])
    p = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(c), dim=-1)
    o = F.linear(p @ v, sd[key + ".to_out.0.weight"], sd[key + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(b, c, h, w) + x


def _count(sd, prefix):
    n = 0
    while any(k.startswith(f"{prefix}.{n}.") for k in sd):
        n += 1
    return n


def encode_moments(sd, x, quant=True):
    """`Encoder` (the Stable Diffusion VAE encoder) + quant_conv: (n, 3, H, W) in [-1, 1] -> moments (n, 8, H/8, W/8);
    latent_dist.mode() is the first half of the channels.  quant=False: the encoder's output before quant_conv."""
    x = _conv(sd, "encoder.conv_in", x, padding=1)
    nb = _count(sd, "encoder.down_blocks")
    for i in range(nb):
        for j in range(_count(sd, f"encoder.down_blocks.{i}.resnets")):
            x = resnet2d(sd, f"encoder.down_blocks.{i}.resnets.{j}", x)
        dk = f"encoder.down_blocks.{i}.downsamplers.0.conv"
        if dk + ".weight" in sd:                               # Downsample2D, padding 0: pad (0, 1, 0, 1), stride 2
            x = _conv(sd, dk, F.pad(x, (0, 1, 0, 1)), stride=2)
    x = resnet2d(sd, "encoder.mid_block.resnets.0", x)
    x = attention(sd, "encoder.mid_block.attentions.0", x)
    x = resnet2d(sd, "encoder.mid_block.resnets.1", x)
    x = _conv(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.conv_norm_out", x, 1e-6)), padding=1)
    return _conv(sd, "quant_conv", x) if quant else x


def decode(sd, z, num_frames):
    """`TemporalDecoder`: (n, 4, h, w) latents of whole clips (n a multiple of num_frames) -> (n, 3, 8h, 8w) frames."""
    x = _conv(sd, "decoder.conv_in", z, padding=1)
    x = spatio_temporal_resblock(sd, "decoder.mid_block.resnets.0", x, num_frames)
    for j in range(1, _count(sd, "decoder.mid_block.resnets")):
        x = attention(sd, f"decoder.mid_block.attentions.{j - 1}", x)
        x = spatio_temporal_resblock(sd, f"decoder.mid_block.resnets.{j}", x, num_frames)
    for i in range(_count(sd, "decoder.up_blocks")):
        for j in range(_count(sd, f"decoder.up_blocks.{i}.resnets")):
            x = spatio_temporal_resblock(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, num_frames)
        uk = f"decoder.up_blocks.{i}.upsamplers.0.conv"
        if uk + ".weight" in sd:                               # Upsample2D: nearest x2, then 3x3
            x = _conv(sd, uk, F.interpolate(x, scale_factor=2.0, mode="nearest"), padding=1)
    x = _conv(sd, "decoder.conv_out", F.silu(_gn(sd, "decoder.conv_norm_out", x, 1e-6)), padding=1)
    bf, c, h, w = x.shape
    x5 = x.reshape(bf // num_frames, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
    x5 = F.conv3d(x5, sd["decoder.time_conv_out.weight"], sd["decoder.time_conv_out.bias"], padding=(1, 0, 0))
    return x5.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)
