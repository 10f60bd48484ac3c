"""ORACLE (test infrastructure only -- never imported by the product path).  PARITY UNPINNED (see blocks.py).

CPU restatement of the two model forwards on the hot path:

  * UNetSpatioTemporalConditionModel.forward ... /root/reference/src/ctrlv/models/unet_spatio_temporal_condition.py:31-171
  * ControlNetModel.__init__ / forward ......... /root/reference/src/ctrlv/models/controlnet.py:52-195, 226-351
  * ControlNetModel.from_unet .................. /root/reference/src/ctrlv/models/controlnet.py:197-224

The module graph of the parent diffusers UNet (not in tree) follows SURVEY.md A.1/A.5.
"""
from types import SimpleNamespace

import torch
from torch import nn

from .blocks import (CrossAttnDownBlockSpatioTemporal, CrossAttnUpBlockSpatioTemporal, DownBlockSpatioTemporal,
                     TimestepEmbedding, Timesteps, UNetMidBlockSpatioTemporal, UpBlockSpatioTemporal, store)

SVD_CONFIG = dict(
    sample_size=96, in_channels=8, out_channels=4,
    down_block_types=("CrossAttnDownBlockSpatioTemporal", "CrossAttnDownBlockSpatioTemporal",
                      "CrossAttnDownBlockSpatioTemporal", "DownBlockSpatioTemporal"),
    up_block_types=("UpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal",
                    "CrossAttnUpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal"),
    block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256,
    projection_class_embeddings_input_dim=768, layers_per_block=2, cross_attention_dim=1024,
    transformer_layers_per_block=1, num_attention_heads=(5, 10, 20, 20), num_frames=25,
)

# The tiny configuration of SURVEY.md section 7 step 1: production head_dim (64), everything else small.
TINY_CONFIG = dict(SVD_CONFIG, sample_size=16, block_out_channels=(64, 128, 128, 128), addition_time_embed_dim=32,
                   projection_class_embeddings_input_dim=96, cross_attention_dim=64, num_attention_heads=(1, 2, 2, 2),
                   num_frames=3)


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def _make_down_block(kind, **kw):
    if kind == "CrossAttnDownBlockSpatioTemporal":
        return CrossAttnDownBlockSpatioTemporal(
            kw["in_channels"], kw["out_channels"], kw["temb_channels"], kw["num_layers"], kw["num_attention_heads"],
            kw["cross_attention_dim"], kw["add_downsample"], kw["time_context_order"])
    if kind == "DownBlockSpatioTemporal":
        return DownBlockSpatioTemporal(kw["in_channels"], kw["out_channels"], kw["temb_channels"], kw["num_layers"],
                                       kw["add_downsample"])
    raise ValueError(f"{kind} does not exist.")


class _EncoderMixin:
    """Shared pieces of the UNet and the ControlNet: embeddings (A.2) and the down path."""

    def _embed(self, sample, timestep, added_time_ids):
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.float32, device=sample.device)
        if len(timesteps.shape) == 0:
            timesteps = timesteps[None].to(sample.device)
        batch_size = sample.shape[0]
        timesteps = timesteps.expand(batch_size)
        t_emb = store(self.time_proj(timesteps).to(dtype=sample.dtype))
        emb = store(self.time_embedding(t_emb))
        time_embeds = self.add_time_proj(added_time_ids.flatten())
        time_embeds = store(time_embeds.reshape((batch_size, -1)).to(emb.dtype))
        return emb + self.add_embedding(time_embeds)

    def _down(self, sample, emb, encoder_hidden_states, image_only_indicator):
        down_block_res_samples = (sample,)
        for blk in self.down_blocks:
            if blk.has_cross_attention:
                sample, res = blk(sample, emb, encoder_hidden_states, image_only_indicator)
            else:
                sample, res = blk(sample, emb, image_only_indicator)
            down_block_res_samples += res
        return sample, down_block_res_samples


class UNetSpatioTemporalConditionModel(nn.Module, _EncoderMixin):
    def __init__(self, time_context_order="sb", **overrides):
        super().__init__()
        cfg = dict(SVD_CONFIG, **overrides)
        self.config = SimpleNamespace(**cfg)
        boc = cfg["block_out_channels"]
        n = len(boc)
        heads = _tup(cfg["num_attention_heads"], n)
        cross = _tup(cfg["cross_attention_dim"], n)
        layers = _tup(cfg["layers_per_block"], n)
        time_embed_dim = boc[0] * 4

        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        self.time_proj = Timesteps(boc[0], True, downscale_freq_shift=0)
        self.time_embedding = TimestepEmbedding(boc[0], time_embed_dim)
        self.add_time_proj = Timesteps(cfg["addition_time_embed_dim"], True, downscale_freq_shift=0)
        self.add_embedding = TimestepEmbedding(cfg["projection_class_embeddings_input_dim"], time_embed_dim)

        self.down_blocks = nn.ModuleList()
        output_channel = boc[0]
        for i, kind in enumerate(cfg["down_block_types"]):
            input_channel, output_channel = output_channel, boc[i]
            self.down_blocks.append(_make_down_block(
                kind, in_channels=input_channel, out_channels=output_channel, temb_channels=time_embed_dim,
                num_layers=layers[i], num_attention_heads=heads[i], cross_attention_dim=cross[i],
                add_downsample=i != n - 1, time_context_order=time_context_order))

        self.mid_block = UNetMidBlockSpatioTemporal(boc[-1], time_embed_dim, heads[-1], cross[-1],
                                                    time_context_order=time_context_order)

        self.up_blocks = nn.ModuleList()
        rboc, rheads, rcross, rlayers = boc[::-1], heads[::-1], cross[::-1], layers[::-1]
        output_channel = rboc[0]
        for i, kind in enumerate(cfg["up_block_types"]):
            prev_output_channel, output_channel = output_channel, rboc[i]
            input_channel = rboc[min(i + 1, n - 1)]
            add_upsample = i != n - 1
            if kind == "UpBlockSpatioTemporal":
                blk = UpBlockSpatioTemporal(input_channel, prev_output_channel, output_channel, time_embed_dim,
                                            rlayers[i] + 1, add_upsample)
            elif kind == "CrossAttnUpBlockSpatioTemporal":
                blk = CrossAttnUpBlockSpatioTemporal(input_channel, prev_output_channel, output_channel,
                                                     time_embed_dim, rlayers[i] + 1, rheads[i], rcross[i],
                                                     add_upsample, time_context_order)
            else:
                raise ValueError(f"{kind} does not exist.")
            self.up_blocks.append(blk)

        self.conv_norm_out = nn.GroupNorm(num_channels=boc[0], num_groups=32, eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids,
                down_block_additional_residuals=None, mid_block_additional_residuals=None, return_dict=False):
        # unet_spatio_temporal_condition.py:61
        is_controlnet = mid_block_additional_residuals is not None and down_block_additional_residuals is not None
        batch_size, num_frames = sample.shape[:2]
        emb = self._embed(sample, timestep, added_time_ids)                                  # :64-85
        sample = sample.flatten(0, 1)                                                        # :89
        emb = emb.repeat_interleave(num_frames, dim=0)                                       # :92
        encoder_hidden_states = encoder_hidden_states.repeat_interleave(num_frames, dim=0)   # :94
        sample = store(self.conv_in(sample), trunk=True)                                                 # :97
        image_only_indicator = torch.zeros(batch_size, num_frames, dtype=sample.dtype, device=sample.device)
        sample, down_block_res_samples = self._down(sample, emb, encoder_hidden_states, image_only_indicator)
        if is_controlnet:                                                                    # :119-127
            down_block_res_samples = tuple(store(s + r, trunk=True) for s, r in zip(down_block_res_samples,
                                                                        down_block_additional_residuals))
        sample = self.mid_block(sample, emb, encoder_hidden_states, image_only_indicator)   # :130-135
        if is_controlnet:
            sample = store(sample + mid_block_additional_residuals, trunk=True)                          # :136-137
        for blk in self.up_blocks:                                                           # :140-158
            res_samples = down_block_res_samples[-len(blk.resnets):]
            down_block_res_samples = down_block_res_samples[: -len(blk.resnets)]
            if blk.has_cross_attention:
                sample = blk(sample, res_samples, emb, encoder_hidden_states, image_only_indicator)
            else:
                sample = blk(sample, res_samples, emb, image_only_indicator)
        sample = store(self.conv_out(store(self.conv_act(self.conv_norm_out(sample)))))      # :161-163
        sample = sample.reshape(batch_size, num_frames, *sample.shape[1:])                   # :166
        return (sample,)


def zero_module(module):
    for p in module.parameters():
        nn.init.zeros_(p)
    return module


class ControlNetModel(nn.Module, _EncoderMixin):
    def __init__(self, time_context_order="sb", **overrides):
        super().__init__()
        cfg = {k: v for k, v in dict(SVD_CONFIG, **overrides).items() if k not in ("out_channels", "up_block_types")}
        self.config = SimpleNamespace(**cfg)
        boc, types = cfg["block_out_channels"], cfg["down_block_types"]
        n = len(types)
        # controlnet.py:80-98
        if len(boc) != len(types):
            raise ValueError("Must provide the same number of `block_out_channels` as `down_block_types`.")
        if not isinstance(cfg["num_attention_heads"], int) and len(cfg["num_attention_heads"]) != n:
            raise ValueError("Must provide the same number of `num_attention_heads` as `down_block_types`.")
        heads = _tup(cfg["num_attention_heads"], n)
        cross = _tup(cfg["cross_attention_dim"], n)
        layers = _tup(cfg["layers_per_block"], n)
        time_embed_dim = boc[0] * 4

        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)                   # :101-106
        self.time_proj = Timesteps(boc[0], True, downscale_freq_shift=0)
        self.time_embedding = TimestepEmbedding(boc[0], time_embed_dim)
        self.add_time_proj = Timesteps(cfg["addition_time_embed_dim"], True, downscale_freq_shift=0)
        self.add_embedding = TimestepEmbedding(cfg["projection_class_embeddings_input_dim"], time_embed_dim)
        self.control_conv_in = nn.Conv2d(cfg["in_channels"] // 2, boc[0], 3, padding=1)      # :136-141

        self.down_blocks = nn.ModuleList()
        self.controlnet_down_blocks = nn.ModuleList()
        output_channel = boc[0]
        self.controlnet_down_blocks.append(zero_module(nn.Conv2d(output_channel, output_channel, 1)))
        for i, kind in enumerate(types):
            input_channel, output_channel = output_channel, boc[i]
            is_final_block = i == n - 1
            self.down_blocks.append(_make_down_block(
                kind, in_channels=input_channel, out_channels=output_channel, temb_channels=time_embed_dim,
                num_layers=layers[i], num_attention_heads=heads[i], cross_attention_dim=cross[i],
                add_downsample=not is_final_block, time_context_order=time_context_order))
            for _ in range(layers[i]):
                self.controlnet_down_blocks.append(zero_module(nn.Conv2d(output_channel, output_channel, 1)))
            if not is_final_block:
                self.controlnet_down_blocks.append(zero_module(nn.Conv2d(output_channel, output_channel, 1)))
        self.controlnet_mid_block = zero_module(nn.Conv2d(boc[-1], boc[-1], 1))
        self.mid_block = UNetMidBlockSpatioTemporal(boc[-1], time_embed_dim, heads[-1], cross[-1],
                                                    time_context_order=time_context_order)

    @classmethod
    def from_unet(cls, unet, load_weights_from_unet=True, time_context_order="sb"):          # :197-224
        c = unet.config
        ctrlnet = cls(time_context_order=time_context_order, in_channels=c.in_channels,
                      down_block_types=c.down_block_types, block_out_channels=c.block_out_channels,
                      addition_time_embed_dim=c.addition_time_embed_dim,
                      projection_class_embeddings_input_dim=c.projection_class_embeddings_input_dim,
                      layers_per_block=c.layers_per_block, cross_attention_dim=c.cross_attention_dim,
                      transformer_layers_per_block=c.transformer_layers_per_block,
                      num_attention_heads=c.num_attention_heads, num_frames=c.num_frames)
        if load_weights_from_unet:
            usd, csd = unet.state_dict(), ctrlnet.state_dict()
            for key in set(csd.keys()) & set(usd.keys()):
                csd[key].copy_(usd[key])
        return ctrlnet

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, control_cond=None,
                conditioning_scale=1.0, return_dict=False):
        batch_size, num_frames = sample.shape[:2]
        emb = self._embed(sample, timestep, added_time_ids)                                  # :262-283
        sample = sample.flatten(0, 1)
        control_cond = control_cond.flatten(0, 1)
        emb = emb.repeat_interleave(num_frames, dim=0)
        encoder_hidden_states = encoder_hidden_states.repeat_interleave(num_frames, dim=0)
        sample = store(self.conv_in(sample) + self.control_conv_in(control_cond), trunk=True)            # :297-299
        image_only_indicator = torch.zeros(batch_size, num_frames, dtype=sample.dtype, device=sample.device)
        sample, down_block_res_samples = self._down(sample, emb, encoder_hidden_states, image_only_indicator)
        sample = self.mid_block(sample, emb, encoder_hidden_states, image_only_indicator)   # :322-327
        down = [blk(s) for s, blk in zip(down_block_res_samples, self.controlnet_down_blocks)]   # :331-337
        mid = self.controlnet_mid_block(sample)                                              # :339
        down = [store(s * conditioning_scale) for s in down]                                 # :343
        mid = store(mid * conditioning_scale)                                                # :344
        return (down, mid)


def seeded_init_(model, seed=0, zero_conv_std=None):
    """SURVEY.md 8(d) weight recipe: torch default nn.Conv*/nn.Linear init under a fixed seed, created in
    state-dict order; norm affines (1, 0); mix_factor = 0.5; ControlNet zero-convs optionally re-drawn N(0, std^2)
    so that the residual path is exercised."""
    g = torch.Generator().manual_seed(seed)
    for name, mod in model.named_modules():
        if isinstance(mod, (nn.Conv2d, nn.Conv3d, nn.Linear)):
            fan_in = mod.weight[0].numel()
            bound = 1.0 / (fan_in ** 0.5)
            with torch.no_grad():
                mod.weight.copy_((torch.rand(mod.weight.shape, generator=g) * 2 - 1) * bound)
                if mod.bias is not None:
                    mod.bias.copy_((torch.rand(mod.bias.shape, generator=g) * 2 - 1) * bound)
            is_zero_conv = name.startswith("controlnet_down_blocks") or name.startswith("controlnet_mid_block")
            if is_zero_conv:
                with torch.no_grad():
                    if zero_conv_std is None:
                        mod.weight.zero_(); mod.bias.zero_()
                    else:
                        mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * zero_conv_std)
                        mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * zero_conv_std)
    return model
