"""Spatio-temporal ControlNet on MI355X.

Drop-in for /root/reference/src/ctrlv/models/controlnet.py:20-351: same constructor arguments and ValueErrors
(:52-98), `from_unet` (:197-224), `forward` signature and outputs (:226-351).  The 12 + 1 zero-initialised 1x1
convs and the `* conditioning_scale` pass (:331-344) are ONE gather-GEMM each with the scale in the epilogue; the
residuals are returned as (N, C, H, W)-shaped tensors with channels-last strides, so ctrlv_amd's UNet adds them
without any layout pass and any other consumer sees ordinary NCHW-indexed tensors.
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple, Union

import torch
from torch import nn

from .. import ops, packing
from .blocks import _f32
from .encoder import SpatioTemporalEncoderBase, _tup


@dataclass
class ControlNetOutput:
    down_block_res_samples: Tuple[torch.Tensor] = None
    mid_block_res_sample: torch.Tensor = None


def zero_module(module):
    for p in module.parameters():
        nn.init.zeros_(p)
    return module


class ControlNetModel(SpatioTemporalEncoderBase):
    _class_name = "ControlNetModel"
    _supports_gradient_checkpointing = True

    def __init__(
        self,
        sample_size: Optional[int] = None,
        in_channels: int = 8,
        down_block_types: Tuple[str] = ("CrossAttnDownBlockSpatioTemporal", "CrossAttnDownBlockSpatioTemporal",
                                        "CrossAttnDownBlockSpatioTemporal", "DownBlockSpatioTemporal"),
        block_out_channels: Tuple[int] = (320, 640, 1280, 1280),
        addition_time_embed_dim: int = 256,
        projection_class_embeddings_input_dim: int = 768,
        layers_per_block: Union[int, Tuple[int]] = 2,
        cross_attention_dim: Union[int, Tuple[int]] = 1024,
        transformer_layers_per_block: Union[int, Tuple[int], Tuple[Tuple]] = 1,
        num_attention_heads: Union[int, Tuple[int]] = (5, 10, 20, 20),
        num_frames: int = 25,
        time_context_order: str = "sb",
    ):
        super().__init__()
        self.register_to_config(
            sample_size=sample_size, in_channels=in_channels, down_block_types=tuple(down_block_types),
            block_out_channels=tuple(block_out_channels), addition_time_embed_dim=addition_time_embed_dim,
            projection_class_embeddings_input_dim=projection_class_embeddings_input_dim,
            layers_per_block=layers_per_block, cross_attention_dim=cross_attention_dim,
            transformer_layers_per_block=transformer_layers_per_block, num_attention_heads=num_attention_heads,
            num_frames=num_frames)
        self.sample_size = sample_size
        self.time_context_order = time_context_order
        # controlnet.py:80-98
        if len(block_out_channels) != len(down_block_types):
            raise ValueError(
                f"Must provide the same number of `block_out_channels` as `down_block_types`. `block_out_channels`: {block_out_channels}. `down_block_types`: {down_block_types}.")
        if not isinstance(num_attention_heads, int) and len(num_attention_heads) != len(down_block_types):
            raise ValueError(
                f"Must provide the same number of `num_attention_heads` as `down_block_types`. `num_attention_heads`: {num_attention_heads}. `down_block_types`: {down_block_types}.")
        if isinstance(cross_attention_dim, list) and len(cross_attention_dim) != len(down_block_types):
            raise ValueError(
                f"Must provide the same number of `cross_attention_dim` as `down_block_types`. `cross_attention_dim`: {cross_attention_dim}. `down_block_types`: {down_block_types}.")
        if not isinstance(layers_per_block, int) and len(layers_per_block) != len(down_block_types):
            raise ValueError(
                f"Must provide the same number of `layers_per_block` as `down_block_types`. `layers_per_block`: {layers_per_block}. `down_block_types`: {down_block_types}.")
        n = len(down_block_types)
        boc = tuple(block_out_channels)
        layers = _tup(layers_per_block, n)
        self._build_encoder(in_channels, down_block_types, boc, addition_time_embed_dim,
                            projection_class_embeddings_input_dim, layers_per_block, cross_attention_dim,
                            num_attention_heads)
        self.control_conv_in = nn.Conv2d(in_channels // 2, boc[0], kernel_size=3, padding=1)     # :136-141
        self.controlnet_down_blocks = nn.ModuleList([zero_module(nn.Conv2d(boc[0], boc[0], kernel_size=1))])
        for i in range(n):
            for _ in range(layers[i]):
                self.controlnet_down_blocks.append(zero_module(nn.Conv2d(boc[i], boc[i], kernel_size=1)))
            if i != n - 1:
                self.controlnet_down_blocks.append(zero_module(nn.Conv2d(boc[i], boc[i], kernel_size=1)))
        self.controlnet_mid_block = zero_module(nn.Conv2d(boc[-1], boc[-1], kernel_size=1))
        self.num_upsamplers = 0

    @classmethod
    def from_unet(cls, unet, load_weights_from_unet: bool = True):                              # :197-224
        c = unet.config
        ctrlnet = cls(
            in_channels=c.in_channels, down_block_types=c.down_block_types, block_out_channels=c.block_out_channels,
            addition_time_embed_dim=c.addition_time_embed_dim,
            projection_class_embeddings_input_dim=c.projection_class_embeddings_input_dim,
            layers_per_block=c.layers_per_block, cross_attention_dim=c.cross_attention_dim,
            transformer_layers_per_block=c.transformer_layers_per_block, num_attention_heads=c.num_attention_heads,
            num_frames=c.num_frames, time_context_order=getattr(unet, "time_context_order", "sb"))
        if load_weights_from_unet:
            usd, csd = unet.state_dict(), ctrlnet.state_dict()
            with torch.no_grad():
                for key in set(csd.keys()) & set(usd.keys()):
                    csd[key].copy_(usd[key])
            ctrlnet._packed = False
        return ctrlnet

    def _extra_input_convs(self):
        return [self.control_conv_in]

    def _pack_extra(self, pk):
        pk["zc"] = [(packing.pack_linear(m.weight), _f32(m.bias)) for m in self.controlnet_down_blocks]
        pk["zc_mid"] = (packing.pack_linear(self.controlnet_mid_block.weight), _f32(self.controlnet_mid_block.bias))

    _plan_kind = "controlnet"

    def _forward_plan(self, sample, timestep, encoder_hidden_states, added_time_ids, control_cond, conditioning_scale,
                      return_dict):
        """One call into the C++ execution plan (ctrlv_controlnet_forward)."""
        plan = self._ensure_plan(sample)
        B, F, _, h, w = sample.shape
        N = B * F
        t32, ehs, ids32 = self._plan_inputs(sample, timestep, encoder_hidden_states, added_time_ids)
        control = control_cond.to(device=sample.device, dtype=sample.dtype).contiguous()
        shapes = [plan.residual_shape(i, B, F, h, w) for i in range(plan.n_down + 1)]
        el = self.el_dtype
        rows = [torch.empty(M, C, dtype=el, device=sample.device) for M, C in shapes]
        plan.controlnet_forward(sample.contiguous(), control, t32, ehs, ids32, float(conditioning_scale), rows[:-1],
                                rows[-1], lane=getattr(self, "_lane", 0))

        def view(o, level_rows):      # (N, C, H, W) shape, channels-last strides
            C = o.shape[1]
            s = level_rows // N
            hh = h
            while hh * (w * hh // h) != s:     # recover (H, W) of the level from its pixel count
                hh //= 2
            o = o.view(N, hh, s // hh, C).permute(0, 3, 1, 2)
            return o if sample.dtype == el else o.to(sample.dtype)

        outs = [view(o, M) for o, (M, _) in zip(rows, shapes)]
        down, mid = outs[:-1], outs[-1]
        if not return_dict:
            return (down, mid)
        return ControlNetOutput(down_block_res_samples=down, mid_block_res_sample=mid)

    @torch.no_grad()
    def forward(
        self,
        sample: torch.FloatTensor,
        timestep: Union[torch.Tensor, float, int],
        encoder_hidden_states: torch.Tensor,
        added_time_ids: torch.Tensor,
        control_cond: torch.FloatTensor = None,
        conditioning_scale: float = 1.0,
        return_dict: bool = True,
    ) -> Union[ControlNetOutput, Tuple]:
        if control_cond is None:
            raise ValueError("control_cond is required (controlnet.py:289 flattens it unconditionally)")
        if sample.dim() != 5 or sample.shape[2] != self.config.in_channels:
            raise ValueError(f"sample must be (batch, frames, {self.config.in_channels}, height, width); got "
                             f"{tuple(sample.shape)}")
        B, F, Cin, h, w = sample.shape
        if tuple(control_cond.shape) != (B, F, Cin // 2, h, w):
            raise ValueError(f"control_cond must have shape {(B, F, Cin // 2, h, w)}; got {tuple(control_cond.shape)}")
        self._check_hw(h, w, len(self.down_blocks) - 1)
        N = B * F
        if self._use_plan():
            return self._forward_plan(sample, timestep, encoder_hidden_states, added_time_ids, control_cond,
                                      conditioning_scale, return_dict)
        ws = self._ensure_ready(sample)
        pk = self._pk
        ctx = self._context(ws, sample, timestep, encoder_hidden_states, added_time_ids)       # :262-294
        x = self._input_rows(ws, [sample.reshape(N, Cin, h, w),                                # :287-299
                                  control_cond.reshape(N, Cin // 2, h, w).to(sample.device)], N, h, w)
        x, H, W, taps = self._run_down_mid(ctx, x, h, w)                                       # :303-327
        scale = float(conditioning_scale)

        def zero_conv(rows, hh, ww, wb):                                                       # :331-344
            C = rows.shape[1]
            o = torch.empty(rows.shape[0], C, dtype=rows.dtype, device=rows.device)
            ops.gemm(rows, wb[0], o, N=wb[0].shape[0], cin=C, bias=wb[1], s_acc=scale)
            o = o.view(N, hh, ww, C).permute(0, 3, 1, 2)       # (N, C, H, W) shape, channels-last strides
            return o if sample.dtype == o.dtype else o.to(sample.dtype)

        down = [zero_conv(r, hh, ww, wb) for (r, hh, ww), wb in zip(taps, pk["zc"])]
        mid = zero_conv(x, H, W, pk["zc_mid"])
        if not return_dict:
            return (down, mid)
        return ControlNetOutput(down_block_res_samples=down, mid_block_res_sample=mid)
