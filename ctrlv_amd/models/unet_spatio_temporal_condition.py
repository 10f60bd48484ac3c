"""UNetSpatioTemporalConditionModel on MI355X.

Drop-in for the reference class /root/reference/src/ctrlv/models/unet_spatio_temporal_condition.py:13-171 (same
`forward` signature incl. the Ctrl-V specific `down_block_additional_residuals` / `mid_block_additional_residuals`
kwargs, same helper methods :15-29, same diffusers config and state-dict keys).  The module graph of the diffusers
parent class follows SURVEY.md A.1/A.5.
"""
from dataclasses import dataclass
from typing import Optional, Tuple, Union

import torch
from torch import nn

from .. import ops, packing
from .blocks import CrossAttnUpBlockSpatioTemporal, UpBlockSpatioTemporal, _f32, _gn_scratch, _lo
from .encoder import SpatioTemporalEncoderBase, _tup


@dataclass
class UNetSpatioTemporalConditionOutput:
    sample: torch.Tensor = None

    def __getitem__(self, i):
        return (self.sample,)[i]


class UNetSpatioTemporalConditionModel(SpatioTemporalEncoderBase):
    _class_name = "UNetSpatioTemporalConditionModel"
    _supports_gradient_checkpointing = True

    def __init__(
        self,
        sample_size: Optional[int] = None,
        in_channels: int = 8,
        out_channels: int = 4,
        down_block_types: Tuple[str] = ("CrossAttnDownBlockSpatioTemporal", "CrossAttnDownBlockSpatioTemporal",
                                        "CrossAttnDownBlockSpatioTemporal", "DownBlockSpatioTemporal"),
        up_block_types: Tuple[str] = ("UpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal",
                                      "CrossAttnUpBlockSpatioTemporal", "CrossAttnUpBlockSpatioTemporal"),
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
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
            block_out_channels=tuple(block_out_channels), addition_time_embed_dim=addition_time_embed_dim,
            projection_class_embeddings_input_dim=projection_class_embeddings_input_dim,
            layers_per_block=layers_per_block, cross_attention_dim=cross_attention_dim,
            transformer_layers_per_block=transformer_layers_per_block, num_attention_heads=num_attention_heads,
            num_frames=num_frames)
        self.sample_size = sample_size
        self.time_context_order = time_context_order
        n = len(down_block_types)
        if len(down_block_types) != len(up_block_types):
            raise ValueError("Must provide the same number of `down_block_types` as `up_block_types`.")
        if len(block_out_channels) != n:
            raise ValueError("Must provide the same number of `block_out_channels` as `down_block_types`.")
        if not isinstance(num_attention_heads, int) and len(num_attention_heads) != n:
            raise ValueError("Must provide the same number of `num_attention_heads` as `down_block_types`.")
        if isinstance(cross_attention_dim, (list, tuple)) and len(cross_attention_dim) != n:
            raise ValueError("Must provide the same number of `cross_attention_dim` as `down_block_types`.")
        if not isinstance(layers_per_block, int) and len(layers_per_block) != n:
            raise ValueError("Must provide the same number of `layers_per_block` as `down_block_types`.")
        if transformer_layers_per_block != 1 and set(_tup(transformer_layers_per_block, n)) != {1}:
            raise ValueError("ctrlv_amd supports transformer_layers_per_block == 1 (the SVD configuration)")
        self._build_encoder(in_channels, down_block_types, block_out_channels, addition_time_embed_dim,
                            projection_class_embeddings_input_dim, layers_per_block, cross_attention_dim,
                            num_attention_heads)
        boc = tuple(block_out_channels)
        ted = boc[0] * 4
        heads, cross, layers = _tup(num_attention_heads, n), _tup(cross_attention_dim, n), _tup(layers_per_block, n)
        rboc, rheads, rcross, rlayers = boc[::-1], heads[::-1], cross[::-1], layers[::-1]
        self.up_blocks = nn.ModuleList()
        output_channel = rboc[0]
        for i, kind in enumerate(up_block_types):
            prev_output_channel, output_channel = output_channel, rboc[i]
            input_channel = rboc[min(i + 1, n - 1)]
            add_upsample = i != n - 1
            if kind == "UpBlockSpatioTemporal":
                blk = UpBlockSpatioTemporal(input_channel, prev_output_channel, output_channel, ted,
                                            rlayers[i] + 1, add_upsample)
            elif kind == "CrossAttnUpBlockSpatioTemporal":
                blk = CrossAttnUpBlockSpatioTemporal(input_channel, prev_output_channel, output_channel, ted,
                                                     rlayers[i] + 1, rheads[i], rcross[i], add_upsample)
            else:
                raise ValueError(f"{kind} does not exist.")
            self.up_blocks.append(blk)
        self.conv_norm_out = nn.GroupNorm(num_channels=boc[0], num_groups=32, eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    # ---- helpers of the reference subclass (unet_spatio_temporal_condition.py:15-29) -------------------------
    def enable_grad(self, temporal_transformer_block=True, all=False):
        parameters_list = []
        for name, param in self.named_parameters():
            if bool('temporal_transformer_block' in name and temporal_transformer_block) or all:
                parameters_list.append(param)
                param.requires_grad = True
            else:
                param.requires_grad = False
        return parameters_list

    def get_parameters_with_grad(self):
        return [param for param in self.parameters() if param.requires_grad]

    def encode_bbox_frame(self, frame_latent, encoded_objects):
        return frame_latent.unsqueeze(1).repeat(1, self.config.num_frames, 1, 1, 1)

    # ---- packing of the tail ------------------------------------------------------------------------------------
    def _pack_extra(self, pk):
        pk["gno"] = (_f32(self.conv_norm_out.weight), _f32(self.conv_norm_out.bias))
        pk["cout_w"] = packing.pack_conv3x3(self.conv_out.weight)       # rows padded to 32
        pk["cout_b"] = packing.pad_bias(self.conv_out.bias)

    def _residual_rows(self, ws, r, M, C):
        """A ControlNet residual given as an (N, C, H, W) tensor -> channels-last rows [M, C] of this model's element
        type (a view if it already is that, which is what ctrlv_amd's ControlNetModel of the same dtype returns)."""
        el = self.el_dtype
        if tuple(r.shape[:2]) != (r.shape[0], C) or r.numel() != M * C:
            raise ValueError(f"additional residual has shape {tuple(r.shape)}, expected {M * C} elements with {C} channels")
        rp = r.permute(0, 2, 3, 1)
        if r.dtype == el and rp.is_contiguous():
            return rp.reshape(M, C)
        rows = ws.alloc((M, C)) if ws is not None else torch.empty(M, C, dtype=el, device=r.device)
        return ops.nchw_to_rows(r.contiguous(), rows, 0)

    def _forward_plan(self, sample, timestep, encoder_hidden_states, added_time_ids, down, mid, return_dict):
        """One call into the C++ execution plan (ctrlv_unet_forward)."""
        plan = self._ensure_plan(sample)
        B, F, _, h, w = sample.shape
        t32, ehs, ids32 = self._plan_inputs(sample, timestep, encoder_hidden_states, added_time_ids)
        down_rows = mid_rows = None
        if down is not None:
            if len(down) != plan.n_down:
                raise ValueError(f"expected {plan.n_down} down_block_additional_residuals, got {len(down)}")
            down_rows = []
            for i, r in enumerate(list(down) + [mid]):
                M, C = plan.residual_shape(i, B, F, h, w)
                down_rows.append(self._residual_rows(None, r, M, C))
            mid_rows = down_rows.pop()
        out = torch.empty(B, F, self.config.out_channels, h, w, dtype=sample.dtype, device=sample.device)
        plan.unet_forward(sample.contiguous(), t32, ehs, ids32, down_rows, mid_rows, out,
                          residual_event=getattr(self, "_residual_event", None), lane=getattr(self, "_lane", 0))
        if not return_dict:
            return (out,)
        return UNetSpatioTemporalConditionOutput(sample=out)

    @torch.no_grad()
    def forward(
        self,
        sample: torch.FloatTensor,
        timestep: Union[torch.Tensor, float, int],
        encoder_hidden_states: torch.Tensor,
        added_time_ids: torch.Tensor,
        down_block_additional_residuals: Optional[Tuple[torch.Tensor]] = None,
        mid_block_additional_residuals: Optional[torch.Tensor] = None,
        return_dict: bool = True,
    ) -> Union[UNetSpatioTemporalConditionOutput, Tuple]:
        # unet_spatio_temporal_condition.py:61
        is_controlnet = mid_block_additional_residuals is not None and down_block_additional_residuals is not None
        if sample.dim() != 5 or sample.shape[2] != self.config.in_channels:
            raise ValueError(f"sample must be (batch, frames, {self.config.in_channels}, height, width); got "
                             f"{tuple(sample.shape)}")
        B, F, Cin, h, w = sample.shape
        self._check_hw(h, w, len(self.down_blocks) - 1)
        N = B * F
        if self._use_plan():
            return self._forward_plan(sample, timestep, encoder_hidden_states, added_time_ids,
                                      down_block_additional_residuals if is_controlnet else None,
                                      mid_block_additional_residuals if is_controlnet else None, return_dict)
        ws = self._ensure_ready(sample)
        pk = self._pk
        ctx = self._context(ws, sample, timestep, encoder_hidden_states, added_time_ids)       # :64-94
        x = self._input_rows(ws, [sample.reshape(N, Cin, h, w)], N, h, w)                      # :89,97
        x, H, W, taps = self._run_down_mid(ctx, x, h, w)                                       # :101-117,130-135
        if is_controlnet:                                                                      # :119-127,136-137
            if len(down_block_additional_residuals) != len(taps):
                raise ValueError(f"expected {len(taps)} down_block_additional_residuals, got "
                                 f"{len(down_block_additional_residuals)}")
            ev = getattr(self, "_residual_event", None)        # DenoiseStepper: the ControlNet ran on a side stream
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
            mk = ws.mark()
            def add_res(t, r):                                 # (split trunk: the sum is re-split into the tensor's two planes)
                rr = self._residual_rows(ws, r, t.shape[0], t.shape[1])
                if _lo(t) is not None:
                    ops.axpby_split(t, _lo(t), rr, 1.0, 1.0, t, _lo(t))
                else:
                    ops.axpby(t, rr, 1.0, 1.0, t)
            for (s, sh, sw), r in zip(taps, down_block_additional_residuals):
                add_res(s, r)
            add_res(x, mid_block_additional_residuals)
            ws.release(mk)
        skips = [t[0] for t in taps]
        for blk in self.up_blocks:                                                             # :140-158
            x, H, W = blk.run(ctx, x, H, W, skips)
        M, c0 = x.shape
        part = _gn_scratch(ctx, N, H * W, c0, 1)                                               # :161-163
        xn = ws.alloc((M, c0))
        ops.groupnorm(x, None, N, H * W, c0, 1, pk["gno"][0], pk["gno"][1], 1e-5, True, xn, part, x_lo=_lo(x))
        co = self.config.out_channels
        co_p = (co + 3) // 4 * 4
        y = ws.alloc((M, co_p))
        ops.gemm(xn, pk["cout_w"], y, N=pk["cout_w"].shape[0], cin=c0, taps=9, mode=1, conv=(H, W, H, W, 1, 0),
                 bias=pk["cout_b"], n_store=co_p)
        out = torch.empty(N, co, H, W, dtype=sample.dtype, device=sample.device)
        ops.rows_to_nchw(y, out)
        out = out.reshape(B, F, co, H, W)                                                      # :166
        if not return_dict:
            return (out,)
        return UNetSpatioTemporalConditionOutput(sample=out)
