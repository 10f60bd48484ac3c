"""Shared encoder machinery of the UNet and the ControlNet: embeddings, input conv, down blocks.

Reference: the identical prologues of
  src/ctrlv/models/unet_spatio_temporal_condition.py:61-117 and src/ctrlv/models/controlnet.py:261-327.
"""
import os

import torch
from torch import nn

from .. import _lib, ops, packing
from .. import profiler as _prof
from ..plan import Plan
from ..workspace import Workspace
from .blocks import (CrossAttnDownBlockSpatioTemporal, DownBlockSpatioTemporal, Downsample2D, FwdCtx,
                     SpatioTemporalResBlock, TransformerSpatioTemporalModel, Upsample2D, _f32, _TimestepEmbedding)
from .modeling_utils import HipModelMixin


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def make_down_block(kind, **kw):
    if kind == "CrossAttnDownBlockSpatioTemporal":
        return CrossAttnDownBlockSpatioTemporal(kw["in_channels"], kw["out_channels"], kw["temb_channels"],
                                                kw["num_layers"], kw["num_attention_heads"],
                                                kw["cross_attention_dim"], kw["add_downsample"])
    if kind == "DownBlockSpatioTemporal":
        return DownBlockSpatioTemporal(kw["in_channels"], kw["out_channels"], kw["temb_channels"], kw["num_layers"],
                                       kw["add_downsample"])
    raise ValueError(f"{kind} does not exist.")


_CLIP_ROWS = 11     # ctrlv_gemm tile 11: the per-clip-rows kernel for every GEMM of the conditioning path, whatever B is


class SpatioTemporalEncoderBase(HipModelMixin):
    """conv_in + time/added-id embeddings + down blocks + mid block, and the HIP execution plumbing."""

    time_context_order = "sb"     # diffusers 0.27.2 ordering quirk of the temporal cross-attention context (H1)
    # Who walks the layer list: "plan" = the C++ execution plan behind ctrlv_unet_forward / ctrlv_controlnet_forward
    # (one ctypes call per forward; default), "python" = the per-op executor of models/blocks.py (same kernels, same
    # descriptors, bit-identical results; used automatically for the per-kernel timer and the per-block trace).
    executor = os.environ.get("CTRLV_EXECUTOR", "plan")
    _plan_kind = "unet"
    # Storage of the RESIDUAL TRUNK (conv_in output, block / AlphaBlender outputs, skip tensors): "same" = one element per
    # value like every activation; "fp16x2" = split into an fp16 hi plane + a one-byte e5m2 lo plane (~15 significant bits in 3 bytes) under fp16
    # branches -- fp16 models, both executors; model-level error against the fp32 oracle 1.3e-3 -> < 1e-3.
    trunk_dtype = os.environ.get("CTRLV_TRUNK", "same")

    def _build_encoder(self, in_channels, down_block_types, block_out_channels, addition_time_embed_dim,
                       projection_class_embeddings_input_dim, layers_per_block, cross_attention_dim,
                       num_attention_heads):
        from .blocks import UNetMidBlockSpatioTemporal
        n = len(down_block_types)
        boc = tuple(block_out_channels)
        heads = _tup(num_attention_heads, n)
        cross = _tup(cross_attention_dim, n)
        layers = _tup(layers_per_block, n)
        time_embed_dim = boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = _TimestepEmbedding(boc[0], time_embed_dim)
        self.add_embedding = _TimestepEmbedding(projection_class_embeddings_input_dim, time_embed_dim)
        # diffusers exposes add_embedding.linear_1.in_features (src/ctrlv/utils/util.py:161) -- nn.Linear provides it
        self.down_blocks = nn.ModuleList()
        output_channel = boc[0]
        for i, kind in enumerate(down_block_types):
            input_channel, output_channel = output_channel, boc[i]
            self.down_blocks.append(make_down_block(
                kind, in_channels=input_channel, out_channels=output_channel, temb_channels=time_embed_dim,
                num_layers=layers[i], num_attention_heads=heads[i], cross_attention_dim=cross[i],
                add_downsample=i != n - 1))
        self.mid_block = UNetMidBlockSpatioTemporal(boc[-1], time_embed_dim, heads[-1], cross[-1])
        self._packed = False
        self._ws = None
        self._wss = {}
        self._plan = None

    # ------------------------------------------------------------------------------------------- packing
    def _apply(self, fn, *a, **k):          # .to() / .cuda() / .half() invalidate the packed weights
        self._packed = False
        self._plan = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = False
        self._plan = None
        return super().load_state_dict(*a, **k)

    def __setattr__(self, name, value):
        if name == "_packed" and value is False:       # whoever invalidates the Python packing invalidates the plan
            object.__setattr__(self, "_plan", None)
        super().__setattr__(name, value)

    def _param_fingerprint(self):
        """Changes whenever a parameter is modified in place (optimizer.step(), copy_) or replaced: the packed weights of
        both executors are keyed on it, so a forward after a training step runs on the UPDATED weights (the reference's
        periodic validation call, tools/train_video_controlnet.py log_validation)."""
        ver, first = 0, 0
        for q in self.parameters():
            ver += q._version
            if first == 0:
                first = q.data_ptr()
        return ver, first

    def _check_params_unchanged(self):
        fp = self._param_fingerprint()
        if getattr(self, "_packed_fp", None) != fp:
            if getattr(self, "_packed_fp", None) is not None:
                self._packed = False                     # (also drops the plan)
            object.__setattr__(self, "_packed_fp", fp)

    @property
    def el_dtype(self):
        """Element type of this model's activations / packed weights: fp16 parameters run on libctrlv_hip_f16.so (the
        reference's autocast dtype), everything else (bf16 parameters, the fp32 masters of a training step) on the bf16
        library."""
        return torch.float16 if self.dtype == torch.float16 else torch.bfloat16

    # ------------------------------------------------------------------------------------------- C++ plan
    def _use_plan(self):
        return (self.executor == "plan" and getattr(self, "_trace", None) is None and _prof._active is None)

    def _ensure_plan(self, sample):
        """The C++ execution plan of this model (csrc/plan.hip): created and loaded with the current parameters on first
        use, rebuilt after .to() / load_state_dict()."""
        el = self.el_dtype
        _lib.load(el)                             # raises if the HIP library is missing: no fallback
        if not sample.is_cuda:
            raise _lib.CtrlvHipError("ctrlv_amd models run on a HIP device only; there is no CPU forward "
                                     f"(got a {sample.device} input)")
        if self.device != sample.device:
            raise ValueError(f"model is on {self.device} but the input is on {sample.device}")
        self._check_params_unchanged()
        if self._plan is not None and self._plan.dtype != el:
            object.__setattr__(self, "_plan", None)
        if self._plan is None:
            plan = Plan(self._plan_kind, self.config, sample.device, self.time_context_order, dtype=el)
            plan.load_state_dict(self.state_dict())
            object.__setattr__(self, "_plan", plan)
            self._plan_order = self.time_context_order
        if self._plan_order != self.time_context_order:
            self._plan.set_time_context_order(self.time_context_order)
            self._plan_order = self.time_context_order
        if self._plan.trunk_mode != self.trunk_dtype:
            if self.trunk_dtype != "same" and el != torch.float16:
                raise ValueError('trunk_dtype="fp16x2" needs an fp16 model (the split planes are fp16 elements)')
            self._plan.set_trunk_mode(self.trunk_dtype)
        return self._plan

    def _plan_inputs(self, sample, timestep, encoder_hidden_states, added_time_ids):
        """Argument checks of _context() + the plain device buffers the C ABI takes (fp32 timestep / added ids,
        encoder states in the sample's dtype)."""
        dev = sample.device
        B = sample.shape[0]
        if encoder_hidden_states.dim() != 3 or encoder_hidden_states.shape[1] != 1:
            raise ValueError("encoder_hidden_states must have shape (batch, 1, cross_attention_dim): the path is "
                             "specialised for the single CLIP image token "
                             f"(unet_spatio_temporal_condition.py:93-94); got {tuple(encoder_hidden_states.shape)}")
        if encoder_hidden_states.shape[0] != B or added_time_ids.shape[0] != B:
            raise ValueError("encoder_hidden_states / added_time_ids batch size does not match sample")
        n_ids = added_time_ids.shape[1]
        add_dim = self.config.addition_time_embed_dim
        if add_dim * n_ids != self.add_embedding.linear_1.in_features:
            raise ValueError(f"Model expects an added time embedding vector of length "
                             f"{self.add_embedding.linear_1.in_features}, but a vector of {add_dim * n_ids} was created.")
        t = timestep if torch.is_tensor(timestep) else torch.tensor(float(timestep))
        t32 = t.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        if t32.numel() not in (1, B):
            raise ValueError(f"timestep must be a scalar or have {B} entries, got {t32.numel()}")
        ids32 = added_time_ids.to(device=dev, dtype=torch.float32).contiguous()
        ehs = encoder_hidden_states.to(device=dev, dtype=sample.dtype).contiguous()
        return t32, ehs, ids32

    def _extra_input_convs(self):
        return []

    def pack(self):
        """(Re)build every packed weight buffer (element type `el_dtype`) from the current parameters."""
        with packing.element_dtype(self.el_dtype):
            self._pack()
        self._pk_el = self.el_dtype

    def _pack(self):
        res_blocks, transformers = [], []
        for m in self.modules():
            if isinstance(m, SpatioTemporalResBlock):
                res_blocks.append(m)
            elif isinstance(m, TransformerSpatioTemporalModel):
                transformers.append(m)
            if isinstance(m, (SpatioTemporalResBlock, TransformerSpatioTemporalModel, Downsample2D, Upsample2D)):
                m.pack()
        pk = {}
        convs = [self.conv_in] + self._extra_input_convs()
        cin_tot = sum(c.weight.shape[1] for c in convs)
        cp = (cin_tot + 7) // 8 * 8
        kp = (9 * cp + 63) // 64 * 64
        pk["cin_cp"], pk["cin_kp"] = cp, kp
        pk["cin_w"] = packing.pack_conv_in([c.weight for c in convs], cp, kp)
        pk["cin_b"] = packing.pad_bias(sum(c.bias.detach().float() for c in convs))
        for name, emb in (("te", self.time_embedding), ("ae", self.add_embedding)):
            pk[name + "1_w"], pk[name + "1_b"] = packing.pack_linear(emb.linear_1.weight), _f32(emb.linear_1.bias)
            pk[name + "2_w"], pk[name + "2_b"] = packing.pack_linear(emb.linear_2.weight), _f32(emb.linear_2.bias)
        # all time_emb_proj of the model as ONE [sum Cout, 1280] GEMM; each res block keeps its column offsets
        ws_, bs_, off = [], [], 0
        for rb in res_blocks:
            offs = []
            for lin in rb.temb_projections():
                ws_.append(lin.weight.detach()); bs_.append(lin.bias.detach().float())
                offs.append(off); off += lin.weight.shape[0]
            rb.temb_off = tuple(offs)
        pk["temb_w"] = packing.pack_linear(torch.cat(ws_, 0))
        pk["temb_b"] = packing.pad_bias(torch.cat(bs_, 0))
        pk["temb_n"] = off
        # all cross-attention to_v as ONE [sum C, cross_dim] GEMM; to_out stays per attention (different inputs)
        vs_, off, xo = [], 0, []
        for tr in transformers:
            offs = []
            for attn in tr.cross_attentions():
                vs_.append(attn.to_v.weight.detach())
                xo.append((off, attn.to_v.weight.shape[0], packing.pack_linear(attn.to_out[0].weight),
                           _f32(attn.to_out[0].bias)))
                offs.append(off); off += attn.to_v.weight.shape[0]
            tr.xattn_off = tuple(offs)
        if vs_:
            pk["xv_w"] = packing.pack_linear(torch.cat(vs_, 0))
        pk["xattn_n"], pk["xattn_out"] = off, xo
        self._pack_extra(pk)
        self._pk = pk
        self._packed = True

    def _pack_extra(self, pk):
        pass

    def _ensure_ready(self, sample):
        el = self.el_dtype
        if self.trunk_dtype not in ("same", "fp16x2"):
            raise ValueError(f'trunk_dtype must be "same" or "fp16x2" (got {self.trunk_dtype!r})')
        if self.trunk_dtype != "same" and el != torch.float16:
            raise ValueError('trunk_dtype="fp16x2" needs an fp16 model (the split planes are fp16 elements)')
        _lib.load(el)                             # raises if the HIP library is missing: no fallback
        if not sample.is_cuda:
            raise _lib.CtrlvHipError("ctrlv_amd models run on a HIP device only; there is no CPU forward "
                                     f"(got a {sample.device} input)")
        if self.device != sample.device:
            raise ValueError(f"model is on {self.device} but the input is on {sample.device}")
        self._check_params_unchanged()
        if not self._packed or getattr(self, "_pk_el", None) != el:
            self.pack()
        # one activation arena per execution lane: DenoiseStepper runs the two CFG halves of a step as concurrent
        # forwards of this one model (shared packed weights) on different streams
        lane = getattr(self, "_lane", 0)
        ws = self._wss.get(lane)
        if ws is None or ws.device != sample.device:
            ws = self._wss[lane] = Workspace(sample.device)
        if lane == 0:
            self._ws = ws
        ws.el = el
        ws.split = self.trunk_dtype == "fp16x2"   # trunk tensors (Workspace.trunk) carry a lo plane, as in the C++ plan
        ws.reset()
        return ws

    # ------------------------------------------------------------------------------------------- embeddings
    @staticmethod
    def _sinusoid(t32, dim, kpad, device, el=torch.bfloat16):
        n = t32.numel()
        if kpad == dim:
            out = torch.empty(n, dim, dtype=el, device=device)
            return ops.timestep_embedding(t32, dim, out)
        tmp = torch.empty(n, dim, dtype=el, device=device)
        ops.timestep_embedding(t32, dim, tmp)
        out = torch.zeros(n, kpad, dtype=el, device=device)
        out[:, :dim] = tmp
        return out

    def _context(self, ws, sample, timestep, encoder_hidden_states, added_time_ids):
        """unet_spatio_temporal_condition.py:64-94 / controlnet.py:262-294 -> FwdCtx with the per-clip tables."""
        pk, dev, el = self._pk, sample.device, ws.el
        B, F = sample.shape[:2]
        boc0 = self.conv_in.weight.shape[0]
        ted = boc0 * 4
        if encoder_hidden_states.dim() != 3 or encoder_hidden_states.shape[1] != 1:
            raise ValueError("encoder_hidden_states must have shape (batch, 1, cross_attention_dim): the path is "
                             "specialised for the single CLIP image token "
                             f"(unet_spatio_temporal_condition.py:93-94); got {tuple(encoder_hidden_states.shape)}")
        if encoder_hidden_states.shape[0] != B or added_time_ids.shape[0] != B:
            raise ValueError("encoder_hidden_states / added_time_ids batch size does not match sample")
        t = timestep if torch.is_tensor(timestep) else torch.tensor(float(timestep))
        t32 = t.to(device=dev, dtype=torch.float32).reshape(-1)
        if t32.numel() == 1:
            t32 = t32.expand(B)
        t32 = t32.contiguous()
        te = self._sinusoid(t32, boc0, pk["te1_w"].shape[1], dev, el)
        h = torch.empty(B, ted, dtype=el, device=dev)
        ops.gemm(te, pk["te1_w"], h, N=ted, cin=te.shape[1], bias=pk["te1_b"], act=1, tile=_CLIP_ROWS)
        emb_t = torch.empty(B, ted, dtype=el, device=dev)
        ops.gemm(h, pk["te2_w"], emb_t, N=ted, cin=ted, bias=pk["te2_b"], tile=_CLIP_ROWS)
        ids = added_time_ids.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        n_ids = added_time_ids.shape[1]
        add_dim = self.config.addition_time_embed_dim
        if add_dim * n_ids != self.add_embedding.linear_1.in_features:
            raise ValueError(f"Model expects an added time embedding vector of length "
                             f"{self.add_embedding.linear_1.in_features}, but a vector of {add_dim * n_ids} was created.")
        kp = pk["ae1_w"].shape[1]
        ae = self._sinusoid(ids, add_dim, add_dim, dev, el).reshape(B, n_ids * add_dim)
        if kp != ae.shape[1]:
            ae_p = torch.zeros(B, kp, dtype=el, device=dev)
            ae_p[:, :ae.shape[1]] = ae
            ae = ae_p
        ops.gemm(ae, pk["ae1_w"], h, N=ted, cin=kp, bias=pk["ae1_b"], act=1, tile=_CLIP_ROWS)
        emb_s = torch.empty(B, ted, dtype=el, device=dev)       # silu(emb + aug_emb)
        ops.gemm(h, pk["ae2_w"], emb_s, N=ted, cin=ted, bias=pk["ae2_b"], R1=emb_t, act=1, tile=_CLIP_ROWS)
        temb = torch.empty(B, pk["temb_w"].shape[0], dtype=torch.float32, device=dev)
        ops.gemm(emb_s, pk["temb_w"], temb, N=pk["temb_w"].shape[0], cin=ted, bias=pk["temb_b"], out_f32=True, tile=_CLIP_ROWS)
        xattn = None
        if pk["xattn_n"]:
            dc = encoder_hidden_states.shape[2]
            ehs = encoder_hidden_states.reshape(B, dc).to(el).contiguous()
            kx = pk["xv_w"].shape[1]
            if kx != dc:
                ehs_p = torch.zeros(B, kx, dtype=el, device=dev)
                ehs_p[:, :dc] = ehs
                ehs = ehs_p
            nx = pk["xv_w"].shape[0]
            v_all = torch.empty(B, nx, dtype=el, device=dev)
            ops.gemm(ehs, pk["xv_w"], v_all, N=nx, cin=kx, tile=_CLIP_ROWS)
            xattn = torch.empty(B, nx, dtype=torch.float32, device=dev)
            for off, c, wo, bo in pk["xattn_out"]:
                ops.gemm(v_all[:, off:off + c], wo, xattn[:, off:off + c], N=c, cin=c, bias=bo, out_f32=True, tile=_CLIP_ROWS)
        ctx = FwdCtx(ws, B, F, temb, xattn, self.time_context_order)
        ctx.trace = getattr(self, "_trace", None)        # tests: per-block outputs (error-growth trace)
        return ctx

    def _input_rows(self, ws, planes, N, h, w):
        """NCHW input planes -> channels-last rows -> conv_in (+control_conv_in) as ONE im2col GEMM."""
        pk = self._pk
        M = N * h * w
        x16 = ws.alloc((M, pk["cin_cp"]))
        x16.zero_()
        off = 0
        for p in planes:
            ops.nchw_to_rows(p.contiguous(), x16, off)
            off += p.shape[1]
        col = ws.alloc((M, pk["cin_kp"]))
        ops.im2col3x3(x16, N, h, w, col)
        c0 = self.conv_in.weight.shape[0]
        x = ws.trunk((M, c0))
        ops.gemm(col, pk["cin_w"], x, N=pk["cin_w"].shape[0], cin=pk["cin_kp"], bias=pk["cin_b"], out_lo=getattr(x, "lo", None))
        return x

    def _run_down_mid(self, ctx, x, h, w):
        taps = [(x, h, w)]
        H, W = h, w
        for blk in self.down_blocks:
            x, H, W, t = blk.run(ctx, x, H, W)
            taps += t
        x = self.mid_block.run(ctx, x, H, W)
        return x, H, W, taps

    @staticmethod
    def _check_hw(h, w, n_down):
        m = 1 << n_down
        if h % m or w % m:
            raise ValueError(f"latent height and width have to be divisible by {m} but are {h} and {w}.")
