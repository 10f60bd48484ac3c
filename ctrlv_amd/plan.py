"""Host binding of the plan-level C ABI (include/ctrlv_hip.h: ctrlv_plan_*, ctrlv_unet_forward,
ctrlv_controlnet_forward): one ctypes call = one model forward, walked in C++ (csrc/plan.hip).

This is what `UNetSpatioTemporalConditionModel.forward` / `ControlNetModel.forward` run by default; the per-op Python
executor in models/blocks.py issues the same kernels with the same descriptors (bit-identical results) and is kept for
per-block tests, the error-growth trace and the per-kernel timing of bench.py.
"""
import ctypes

import torch

from . import _lib
from ._lib import ModelConfig, TensorDesc, check

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def config_struct(kind, cfg, time_context_order="sb"):
    """diffusers-style model config (dict / FrozenConfig) -> ctrlv_model_config."""
    down = tuple(cfg["down_block_types"])
    n = len(down)
    if n > _lib.CTRLV_MAX_BLOCKS:
        raise ValueError(f"at most {_lib.CTRLV_MAX_BLOCKS} blocks are supported, got {n}")
    c = ModelConfig()
    c.kind = 0 if kind == "unet" else 1
    c.in_channels = cfg["in_channels"]
    c.out_channels = cfg.get("out_channels", 4)
    c.n_blocks = n
    boc = tuple(cfg["block_out_channels"])
    heads = _tup(cfg["num_attention_heads"], n)
    layers = _tup(cfg["layers_per_block"], n)
    cross = _tup(cfg["cross_attention_dim"], n)
    if len(set(cross)) != 1:
        raise ValueError("ctrlv_amd supports one cross_attention_dim for all blocks (the SVD configuration)")
    up = tuple(cfg.get("up_block_types", ()))
    for i in range(n):
        c.block_out_channels[i] = boc[i]
        c.down_cross_attn[i] = 1 if down[i] == "CrossAttnDownBlockSpatioTemporal" else 0
        c.up_cross_attn[i] = 1 if (kind == "unet" and up[i] == "CrossAttnUpBlockSpatioTemporal") else 0
        c.layers_per_block[i] = layers[i]
        c.num_attention_heads[i] = heads[i]
    c.cross_attention_dim = cross[0]
    c.addition_time_embed_dim = cfg["addition_time_embed_dim"]
    c.projection_class_embeddings_input_dim = cfg["projection_class_embeddings_input_dim"]
    c.num_frames = cfg.get("num_frames", 25) or 25
    c.time_context_order = 0 if time_context_order == "sb" else 1
    return c


class Plan:
    """Owns one `ctrlv_plan` (module graph + packed weights in library-owned device memory)."""

    def __init__(self, kind, config, device, time_context_order="sb", dtype=torch.bfloat16):
        """dtype: the ELEMENT type of the plan's activations and packed weights -- torch.bfloat16 (libctrlv_hip.so) or
        torch.float16 (libctrlv_hip_f16.so)."""
        self.kind = kind
        self.device = torch.device(device)
        self.dtype = dtype
        self._lib = _lib.load(dtype)
        self._h = ctypes.c_void_p()
        self._cfg = config_struct(kind, config, time_context_order)
        check(self._lib.ctrlv_plan_create(ctypes.byref(self._cfg), self.device.index or 0, ctypes.byref(self._h)),
              "ctrlv_plan_create")
        self.trunk_mode = "same"
        self._ws = {}            # lane -> uint8 workspace tensor
        self._ws_bytes = {}      # (B, F, H, W) -> bytes
        self.n_down = self._lib.ctrlv_plan_num_down_residuals(self._h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.ctrlv_plan_destroy(h)
            except Exception:       # noqa: BLE001  (interpreter shutdown)
                pass

    def load_state_dict(self, state_dict):
        """Hand every parameter to the library by its diffusers key; packing happens on the device in C++."""
        items = [(k, v.detach()) for k, v in state_dict.items() if torch.is_tensor(v)]
        keep = []
        arr = (TensorDesc * len(items))()
        for i, (k, v) in enumerate(items):
            if v.dtype not in _DT:
                v = v.float()
            v = v.contiguous()
            keep.append(v)
            arr[i].name = k.encode()
            arr[i].data = v.data_ptr()
            arr[i].dtype = _DT[v.dtype]
            arr[i].on_device = 1 if v.is_cuda else 0
            arr[i].numel = v.numel()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)          # parameters may still be in flight on torch's streams
        check(self._lib.ctrlv_plan_load_weights(self._h, arr, len(items)), "ctrlv_plan_load_weights")
        del keep

    def profile(self, enable=True):
        """Bracket every kernel the plan issues with HIP events (eager forwards only; not under graph capture)."""
        check(self._lib.ctrlv_plan_profile(self._h, 1 if enable else 0), "ctrlv_plan_profile")

    def profile_read(self):
        """[(family, ms, flops, bytes, (M, N, K, flags))] of the launches since profile(True), in launch order."""
        n = self._lib.ctrlv_plan_profile_read(self._h, None, 0)
        if n <= 0:
            return []
        arr = (_lib.ProfileRecord * n)()
        m = self._lib.ctrlv_plan_profile_read(self._h, arr, n)
        if m < 0:
            check(m, "ctrlv_plan_profile_read")
        return [(_lib.FAMILIES[r.family], r.ms, r.flops, r.bytes, (r.M, r.N, r.K, r.flags)) for r in arr[:m]]

    def set_trunk_mode(self, mode):
        """Storage of the residual trunk: "same" (one element per value) or "fp16x2" (split: an fp16 hi plane + a one-byte
        e5m2 lo plane, ~15 significant bits in 3 bytes; fp16 plans only; the name is the API's since round 5) -- ctrlv_plan_set_trunk_mode."""
        if mode not in ("same", "fp16x2"):
            raise ValueError(f'trunk_dtype must be "same" or "fp16x2", got {mode!r}')
        check(self._lib.ctrlv_plan_set_trunk_mode(self._h, 1 if mode == "fp16x2" else 0), "ctrlv_plan_set_trunk_mode")
        self._ws_bytes.clear()          # the workspace grows with the lo planes
        self.trunk_mode = mode

    def set_time_context_order(self, order):
        check(self._lib.ctrlv_plan_set_time_context_order(self._h, 0 if order == "sb" else 1),
              "ctrlv_plan_set_time_context_order")

    def workspace_bytes(self, B, F, H, W):
        key = (B, F, H, W)
        if key not in self._ws_bytes:
            n = self._lib.ctrlv_plan_workspace_bytes(self._h, B, F, H, W)
            if n == 0:
                check(-2, "ctrlv_plan_workspace_bytes", lib=self._lib)
            self._ws_bytes[key] = n
        return self._ws_bytes[key]

    def workspace(self, B, F, H, W, lane=0):
        need = self.workspace_bytes(B, F, H, W)
        ws = self._ws.get(lane)
        if ws is None or ws.numel() < need or ws.device != self.device:
            ws = self._ws[lane] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def residual_shape(self, i, B, F, H, W):
        rows, ch = ctypes.c_int64(), ctypes.c_int32()
        check(self._lib.ctrlv_plan_residual_shape(self._h, i, B, F, H, W, ctypes.byref(rows), ctypes.byref(ch)),
              "ctrlv_plan_residual_shape")
        return rows.value, ch.value

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def unet_forward(self, sample, t32, ehs, ids32, down_rows, mid_rows, out, residual_event=None, lane=0):
        B, F, _, H, W = sample.shape
        ws = self.workspace(B, F, H, W, lane)
        down = None
        if down_rows is not None:
            down = (ctypes.c_void_p * len(down_rows))(*[r.data_ptr() for r in down_rows])
        ev = ctypes.c_void_p(residual_event.cuda_event) if residual_event is not None else None
        check(self._lib.ctrlv_unet_forward(
            self._h, sample.data_ptr(), _DT[sample.dtype], t32.data_ptr(), t32.numel(), ehs.data_ptr(),
            ids32.data_ptr(), ids32.shape[1], down, mid_rows.data_ptr() if mid_rows is not None else None, ev,
            out.data_ptr(), B, F, H, W, ws.data_ptr(), ws.numel(), self._stream()), "ctrlv_unet_forward")
        return out

    def unet_encoder_forward(self, sample, t32, ehs, ids32, out_taps, out_mid, lane=0):
        """Down + mid path only: writes the skip tensors / mid output into caller-owned row tensors (training step)."""
        B, F, _, H, W = sample.shape
        ws = self.workspace(B, F, H, W, lane)
        outs = (ctypes.c_void_p * len(out_taps))(*[r.data_ptr() for r in out_taps])
        check(self._lib.ctrlv_unet_encoder_forward(
            self._h, sample.data_ptr(), _DT[sample.dtype], t32.data_ptr(), t32.numel(), ehs.data_ptr(),
            ids32.data_ptr(), ids32.shape[1], outs, out_mid.data_ptr(), B, F, H, W, ws.data_ptr(), ws.numel(),
            self._stream()), "ctrlv_unet_encoder_forward")

    def controlnet_forward(self, sample, control, t32, ehs, ids32, scale, out_down, out_mid, lane=0):
        B, F, _, H, W = sample.shape
        ws = self.workspace(B, F, H, W, lane)
        outs = (ctypes.c_void_p * len(out_down))(*[r.data_ptr() for r in out_down])
        check(self._lib.ctrlv_controlnet_forward(
            self._h, sample.data_ptr(), control.data_ptr(), _DT[sample.dtype], t32.data_ptr(), t32.numel(),
            ehs.data_ptr(), ids32.data_ptr(), ids32.shape[1], float(scale), outs, out_mid.data_ptr(), B, F, H, W,
            ws.data_ptr(), ws.numel(), self._stream()), "ctrlv_controlnet_forward")
