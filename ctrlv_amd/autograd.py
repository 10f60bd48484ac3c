"""Autograd over the HIP kernels -- first slice of the training path (BASELINE config 5; the reference's training step is
tools/train_video_controlnet.py:451-488: ControlNet forward + backward with fp32 master parameters under bf16 compute).

Scope of this slice ("smallest verifiable slice"): the gather-GEMM family (nn.Linear / 1x1 conv
incl. the ControlNet zero-convs, 3x3 Conv2d, (3,1,1) Conv3d) with its fused epilogue operands {bias, residual R1,
per-clip row vector V, s_acc}, GroupNorm(+SiLU) in its 4-D and 5-D forms, and the folded AlphaBlender -- i.e. a complete
`SpatioTemporalResBlock` and the zero-convs -- checked against torch.autograd on the oracle (tests/test_backward_gpu.py).

  dgrad  = the forward kernel on role-swapped weights (Linear: W^T; convs: taps reversed, channels transposed)
  wgrad  = ctrlv_gemm_wgrad (transposed-LDS-read MFMA kernel), bias / row-vector gradients = ctrlv_colsum
  norms  = ctrlv_groupnorm_bwd on the statistics the forward saved
Activations and activation gradients are bf16 rows; parameter gradients are fp32 in the PyTorch layouts.
Second slice: LayerNorm, GEGLU, the attention cores (flash-style backward, csrc/attention_bwd.hip), nn.Linear with the
whole fused epilogue (FusedLinear / BlendLinear) => a complete `TransformerSpatioTemporalModel`; stride-2 and
upsample-fused convs.  The model-level training step built from these lives in ctrlv_amd/training.py.
"""
import contextlib
import math
import os
import weakref

import torch

from . import ops, packing


def _rows(M, C, like):
    return torch.empty(M, C, dtype=torch.bfloat16, device=like.device)


# Packed (bf16, kernel-layout) forms of FROZEN parameters are built once: the UNet of the training step never changes,
# and re-packing its decoder (forward form + role-swapped dgrad form) every step is ~10 ms of small torch kernels.
# Trainable parameters are packed from the fp32 masters on every use.  (Writes through `.data` do not bump _version:
# call clear_pack_cache() after editing a frozen model that way.)
_PACK_CACHE = {}
_WGRAD_DIRECT = os.environ.get("CTRLV_WGRAD_DIRECT", "0") == "1"     # conv dW straight in [N, cin, taps] (A/B handle)
_SKIP_FUSE = os.environ.get("CTRLV_SKIP_FUSE", "1") != "0"           # 0: autograd sums the skip gradients itself (A/B handle)


def clear_pack_cache():
    _PACK_CACHE.clear()


def _packed(weight, kind, fn):
    if weight.requires_grad or not isinstance(weight, torch.nn.Parameter):     # (temporaries may recycle an address)
        return fn(weight)
    key = (id(weight), kind)
    hit = _PACK_CACHE.get(key)
    # the entry is valid only for THIS parameter object (ids are recycled after garbage collection) at THIS version and
    # storage (in-place updates bump _version; .to() / load_state_dict may swap the storage)
    if hit is not None and hit[0]() is weight and hit[1] == (weight._version, weight.data_ptr(), weight.dtype):
        return hit[2]
    if len(_PACK_CACHE) > 4096:
        _PACK_CACHE.clear()
    out = fn(weight)
    _PACK_CACHE[key] = (weakref.ref(weight), (weight._version, weight.data_ptr(), weight.dtype), out)
    return out


def f32(p):
    """fp32 form of a parameter for the per-clip torch ops and the norm kernels: trainable / fp32 parameters as they are
    (autograd sees them), FROZEN 16-bit ones converted once (the UNet of the training step: ~500 small conversion kernels
    per step otherwise)."""
    if p.dtype == torch.float32 or p.requires_grad or not isinstance(p, torch.nn.Parameter):
        return p.float()
    return _packed(p, "f32", lambda w: w.detach().float().contiguous())


def _frozen(key_param, kind, fn):
    """A tensor derived from FROZEN parameters (fused q|k|v weight, GEGLU row interleave): built once and kept as a frozen
    Parameter, so that the packed kernel layouts derived from IT are cached as well.  Trainable: rebuilt on every use."""
    if key_param.requires_grad or not isinstance(key_param, torch.nn.Parameter):
        return fn()
    return _packed(key_param, kind, lambda _w: torch.nn.Parameter(fn().detach(), requires_grad=False))


def _geglu_bias32(bias):
    return _packed(bias, "geglu_b32", lambda b: packing.geglu_interleave(b.detach()).float().contiguous())


def _bias32(bias):
    """padded fp32 bias row of a GEMM launch (packing.pad_bias); cached for frozen parameters"""
    return _packed(bias, "bias32", packing.pad_bias)


def _pack_fwd(weight, mode, geglu=False):
    """bf16 forward GEMM layout of a parameter: one kernel (ctrlv_pack_weight) where the shape allows, else packing.py."""
    out = ops.pack_weight(weight, 0, geglu) if weight.is_cuda else None
    if out is not None:
        return out
    if geglu:
        return packing.pack_geglu(weight, weight.new_zeros(weight.shape[0]))[0]
    return (packing.pack_linear if mode == 0 else (packing.pack_conv3x3 if mode == 1 else packing.pack_conv_temporal))(weight)


class GatherGemm(torch.autograd.Function):
    """out = s_acc * (gather-GEMM(A, weight) + bias) + R1 + V[(m // vdiv)]   (what the res block's convs fuse).

    weight / bias: parameters in the PyTorch layout ([N, C], [N, C, 3, 3] or [N, C, 3, 1, 1]), any float dtype.
    geom: dict(mode, conv=(H, W, Ho, Wo, 1, 0) | None, temporal=(F, S) | None, vdiv)."""

    @staticmethod
    def _pack(weight, mode):
        return _packed(weight, ("fwd", mode), lambda w: _pack_fwd(w, mode))

    @staticmethod
    def forward(ctx, A, weight, bias, R1, V, s_acc, geom):
        mode = geom["mode"]
        N, cin = weight.shape[0], weight.shape[1]
        taps = {0: 1, 1: 9, 2: 3}[mode]
        m_out = A.shape[0]
        if mode == 1:        # stride-2 / upsample-fused convs change the row count
            H, W, Ho, Wo = geom["conv"][:4]
            m_out = A.shape[0] // (H * W) * Ho * Wo
        out = _rows(m_out, N, A)
        ops.gemm(A, GatherGemm._pack(weight, mode), out, N=(N + 31) // 32 * 32, cin=cin, taps=taps, mode=mode,
                 conv=geom.get("conv"), temporal=geom.get("temporal"),
                 bias=None if bias is None else _bias32(bias), R1=R1, s_acc=float(s_acc),
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
        # dgrad: the forward kernel with the weight's roles swapped.  The contraction runs over the forward's OUTPUT
        # channels: the GEMM needs a multiple of 64 of them (conv_out has 4: zero-padded).
        npad = (N + 63) // 64 * 64
        wd, dYd = weight.detach(), dY
        if npad != N:
            wd = torch.cat([wd, wd.new_zeros((npad - N,) + tuple(wd.shape[1:]))], 0)
            dYd = torch.zeros(dY.shape[0], npad, dtype=dY.dtype, device=dY.device)
            dYd[:, :N] = dY
        conv = geom.get("conv")

        def swapped(_):
            direct = ops.pack_weight(weight, 1) if weight.is_cuda else None     # one kernel incl. the zero padding of N
            if direct is not None:
                return direct
            if mode == 0:
                return packing.pack_linear(wd.reshape(npad, cin).t())
            if mode == 1:
                return packing.pack_conv3x3(wd.flip(2, 3).transpose(0, 1))
            return packing.pack_conv_temporal(wd.flip(2).transpose(0, 1))

        wt = _packed(weight, ("dgrad", mode), swapped)
        if mode == 1:
            H, W, Ho, Wo, stride, up = conv
            n_img = dY.shape[0] // (Ho * Wo)
            if stride == 2:
                # transposed conv = stride-1 dgrad on the zero-inserted gradient: dYz[2 yo, 2 xo] = dY[yo, xo]
                z = torch.zeros(n_img, H, W, npad, dtype=dY.dtype, device=dY.device)
                z[:, 0:2 * Ho:2, 0:2 * Wo:2] = dYd.view(n_img, Ho, Wo, npad)
                dYd = z.view(n_img * H * W, npad)
                conv = (H, W, H, W, 1, 0)
            elif up:
                conv = (Ho, Wo, Ho, Wo, 1, 0)        # dgrad on the upsampled grid, 2x2 sum-pool below
        dA = _rows(dYd.shape[0], cin, A)
        ops.gemm(dYd, wt, dA, N=(cin + 31) // 32 * 32, cin=npad, taps=taps, mode=mode, conv=conv,
                 temporal=geom.get("temporal"), s_acc=s_acc)
        if mode == 1 and geom["conv"][5]:
            H, W = geom["conv"][0], geom["conv"][1]
            n_img = A.shape[0] // (H * W)
            dA = dA.view(n_img, H, 2, W, 2, cin).float().sum((2, 4)).to(torch.bfloat16).view(n_img * H * W, cin)
    if need_dW:
        # one kernel: dW (scaled by s_acc) and, riding along in the workgroups that stream dY anyway, the bias gradient.
        # Linear weights are written in place; conv weights in the packed tap-major order and permuted afterwards
        # (writing [N, cin, taps] directly scatters the atomics at a 36-byte stride: A/B in DESIGN 3.6)
        direct = taps == 1 or _WGRAD_DIRECT
        # (deterministic form: the ordered slab sum WRITES dW / dbias -- no zero fill; the atomic form accumulates)
        new = torch.empty if ops.DETERMINISTIC else torch.zeros
        dWp = new(weight.shape if direct else (N, taps * cin), dtype=torch.float32, device=A.device)
        dbp = new(N, dtype=torch.float32, device=A.device) if need_db else None
        ops.gemm_wgrad(A, dY, dWp, N=N, cin=cin, taps=taps, mode=mode, conv=geom.get("conv"),
                       temporal=geom.get("temporal"), dbias=dbp, scale=s_acc, torch_layout=direct, assign=ops.DETERMINISTIC)
        if direct:
            dW = dWp
        elif mode == 1:
            dW = dWp.reshape(N, 3, 3, cin).permute(0, 3, 1, 2)
        else:
            dW = dWp.reshape(N, 3, cin).permute(0, 2, 1).reshape(N, cin, 3, 1, 1)
        dW = dW.to(weight.dtype)
        if need_db:
            db = dbp.to(weight.dtype)
    elif need_db:
        db = torch.zeros(N, dtype=torch.float32, device=A.device)
        ops.colsum(dY, db, scale=s_acc)
        db = db.to(weight.dtype)
    return dA, dW, db


class GroupNormSiLU(torch.autograd.Function):
    """y = [silu](GroupNorm32(x)) on channels-last rows; imgs_per_stat = 1 (4-D) or F (5-D statistics).
    skip=True: returns (y, x_skip) -- x_skip is x again, to be used by the SKIP connection around the branch this norm opens
    (the residual operand of the branch's last GEMM).  Both gradients of x then arrive in THIS backward, and the norm's
    backward kernel adds the skip's while it writes dx (ctrlv_groupnorm_bwd_add) instead of autograd summing the two in a
    separate pass over both tensors (181 such sums, 6.7 ms of the cfg5 step)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, n_img, S, imgs_per_stat, eps, silu, *opt):
        skip = bool(opt[0]) if opt else False
        ctx.n_opt = len(opt)
        C = x.shape[1]
        part = torch.empty(ops.groupnorm_scratch_floats(n_img, S, C, imgs_per_stat), dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        g32, b32 = f32(gamma).detach().contiguous(), f32(beta).detach().contiguous()
        ops.groupnorm(x, None, n_img, S, C, imgs_per_stat, g32, b32, eps, silu, y, part)
        ctx.save_for_backward(x, g32, b32, part)
        ctx.cfg = (n_img, S, C, imgs_per_stat, silu, gamma.dtype, bool(skip))
        ctx.set_materialize_grads(False)
        return (y, x) if skip else y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, g32, b32, part = ctx.saved_tensors
        n_img, S, C, ips, silu, pdt, skip = ctx.cfg
        tail = (None,) * (5 + ctx.n_opt)
        if dy is None:                    # (only the skip connection carried a gradient)
            return (dskip, None, None) + tail
        dx = torch.empty_like(x)
        dg = torch.zeros(C, dtype=torch.float32, device=x.device)
        db = torch.zeros(C, dtype=torch.float32, device=x.device)
        ops.groupnorm_bwd(x, dy.contiguous(), n_img, S, C, ips, part, g32, b32, silu, dx, dg, db,
                          add=None if dskip is None else dskip.contiguous())
        return (dx, dg.to(pdt), db.to(pdt)) + tail


_MIX_CACHE = {}      # id(mix_factor parameter) -> (version, sigmoid value); filled by prefetch_mix_factors()


def prefetch_mix_factors(*models):
    """ONE device-to-host copy for all AlphaBlender mix factors of the given models (about 60 scalars).  BlendGemm /
    BlendLinear fold sigmoid(mix_factor) into GEMM epilogue scalars, i.e. need it on the host: read one by one, that was a
    blocking sync per res block / transformer (it drained the queue ~60 times per step and serialised the backward /
    all-reduce overlap).  training.train_step calls this once per step."""
    ps = [p for m in models for n, p in m.named_parameters() if n.endswith("mix_factor")]
    if not ps:
        return
    vals = torch.sigmoid(torch.stack([p.detach().float().reshape(()) for p in ps])).cpu().tolist()
    _MIX_CACHE.clear()
    for p, v in zip(ps, vals):
        _MIX_CACHE[id(p)] = (p._version, p.data_ptr(), float(v))


def _mix_alpha(mix_factor):
    """sigmoid(mix_factor) as a python float: from the per-step cache if it is current, else one (blocking) read."""
    hit = _MIX_CACHE.get(id(mix_factor))
    if hit is not None and hit[0] == mix_factor._version and hit[1] == mix_factor.data_ptr():
        return hit[2]
    return 1.0 / (1.0 + math.exp(-float(mix_factor.detach().float().cpu())))


class BlendGemm(torch.autograd.Function):
    """AlphaBlender folded into the last temporal conv: out = xs + (1 - a) * (conv(hn) + bias), a = sigmoid(mix_factor)
    (a * xs + (1 - a) * (xs + conv) of SURVEY A.3).  Gradients for hn, the conv parameters, xs AND mix_factor."""

    @staticmethod
    def forward(ctx, hn, weight, bias, xs, mix_factor, geom):
        a = _mix_alpha(mix_factor)
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
    table of TransformerSpatioTemporalModel (fp32 [F, C]) or None.  skip=True: returns (y, x_skip), see GroupNormSiLU."""

    @staticmethod
    def forward(ctx, x, gamma, beta, V, vdiv, vmod, *opt):
        skip = bool(opt[0]) if opt else False
        ctx.n_opt = len(opt)
        y = torch.empty_like(x)
        g32, b32 = f32(gamma).detach().contiguous(), f32(beta).detach().contiguous()
        ops.layernorm(x, g32, b32, 1e-5, y, V=V, vdiv=vdiv, vmod=vmod)
        ctx.save_for_backward(x, g32, V if V is not None else torch.empty(0, device=x.device))
        ctx.cfg = (V is not None, vdiv, vmod, gamma.dtype)
        ctx.set_materialize_grads(False)
        return (y, x) if skip else y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, g32, V = ctx.saved_tensors
        has_v, vdiv, vmod, pdt = ctx.cfg
        tail = (None,) * (2 + ctx.n_opt)
        if dy is None:
            return (dskip, None, None, None) + tail
        C = x.shape[1]
        dx = torch.empty_like(x)
        dg = torch.zeros(C, dtype=torch.float32, device=x.device)
        db = torch.zeros(C, dtype=torch.float32, device=x.device)
        dV = None
        if has_v and ctx.needs_input_grad[3]:
            # (the table's gradient is the column sum of the NORM's dx: without the skip gradient)
            ops.layernorm_bwd(x, dy.contiguous(), g32, 1e-5, dx, dg, db, V=V, vdiv=vdiv, vmod=vmod)
            dV = torch.zeros_like(V)
            ops.colsum(dx, dV, vmode=1, vdiv=vdiv, vmod=vmod)
            if dskip is not None:
                dx = dx + dskip
        else:
            ops.layernorm_bwd(x, dy.contiguous(), g32, 1e-5, dx, dg, db, V=V if has_v else None, vdiv=vdiv, vmod=vmod,
                              add=None if dskip is None else dskip.contiguous())
        return (dx, dg.to(pdt), db.to(pdt), dV) + tail


# ---- gradient checkpointing of the GEGLU feed-forwards (`enable_gradient_checkpointing()`, reference:
# tools/train_video_controlnet.py:185-186).  The two 4C / 8C-wide intermediates of a feed-forward -- u (saved by the
# output projection for its wgrad) and the raw projection (saved for the GEGLU backward) -- are 31 GB of the step's 101 GB
# at the reference's size.  Checkpointed, NEITHER is kept: the forward does not even write the raw projection, and the
# backward recomputes both with the forward's own launch (one GEGLU GEMM with raw_out: the same bits).  u is saved by
# whichever Function consumes it (FusedLinear / BlendLinear), so it is swapped for a recompute cell by a saved-tensor hook
# that is active around the feed-forward pair only.
_CKPT = [False]


@contextlib.contextmanager
def gradient_checkpointing(enabled):
    """Training forwards built inside run their GEGLU feed-forwards checkpointed (see above)."""
    prev = _CKPT[0]
    _CKPT[0] = bool(enabled)
    try:
        yield
    finally:
        _CKPT[0] = prev


class _GegluCell:
    """Recompute cell of one checkpointed GEGLU projection: holds its small inputs, hands out u / raw once each."""

    def __init__(self, x, weight, bias):
        self.x, self.w, self.b = x, weight, bias
        self.u = self.raw = None
        self.done = False
        self.u_ptr = 0

    def _recompute(self):
        if self.done:
            return
        two_i = self.w.shape[0]
        wp, bp = _packed(self.w, "fwd_geglu", lambda w: _pack_fwd(w, 0, geglu=True)), _geglu_bias32(self.b)
        M = self.x.shape[0]
        self.u, self.raw = _rows(M, two_i // 2, self.x), _rows(M, two_i, self.x)
        ops.gemm(self.x, wp, self.u, N=two_i, cin=wp.shape[1], bias=bp, geglu=1, raw_out=self.raw)
        self.done = True

    def take(self, which):
        """Hands out u / raw once per recompute: a second take of the same tensor (ctx.saved_tensors read twice,
        backward(retain_graph=True) run again) recomputes instead of returning None."""
        if self.done and getattr(self, which) is None:
            self.done = False                        # that tensor was already handed out: run the projection again
            self.u = self.raw = None
        self._recompute()
        t = getattr(self, which)
        setattr(self, which, None)
        return t


_REC = [True]        # grad mode of the caller of the current feed-forward region (set by _ff_region)
_CELLS = {}          # data_ptr of a live checkpointed u -> its cell (only while the feed-forward pair is being built)


def _ckpt_pack(t):
    cell = _CELLS.pop(t.data_ptr(), None) if t.is_cuda else None
    return (cell,) if cell is not None else t


def _ckpt_unpack(obj):
    return obj[0].take("u") if isinstance(obj, tuple) else obj


@contextlib.contextmanager
def _ff_region():
    """Around one GEGLU projection + its output projection: under checkpointing the consumer's saved u becomes a cell."""
    prev, _REC[0] = _REC[0], torch.is_grad_enabled()
    try:
        yield from _ff_region_body()
    finally:
        _REC[0] = prev


def _ff_region_body():
    if not _CKPT[0]:
        yield
        return
    with torch.autograd.graph.saved_tensors_hooks(_ckpt_pack, _ckpt_unpack):
        try:
            yield
        finally:
            # every checkpointed u must have been claimed by its consumer's save_for_backward: a consumer that saved a copy
            # (a .contiguous() / reshaped u) would silently keep the tensor and make checkpointing a no-op
            leaked, _ = len(_CELLS), _CELLS.clear()
        if leaked:
            raise RuntimeError(f"gradient checkpointing: {leaked} GEGLU intermediate(s) were not claimed by the output "
                               "projection's saved tensors (the consumer saved a copy?)")


class GegluProj(torch.autograd.Function):
    """u = a * gelu_erf(g), (a | g) = x @ W^T + b   (diffusers GEGLU: `proj` Linear(C -> 2 I), chunk, exact gelu).  The
    forward is ONE GEMM with the GEGLU epilogue that ALSO writes the raw projection (bf16, packed column order) for the
    backward -- u itself comes from the fp32 accumulators exactly as in the inference path; shapes the ping-pong tiles do
    not serve (K < 128) recompute the raw projection in the backward instead.  Under gradient checkpointing (above) the
    raw projection is neither written nor kept: the backward takes it from the recompute cell."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        two_i, cin = weight.shape
        wp, bp = _packed(weight, "fwd_geglu", lambda w: _pack_fwd(w, 0, geglu=True)), _geglu_bias32(bias)
        u = _rows(x.shape[0], two_i // 2, x)
        # (a cell only where a backward will run: under torch.no_grad() / with no input requiring grad -- an evaluation pass
        #  inside a checkpointed training context -- Function.forward still executes, but no consumer ever saves u, so a cell
        #  registered here would stay unclaimed and _ff_region would report it as leaked.  ADVICE r05.  Grad mode is always
        #  off INSIDE a Function.forward: _ff_region records the caller's.)
        records = _REC[0] and any(ctx.needs_input_grad)
        ckpt = _CKPT[0] and records and cin >= 128 and cin % 32 == 0
        raw = _rows(x.shape[0], two_i, x) if (records and cin >= 128 and cin % 32 == 0 and not ckpt) else None
        ops.gemm(x, wp, u, N=two_i, cin=wp.shape[1], bias=bp, geglu=1, raw_out=raw)
        ctx.cell = None
        if ckpt:
            ctx.cell = _GegluCell(x, weight, bias)
            _CELLS[u.data_ptr()] = ctx.cell
        ctx.save_for_backward(x, weight, bias, raw if raw is not None else torch.empty(0, device=x.device))
        ctx.has_raw = raw is not None
        return u

    @staticmethod
    def backward(ctx, du):
        x, weight, bias, raw = ctx.saved_tensors
        two_i, cin = weight.shape
        inner = two_i // 2
        wi = _frozen(weight, "geglu_il", lambda: packing.geglu_interleave(weight.detach()))   # rows in the packed (value, gate) block order
        if ctx.cell is not None:
            raw = ctx.cell.take("raw")                                 # (recomputed together with u: one launch)
            ctx.cell = None
        elif not ctx.has_raw:
            wp, bp = packing.pack_geglu(weight, bias)
            raw = _rows(x.shape[0], two_i, x)
            ops.gemm(x, wp, raw, N=two_i, cin=wp.shape[1], bias=bp)    # activation recompute (no GEGLU epilogue)
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
    # (x and xs each feed a norm AND the skip connection around the branch it opens: the norm hands the skip its alias, so
    #  that both gradients meet in the norm's backward kernel -- GroupNormSiLU, skip=True)
    if s.conv_shortcut is None and _SKIP_FUSE:
        xn, res = GroupNormSiLU.apply(x, s.norm1.weight, s.norm1.bias, N, S, 1, block.eps, True, True)
    else:
        xn, res = GroupNormSiLU.apply(x, s.norm1.weight, s.norm1.bias, N, S, 1, block.eps, True), x
    h = GatherGemm.apply(xn, s.conv1.weight, s.conv1.bias, None, temb_tables[0], 1.0, g2d)
    hn = GroupNormSiLU.apply(h, s.norm2.weight, s.norm2.bias, N, S, 1, block.eps, True)
    if s.conv_shortcut is not None:
        res = GatherGemm.apply(x, s.conv_shortcut.weight.reshape(block.cout, block.cin), s.conv_shortcut.bias, None, None,
                               1.0, dict(mode=0))
    xs = GatherGemm.apply(hn, s.conv2.weight, s.conv2.bias, res, None, 1.0, g2d)
    if _SKIP_FUSE:
        hn, xs = GroupNormSiLU.apply(xs, t.norm1.weight, t.norm1.bias, N, S, F, block.eps, True, True)
    else:
        hn = GroupNormSiLU.apply(xs, t.norm1.weight, t.norm1.bias, N, S, F, block.eps, True)
    h = GatherGemm.apply(hn, t.conv1.weight, t.conv1.bias, None, temb_tables[1], 1.0, g3d)
    hn = GroupNormSiLU.apply(h, t.norm2.weight, t.norm2.bias, N, S, F, block.eps, True)
    return BlendGemm.apply(hn, t.conv2.weight, t.conv2.bias, xs, block.time_mixer.mix_factor, g3d)


def zero_conv_train_forward(conv, x, scale=1.0):
    """A ControlNet zero-conv (1x1) times conditioning_scale (controlnet.py:331-344) with gradients."""
    C = conv.weight.shape[0]
    return GatherGemm.apply(x, conv.weight.reshape(C, -1), conv.bias, None, None, float(scale), dict(mode=0))


# =============================================================================== transformer (second slice)
def _scaled(dy, s):
    """s * dy as a new bf16 tensor (dy itself when s == 1)."""
    if s == 1.0:
        return dy
    out = torch.empty_like(dy)
    ops.axpby(dy, dy, float(s), 0.0, out)
    return out


class FusedLinear(torch.autograd.Function):
    """out = s_acc * (A @ W^T + b) + s1 * R1 + s2 * R2 + V[(m // vdiv) % vmod]   (nn.Linear with the epilogue operands the
    transformer fuses: residuals, the per-clip cross-attention vector, the frame positional embedding)."""

    @staticmethod
    def forward(ctx, A, weight, bias, R1, R2, V, cfg):
        s_acc, s1, s2 = float(cfg.get("s_acc", 1.0)), float(cfg.get("s1", 1.0)), float(cfg.get("s2", 1.0))
        vdiv, vmod = cfg.get("vdiv", 1), cfg.get("vmod", 1 << 30)
        vmode, vS = (cfg.get("vmode", 1), cfg.get("vS", 1)) if V is not None else (0, 1)
        N, cin = weight.shape
        out = _rows(A.shape[0], N, A)
        ops.gemm(A, _packed(weight, ("fwd", 0), lambda w: _pack_fwd(w, 0)), out, N=(N + 31) // 32 * 32, cin=cin,
                 bias=None if bias is None else _bias32(bias), s_acc=s_acc, R1=R1, s1=s1, R2=R2, s2=s2,
                 V=V, vmode=vmode, vdiv=vdiv, vmod=vmod, vS=vS)
        ctx.save_for_backward(A, weight)
        ctx.cfg = (s_acc, s1, s2, vdiv, vmod, bias is not None, R1 is not None, R2 is not None,
                   None if V is None else tuple(V.shape), vmode, vS)
        return out

    @staticmethod
    def backward(ctx, dY):
        A, weight = ctx.saved_tensors
        s_acc, s1, s2, vdiv, vmod, has_b, has_r1, has_r2, vshape, vmode, vS = ctx.cfg
        need = ctx.needs_input_grad
        dY = dY.contiguous()
        dA, dW, db = gemm_grads(A, weight, dY, dict(mode=0), s_acc, need[0], need[1], has_b and need[2])
        dR1 = _scaled(dY, s1) if (has_r1 and need[3]) else None
        dR2 = _scaled(dY, s2) if (has_r2 and need[4]) else None
        dV = None
        if vshape is not None and need[5]:
            dV = torch.zeros(vshape, dtype=torch.float32, device=A.device)
            if vmode == 1:
                ops.colsum(dY, dV, vmode=1, vdiv=vdiv, vmod=min(vmod, vshape[0]))
            else:
                # vidx = ((m // vdiv) * vS + m % vS) % vmod (the diffusers-0.27.2 (s, b) context order): with vS and
                # vdiv multiples of vmod this is m % vmod -- table row j collects every vmod-th row starting at j
                if vS % vmod or vdiv % vmod:
                    raise NotImplementedError("row-vector gradient for vmode 2 needs vS and vdiv to be multiples of vmod")
                for j in range(vmod):
                    ops.colsum(dY[j::vmod], dV[j:j + 1])
        return dA, dW, db, dR1, dR2, dV, None


class BlendLinear(torch.autograd.Function):
    """Temporal FF output with the transformer's AlphaBlender folded in:
        out = a * h2 + (1 - a) * (g1 + u @ W^T + b),   a = sigmoid(mix_factor)."""

    @staticmethod
    def forward(ctx, u, weight, bias, g1, h2, mix_factor):
        a = _mix_alpha(mix_factor)
        N, cin = weight.shape
        out = _rows(u.shape[0], N, u)
        ops.gemm(u, _packed(weight, ("fwd", 0), lambda w: _pack_fwd(w, 0)), out, N=N, cin=cin, bias=_bias32(bias), s_acc=1.0 - a,
                 R1=g1, s1=1.0 - a, R2=h2, s2=a)
        ctx.save_for_backward(u, weight, h2, out, mix_factor)
        ctx.a = a
        return out

    @staticmethod
    def backward(ctx, dY):
        u, weight, h2, out, mix = ctx.saved_tensors
        a = ctx.a
        need = ctx.needs_input_grad
        dY = dY.contiguous()
        du, dW, db = gemm_grads(u, weight, dY, dict(mode=0), 1.0 - a, need[0], need[1], need[2])
        dg1 = _scaled(dY, 1.0 - a) if need[3] else None
        dh2 = _scaled(dY, a) if need[4] else None
        dmix = None
        if need[5]:
            # dL/da = sum dY * (h2 - g1 - lin) = sum dY * (h2 - out) / (1 - a);  da/dmix = a (1 - a)
            acc = torch.zeros(1, dtype=torch.float32, device=dY.device)
            ops.dot_diff(dY, h2, out, acc, scale=a)
            dmix = acc.to(mix.dtype).reshape(mix.shape)
        return du, dW, db, dg1, dh2, dmix


class SpatialAttention(torch.autograd.Function):
    """softmax(q k^T / 8) v per (image, head) on qkv rows [n_img * S, 3C] (BasicTransformerBlock.attn1 core)."""

    @staticmethod
    def forward(ctx, qkv, n_img, S, C):
        out = _rows(qkv.shape[0], C, qkv)
        lse = torch.empty(n_img, C // 64, S, dtype=torch.float32, device=qkv.device)
        ops.attention_spatial_lse(qkv, out, lse, n_img, S, C)
        ctx.save_for_backward(qkv, out, lse)
        ctx.cfg = (n_img, S, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        n_img, S, C = ctx.cfg
        dqkv = torch.empty_like(qkv)
        ops.attention_spatial_bwd(qkv, out, dout.contiguous(), lse, dqkv, n_img, S, C)
        return dqkv, None, None, None


class TemporalAttention(torch.autograd.Function):
    """Attention over the F frames of every (clip, pixel); rows ordered (b, f, s) (TemporalBasicTransformerBlock.attn1)."""

    @staticmethod
    def forward(ctx, qkv, B, F, S, C):
        out = _rows(qkv.shape[0], C, qkv)
        ops.attention_temporal(qkv, out, B, F, S, C)
        ctx.save_for_backward(qkv, out)
        ctx.cfg = (B, F, S, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out = ctx.saved_tensors
        B, F, S, C = ctx.cfg
        dqkv = torch.empty_like(qkv)
        ops.attention_temporal_bwd(qkv, out, dout.contiguous(), dqkv, B, F, S, C)
        return dqkv, None, None, None, None


def sinusoid(t, dim):
    """diffusers `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)` in fp32: [cos | sin]."""
    half = dim // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    arg = t.float()[:, None] * freq[None]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def transformer_train_forward(tr, x, ehs, B, F, H, W, time_context_order="sb"):
    """Training-mode forward of `ctrlv_amd.models.blocks.TransformerSpatioTemporalModel`: the same kernels and fusion as
    `tr.run` with every op recording its backward.  x: bf16 rows [B*F*H*W, C]; ehs: fp32 [B, cross_dim] (the single CLIP
    token per clip).  The tiny per-clip tensors (the two degenerate cross-attention vectors to_out(to_v(ehs)) and the
    frame positional embedding MLP) are plain fp32 torch ops, so their parameters get gradients through torch autograd;
    to_q / to_k / norm2 of the cross-attentions receive exactly zero gradient (softmax over one key is constant)."""
    import torch.nn.functional as Fn
    sb, tb = tr.transformer_blocks[0], tr.temporal_transformer_blocks[0]
    N, S, C = B * F, H * W, tr.C
    quirk = time_context_order == "sb" and B > 1       # diffusers 0.27.2: time_context rows (s, b), tokens (b, s) -- H1

    def xvec(attn):
        return Fn.linear(Fn.linear(ehs.float(), f32(attn.to_v.weight)), f32(attn.to_out[0].weight),
                         f32(attn.to_out[0].bias)).contiguous()

    tpe = tr.time_pos_embed
    emb = Fn.linear(Fn.silu(Fn.linear(sinusoid(torch.arange(F, device=x.device), C), f32(tpe.linear_1.weight),
                                      f32(tpe.linear_1.bias))), f32(tpe.linear_2.weight),
                    f32(tpe.linear_2.bias)).contiguous()                                   # fp32 [F, C]
    big = 1 << 30

    def lns(h, n, V=None, vdiv=1, vmod=big):             # (normed rows, the alias of h for the skip connection)
        if not _SKIP_FUSE:
            return LayerNormFn.apply(h, n.weight, n.bias, V, vdiv, vmod), h
        return LayerNormFn.apply(h, n.weight, n.bias, V, vdiv, vmod, True)

    def qkv_w(attn):
        return _frozen(attn.to_q.weight, "qkv_cat", lambda: torch.cat([attn.to_q.weight, attn.to_k.weight, attn.to_v.weight], 0))

    # Every trunk tensor (x, h0, h1, h2, g0, g1) feeds the norm that opens a branch AND the skip connection around it: the norm
    # hands the skip its alias (skip=True), so both gradients meet in the norm's backward kernel instead of a separate sum.
    if _SKIP_FUSE:
        t, x = GroupNormSiLU.apply(x, tr.norm.weight, tr.norm.bias, N, S, 1, 1e-6, False, True)
    else:
        t = GroupNormSiLU.apply(x, tr.norm.weight, tr.norm.bias, N, S, 1, 1e-6, False)
    h0 = FusedLinear.apply(t, tr.proj_in.weight, tr.proj_in.bias, None, None, None, {})
    # ---- spatial BasicTransformerBlock
    n, h0 = lns(h0, sb.norm1)
    qkv = FusedLinear.apply(n, qkv_w(sb.attn1), None, None, None, None, {})
    a = SpatialAttention.apply(qkv, N, S, C)
    h1 = FusedLinear.apply(a, sb.attn1.to_out[0].weight, sb.attn1.to_out[0].bias, h0, None, xvec(sb.attn2),
                           dict(vdiv=F * S))
    with _ff_region():
        n, h1 = lns(h1, sb.norm3)
        u = GegluProj.apply(n, sb.ff.net[0].proj.weight, sb.ff.net[0].proj.bias)
        h2 = FusedLinear.apply(u, sb.ff.net[2].weight, sb.ff.net[2].bias, h1, None, None, {})
    del u
    # ---- temporal block: rows stay ordered (b, f, s); the frame embedding is added inside the consumers
    with _ff_region():
        n, h2 = lns(h2, tb.norm_in, emb, S, F)
        u = GegluProj.apply(n, tb.ff_in.net[0].proj.weight, tb.ff_in.net[0].proj.bias)
        g0 = FusedLinear.apply(u, tb.ff_in.net[2].weight, tb.ff_in.net[2].bias, h2, None, emb, dict(vdiv=S, vmod=F))
    del u
    n, g0 = lns(g0, tb.norm1)
    qkv = FusedLinear.apply(n, qkv_w(tb.attn1), None, None, None, None, {})
    a = TemporalAttention.apply(qkv, B, F, S, C)
    g1 = FusedLinear.apply(a, tb.attn1.to_out[0].weight, tb.attn1.to_out[0].bias, g0, None, xvec(tb.attn2),
                           dict(vmode=2, vdiv=F * S, vS=S, vmod=B) if quirk else dict(vdiv=F * S))
    with _ff_region():
        n, g1 = lns(g1, tb.norm3)
        u = GegluProj.apply(n, tb.ff.net[0].proj.weight, tb.ff.net[0].proj.bias)
        h3 = BlendLinear.apply(u, tb.ff.net[2].weight, tb.ff.net[2].bias, g1, h2, tr.time_mixer.mix_factor)
    del u, n
    return FusedLinear.apply(h3, tr.proj_out.weight, tr.proj_out.bias, x, None, None, {})

