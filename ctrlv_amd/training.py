"""cfg5: one training step of the ControlNet on the HIP kernels (reference: tools/train_video_controlnet.py:366-488).

    ControlNet forward (trainable, fp32 master parameters, bf16 compute)
    -> frozen UNet forward: encoder + mid on the inference executor (no gradients flow there: the residuals join the skip
       tensors AFTER the down path, unet_spatio_temporal_condition.py:119-137), decoder in training mode (dgrad only)
    -> EDM-preconditioned MSE (train_video_controlnet.py:468-478) -> backward -> AdamW (torch.optim on the fp32 masters).

Everything activation-sized runs in the kernels of libctrlv_hip.so through ctrlv_amd.autograd; the per-clip vectors (time
and added-id embeddings, time_emb_proj, the one-key cross-attention vectors, the frame positional embedding MLP) are
plain fp32 torch ops on [B, 1280]-sized tensors.  Multi-GPU: `GradientBuckets` all-reduces flat fp32 buckets (25 MB, the
reference's DDP default) over torch.distributed WHILE the backward pass runs (hooks fire as gradients complete);
`allreduce_gradients` is the simple reduce-after-backward form (RCCL on the GPU box, gloo in the CPU tests).
"""
import math

import torch
import torch.nn.functional as Fn

from . import ops
from .autograd import (FusedLinear, GatherGemm, GroupNormSiLU, f32, gradient_checkpointing, res_block_train_forward, sinusoid,
                       transformer_train_forward, zero_conv_train_forward)
from .models.blocks import TransformerSpatioTemporalModel  # noqa: F401  (documentation anchor)


# ------------------------------------------------------------------------------------------------- embeddings
def clip_embeddings(model, timestep, added_time_ids, B, device):
    """silu(time_embedding(t) + add_embedding(ids)) -> fp32 [B, 1280] (controlnet.py:262-286); torch autograd."""
    boc0 = model.conv_in.weight.shape[0]
    t = timestep if torch.is_tensor(timestep) else torch.tensor(float(timestep))
    t = t.to(device=device, dtype=torch.float32).reshape(-1)
    if t.numel() == 1:
        t = t.expand(B)

    def mlp(e, x):
        return Fn.linear(Fn.silu(Fn.linear(x, f32(e.linear_1.weight), f32(e.linear_1.bias))),
                         f32(e.linear_2.weight), f32(e.linear_2.bias))

    emb = mlp(model.time_embedding, sinusoid(t, boc0))
    ids = added_time_ids.to(device=device, dtype=torch.float32)
    add_dim = model.config.addition_time_embed_dim
    aug = mlp(model.add_embedding, sinusoid(ids.reshape(-1), add_dim).reshape(B, -1))
    return Fn.silu(emb + aug)


def _temb_tables(block, emb_s):
    return [Fn.linear(emb_s, f32(m.time_emb_proj.weight), f32(m.time_emb_proj.bias)).contiguous()
            for m in (block.spatial_res_block, block.temporal_res_block)]


# ------------------------------------------------------------------------------------------------- blocks
def _res(block, x, emb_s, B, F, H, W):
    return res_block_train_forward(block, x, _temb_tables(block, emb_s), B, F, H, W)


def _conv(conv, x, H, W, Ho, Wo, stride, up):
    return GatherGemm.apply(x, conv.weight, conv.bias, None, None, 1.0, dict(mode=1, conv=(H, W, Ho, Wo, stride, up)))


def down_block_train(blk, x, emb_s, ehs, B, F, H, W, order):
    taps = []
    attns = getattr(blk, "attentions", None)
    for i, resnet in enumerate(blk.resnets):
        x = _res(resnet, x, emb_s, B, F, H, W)
        if attns is not None:
            x = transformer_train_forward(attns[i], x, ehs, B, F, H, W, order)
        taps.append((x, H, W))
    if blk.downsamplers is not None:
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        x = _conv(blk.downsamplers[0].conv, x, H, W, Ho, Wo, 2, 0)
        H, W = Ho, Wo
        taps.append((x, H, W))
    return x, H, W, taps


def mid_block_train(blk, x, emb_s, ehs, B, F, H, W, order):
    x = _res(blk.resnets[0], x, emb_s, B, F, H, W)
    for attn, resnet in zip(blk.attentions, blk.resnets[1:]):
        x = transformer_train_forward(attn, x, ehs, B, F, H, W, order)
        x = _res(resnet, x, emb_s, B, F, H, W)
    return x


def up_block_train(blk, x, emb_s, ehs, B, F, H, W, skips, order):
    attns = getattr(blk, "attentions", None)
    for i, resnet in enumerate(blk.resnets):
        x = torch.cat([x, skips.pop()], dim=1)               # unet_3d_blocks: torch.cat([hidden, skip], dim=1)
        x = _res(resnet, x, emb_s, B, F, H, W)
        if attns is not None:
            x = transformer_train_forward(attns[i], x, ehs, B, F, H, W, order)
    if blk.upsamplers is not None:
        x = _conv(blk.upsamplers[0].conv, x, H, W, 2 * H, 2 * W, 1, 1)
        H, W = 2 * H, 2 * W
    return x, H, W


# ------------------------------------------------------------------------------------------------- models
def _input_cols(planes, N, h, w, cp, kp, device):
    """NCHW planes -> channels-last rows [M, cp] -> im2col [M, kp] (the tiny-channel input convs run as ONE GEMM)."""
    M = N * h * w
    x16 = torch.zeros(M, cp, dtype=torch.bfloat16, device=device)
    off = 0
    for p in planes:
        ops.nchw_to_rows(p.contiguous(), x16, off)
        off += p.shape[1]
    col = torch.empty(M, kp, dtype=torch.bfloat16, device=device)
    ops.im2col3x3(x16, N, h, w, col)
    return col


def _input_conv_weight(convs, cp, kp):
    """[N, kp] im2col weight of conv_in (+ control_conv_in) built with differentiable torch ops (packing.pack_conv_in)."""
    n = convs[0].weight.shape[0]
    w = torch.cat([c.weight for c in convs], dim=1)                                   # [N, sum C, 3, 3]
    w = Fn.pad(w.permute(0, 2, 3, 1), (0, cp - w.shape[1]))                           # [N, 3, 3, cp]
    w = Fn.pad(w.reshape(n, 9 * cp), (0, kp - 9 * cp))
    b = sum(c.bias for c in convs)
    return w, b


def controlnet_train_forward(model, sample, timestep, encoder_hidden_states, added_time_ids, control_cond,
                             conditioning_scale=1.0):
    """`ControlNetModel.forward` (controlnet.py:226-351) with gradients.  Returns (down, mid): lists of
    (rows [N*H*W, C] bf16, H, W) in the order of `down_block_res_samples`, and the mid tuple."""
    B, F, Cin, h, w = sample.shape
    N, dev = B * F, sample.device
    order = model.time_context_order
    emb_s = clip_embeddings(model, timestep, added_time_ids, B, dev)
    ehs = encoder_hidden_states.reshape(B, -1).float()
    convs = [model.conv_in] + model._extra_input_convs()
    cin_tot = sum(c.weight.shape[1] for c in convs)
    cp = (cin_tot + 7) // 8 * 8
    kp = (9 * cp + 63) // 64 * 64
    col = _input_cols([sample.reshape(N, Cin, h, w), control_cond.reshape(N, -1, h, w).to(dev)], N, h, w, cp, kp, dev)
    wi, bi = _input_conv_weight(convs, cp, kp)
    x = FusedLinear.apply(col, wi, bi, None, None, None, {})
    taps, H, W = [(x, h, w)], h, w
    # `model.enable_gradient_checkpointing()` (the reference's trainer calls it: tools/train_video_controlnet.py:185-186):
    # the GEGLU feed-forward intermediates are recomputed in the backward instead of kept (autograd.py)
    with gradient_checkpointing(getattr(model, "gradient_checkpointing", False)):
        for blk in model.down_blocks:
            x, H, W, t = down_block_train(blk, x, emb_s, ehs, B, F, H, W, order)
            taps += t
        x = mid_block_train(model.mid_block, x, emb_s, ehs, B, F, H, W, order)
    down = [(zero_conv_train_forward(zc, r, conditioning_scale), hh, ww)
            for (r, hh, ww), zc in zip(taps, model.controlnet_down_blocks)]
    mid = (zero_conv_train_forward(model.controlnet_mid_block, x, conditioning_scale), H, W)
    return down, mid


def unet_train_forward(unet, sample, timestep, encoder_hidden_states, added_time_ids, down_res, mid_res):
    """Frozen-UNet forward that is differentiable w.r.t. the ControlNet residuals.  Encoder + mid run on the inference
    executor under no_grad (no gradient path: the residuals are added to its OUTPUTS); the decoder runs in training
    mode.  Returns the model prediction as channels-last rows [B*F*h*w, out_channels] (bf16)."""
    B, F, Cin, h, w = sample.shape
    N, dev = B * F, sample.device
    order = unet.time_context_order
    with torch.no_grad():
        if unet._use_plan():
            # the library-owned C++ plan walks the frozen encoder (ctrlv_unet_encoder_forward): one host call
            plan = unet._ensure_plan(sample)
            t32, ehs_p, ids32 = unet._plan_inputs(sample, timestep, encoder_hidden_states, added_time_ids)
            shapes = [plan.residual_shape(i, B, F, h, w) for i in range(plan.n_down + 1)]
            rows = [torch.empty(M, C, dtype=torch.bfloat16, device=dev) for M, C in shapes]
            plan.unet_encoder_forward(sample.contiguous(), t32, ehs_p, ids32, rows[:-1], rows[-1])
            hw = [(h, w)]
            for M, _ in shapes[1:]:
                hh, ww = hw[-1]
                while hh * ww * N != M:
                    hh, ww = (hh + 2 - 3) // 2 + 1, (ww + 2 - 3) // 2 + 1
                hw.append((hh, ww))
            taps = [(r, a, b) for r, (a, b) in zip(rows[:-1], hw[:-1])]
            x, (H, W) = rows[-1], hw[-1]
        else:                        # per-op executor (profiling / tracing modes)
            ws = unet._ensure_ready(sample)
            ctx = unet._context(ws, sample, timestep, encoder_hidden_states, added_time_ids)
            x = unet._input_rows(ws, [sample.reshape(N, Cin, h, w)], N, h, w)
            x, H, W, taps = unet._run_down_mid(ctx, x, h, w)
        emb_s = clip_embeddings(unet, timestep, added_time_ids, B, dev)
    ehs = encoder_hidden_states.reshape(B, -1).float()
    if len(down_res) != len(taps):
        raise ValueError(f"expected {len(taps)} down_block_additional_residuals, got {len(down_res)}")
    skips = [s + r for (s, _, _), (r, _, _) in zip(taps, down_res)]            # :119-127 (out of place: s is arena memory)
    x = x + mid_res[0]                                                         # :136-137
    with gradient_checkpointing(getattr(unet, "gradient_checkpointing", False)):
        for blk in unet.up_blocks:
            x, H, W = up_block_train(blk, x, emb_s, ehs, B, F, H, W, skips, order)
    c0 = x.shape[1]
    xn = GroupNormSiLU.apply(x, unet.conv_norm_out.weight, unet.conv_norm_out.bias, N, H * W, 1, 1e-5, True)
    return GatherGemm.apply(xn, unet.conv_out.weight, unet.conv_out.bias, None, None, 1.0,
                            dict(mode=1, conv=(H, W, H, W, 1, 0)))


# ------------------------------------------------------------------------------------------------- loss / step
def rows_of(x5):
    """(B, F, C, h, w) -> channels-last rows [B*F*h*w, C] (torch permute: tiny latent tensors)."""
    B, F, C, h, w = x5.shape
    return x5.permute(0, 1, 3, 4, 2).reshape(B * F * h * w, C)


def edm_loss(pred_rows, noisy_latents, target_latents, sigmas):
    """train_video_controlnet.py:468-478: denoised = c_out * pred + c_skip * noisy, weighting (1 + s^2) / s^2, mean over
    everything but the batch, then over the batch.  latents (B, F, 4, h, w) fp32; sigmas [B]."""
    B = noisy_latents.shape[0]
    per = noisy_latents[0].numel()
    s = sigmas.float().reshape(B, 1).repeat_interleave(per // noisy_latents.shape[2], 0)          # one per row
    c_out, c_skip = -s / (s * s + 1) ** 0.5, 1.0 / (s * s + 1)
    den = pred_rows.float() * c_out + c_skip * rows_of(noisy_latents).float()
    wgt = (1 + s * s) / (s * s)
    err = wgt * (den - rows_of(target_latents).float()) ** 2
    return err.reshape(B, -1).mean(dim=1).mean()


def train_step(controlnet, unet, batch, optimizer=None, conditioning_scale=1.0, world_size=1, buckets=None,
               accumulate=False, loss_scale=1.0):
    """One optimisation step.  Data parallel: pass `buckets=GradientBuckets(params)` (all-reduce overlapped with the
    backward pass), or only `world_size > 1` for the simple reduce-after-backward path.
    Gradient accumulation (`accelerator.accumulate`, gradient_accumulation_steps of train_video_controlnet.py:376): call
    with accumulate=True and loss_scale=1/steps for every micro-batch but the last -- gradients add up locally, nothing is
    all-reduced and the optimizer does not step; the last micro-batch (accumulate=False) reduces the sums and steps.  batch: dict(latents (B,F,4,h,w) clean, noise, sigmas [B], image_latents (B,F,4,h,w)
    (conditioning frame repeated), control_cond (B,F,4,h,w), encoder_hidden_states (B,1,D), added_time_ids (B,3)).
    Returns the loss as a device tensor; the one host read of the step is the batched mix-factor fetch at its start."""
    from .autograd import prefetch_mix_factors
    prefetch_mix_factors(controlnet, unet)          # the step's only device-to-host read (about 60 scalars, one copy)
    lat, noise, sig = batch["latents"].float(), batch["noise"].float(), batch["sigmas"].float()
    B = lat.shape[0]
    s5 = sig.reshape(B, 1, 1, 1, 1)
    noisy = lat + noise * s5                                                    # EulerDiscreteScheduler.add_noise
    inp = noisy / (s5 * s5 + 1) ** 0.5                                          # :410
    timesteps = 0.25 * torch.log(sig)                                           # continuous timestep of sigma
    sample = torch.cat([inp, batch["image_latents"].float()], dim=2).to(torch.bfloat16)
    down, mid = controlnet_train_forward(controlnet, sample, timesteps, batch["encoder_hidden_states"],
                                         batch["added_time_ids"], batch["control_cond"].to(torch.bfloat16),
                                         conditioning_scale)
    pred = unet_train_forward(unet, sample, timesteps, batch["encoder_hidden_states"], batch["added_time_ids"], down, mid)
    loss = edm_loss(pred, noisy, lat, sig)
    if buckets is not None:
        buckets.enabled = not accumulate
    (loss * loss_scale if loss_scale != 1.0 else loss).backward()
    if accumulate:
        return loss.detach()
    if buckets is not None:
        buckets.finish()
    elif world_size > 1:
        allreduce_gradients([p for p in controlnet.parameters() if p.requires_grad])
    # parameters without a gradient path get ZERO gradients (what the reference's autograd gives them: weight decay then
    # applies), whatever the world size -- the data-parallel paths write zeros for them, so does the single-GPU one
    for p in controlnet.parameters():
        if p.requires_grad and p.grad is None:
            p.grad = torch.zeros_like(p)
    if optimizer is not None:
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    return loss.detach()


# ------------------------------------------------------------------------------------------------- data parallel
class GradientBuckets:
    """DDP-style overlap of the gradient all-reduce with the backward pass (reference: torch DDP under `accelerate`,
    train_video_controlnet.py:225,485).  Parameters are grouped into flat fp32 buckets of `bucket_bytes` in REVERSE
    registration order (the order in which autograd finishes them); a post-accumulate-grad hook on every parameter counts
    its bucket down and launches the bucket's asynchronous all-reduce (RCCL: on its own stream) the moment the last
    gradient of the bucket exists, while the backward pass keeps running.  `finish()` -- after `loss.backward()` --
    flushes buckets whose parameters got no gradient (zeros: every rank issues identical collectives), waits, divides by
    the world size and writes the averaged gradients back.  One instance per model; re-armed by `finish()`.

    Collective ORDER is fixed: bucket i is launched only after buckets 0 .. i-1 (the expected completion order), whatever
    the order in which the hooks fire -- ranks whose graphs differ (a branch that is taken on one rank only) still issue
    the same sequence of all-reduces.  A parameter learnt as "unused" that receives a gradient after all, after its
    bucket has gone out, marks the bucket DIRTY: `finish()` agrees on the dirty set across ranks (one MAX all-reduce of a
    byte mask) and reduces those buckets again with the gradient in place, so no gradient is ever dropped."""

    def __init__(self, params, bucket_bytes=25 * 1024 * 1024, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.active else 1
        self.buckets, cur, size = [], [], 0
        for p in reversed([p for p in params if p.requires_grad]):
            nbytes = p.numel() * 4
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self.bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self.enabled = True               # False = accumulate locally (accelerate's no_sync micro-batches): no collective
        self.launch_order = []
        self.unused = set()               # ids of parameters without a gradient path, learnt in the first step
        self._arm()
        self.handles = [p.register_post_accumulate_grad_hook(self._hook) for b in self.buckets for p in b] if self.active else []

    def _arm(self):
        # parameters that received no gradient in the previous step (the one-key cross-attentions' to_q / to_k / norm2:
        # softmax over one key is constant) never fire their hook: they are not waited for, so their buckets are
        # all-reduced DURING the backward pass like every other one (their slots carry zeros)
        self.pending = [len(b) - sum(1 for p in b if id(p) in self.unused) for b in self.buckets]
        self.inflight = [None] * len(self.buckets)
        self.launch_order = []
        self.fired = set()
        self.next = 0                     # buckets [0, next) have been launched: the collective order is the bucket order
        self.dirty = set()                # buckets that went out before a (formerly unused) parameter's gradient arrived

    def _launch(self, i):
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).float().reshape(-1) for p in self.buckets[i]])
        self.inflight[i] = (flat, self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.launch_order.append(i)

    def _hook(self, p):
        if not self.enabled:
            return
        i = self.bucket_of[id(p)]
        self.fired.add(id(p))
        if id(p) in self.unused:          # it has a gradient after all: count it again from the next step on, and
            self.unused.discard(id(p))    # reduce its bucket again if that has already gone out without this gradient
            if self.inflight[i] is not None:
                self.dirty.add(i)
            return
        self.pending[i] -= 1
        while self.next < len(self.buckets) and self.pending[self.next] <= 0:
            self._launch(self.next)
            self.next += 1

    def finish(self):
        if not self.active or not self.enabled:
            return 0
        while self.next < len(self.buckets):
            self._launch(self.next)
            self.next += 1
        # late gradients: every rank must re-reduce the same buckets, in the same order
        mask = torch.zeros(len(self.buckets), dtype=torch.uint8, device=self.inflight[0][0].device)
        for i in self.dirty:
            mask[i] = 1
        self.dist.all_reduce(mask, op=self.dist.ReduceOp.MAX, group=self.group)
        redo = [i for i, v in enumerate(mask.tolist()) if v]
        for i in redo:
            self.inflight[i][1].wait()
            self._launch(i)
        for i, b in enumerate(self.buckets):
            flat, work = self.inflight[i]
            work.wait()
            flat.div_(self.world)
            off = 0
            for p in b:
                n = p.numel()
                g = flat[off:off + n].reshape(p.shape).to(p.dtype)
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.copy_(g)
                off += n
        n = len(self.buckets)
        self.unused = {id(p) for b in self.buckets for p in b if id(p) not in self.fired}
        self._arm()
        return n

    def remove(self):
        for h in self.handles:
            h.remove()
        self.handles = []


def allreduce_gradients(params, bucket_bytes=25 * 1024 * 1024, group=None):
    """Average the gradients over the process group in flat fp32 buckets (reference: DDP's 25 MB default under
    `accelerate`, train_video_controlnet.py:225,485).  All buckets are launched asynchronously before the first wait, so
    the transfers queue back to back on the collective stream; parameters without a gradient contribute zeros (every
    rank must issue identical collectives)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    buckets, cur, size = [], [], 0
    for p in params:
        nbytes = p.numel() * 4
        if cur and size + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(p)
        size += nbytes
    if cur:
        buckets.append(cur)
    pending = []
    for b in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).float().reshape(-1) for p in b])
        pending.append((b, flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)))
    for b, flat, work in pending:
        work.wait()
        flat.div_(world)
        off = 0
        for p in b:
            n = p.numel()
            g = flat[off:off + n].reshape(p.shape).to(p.dtype)
            if p.grad is None:
                p.grad = g
            else:
                p.grad.copy_(g)
            off += n
    return len(buckets)
