"""cfg5 training step (reference: tools/train_video_controlnet.py:451-488) on the HIP kernels against torch.autograd on the
oracle: ControlNet forward (trainable) -> frozen UNet forward with the residuals -> EDM loss -> backward.  Checked: the loss,
the gradient of EVERY ControlNet parameter, that the frozen UNet receives no gradient, one AdamW step.

Tolerance: the gradients travel through ~40 blocks of bf16-stored activations and activation gradients; per-parameter
rel-L2 against the fp32 oracle is printed next to the yardstick (the same step in plain torch bf16 on the GPU); bounds:
concatenation of all gradients < 3e-2 and < 1.5x the yardstick; per parameter < 8e-2 and < 3x the yardstick."""
import math

import pytest
import torch

from tests.parity_utils import make_pair, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(config, B, F, h, w, seed=5):
    g = torch.Generator().manual_seed(seed)
    bf = lambda x: x.to(torch.bfloat16).float()   # noqa: E731
    dc = config["cross_attention_dim"]
    return dict(latents=bf(torch.randn(B, F, 4, h, w, generator=g)), noise=bf(torch.randn(B, F, 4, h, w, generator=g)),
                sigmas=torch.tensor([1.3] * B), image_latents=bf(torch.randn(B, 1, 4, h, w, generator=g)).repeat(1, F, 1, 1, 1),
                control_cond=bf(torch.randn(B, F, 4, h, w, generator=g)),
                encoder_hidden_states=bf(torch.randn(B, 1, dc, generator=g)),
                added_time_ids=torch.tensor([[6.0, 127.0, 0.02]] * B))


def _oracle_step(ou, oc, b, scale):
    """The reference's step on the oracle modules, in whatever dtype / device they live (fp32 CPU = the reference
    result; bf16 on the GPU = the yardstick: what plain torch bf16 training gives)."""
    p0 = next(oc.parameters())
    b = {k: v.to(device=p0.device, dtype=p0.dtype) for k, v in b.items()}
    lat, noise, sig = b["latents"], b["noise"], b["sigmas"]
    B = lat.shape[0]
    s5 = sig.reshape(B, 1, 1, 1, 1)
    noisy = lat + noise * s5
    inp = noisy / (s5 * s5 + 1) ** 0.5
    sample = torch.cat([inp, b["image_latents"]], dim=2).to(torch.bfloat16).to(p0.dtype)
    t = 0.25 * torch.log(sig.float())[0]
    for p in ou.parameters():
        p.requires_grad_(False)
    d, m = oc(sample, t, b["encoder_hidden_states"], b["added_time_ids"], control_cond=b["control_cond"],
              conditioning_scale=scale)
    pred = ou(sample, t, b["encoder_hidden_states"], b["added_time_ids"], d, m)[0]
    c_out, c_skip = -s5 / (s5 * s5 + 1) ** 0.5, 1 / (s5 * s5 + 1)
    den = pred * c_out + c_skip * noisy
    wgt = (1 + s5 ** 2) * s5 ** -2.0
    loss = (wgt.float() * (den.float() - lat.float()) ** 2).reshape(B, -1).mean(dim=1).mean()
    loss.backward()
    return loss.detach()


def test_train_step_matches_oracle_autograd(hip_lib):
    import ctrlv_ref as R
    from ctrlv_amd.training import train_step
    config = dict(R.TINY_CONFIG)
    B, F, h, w = 1, 3, 16, 16
    ou, oc, hu, hc = make_pair(config, DEV, seed=3)
    b = _batch(config, B, F, h, w)
    loss_ref = _oracle_step(ou, oc, b, 0.8)

    hc.float()                                    # fp32 master parameters, bf16 compute
    for p in hc.parameters():
        p.requires_grad_(True)
    for p in hu.parameters():
        p.requires_grad_(False)
    bd = {k: v.to(DEV) for k, v in b.items()}
    loss = train_step(hc, hu, bd, optimizer=None, conditioning_scale=0.8)
    torch.cuda.synchronize()
    print(f"  loss: HIP {float(loss):.6f}  oracle {float(loss_ref):.6f}")
    assert abs(float(loss) - float(loss_ref)) <= 5e-3 * abs(float(loss_ref))
    assert all(p.grad is None for p in hu.parameters())
    refp = dict(oc.named_parameters())
    gmax = max(float(q.grad.abs().max()) for q in refp.values() if q.grad is not None)
    errs, got_all, ref_all = {}, [], []
    for name, p in hc.named_parameters():
        rg = refp[name].grad
        if rg is None or float(rg.abs().max()) <= 1e-6 * gmax:          # one-key cross-attention: to_q, to_k, norm2
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        pg = p.grad.float().cpu().reshape(rg.shape)
        errs[name] = rel_l2(pg, rg)
        got_all.append(pg.reshape(-1)); ref_all.append(rg.reshape(-1))
    # yardstick: the same step in plain torch bf16 on the GPU (bf16 parameters, activations and autograd)
    import copy
    yu, yc = copy.deepcopy(ou).to(DEV, torch.bfloat16), copy.deepcopy(oc).to(DEV, torch.bfloat16)
    for q in yc.parameters():
        q.grad = None
    _oracle_step(yu, yc, b, 0.8)
    yp = dict(yc.named_parameters())
    yard = {n: rel_l2(yp[n].grad.float().cpu(), refp[n].grad) for n in errs}
    ytot = rel_l2(torch.cat([yp[n].grad.float().cpu().reshape(-1) for n in errs]), torch.cat(ref_all))
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:10]
    for k, v in worst:
        print(f"  {v:.2e} (torch bf16: {yard[k]:.2e})  d/d {k}")
    tot = rel_l2(torch.cat(got_all), torch.cat(ref_all))
    print(f"  {len(errs)} parameter gradients, all concatenated: rel-L2 {tot:.2e}   (torch bf16: {ytot:.2e})")
    assert tot < 3e-2 and tot < 1.5 * ytot
    # per parameter.  The scalar mix factors are cancelling sums over whole activation tensors (relative error is
    # unbounded for both implementations -- the yardstick shows 10 %..600 % on some): they are judged on the absolute
    # error against the largest mix-factor gradient of the model; everything else relatively.
    hp = dict(hc.named_parameters())
    mixmax = max(float(refp[n].grad.abs().max()) for n in errs if n.endswith("mix_factor"))
    mix_worst = 0.0
    for n, v in errs.items():
        if n.endswith("mix_factor"):
            ae = float((hp[n].grad.float().cpu().reshape(-1) - refp[n].grad.reshape(-1)).abs().max())
            mix_worst = max(mix_worst, ae / mixmax)
            assert ae < 5e-2 * mixmax, (n, ae, mixmax)
        else:
            # (tiny widths: C/32 = 1-4 channels per group make single norm / projection gradients noisy -- worst measured
            # 6.6e-2 where torch's own bf16 run is at 8.0e-2; the production-width test bounds every parameter by 5e-2)
            assert v < 8e-2 and v < max(2.0 * yard[n], 2e-2), (n, v, yard[n])
    print(f"  mix factors: worst absolute error / largest mix-factor gradient = {mix_worst:.2e} (bound 5e-2)")


def test_adamw_step_moves_the_zero_convs(hip_lib):
    """Two optimisation steps with torch.optim.AdamW on the fp32 masters: the loss is finite, every trainable parameter
    with a gradient moves, the frozen UNet does not."""
    import ctrlv_ref as R
    from ctrlv_amd.training import train_step
    config = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(config, DEV, seed=4, zero_conv_std=0.0)       # zero-convs at their real initial value
    hc.float()
    for p in hu.parameters():
        p.requires_grad_(False)
    before_u = [p.detach().clone() for p in hu.parameters()]
    zc0 = hc.controlnet_down_blocks[0].weight.detach().clone()
    opt = torch.optim.AdamW([p for p in hc.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-2)
    bd = {k: v.to(DEV) for k, v in _batch(config, 1, 3, 16, 16, seed=9).items()}
    losses = [float(train_step(hc, hu, bd, optimizer=opt)) for _ in range(2)]
    print("  losses:", losses)
    assert all(math.isfinite(v) for v in losses)
    assert float((hc.controlnet_down_blocks[0].weight.detach() - zc0).abs().max()) > 0.0      # zero-convs start to learn
    assert all(torch.equal(a, p) for a, p in zip(before_u, hu.parameters()))


def test_forward_after_an_optimizer_step_uses_the_updated_weights(hip_lib):
    """train_step updates the fp32 masters in place; the packed weights of BOTH inference executors (Python
    packing and the library-owned C++ plan) must follow.  forward -> two AdamW steps -> forward again: the ControlNet's
    output changes and matches the oracle evaluated with the updated parameters (the reference's periodic
    log_validation call, tools/train_video_controlnet.py)."""
    import ctrlv_ref as R
    from tests.parity_utils import make_inputs, parity_err
    from ctrlv_amd.training import train_step
    config = dict(R.TINY_CONFIG)
    ou, oc, hu, hc = make_pair(config, DEV, seed=6)
    hc.float()
    for p in hu.parameters():
        p.requires_grad_(False)
    sample, t, ehs, ids, cond = make_inputs(config, 1, 3, 16, 16)
    dev = lambda x: x.to(device=DEV, dtype=torch.float32)   # noqa: E731

    def fwd(executor):
        hc.executor = executor
        with torch.no_grad():
            d, m = hc(dev(sample), t.to(DEV), dev(ehs), ids.to(DEV), control_cond=dev(cond), conditioning_scale=1.0,
                      return_dict=False)
        torch.cuda.synchronize()
        return m.float().cpu()

    before = {ex: fwd(ex) for ex in ("plan", "python")}
    opt = torch.optim.AdamW([p for p in hc.parameters() if p.requires_grad], lr=3e-3, weight_decay=0.0)
    bd = {k: v.to(DEV) for k, v in _batch(config, 1, 3, 16, 16, seed=9).items()}
    for _ in range(2):
        train_step(hc, hu, bd, optimizer=opt)
    oc.load_state_dict({k: v.detach().float().cpu().to(torch.bfloat16).float() for k, v in hc.state_dict().items()})
    with torch.no_grad():
        _, ref = oc(sample, t, ehs, ids, control_cond=cond, conditioning_scale=1.0)
    for ex in ("plan", "python"):
        after = fwd(ex)
        assert float((after - before[ex]).abs().max()) > 0.0, ex           # the forward saw the optimizer's update
        assert parity_err(after, ref.float(), f"mid residual after training, {ex} executor") < 2e-2


def test_gradient_checkpointing_gives_the_same_step_with_less_memory(hip_lib):
    """`enable_gradient_checkpointing()` (tools/train_video_controlnet.py:185-186): the GEGLU feed-forward intermediates are
    recomputed in the backward by the forward's own launch (the same bits), so the loss AND EVERY GRADIENT are bit-identical
    to the plain step (round 5: the parameter-gradient reductions are ordered sums, ctrlv_amd.ops.DETERMINISTIC; with the
    fp32 atomics of round 4 two plain steps differed in their last bits) -- and the step's peak memory is lower."""
    import ctrlv_ref as R
    from ctrlv_amd.training import train_step
    config = dict(R.TINY_CONFIG)
    _, _, hu, hc = make_pair(config, DEV, seed=3)
    b = {k: v.to(DEV) for k, v in _batch(config, 2, 4, 32, 32).items()}
    hc.float()
    for p in hc.parameters():
        p.requires_grad_(True)
    for p in hu.parameters():
        p.requires_grad_(False)
    res = {}
    train_step(hc, hu, b, conditioning_scale=0.8)       # warm-up: packed-weight caches and scratch buffers are allocated once
    for ck in (True, False):
        (hu.enable_gradient_checkpointing if ck else hu.disable_gradient_checkpointing)()
        (hc.enable_gradient_checkpointing if ck else hc.disable_gradient_checkpointing)()
        for p in hc.parameters():
            p.grad = None
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loss = train_step(hc, hu, b, conditioning_scale=0.8)
        torch.cuda.synchronize()
        res[ck] = (float(loss), [p.grad.clone() for p in hc.parameters()], torch.cuda.max_memory_allocated() - base)
    (l0, g0, m0), (l1, g1, m1) = res[False], res[True]
    assert l0 == l1 and math.isfinite(l0)
    worst = 0.0
    for a, c in zip(g0, g1):
        if float(a.abs().max()) > 0:
            worst = max(worst, rel_l2(c, a))
    print(f"  worst per-parameter rel-L2 between the checkpointed and the plain step: {worst:.2e}")
    assert worst < 1e-5
    assert all(torch.equal(a, c) for a, c in zip(g0, g1))
    print(f"  peak above the resident set: plain {m0 / 2**20:.1f} MiB, checkpointed {m1 / 2**20:.1f} MiB")
    assert m1 < 0.9 * m0


def test_geglu_checkpointing_is_bit_identical_at_c1280_and_quiet_without_grad(hip_lib):
    """ADVICE r05.  (1) At Cin = 1280 the GEGLU projection runs on the 16x16x32 core; the checkpoint's recompute launch
    carries raw_out, the checkpointed forward does not -- the routing must not depend on raw_out, so the recomputed u / raw
    have the forward's bits and the gradients of the checkpointed and the plain feed-forward are `torch.equal` (the tiny
    config of the step-level test has no Cin >= 1280 layer).  (2) A forward under torch.no_grad() inside a checkpointed
    context (the trainer's periodic validation) registers no recompute cell: nothing leaks, nothing raises."""
    from ctrlv_amd import autograd as A
    C, M = 1280, 512
    gen = torch.Generator().manual_seed(11)
    x0 = (torch.randn(M, C, generator=gen)).to(torch.bfloat16).to(DEV)
    w1 = (torch.randn(8 * C, C, generator=gen) / C ** 0.5).to(DEV)
    b1 = (torch.randn(8 * C, generator=gen) * 0.1).to(DEV)
    w2 = (torch.randn(C, 4 * C, generator=gen) / (4 * C) ** 0.5).to(DEV)
    b2 = (torch.randn(C, generator=gen) * 0.1).to(DEV)
    dy = torch.randn(M, C, generator=gen).to(torch.bfloat16).to(DEV)

    def run(ck):
        x = x0.clone().requires_grad_(True)
        ps = [p.clone().requires_grad_(True) for p in (w1, b1, w2, b2)]
        with A.gradient_checkpointing(ck):
            with A._ff_region():
                u = A.GegluProj.apply(x, ps[0], ps[1])
                y = A.FusedLinear.apply(u, ps[2], ps[3], None, None, None, {})
        y.backward(dy)
        return [y.detach()] + [t.grad for t in [x] + ps]
    plain, ckpt = run(False), run(True)
    for a, c in zip(plain, ckpt):
        assert torch.equal(a, c)
    with A.gradient_checkpointing(True), torch.no_grad():
        with A._ff_region():
            u = A.GegluProj.apply(x0, w1, b1)
            y = A.FusedLinear.apply(u, w2, b2, None, None, None, {})
    assert torch.equal(y, plain[0]) and not A._CELLS
