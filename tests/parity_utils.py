"""Shared parity harness: the HIP models vs the CPU oracle on identical seeded weights and inputs.

Used by tests/, __graft_entry__.smoke() and bench.py's checker leg (never the measured path).

Error measures.  `parity_err(a, ref)` is what every `-m gpu` test bounds.  It returns the LARGER of
  * the relative L2 error  ||a - ref|| / ||ref||, and
  * the worst element error in units of the tolerance's own scale:  max |a - ref| / (6 rms(ref) + 2 |ref|)
so that one assertion `parity_err(a, ref) < tol` states both `rel-L2 < tol` and, element-wise,
`|a - ref| < tol * (6 rms(ref) + 2 |ref|)` (a torch.testing.assert_close with atol = 6 tol rms, rtol = 2 tol).
A rel-L2 alone cannot see row-local defects (one corrupted row in 460 800 contributes 1.5e-3): a corrupted element
is off by ~rms(ref), i.e. 50x the element bound at tol = 3e-3.  For error that is Gaussian with the rel-L2's sigma the
maximum over 1e7 elements sits at ~5.5 sigma, inside the 6 rms allowance; the 2 |ref| term covers bf16's value-
proportional output rounding (2^-9 |ref|) on outliers.
"""
import copy

import torch


def _prep(a, b):
    if a.device != b.device:          # mixed devices: compare on the host; otherwise stay where the tensors are
        a, b = a.cpu(), b.cpu()
    return a.float(), b.float()


def rel_l2(a, b):
    a, b = _prep(a, b)
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def max_err(a, b):
    """max |a-b| / (6 rms(b) + 2 |b|): the element-wise companion of rel_l2 (same tolerance applies to both)."""
    a, b = _prep(a, b)
    rms = b.pow(2).mean().sqrt().clamp_min(1e-20)
    return ((a - b).abs() / (6.0 * rms + 2.0 * b.abs())).max().item()


def parity_err(a, b, what=None):
    r, m = rel_l2(a, b), max_err(a, b)
    if what is not None:
        print(f"  {what}: rel-L2 {r:.3e}  max-elem {m:.3e}")
    if not (r == r and m == m):
        return float("inf")
    return max(r, m)


def make_pair(config, device, seed=0, zero_conv_std=0.02, time_context_order="sb", dtype=torch.bfloat16,
              lean=False):
    """(oracle_unet, oracle_ctrl, hip_unet, hip_ctrl) sharing one seeded state dict.  The oracle weights are
    rounded to `dtype` first (bf16, or fp16 for the libctrlv_hip_f16.so models) so both sides see exactly the same
    parameters.  `lean`: build the HIP models straight on
    the device (no fp32 host copy) -- for the full-width configuration."""
    import ctrlv_ref as R
    from ctrlv_amd.models import ControlNetModel, UNetSpatioTemporalConditionModel
    ou = R.UNetSpatioTemporalConditionModel(time_context_order=time_context_order, **config)
    R.seeded_init_(ou, seed)
    oc = R.ControlNetModel.from_unet(ou, time_context_order=time_context_order)
    R.seeded_init_(oc, seed + 1, zero_conv_std=zero_conv_std)
    for m in (ou, oc):
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(p.to(dtype).float())
        m.eval()
    if lean:
        from ctrlv_amd.utils import build_on_device
        hu = build_on_device(UNetSpatioTemporalConditionModel, device, dtype, time_context_order=time_context_order,
                             **config)
        hc = build_on_device(ControlNetModel, device, dtype, time_context_order=time_context_order,
                             **{k: v for k, v in config.items() if k not in ("out_channels", "up_block_types")})
        hu.load_state_dict(ou.state_dict())
        hc.load_state_dict(oc.state_dict())
        return ou, oc, hu.eval(), hc.eval()
    hu = UNetSpatioTemporalConditionModel(time_context_order=time_context_order, **config)
    hc = ControlNetModel.from_unet(hu, load_weights_from_unet=False)
    hu.load_state_dict(ou.state_dict())
    hc.load_state_dict(oc.state_dict())
    hu.to(device=device, dtype=dtype).eval()
    hc.to(device=device, dtype=dtype).eval()
    return ou, oc, hu, hc


def set_context_order(models, order):
    """Switch the temporal cross-attention context order on already-built oracle / HIP models."""
    for m in models:
        m.time_context_order = order
        for sub in m.modules():
            if hasattr(sub, "time_context_order"):
                sub.time_context_order = order


def make_inputs(config, B, F, h, w, seed=123, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    dc = config["cross_attention_dim"]
    if not isinstance(dc, int):
        dc = dc[0]
    sample = torch.randn(B, F, config["in_channels"], h, w, generator=g)
    cond = torch.randn(B, F, config["in_channels"] // 2, h, w, generator=g)
    ehs = torch.randn(B, 1, dc, generator=g)
    ids = torch.tensor([[6.0, 127.0, 0.02]] * B)
    if B > 1:                                   # CFG layout: unconditional half is zeros
        ehs[: B // 2] = 0
        cond[: B // 2] = 0
    t = torch.tensor(1.6377)
    bf = lambda x: x.to(dtype).float()   # noqa: E731  (inputs exactly representable in the HIP models' dtype)
    return bf(sample), t, bf(ehs), ids, bf(cond)


@torch.no_grad()
def oracle_forward(ou, oc, inputs, scale=0.8, with_unet_no_ctrl=True):
    """ControlNet -> UNet (+ UNet without residuals) on whatever device / dtype the oracle modules live on."""
    sample, t, ehs, ids, cond = inputs
    p = next(ou.parameters())
    cv = lambda x: x.to(device=p.device, dtype=p.dtype)   # noqa: E731
    d, m = oc(cv(sample), t.to(p.device), cv(ehs), cv(ids), control_cond=cv(cond), conditioning_scale=scale)
    y = ou(cv(sample), t.to(p.device), cv(ehs), cv(ids), d, m)[0]
    y0 = ou(cv(sample), t.to(p.device), cv(ehs), cv(ids))[0] if with_unet_no_ctrl else None
    return dict(down=[x.float().cpu() for x in d], mid=m.float().cpu(), unet=y.float().cpu(),
                unet_no_ctrl=None if y0 is None else y0.float().cpu())


@torch.no_grad()
def hip_forward(hu, hc, inputs, device, scale=0.8, with_unet_no_ctrl=True):
    sample, t, ehs, ids, cond = inputs
    mdt = hu.dtype if hu.dtype in (torch.bfloat16, torch.float16) else torch.bfloat16
    dev = lambda x: x.to(device=device, dtype=mdt)   # noqa: E731
    d, m = hc(dev(sample), t.to(device), dev(ehs), ids.to(device), control_cond=dev(cond), conditioning_scale=scale,
              return_dict=False)
    y = hu(dev(sample), t.to(device), dev(ehs), ids.to(device), d, m, return_dict=False)[0]
    y0 = hu(dev(sample), t.to(device), dev(ehs), ids.to(device)).sample if with_unet_no_ctrl else None
    torch.cuda.synchronize()
    return dict(down=[x.float().cpu() for x in d], mid=m.float().cpu(), unet=y.float().cpu(),
                unet_no_ctrl=None if y0 is None else y0.float().cpu())


def compare(got, ref, fn=parity_err):
    e = {"controlnet_down": max(fn(a, b) for a, b in zip(got["down"], ref["down"])),
         "controlnet_mid": fn(got["mid"], ref["mid"]), "unet": fn(got["unet"], ref["unet"])}
    if got.get("unet_no_ctrl") is not None and ref.get("unet_no_ctrl") is not None:
        e["unet_no_ctrl"] = fn(got["unet_no_ctrl"], ref["unet_no_ctrl"])
    return e


@torch.no_grad()
def run_parity(config, device="cuda:0", B=2, F=3, h=16, w=16, time_context_order="sb", verbose=False, pair=None,
               torch_bf16=True, lean=False, with_unet_no_ctrl=True, dtype=torch.bfloat16):
    """Runs the HIP models and three oracle variants on the same inputs and returns the error tables

      fp32        HIP  vs the fp32 oracle                                  (rel-L2 and element bound, `parity_err`)
      storage     HIP  vs the fp32 oracle with `dtype` rounding at exactly the HIP path's storage points
      torch_bf16  PyTorch's own bf16 execution of the oracle (on the GPU: rocBLAS / MIOpen / SDPA) vs the fp32 oracle
                  -- the yardstick for "what bf16 costs through this network"; rel-L2 only
      fp32_l2     rel-L2 part of `fp32` alone (comparable with torch_bf16)
    """
    import ctrlv_ref as R
    ou, oc, hu, hc = pair if pair is not None else make_pair(config, device, time_context_order=time_context_order,
                                                             lean=lean, dtype=dtype)
    if pair is not None:
        set_context_order((ou, oc, hu, hc), time_context_order)
        dtype = hu.dtype
    inputs = make_inputs(config, B, F, h, w, dtype=dtype)
    ref = oracle_forward(ou, oc, inputs, with_unet_no_ctrl=with_unet_no_ctrl)
    with R.storage_rounding(dtype):
        ref_q = oracle_forward(ou, oc, inputs, with_unet_no_ctrl=with_unet_no_ctrl)
    got = hip_forward(hu, hc, inputs, device, with_unet_no_ctrl=with_unet_no_ctrl)
    out = {"fp32": compare(got, ref), "storage": compare(got, ref_q), "fp32_l2": compare(got, ref, rel_l2)}
    if torch_bf16:
        ob, cb = copy.deepcopy(ou).to(device, torch.bfloat16), copy.deepcopy(oc).to(device, torch.bfloat16)
        out["torch_bf16"] = compare(oracle_forward(ob, cb, inputs, with_unet_no_ctrl=with_unet_no_ctrl), ref, rel_l2)
        del ob, cb
    if verbose:
        for k, v in out.items():
            print(f"parity[{k}] " + "  ".join(f"{n}={e:.2e}" for n, e in v.items()))
    return out


@torch.no_grad()
def run_tiny_parity(device="cuda:0", B=2, F=3, h=16, w=16, time_context_order="sb", verbose=False,
                    dtype=torch.bfloat16):
    """Tiny-config HIP-vs-fp32-oracle errors (smoke() and the model tests)."""
    import ctrlv_ref as R
    return run_parity(dict(R.TINY_CONFIG), device, B, F, h, w, time_context_order, verbose, torch_bf16=False,
                      dtype=dtype)["fp32"]


@torch.no_grad()
def error_growth_trace(ou, hu, inputs, device, oc=None, hc=None):
    """Per-block relative L2 error of the HIP model against the fp32 oracle, in execution order: the output of every
    SpatioTemporalResBlock / TransformerSpatioTemporalModel of the ControlNet (if given) and the UNet.  Returns
    [(name, rel_l2)].  A wiring / kernel defect shows up as a jump at one block; bf16 storage noise as a slow walk."""
    import ctrlv_ref as R
    sample, t, ehs, ids, cond = inputs
    out = []
    for o_model, h_model, kind in ((oc, hc, "controlnet"), (ou, hu, "unet")):
        if o_model is None:
            continue
        ref, hooks, names = [], [], {}
        for name, mod in o_model.named_modules():
            if isinstance(mod, (R.SpatioTemporalResBlock, R.TransformerSpatioTemporalModel)):
                names[mod] = name
                hooks.append(mod.register_forward_hook(lambda m, a, y: ref.append((names[m], y.detach().float()))))
        hnames = {m: n for n, m in h_model.named_modules()}
        h_model._trace = []
        mdt = h_model.dtype if h_model.dtype in (torch.bfloat16, torch.float16) else torch.bfloat16
        dev = lambda x, mdt=mdt: x.to(device=device, dtype=mdt)   # noqa: E731
        try:
            if kind == "controlnet":
                d_ref, m_ref = o_model(sample, t, ehs, ids, control_cond=cond, conditioning_scale=0.8)
                d_hip, m_hip = h_model(dev(sample), t.to(device), dev(ehs), ids.to(device), control_cond=dev(cond),
                                       conditioning_scale=0.8, return_dict=False)
                res_ref, res_hip = (d_ref, m_ref), (d_hip, m_hip)
            else:
                r_ref = res_ref if oc is not None else (None, None)
                r_hip = res_hip if oc is not None else (None, None)
                o_model(sample, t, ehs, ids, *r_ref)
                h_model(dev(sample), t.to(device), dev(ehs), ids.to(device), *r_hip)
            torch.cuda.synchronize()
            got = [(hnames[m], rows, H, W) for m, rows, H, W in h_model._trace]
        finally:
            h_model._trace = None
            for h in hooks:
                h.remove()
        assert [n for n, _ in ref] == [g[0] for g in got], "block execution order differs between oracle and HIP"
        for (name, y), (_, rows, H, W) in zip(ref, got):
            n = y.shape[0]
            x = rows.float().cpu().reshape(n, H, W, -1).permute(0, 3, 1, 2)
            out.append((f"{kind}.{name}", rel_l2(x, y)))
    return out
