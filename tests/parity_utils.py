"""Shared parity harness: the HIP models vs the CPU oracle on identical seeded weights and inputs.

Used by tests/test_models_gpu.py, __graft_entry__.smoke() and bench.py (checker only, never the measured path)."""
import torch


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def make_pair(config, device, seed=0, zero_conv_std=0.02, time_context_order="sb", dtype=torch.bfloat16):
    """(oracle_unet, oracle_ctrl, hip_unet, hip_ctrl) sharing one seeded state dict.  The oracle weights are
    rounded to bf16 first so both sides see exactly the same parameters."""
    import ctrlv_ref as R
    from ctrlv_amd.models import ControlNetModel, UNetSpatioTemporalConditionModel
    ou = R.UNetSpatioTemporalConditionModel(time_context_order=time_context_order, **config)
    R.seeded_init_(ou, seed)
    oc = R.ControlNetModel.from_unet(ou, time_context_order=time_context_order)
    R.seeded_init_(oc, seed + 1, zero_conv_std=zero_conv_std)
    for m in (ou, oc):
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(p.to(torch.bfloat16).float())
        m.eval()
    hu = UNetSpatioTemporalConditionModel(time_context_order=time_context_order, **config)
    hc = ControlNetModel.from_unet(hu, load_weights_from_unet=False)
    hu.load_state_dict(ou.state_dict())
    hc.load_state_dict(oc.state_dict())
    hu.to(device=device, dtype=dtype).eval()
    hc.to(device=device, dtype=dtype).eval()
    return ou, oc, hu, hc


def make_inputs(config, B, F, h, w, seed=123):
    g = torch.Generator().manual_seed(seed)
    dc = config["cross_attention_dim"]
    sample = torch.randn(B, F, config["in_channels"], h, w, generator=g)
    cond = torch.randn(B, F, config["in_channels"] // 2, h, w, generator=g)
    ehs = torch.randn(B, 1, dc, generator=g)
    ids = torch.tensor([[6.0, 127.0, 0.02]] * B)
    if B > 1:                                   # CFG layout: unconditional half is zeros
        ehs[: B // 2] = 0
        cond[: B // 2] = 0
    t = torch.tensor(1.6377)
    bf = lambda x: x.to(torch.bfloat16).float()   # noqa: E731
    return bf(sample), t, bf(ehs), ids, bf(cond)


@torch.no_grad()
def run_tiny_parity(device="cuda:0", B=2, F=3, h=16, w=16, time_context_order="sb", verbose=False):
    import ctrlv_ref as R
    cfg = {k: v for k, v in R.TINY_CONFIG.items()}
    ou, oc, hu, hc = make_pair(cfg, device, time_context_order=time_context_order)
    sample, t, ehs, ids, cond = make_inputs(cfg, B, F, h, w)
    d_ref, m_ref = oc(sample, t, ehs, ids, control_cond=cond, conditioning_scale=0.8)
    y_ref = ou(sample, t, ehs, ids, d_ref, m_ref)[0]
    y0_ref = ou(sample, t, ehs, ids)[0]
    dev = lambda x: x.to(device=device, dtype=torch.bfloat16)   # noqa: E731
    d_hip, m_hip = hc(dev(sample), t.to(device), dev(ehs), ids.to(device), control_cond=dev(cond),
                      conditioning_scale=0.8, return_dict=False)
    y_hip = hu(dev(sample), t.to(device), dev(ehs), ids.to(device), d_hip, m_hip, return_dict=False)[0]
    y0_hip = hu(dev(sample), t.to(device), dev(ehs), ids.to(device)).sample
    torch.cuda.synchronize()
    err = {
        "controlnet_down": max(rel_l2(a, b) for a, b in zip(d_hip, d_ref)),
        "controlnet_mid": rel_l2(m_hip, m_ref),
        "unet": rel_l2(y_hip, y_ref),
        "unet_no_ctrl": rel_l2(y0_hip, y0_ref),
    }
    if verbose:
        print("tiny parity (rel-L2 vs fp32 oracle):", err)
    return err
