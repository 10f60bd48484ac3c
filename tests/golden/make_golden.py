"""Generates tests/golden/*.npz from the CPU oracle (oracle/ctrlv_ref), fp32, seeded.

PROVENANCE / CAVEAT (recorded with every fixture): the reference (oooolga/Ctrl-V) has no tests or golden vectors and
its arithmetic lives in diffusers==0.27.2, which cannot be installed here, so these vectors come from this repo's
literal restatement of those blocks -- they pin the ORACLE against regressions and pin the HIP path to the oracle;
they do not pin the oracle to diffusers ("parity unpinned").  Anyone with diffusers 0.27.2 can validate them: the
weights are `seeded_init_` of modules whose state-dict keys are the diffusers names.

Run:  python tests/golden/make_golden.py      (CPU, ~10 s)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ctrlv_ref as R  # noqa: E402



def npy(t):
    return t.detach().float().numpy()


@torch.no_grad()
def scheduler_tables():
    out = {}
    for n in (25, 30, 50):
        s = R.EulerDiscreteScheduler()
        s.set_timesteps(n)
        out[f"sigmas_{n}"] = npy(s.sigmas)
        out[f"timesteps_{n}"] = npy(s.timesteps)
        out[f"init_noise_sigma_{n}"] = np.float32(float(s.init_noise_sigma))
    out["guidance_1_3_25"] = npy(R.guidance_scale_tensor(1.0, 3.0, 25, 1))
    np.savez_compressed(os.path.join(HERE, "scheduler_tables.npz"), **out)


@torch.no_grad()
def block_vectors():
    g = torch.Generator().manual_seed(7)
    B, F, H, W, C, temb_c = 1, 2, 8, 8, 64, 256
    out = {}
    ind = torch.zeros(B, F)
    x = torch.randn(B * F, C, H, W, generator=g)
    temb = torch.randn(B, temb_c, generator=g).repeat_interleave(F, 0)
    ehs = torch.randn(B, 1, 64, generator=g).repeat_interleave(F, 0)
    rb = R.SpatioTemporalResBlock(C, 128, temb_c, eps=1e-6)
    R.seeded_init_(rb, 11)
    out["res_x"], out["res_temb"], out["res_y"] = npy(x), npy(temb), npy(rb(x, temb, ind))
    tr = R.TransformerSpatioTemporalModel(1, 64, C, 64)
    R.seeded_init_(tr, 12)
    out["tr_x"], out["tr_ehs"], out["tr_y"] = npy(x), npy(ehs), npy(tr(x, ehs, ind))
    out.update(block_vectors_q())
    np.savez_compressed(os.path.join(HERE, "block_vectors.npz"), **out)


def q_(t):
    return t.to(torch.bfloat16).float()


@torch.no_grad()
def q_blocks():
    """The three block instances of the `q*` fixtures: weights `seeded_init_` then rounded to bf16 (what the HIP path
    holds), so the vectors isolate arithmetic / activation-storage differences from weight rounding."""
    rb = R.seeded_init_(R.SpatioTemporalResBlock(64, 128, 256, eps=1e-6), 21)          # with a 1x1 shortcut
    rc = R.seeded_init_(R.SpatioTemporalResBlock(64 + 128, 64, 256, eps=1e-5), 22)    # skip-concat input (up path)
    tr = R.seeded_init_(R.TransformerSpatioTemporalModel(2, 64, 128, 64), 23)
    for m in (rb, rc, tr):
        for p_ in m.parameters():
            p_.copy_(q_(p_))
    return rb, rc, tr


@torch.no_grad()
def block_vectors_q():
    """Batch-2 (CFG-shaped) block fixtures on bf16-rounded weights and inputs, B = 2 clips x F = 3 frames x 8 x 8.
    `*_y` = fp32 arithmetic; `*_ys` = fp32 arithmetic with bf16 rounding at the HIP path's storage points
    (ctrlv_ref.storage_rounding)."""
    g = torch.Generator().manual_seed(9)
    B, F, H, W = 2, 3, 8, 8
    ind = torch.zeros(B, F)
    rb, rc, tr = q_blocks()
    x = q_(torch.randn(B * F, 64, H, W, generator=g))
    skip = q_(torch.randn(B * F, 128, H, W, generator=g) * 1.5)
    xt = q_(torch.randn(B * F, 128, H, W, generator=g))
    temb_b = q_(torch.randn(B, 256, generator=g))
    ehs_b = q_(torch.randn(B, 1, 64, generator=g))
    temb, ehs = temb_b.repeat_interleave(F, 0), ehs_b.repeat_interleave(F, 0)
    out = {"q_x": npy(x), "q_skip": npy(skip), "q_xt": npy(xt), "q_temb": npy(temb_b), "q_ehs": npy(ehs_b)}
    for tag, ctx in (("y", None), ("ys", R.storage_rounding(torch.bfloat16))):
        if ctx is not None:
            ctx.__enter__()
        out["qres_" + tag] = npy(rb(x, temb, ind))
        out["qcat_" + tag] = npy(rc(torch.cat([x, skip], 1), temb, ind))
        for order in ("sb", "bs"):
            tr.time_context_order = order
            out[f"qtr_{order}_{tag}"] = npy(tr(xt, ehs, ind))
        if ctx is not None:
            ctx.__exit__(None, None, None)
    return out


@torch.no_grad()
def model_vectors():
    cfg = dict(R.TINY_CONFIG)
    out = {}
    for order in ("sb", "bs"):
        unet = R.UNetSpatioTemporalConditionModel(time_context_order=order, **cfg)
        R.seeded_init_(unet, 0)
        ctrl = R.ControlNetModel.from_unet(unet, time_context_order=order)
        R.seeded_init_(ctrl, 1, zero_conv_std=0.02)
        for B in (1, 2):
            g = torch.Generator().manual_seed(100 + B)
            sample = torch.randn(B, 3, 8, 8, 8, generator=g)
            cond = torch.randn(B, 3, 4, 8, 8, generator=g)
            ehs = torch.randn(B, 1, 64, generator=g)
            ids = torch.tensor([[6.0, 127.0, 0.02]] * B)
            t = torch.tensor(1.6377)
            down, mid = ctrl(sample, t, ehs, ids, control_cond=cond, conditioning_scale=1.0)
            k = f"{order}_b{B}"
            out[k + "_unet"] = npy(unet(sample, t, ehs, ids)[0])
            out[k + "_unet_ctrl"] = npy(unet(sample, t, ehs, ids, down, mid)[0])
            out[k + "_mid"] = npy(mid)
            out[k + "_down0"], out[k + "_down11"] = npy(down[0]), npy(down[11])
    # 3-step sampling trajectory (CFG, ControlNet) with the quirk order
    unet = R.UNetSpatioTemporalConditionModel(**cfg)
    R.seeded_init_(unet, 0)
    ctrl = R.ControlNetModel.from_unet(unet)
    R.seeded_init_(ctrl, 1, zero_conv_std=0.02)
    g = torch.Generator().manual_seed(1234)
    sched = R.EulerDiscreteScheduler()
    lat = torch.randn(1, 3, 4, 16, 16, generator=g) * sched.init_noise_sigma
    img = torch.randn(1, 4, 16, 16, generator=g)
    image_latents = torch.cat([torch.zeros_like(img), img]).unsqueeze(1).repeat(1, 3, 1, 1, 1)
    e = torch.randn(1, 1, 64, generator=g)
    ehs = torch.cat([torch.zeros_like(e), e])
    c = torch.randn(1, 3, 4, 16, 16, generator=g)
    cond = torch.cat([torch.zeros_like(c), c])
    ids = torch.tensor([[6.0, 127.0, 0.02]] * 2)
    rec = []
    R.sample_loop(unet, ctrl, sched, lat, image_latents, ehs, ids, cond, 3, record=rec)
    out["traj_init"] = npy(lat)
    for i, r in enumerate(rec):
        out[f"traj_step{i}"] = npy(r)
    np.savez_compressed(os.path.join(HERE, "model_vectors.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    scheduler_tables()
    block_vectors()
    model_vectors()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
