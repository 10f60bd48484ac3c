#!/usr/bin/env python
"""Developer aid: first KERNEL CALL whose output differs between a 2-clip forward and the 1-clip forward of clip 0
(bit-level), through the per-op Python executor."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops  # noqa: E402
from ctrlv_amd.models import ControlNetModel, UNetSpatioTemporalConditionModel  # noqa: E402
from ctrlv_amd.utils import build_on_device, random_init_  # noqa: E402

DEV = "cuda:0"
h, w, FR = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 64, 25
unet = random_init_(build_on_device(UNetSpatioTemporalConditionModel, DEV, num_frames=FR), seed=0)
unet.time_context_order = "bs"
unet.executor = "python"
ctrl = random_init_(build_on_device(ControlNetModel, DEV, num_frames=FR), seed=1, zero_conv_std=0.02)
ctrl.time_context_order = "bs"
ctrl.executor = "python"
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=DEV).to(torch.bfloat16)   # noqa: E731
sample, ehs, cond = rn(2, FR, 8, h, w), rn(2, 1, 1024), rn(2, FR, 4, h, w)
ids = torch.tensor([[6.0, 127.0, 0.02]] * 2, device=DEV, dtype=torch.bfloat16)
t = torch.tensor(0.25 * math.log(20.0), device=DEV)
log = []
orig = {}


def wrap(name, out_index):
    f = getattr(ops, name)
    orig[name] = f

    def g_(*a, **k):
        r = f(*a, **k)
        out = a[out_index] if out_index is not None else r
        desc = {kk: (vv if isinstance(vv, (int, float, tuple)) else (None if vv is None else "T")) for kk, vv in k.items()}
        log.append((name, desc, out.clone()))
        return r
    setattr(ops, name, g_)


wrap("gemm", 2)
wrap("layernorm", 4)
wrap("attention_spatial", 1)
wrap("attention_temporal", 1)
_gn = ops.groupnorm


def gn_(x, x2, n_img, S, C, ips, gamma, beta, eps, silu, y, partials):
    r = _gn(x, x2, n_img, S, C, ips, gamma, beta, eps, silu, y, partials)
    log.append(("groupnorm", dict(n_img=n_img, S=S, C=C, ips=ips, silu=silu), y.clone()))
    return r


ops.groupnorm = gn_
runs = {}
with torch.no_grad():
    unet(sample[:1, :, :, :8, :8], t, ehs[:1], ids[:1])        # packs weights, caches the frame-embedding tables
    ctrl(sample[:1, :, :, :8, :8], t, ehs[:1], ids[:1], control_cond=cond[:1, :, :, :8, :8])

    for key, sl in (("b2", slice(0, 2)), ("b1", slice(0, 1))):
        log.clear()
        down, mid = ctrl(sample[sl], t, ehs[sl], ids[sl], control_cond=cond[sl], return_dict=False)
        unet(sample[sl], t, ehs[sl], ids[sl], down, mid)
        torch.cuda.synchronize()
        runs[key] = list(log)
bad = 0
for i, ((n2, d2, o2), (n1, d1, o1)) in enumerate(zip(runs["b2"], runs["b1"])):
    if o2.dim() != 2 or o1.shape[0] * 2 != o2.shape[0] or o1.shape[1] != o2.shape[1] or n1 != n2:
        continue                                   # per-clip tables ([B, *]) and constants
    half = o2[: o1.shape[0]]
    if not torch.equal(half, o1):
        d = (half.float() - o1.float()).abs().max().item()
        print(f"op {i}: {n2} out {tuple(o1.shape)} max|diff| {d:.3e}\n    B=2 {d2}\n    B=1 {d1}")
        bad += 1
        if bad >= 4:
            break
print("ops compared:", len(runs["b1"]), "mismatching shown:", bad)
