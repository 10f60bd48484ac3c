#!/usr/bin/env python
"""Times the spatial / temporal attention kernels at the cfg3 layer shapes (developer tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
for name, n_img, S, C in [("L0", 50, 9216, 320), ("L1", 50, 2304, 640), ("L2", 50, 576, 1280), ("mid", 50, 144, 1280)]:
    qkv = torch.randn(n_img * S, 3 * C, generator=g, device=DEV).to(torch.bfloat16)
    out = torch.empty(n_img * S, C, dtype=torch.bfloat16, device=DEV)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fl = 4.0 * n_img * (C // 64) * S * S * 64
    qkv_pre = qkv.clone()
    qkv_pre[:, :C] = (qkv[:, :C].float() * ops.Q_PRESCALE).to(torch.bfloat16)
    for rnd in range(2):                     # interleaved rounds: plain / prescaled
        for pre, src in ((False, qkv), (True, qkv_pre)):
            ops.attention_spatial(src, out, n_img, S, C, prescaled=pre)
            torch.cuda.synchronize()
            s.record()
            for _ in range(5):
                ops.attention_spatial(src, out, n_img, S, C, prescaled=pre)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 5
            print(f"spatial {name} {'prescaled' if pre else 'plain    '}: {ms:8.3f} ms  {fl / ms / 1e9:7.0f} TFLOP/s")
    s.record()
    for _ in range(5):
        ops.attention_temporal(qkv, out, 2, 25, S, C)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"temporal {name}: {ms:8.3f} ms  {4 * n_img * S * C * 2 / ms / 1e6:7.0f} GB/s")
