#!/usr/bin/env python
"""Times GroupNorm (stats + finalize + apply) and LayerNorm at the cfg3 layer shapes (developer tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
for name, n_img, S, C in [("L0", 50, 9216, 320), ("L0cat", 50, 9216, 640), ("L1", 50, 2304, 640), ("L2", 50, 576, 1280)]:
    x = torch.randn(n_img * S, C, generator=g, device=DEV).to(torch.bfloat16)
    y = torch.empty_like(x)
    gamma = torch.ones(C, device=DEV)
    beta = torch.zeros(C, device=DEV)
    for ips in (1, 25):
        part = torch.empty(ops.groupnorm_scratch_floats(n_img, S, C, ips), dtype=torch.float32, device=DEV)
        ops.groupnorm(x, None, n_img, S, C, ips, gamma, beta, 1e-5, True, y, part)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            ops.groupnorm(x, None, n_img, S, C, ips, gamma, beta, 1e-5, True, y, part)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        print(f"groupnorm {name} ips={ips:2d}: {ms * 1e3:8.1f} us  {2 * x.numel() * 2 / ms / 1e6:7.0f} GB/s (1R+1W)")
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.layernorm(x, gamma, beta, 1e-5, y)
    s.record()
    for _ in range(10):
        ops.layernorm(x, gamma, beta, 1e-5, y)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"layernorm {name}: {ms * 1e3:8.1f} us  {2 * x.numel() * 2 / ms / 1e6:7.0f} GB/s")
