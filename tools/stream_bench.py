#!/usr/bin/env python
"""How fast do the streaming kernels run against a plain device copy of the same bytes?  (L0 activation: 460 800 x 320,
295 MB -- larger than the 256 MB Infinity Cache; buffers rotated so that nothing is served from a cache.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops  # noqa: E402

DEV = "cuda:0"
M, C = 50 * 9216, 320
g = torch.Generator(device=DEV).manual_seed(0)
NS = 6
xs = [torch.randn(M, C, generator=g, device=DEV).to(torch.bfloat16) for _ in range(NS)]
ys = [torch.empty_like(x) for x in xs]
gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
part = torch.empty(ops.groupnorm_scratch_floats(50, 9216, C, 1), dtype=torch.float32, device=DEV)


def timed(name, fn, nbytes):
    for i in range(NS):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for r in range(4):
        for i in range(NS):
            fn(i)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / (4 * NS)
    print(f"{name:34s} {ms * 1e3:8.1f} us   {nbytes / ms / 1e9:6.2f} TB/s")


b = 2 * M * C * 2
timed("torch copy_ (1R + 1W)", lambda i: ys[i].copy_(xs[i]), b)
timed("ctrlv_layernorm (1R + 1W)", lambda i: ops.layernorm(xs[i], gamma, beta, 1e-5, ys[i]), b)
timed("ctrlv_axpby (2R + 1W)", lambda i: ops.axpby(xs[i], xs[(i + 1) % NS], 1.0, 1.0, ys[i]), 3 * M * C * 2)
timed("ctrlv_groupnorm (2R + 1W)", lambda i: ops.groupnorm(xs[i], None, 50, 9216, C, 1, gamma, beta, 1e-5, True, ys[i], part),
      3 * M * C * 2)
timed("torch add (2R + 1W)", lambda i: torch.add(xs[i], xs[(i + 1) % NS], out=ys[i]), 3 * M * C * 2)
