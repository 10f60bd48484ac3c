#!/usr/bin/env python
"""Times the C = 320 feed-forward pair at M = 460 800: two ctrlv_gemm launches against ctrlv_ff_fused (developer tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops, packing  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
M, C, I = int(os.environ.get("FF_M", 50 * 9216)), 320, 1280
r = lambda *s: torch.randn(*s, generator=g, device=DEV)
w1p, b1p = packing.pack_geglu(r(2 * I, C) / C ** 0.5, r(2 * I))
w2p = packing.pack_linear(r(C, I) / I ** 0.5)
b1, b2 = b1p.float().contiguous(), r(C)
w1f, w2f = ops.ff_fused_pack(w1p, b1, w2p)
NSET = int(os.environ.get("FF_SETS", 4))       # rotate through buffer sets far larger than the 256 MB Infinity Cache
R2 = os.environ.get("FF_R2", "0") == "1"           # the AlphaBlender variant: + s2 * R2
sets = [dict(x=r(M, C).bfloat16(), r1=r(M, C).bfloat16(), r2=r(M, C).bfloat16(), u=torch.empty(M, I, dtype=torch.bfloat16, device=DEV),
             out=torch.empty(M, C, dtype=torch.bfloat16, device=DEV)) for _ in range(NSET)]


def two(b):
    ops.gemm(b["x"], w1p, b["u"], N=2 * I, cin=C, bias=b1, geglu=1)
    ops.gemm(b["u"], w2p, b["out"], N=C, cin=I, bias=b2, R1=b["r1"], **(dict(R2=b["r2"], s2=0.5, s_acc=0.5, s1=0.5) if R2 else {}))


def fused(b):
    ops.ff_fused(b["x"], w1f, w2f, b["out"], bias=b2, R1=b["r1"], **(dict(R2=b["r2"], s2=0.5, s_acc=0.5, s1=0.5) if R2 else {}))


for name, fn in (("two launches", two), ("fused", fused), ("two launches", two), ("fused", fused)):
    for b in sets:
        fn(b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        for b in sets:
            fn(b)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / (3 * NSET)
    print(f"{name:14s} {ms * 1e3:8.1f} us   {2.0 * M * C * (2 * I + I) / ms / 1e9:6.0f} TFLOP/s   ({NSET} buffer sets in rotation)")
