#!/usr/bin/env python
"""Developer probe: what does the epilogue's operand set cost a layer?  Times the 72 x 128-level 3x3 conv (460800 x 320 x 2880),
the temporal conv (K = 960) and the 320 -> 320 Linear with {bias}, {bias, V}, {bias, R1} epilogues (buffer sets in rotation)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops, packing  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g, device=DEV)      # noqa: E731
B, F, H, W, C = 2, 25, 72, 128, 320
S, M = H * W, 2 * 25 * 72 * 128
NSET = 3
xs = [r(M, C).bfloat16() for _ in range(NSET)]
r1s = [r(M, C).bfloat16() for _ in range(NSET)]
outs = [torch.empty(M, C, dtype=torch.bfloat16, device=DEV) for _ in range(NSET)]
vt = r(B, C)
bias = r(C)
w3 = packing.pack_conv3x3(r(C, C, 3, 3) / (9 * C) ** 0.5)
wt = packing.pack_conv_temporal(r(C, C, 3, 1, 1) / (3 * C) ** 0.5)
wl = packing.pack_linear(r(C, C) / C ** 0.5)
layers = {"conv3x3": (w3, dict(N=C, cin=C, taps=9, mode=1, conv=(H, W, H, W, 1, 0))),
          "temporal": (wt, dict(N=C, cin=C, taps=3, mode=2, temporal=(F, S))),
          "linear": (wl, dict(N=C, cin=C))}
epis = {"bias": {}, "bias+V": dict(V=vt, vmode=1, vdiv=F * S), "bias+R1": None}
for name, (w, kw) in layers.items():
    for en, ek in epis.items():
        def run(i):
            e = dict(R1=r1s[i]) if ek is None else ek
            ops.gemm(xs[i], w, outs[i], bias=bias, **kw, **e)
        for i in range(NSET):
            run(i)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            for i in range(NSET):
                run(i)
        e.record(); torch.cuda.synchronize()
        print(f"{name:9s} {en:8s} {s.elapsed_time(e) / (3 * NSET) * 1e3:8.1f} us", flush=True)
