#!/usr/bin/env python
"""Developer aid: is a layer bit-identical whichever GEMM tile serves it?  (2-stage tile 1 vs ping-pong tiles 5 / 6)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops, packing  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=DEV)   # noqa: E731
F_, S, C = 25, 40, 1280
M = 2 * F_ * S
A = rn(M, C).to(torch.bfloat16)
R1 = rn(M, C).to(torch.bfloat16)
bias = rn(C)
V = rn(2, C)
cases = {
    "temporal s_acc+R1": dict(w=packing.pack_conv_temporal(rn(C, C, 3, 1, 1) / 60), kw=dict(taps=3, mode=2, temporal=(F_, S), s_acc=0.3775406687981454, R1=R1)),
    "temporal R1": dict(w=packing.pack_conv_temporal(rn(C, C, 3, 1, 1) / 60), kw=dict(taps=3, mode=2, temporal=(F_, S), R1=R1)),
    "temporal s_acc": dict(w=packing.pack_conv_temporal(rn(C, C, 3, 1, 1) / 60), kw=dict(taps=3, mode=2, temporal=(F_, S), s_acc=0.3775406687981454)),
    "temporal V": dict(w=packing.pack_conv_temporal(rn(C, C, 3, 1, 1) / 60), kw=dict(taps=3, mode=2, temporal=(F_, S), V=V, vmode=1, vdiv=F_ * S)),
    "linear s_acc+R1": dict(w=packing.pack_linear(rn(C, C) / 36), kw=dict(s_acc=0.3775406687981454, R1=R1)),
    "linear plain": dict(w=packing.pack_linear(rn(C, C) / 36), kw=dict()),
    "conv3x3 V": dict(w=packing.pack_conv3x3(rn(C, C, 3, 3) / 100), kw=dict(taps=9, mode=1, conv=(5, 8, 5, 8, 1, 0), V=V, vmode=1, vdiv=F_ * S)),
}
for name, c in cases.items():
    outs = []
    for tile in (1, 5, 6):
        o = torch.empty(M, C, dtype=torch.bfloat16, device=DEV)
        ops.gemm(A, c["w"], o, N=C, cin=C, bias=bias, tile=tile, **c["kw"])
        outs.append(o)
    torch.cuda.synchronize()
    d15 = (outs[0].float() - outs[1].float()).abs()
    print(f"{name:22s} tile1==tile5 {torch.equal(outs[0], outs[1])}  tile5==tile6 {torch.equal(outs[1], outs[2])}  "
          f"n_diff {(d15 > 0).sum().item()} max {d15.max().item():.3e}")
