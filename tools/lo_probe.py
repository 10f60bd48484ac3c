#!/usr/bin/env python
"""What the split-plane operands of a trunk-writing GEMM cost, one by one (developer probe, fp16 library): the L0 320 -> 320
projection (M = 460 800) with {R1} alone, + R1_lo, + out_lo, + both; the lo planes are one e5m2 byte per element.
usage: python tools/lo_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops, packing  # noqa: E402

DEV, EL = "cuda:0", torch.float16
M, N, K = 50 * 9216, 320, 320
g = torch.Generator(device=DEV).manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g, device=DEV)      # noqa: E731
with packing.element_dtype(EL):
    W = packing.pack_linear(r(N, K).cpu() / K ** 0.5).to(DEV)
bias = r(N)
NSET = 3          # buffer sets in rotation: 295 MB tensors, nothing stays in the 256 MB Infinity Cache
sets = [dict(A=r(M, K).to(EL), R1=r(M, N).to(EL), R1_lo=torch.zeros(M, N, dtype=torch.uint8, device=DEV),
             out=torch.empty(M, N, dtype=EL, device=DEV), out_lo=torch.empty(M, N, dtype=torch.uint8, device=DEV)) for _ in range(NSET)]


def run(name, use_r1lo, use_outlo, n=12):
    def once(b):
        kw = dict(N=N, cin=K, bias=bias, R1=b["R1"])
        if use_r1lo:
            kw["R1_lo"] = b["R1_lo"]
        if use_outlo:
            kw["out_lo"] = b["out_lo"]
        ops.gemm(b["A"], W, b["out"], **kw)
    for b in sets:
        once(b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        once(sets[i % NSET])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    mb = M * (K * 2 + N * 2 + N * 2 + (N if use_r1lo else 0) + (N if use_outlo else 0)) / 1e6
    print(f"{name:28s} {us:8.1f} us   {mb:7.0f} MB   {mb / us:5.2f} TB/s", flush=True)


for rep in range(2):
    run("{R1}", False, False)
    run("{R1} + R1_lo", True, False)
    run("{R1} + out_lo", False, True)
    run("{R1} + R1_lo + out_lo", True, True)
