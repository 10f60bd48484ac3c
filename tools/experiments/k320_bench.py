#!/usr/bin/env python
"""Times the Cin = 320 Linear layers of the 72 x 128 level on csrc/gemm_k320.hip (tile 14) against the ping-pong tile (6 / 5),
buffer sets in rotation (developer tool).  usage: python tools/k320_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g, device=DEV)  # noqa: E731
M, C = 50 * 9216, 320
for name, N, res in (("q|k|v 320->960", 960, False), ("out 320->320", 320, False), ("out 320->320 + R1", 320, True)):
    W = (r(N, C) / C ** 0.5).bfloat16()
    b = r(N)
    sets = [dict(x=r(M, C).bfloat16(), out=torch.empty(M, N, dtype=torch.bfloat16, device=DEV),
                 r1=r(M, N).bfloat16() if res else None) for _ in range(3)]
    for rnd in range(2):
        for tile in (14, 6):
            def run(s_):
                kw = dict(N=N, cin=C, bias=b, tile=tile)
                if res:
                    kw["R1"] = s_["r1"]
                ops.gemm(s_["x"], W, s_["out"], **kw)
            for s_ in sets:
                run(s_)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                for s_ in sets:
                    run(s_)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 12
            gb = (M * C * 2 + M * N * 2 * (2 if res else 1)) / 1e9
            print(f"{name:20s} tile {tile:2d}: {ms * 1e3:7.1f} us  {2.0 * M * N * C / ms / 1e9:6.0f} TFLOP/s  {gb / ms * 1e3:6.0f} GB/s")
