#!/usr/bin/env python
"""Experiment: GEGLU projection -> FF-out as M-chunked pairs, so that the chunk of u (the 4C-wide GEGLU output, 590 MB at
L0 for 50 frame-images) written by the first GEMM is still in the 256 MB Infinity Cache when the second reads it.
Prints the pair time for 1 / 2 / 4 / 8 / 16 chunks at the L0 and L1 shapes of the cfg3 step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops  # noqa: E402

DEV = "cuda:0"


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    for name, M, C in (("L0", 50 * 9216, 320), ("L1", 50 * 2304, 640), ("L2", 50 * 576, 1280)):
        x = torch.randn(M, C, generator=g, device=DEV).to(torch.bfloat16)
        r1 = torch.randn(M, C, generator=g, device=DEV).to(torch.bfloat16)
        w1 = (torch.randn(8 * C, C, generator=g, device=DEV) / C ** 0.5).to(torch.bfloat16)
        b1 = torch.randn(8 * C, generator=g, device=DEV)
        w2 = (torch.randn(C, 4 * C, generator=g, device=DEV) / (4 * C) ** 0.5).to(torch.bfloat16)
        b2 = torch.randn(C, generator=g, device=DEV)
        u = torch.empty(M, 4 * C, dtype=torch.bfloat16, device=DEV)
        out = torch.empty(M, C, dtype=torch.bfloat16, device=DEV)
        ref = None
        for nch in (1, 2, 4, 8, 16):
            rows = (M // nch + 255) // 256 * 256

            def run():
                for m0 in range(0, M, rows):
                    m1 = min(M, m0 + rows)
                    ops.gemm(x[m0:m1], w1, u[m0:m1], N=8 * C, cin=C, bias=b1, geglu=1)
                    ops.gemm(u[m0:m1], w2, out[m0:m1], N=C, cin=4 * C, bias=b2, R1=r1[m0:m1])
            run()
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            same = torch.equal(ref, out)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                run()
            e.record()
            torch.cuda.synchronize()
            print(f"{name} C={C:5d}  chunks {nch:2d}  rows/chunk {rows:7d}  pair {s.elapsed_time(e) / 5 * 1e3:8.1f} us  identical {same}")


if __name__ == "__main__":
    main()
