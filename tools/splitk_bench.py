#!/usr/bin/env python
"""Split contraction (csrc/gemm.hip splitk_plan), isolated: the small-level convs of the benchmark's size (9 x 16) and of
the reference's default 320 x 512 size (10 x 16, 5 x 8) for the 50 frame-images of a CFG'd clip, split on / off."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops, packing  # noqa: E402

DEV = torch.device("cuda", 0)
EL = torch.bfloat16


def timed(fn, n=20, warm=5):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    F, n = 25, 50
    for (H, W, cin, cout) in ((9, 16, 1280, 1280), (9, 16, 2560, 1280), (10, 16, 1280, 1280), (5, 8, 1280, 1280), (5, 8, 2560, 1280)):
        S, M = H * W, n * H * W
        xs = [torch.randn(M, cin, device=DEV).to(EL) for _ in range(3)]
        outs = [torch.empty(M, cout, dtype=EL, device=DEV) for _ in range(3)]
        R1 = torch.randn(M, cout, device=DEV).to(EL)
        V = torch.randn(2, cout, device=DEV)
        w3 = packing.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=DEV) / (9 * cin) ** 0.5)
        wt = packing.pack_conv_temporal(torch.randn(cout, cin, 3, 1, 1, device=DEV) / (3 * cin) ** 0.5)
        b = torch.randn(cout, device=DEV)
        cases = {
            "conv3x3+R1": (w3, dict(N=cout, cin=cin, taps=9, mode=1, conv=(H, W, H, W, 1, 0), bias=b, R1=R1)),
            "temporal+V": (wt, dict(N=cout, cin=cin, taps=3, mode=2, temporal=(F, S), bias=b, V=V, vmode=1, vdiv=F * S)),
        }
        for name, (w, kw) in cases.items():
            sl = ops.gemm_splitk_slices(xs[0], w, outs[0], **kw)
            t0 = timed(lambda i: ops.gemm(xs[i % 3], w, outs[i % 3], splitk=False, **kw))
            t1 = timed(lambda i: ops.gemm(xs[i % 3], w, outs[i % 3], **kw))
            fl = 2.0 * M * cout * kw["taps"] * cin
            print(f"{H}x{W} {cin}->{cout} {name:11s} slices {sl:2d}: {t0:7.1f} us ({fl / t0 / 1e6:5.0f} TF/s) -> {t1:7.1f} us ({fl / t1 / 1e6:5.0f} TF/s)")


if __name__ == "__main__":
    main()
