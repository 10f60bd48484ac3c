#!/usr/bin/env python
"""Producer-side GroupNorm statistics, isolated: a res block's conv (3x3 + temb row vector; 3x3 + residual; temporal + temb)
with and without gn_partials, and the GroupNorm that follows through both paths, at the benchmark's level shapes
(50 frame-images).  Tensors rotate through 3 copies so that nothing is served from the Infinity Cache by repetition."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops, packing  # noqa: E402

DEV = torch.device("cuda", 0)
EL = torch.bfloat16


def timed(fn, n=12, warm=3):
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
    for (H, W, C) in ((72, 128, 320), (36, 64, 640), (18, 32, 1280)):
        S, M = H * W, n * H * W
        xs = [torch.randn(M, C, device=DEV).to(EL) for _ in range(3)]
        outs = [torch.empty(M, C, dtype=EL, device=DEV) for _ in range(3)]
        ys = [torch.empty(M, C, dtype=EL, device=DEV) for _ in range(3)]
        R1 = torch.randn(M, C, device=DEV).to(EL)
        V = torch.randn(2, C, device=DEV)
        w3 = packing.pack_conv3x3(torch.randn(C, C, 3, 3, device=DEV) / (9 * C) ** 0.5)
        wt = packing.pack_conv_temporal(torch.randn(C, C, 3, 1, 1, device=DEV) / (3 * C) ** 0.5)
        b = torch.randn(C, device=DEV)
        gamma, beta = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
        cases = {
            "conv3x3+V": (w3, dict(N=C, cin=C, taps=9, mode=1, conv=(H, W, H, W, 1, 0), bias=b, V=V, vmode=1, vdiv=F * S), 1),
            "conv3x3+R1": (w3, dict(N=C, cin=C, taps=9, mode=1, conv=(H, W, H, W, 1, 0), bias=b, R1=R1), F),
            "temporal+V": (wt, dict(N=C, cin=C, taps=3, mode=2, temporal=(F, S), bias=b, V=V, vmode=1, vdiv=F * S), F),
        }
        for name, (w, kw, ips) in cases.items():
            part = torch.empty(ops.groupnorm_fused_scratch_floats(n, S, ips), dtype=torch.float32, device=DEV)
            p2 = torch.empty(ops.groupnorm_scratch_floats(n, S, C, ips), dtype=torch.float32, device=DEV)
            assert ops.gemm_gn_partials_serves(xs[0], w, outs[0], **kw)
            t_plain = timed(lambda i: ops.gemm(xs[i % 3], w, outs[i % 3], **kw))
            t_fused = timed(lambda i: ops.gemm(xs[i % 3], w, outs[i % 3], gn_partials=part, **kw))
            g_two = timed(lambda i: ops.groupnorm(outs[i % 3], None, n, S, C, ips, gamma, beta, 1e-6, True, ys[i % 3], p2))
            g_one = timed(lambda i: ops.groupnorm_from_partials(outs[i % 3], n, S, C, ips, gamma, beta, 1e-6, True, ys[i % 3], part))
            print(f"{H}x{W} C={C} {name:12s} gemm {t_plain:7.1f} -> {t_fused:7.1f} us ({t_fused - t_plain:+6.1f})   "
                  f"groupnorm {g_two:6.1f} -> {g_one:6.1f} us ({g_one - g_two:+6.1f})   net {t_fused - t_plain + g_one - g_two:+6.1f} us")


if __name__ == "__main__":
    main()
