#!/usr/bin/env python
"""Diagnostic: do an MFMA-bound chain and an HBM-bound chain overlap when they run on two streams?  One "chain" = the
kernels of an L0 res block at 25 frames (conv3x3 GEMM -> GroupNorm -> conv3x3 GEMM -> GroupNorm -> linear GEMMs ->
LayerNorm), repeated.  Variants: both chains on one stream; one chain per stream; (CTRLV_PP_MAX_WG caps the persistent
GEMM grids, set it from the outside)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops  # noqa: E402

DEV = "cuda:0"
F = int(os.environ.get("FRAMES", "25"))
S, C = 9216, 320
M = F * S


def make_chain(seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, device=DEV)
    c = dict(x=r(M, C).bfloat16(), y=torch.empty(M, C, device=DEV, dtype=torch.bfloat16),
             z=torch.empty(M, C, device=DEV, dtype=torch.bfloat16),
             u=torch.empty(M, 4 * C, device=DEV, dtype=torch.bfloat16),
             w3=(r(C, 9 * C) / (9 * C) ** 0.5).bfloat16(), w1=(r(C, C) / C ** 0.5).bfloat16(),
             wg=(r(8 * C, C) / C ** 0.5).bfloat16(), wo=(r(C, 4 * C) / (4 * C) ** 0.5).bfloat16(),
             b=r(C), bg=r(8 * C), gamma=torch.ones(C, device=DEV), beta=torch.zeros(C, device=DEV))
    c["part"] = torch.empty(ops.groupnorm_scratch_floats(F, S, C, 1), dtype=torch.float32, device=DEV)
    return c


def run_chain(c, reps):
    geo = (72, 128, 72, 128, 1, 0)
    for _ in range(reps):
        ops.groupnorm(c["x"], None, F, S, C, 1, c["gamma"], c["beta"], 1e-5, True, c["y"], c["part"])
        ops.gemm(c["y"], c["w3"], c["z"], N=C, cin=C, taps=9, mode=1, conv=geo, bias=c["b"])
        ops.groupnorm(c["z"], None, F, S, C, 1, c["gamma"], c["beta"], 1e-5, True, c["y"], c["part"])
        ops.gemm(c["y"], c["w3"], c["z"], N=C, cin=C, taps=9, mode=1, conv=geo, bias=c["b"], R1=c["x"])
        ops.layernorm(c["z"], c["gamma"], c["beta"], 1e-5, c["y"])
        ops.gemm(c["y"], c["w1"], c["z"], N=C, cin=C, bias=c["b"], R1=c["x"])
        ops.layernorm(c["z"], c["gamma"], c["beta"], 1e-5, c["y"])
        ops.gemm(c["y"], c["wg"], c["u"], N=8 * C, cin=C, bias=c["bg"], geglu=1)
        ops.gemm(c["u"], c["wo"], c["y"], N=C, cin=4 * C, bias=c["b"], R1=c["z"])


def main():
    reps = 6
    a, b = make_chain(1), make_chain(2)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    run_chain(a, 1); run_chain(b, 1)
    torch.cuda.synchronize()

    def graph_of(fn):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g

    def one_stream():
        run_chain(a, reps); run_chain(b, reps)

    def two_streams():
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        run_chain(a, reps)
        with torch.cuda.stream(s2):
            run_chain(b, reps)
        cur.wait_stream(s2)

    def two_streams_offset():                 # chain b starts half a block late
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        run_chain(a, reps)
        with torch.cuda.stream(s2):
            ops.groupnorm(b["x"], None, F, S, C, 1, b["gamma"], b["beta"], 1e-5, True, b["y"], b["part"])
            ops.gemm(b["y"], b["w3"], b["z"], N=C, cin=C, taps=9, mode=1, conv=(72, 128, 72, 128, 1, 0), bias=b["b"])
            run_chain(b, reps)
        cur.wait_stream(s2)

    for name, fn in (("one stream", one_stream), ("two streams", two_streams), ("two streams, offset", two_streams_offset)):
        g = graph_of(fn)
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{name:20s} {min(ts):8.2f} ms   (cap {os.environ.get('CTRLV_PP_MAX_WG', '-')}, {F} frames per chain)")


if __name__ == "__main__":
    main()
