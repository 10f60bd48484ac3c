#!/usr/bin/env python
"""Sweep the gather-GEMM tile configurations over the layer shapes of the cfg3 step (MI355X; developer tool).

Prints TFLOP/s per (shape, tile) measured with HIP events on random data; used to build the tile-selection table in
ctrlv_amd/csrc/gemm.hip (`pick_tile`).  usage: python tools/gemm_sweep.py [--reps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctrlv_amd import ops  # noqa: E402

DEV = "cuda:0"
# (name, M, N, K(cin), taps, mode, geometry, geglu)
PIX = int(os.environ.get("SWEEP_PIXELS", "9216"))          # latent pixels per frame-image at L0 (72x128; 40x64 = 2560)
N0, N1, N2, N3 = 50 * PIX, 50 * PIX // 4, 50 * PIX // 16, 50 * PIX // 64
_G = {9216: ((72, 128), (36, 64), (18, 32), (9, 16)), 2560: ((40, 64), (20, 32), (10, 16), (5, 8))}[PIX]


def _geo(level):
    h, w = _G[level]
    return (h, w, h, w, 1, 0)


SHAPES = [
    ("L0 ff.proj geglu 320->2560", N0, 2560, 320, 1, 0, None, 1),
    ("L0 ff.out 1280->320", N0, 320, 1280, 1, 0, None, 0),
    ("L0 qkv 320->960", N0, 960, 320, 1, 0, None, 0),
    ("L0 proj 320->320", N0, 320, 320, 1, 0, None, 0),
    ("L0 conv3x3 320->320", N0, 320, 320, 9, 1, _geo(0), 0),
    ("L0 conv3x3 960->320", N0, 320, 960, 9, 1, _geo(0), 0),
    ("L0 convT 320->320", N0, 320, 320, 3, 2, (25, PIX), 0),
    ("L1 ff.proj geglu 640->5120", N1, 5120, 640, 1, 0, None, 1),
    ("L1 ff.out 2560->640", N1, 640, 2560, 1, 0, None, 0),
    ("L1 qkv 640->1920", N1, 1920, 640, 1, 0, None, 0),
    ("L1 conv3x3 640->640", N1, 640, 640, 9, 1, _geo(1), 0),
    ("L1 conv3x3 1920->640", N1, 640, 1920, 9, 1, _geo(1), 0),
    ("L2 ff.proj geglu 1280->10240", N2, 10240, 1280, 1, 0, None, 1),
    ("L2 ff.out 5120->1280", N2, 1280, 5120, 1, 0, None, 0),
    ("L2 qkv 1280->3840", N2, 3840, 1280, 1, 0, None, 0),
    ("L2 conv3x3 1280->1280", N2, 1280, 1280, 9, 1, _geo(2), 0),
    ("L2 conv3x3 2560->1280", N2, 1280, 2560, 9, 1, _geo(2), 0),
    ("L3 conv3x3 1280->1280", N3, 1280, 1280, 9, 1, _geo(3), 0),
    ("L3 conv3x3 2560->1280", N3, 1280, 2560, 9, 1, _geo(3), 0),
    ("L3 ff.proj geglu 1280->10240", N3, 10240, 1280, 1, 0, None, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--tiles", type=str, default="5,6,7,8")
    ap.add_argument("--dbg", type=int, default=0, help="profiling aid bits: 2 skip stores, 4 skip DMA, 8 skip MFMA")
    ap.add_argument("--epi", type=int, default=0, help="epilogue operands: bit 0 row-vector table, bit 1 R1, bit 2 R2")
    ap.add_argument("--only", type=str, default="", help="substring filter on the shape name")
    args = ap.parse_args()
    tiles = [int(t) for t in args.tiles.split(",")]
    g = torch.Generator(device=DEV).manual_seed(0)
    print(f"{'shape':34s} " + " ".join(f"tile{t:>2d}" for t in tiles) + "   (TFLOP/s)")
    for name, M, N, K, taps, mode, geo, geglu in SHAPES:
        if args.only and args.only not in name:
            continue
        A = torch.randn(M, K, generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
        W = (torch.randn(N, taps * K, generator=g, device=DEV, dtype=torch.float32) / (taps * K) ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, generator=g, device=DEV, dtype=torch.float32)
        out = torch.empty(M, N // 2 if geglu else N, dtype=torch.bfloat16, device=DEV)
        kw = dict(N=N, cin=K, taps=taps, mode=mode, bias=bias, geglu=geglu, _dbg=args.dbg)
        if mode == 1:
            kw["conv"] = geo
        if mode == 2:
            kw["temporal"] = geo
        if args.epi and not geglu:
            if args.epi & 1:      # one row per image of 9216 / 2304 / ... pixels (the temb broadcast add)
                pix = M // 50
                kw.update(V=torch.randn(50, N, generator=g, device=DEV, dtype=torch.float32), vmode=1, vdiv=pix, vmod=50)
            if args.epi & 2:
                kw.update(R1=torch.randn(M, N, generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16), s1=1.0)
            if args.epi & 4:
                kw.update(R2=torch.randn(M, N, generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16), s2=0.5)
        res = []
        for t in tiles:
            if geglu and N % 32:
                res.append(float("nan")); continue
            try:
                ops.gemm(A, W, out, tile=t, **kw)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(args.reps):
                    ops.gemm(A, W, out, tile=t, **kw)
                e.record()
                torch.cuda.synchronize()
                ms = s.elapsed_time(e) / args.reps
                res.append(2.0 * M * N * taps * K / ms / 1e9)
            except Exception as ex:       # noqa: BLE001
                res.append(float("nan"))
                print("   ", name, "tile", t, "failed:", ex)
        print(f"{name:34s} " + " ".join(f"{r:6.0f}" for r in res))
        del A, W, out


if __name__ == "__main__":
    main()
