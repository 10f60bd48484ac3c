"""Weight-gradient launches of the cfg5 training step (B = 1, 25 frames, 72 x 128 latent), one by one: time per launch and
TFLOP/s of ctrlv_gemm_wgrad (main kernel + ordered slab sum).  CTRLV_WGRAD_PP=0 runs every layer on the register-staged
kernel of backward.hip (A/B of csrc/wgrad_pp.hip).   python tools/wgrad_bench.py [--iters 10] [--dtype bf16|fp16]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def shapes(F=25, h=72, w=128):
    out = []
    for lvl, C in enumerate((320, 640, 1280, 1280)):
        H, W = h >> lvl, w >> lvl
        M = F * H * W
        cin0 = C if lvl in (0, 3) else C // 2
        out += [(f"L{lvl} conv3x3 {cin0}->{C}", M, C, cin0, 9, (H, W)), (f"L{lvl} conv3x3 {C}->{C}", M, C, C, 9, (H, W)),
                (f"L{lvl} temporal {C}->{C}", M, C, C, 3, (F, H * W))]
        if lvl < 3:
            out += [(f"L{lvl} linear {C}->{C}", M, C, C, 1, None), (f"L{lvl} GEGLU {C}->{8 * C}", M, 8 * C, C, 1, None),
                    (f"L{lvl} ff out {4 * C}->{C}", M, C, 4 * C, 1, None)]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--filter", default="", help="only layers whose name contains this")
    args = ap.parse_args()
    from ctrlv_amd import ops
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    dev = "cuda:0"
    rows, tot = [], 0.0
    for name, M, N, cin, taps, geo in shapes():
        if args.filter not in name:
            continue
        A = torch.randn(M, cin, device=dev).to(dt)
        dY = torch.randn(M, N, device=dev).to(dt)
        dW = torch.zeros(N, taps * cin, dtype=torch.float32, device=dev)
        db = torch.zeros(N, dtype=torch.float32, device=dev)
        kw = dict(N=N, cin=cin, taps=taps)
        if taps == 9:
            kw.update(mode=1, conv=(geo[0], geo[1], geo[0], geo[1], 1, 0))
        elif taps == 3:
            kw.update(mode=2, temporal=geo)
        for _ in range(2):
            ops.gemm_wgrad(A, dY, dW, dbias=db, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            ops.gemm_wgrad(A, dY, dW, dbias=db, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        tf = 2.0 * M * N * taps * cin / ms / 1e9
        rows.append(dict(layer=name, M=M, N=N, K=taps * cin, ms=round(ms, 4), tflops=round(tf, 1)))
        tot += ms
        print(f"{name:28s} M={M:7d} N={N:6d} K={taps * cin:6d}  {ms * 1e3:9.1f} us  {tf:7.1f} TFLOP/s", flush=True)
        del A, dY, dW
    print(json.dumps(dict(wgrad_pp=os.environ.get("CTRLV_WGRAD_PP", "1"), dtype=args.dtype, sum_ms=round(tot, 3), rows=rows)))


if __name__ == "__main__":
    main()
