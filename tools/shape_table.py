#!/usr/bin/env python
"""Diagnostic: one eager denoising step of bench.py's workload with per-launch HIP events, GEMM launches grouped by
shape: where the GEMM families' time goes, and how far each shape is from ITS OWN bound (the larger of algorithmic FLOPs
at the MFMA peak and algorithmic bytes at the HBM peak)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    args = bench.parse()
    from ctrlv_amd import profiler
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    unet, ctrl = bench.build_models(dev, args.workload, args.frames, bench.torch_dtype(args.dtype))
    if args.trunk != "same":                       # (bench.main's rule: the split planes are fp16 elements)
        for m in (unet, ctrl):
            if m is not None:
                m.trunk_dtype = args.trunk
    args.hip_graph = False
    st = bench.make_stepper(unet, ctrl, dev, args, clip_index=0)
    st.use_hip_graph = False
    for i in range(2):
        bench.run_step(st, i)
    torch.cuda.synchronize()
    timer = profiler.PlanTimer(unet, ctrl)         # the C++ plan's own launches (ctrlv_plan_profile)
    with timer:
        bench.run_step(st, 2)
    agg = {}
    for fam, ms, fl, by, (M, N, K, flags) in timer.launches:
        if not fam.startswith("gemm"):
            if fam in ("groupnorm", "layernorm"):     # one line per shape: where the norm families' time goes
                d = agg.setdefault((fam, M, N, 0, flags & 1, 0, 0, 0), [0, 0.0, 0.0, 0.0])
                d[0] += 1; d[1] += ms; d[3] += by
            continue
        det = ("ff_fused" if flags & 0x100 else fam, M, N, K, flags & 1, (flags >> 1) & 3, (flags >> 3) & 3, 0)
        d = agg.setdefault(det, [0, 0.0, 0.0, 0.0])
        d[0] += 1; d[1] += ms; d[2] += fl; d[3] += by
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print(f"{'family':20s} {'M':>7s} {'N':>6s} {'K':>6s} gg R V act  calls     ms   TFLOP/s    GB/s   bound-ms  of-bound")
    lost = 0.0
    for (fam, M, N, K, gg, nr, vm, act), (n, ms, fl, by) in rows:
        bound = max(fl / 2.5e15, by / 8e12) * 1e3
        lost += ms - bound
        print(f"{fam:20s} {M:7d} {N:6d} {K:6d} {gg:2d} {nr:1d} {vm:1d} {act:3d} {n:6d} {ms:7.2f} {fl / ms / 1e9:8.0f} {by / ms / 1e6:8.0f} "
              f"{bound:9.2f} {bound / ms:8.2f}")
    print(f"total {tot:.1f} ms, {tot - lost:.1f} ms at the bounds")


if __name__ == "__main__":
    main()
