#!/usr/bin/env python
"""In-kernel clock of the MFMA kernels (MI355X; developer tool / round evidence).

MI355X_MICROARCH "DVFS give-back" item 6: the clock the chip holds INSIDE a kernel is d(s_memtime) / d(s_memrealtime) x
100 MHz, read in a diagnostic build after >= 2 s of back-to-back launches on random data.  This builds a variant library with
-DCTRLV_CLOCK_STAMP (csrc/common.h: thread 0 of every workgroup adds its two deltas to a buffer nothing else reads; the
product library contains no stamp), runs each probe shape for `--seconds` of warm launches, resets the buffer, runs `--reps`
more and prints clock and TFLOP/s: what "matrix-pipe-bound at the sustained clock"  is measured with.
usage: python tools/clock_probe.py [--seconds 2] [--reps 20] [--json out.json]"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build():
    lib = os.path.join(ROOT, "ctrlv_amd", "lib", "ab", "libctrlv_clock.so")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ab_build.py"), "clock", "-DCTRLV_CLOCK_STAMP=1"],
                          stdout=subprocess.DEVNULL)
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--json", default="")
    ap.add_argument("--no-build", action="store_true")
    args = ap.parse_args()
    lib_path = os.path.join(ROOT, "ctrlv_amd", "lib", "ab", "libctrlv_clock.so")
    if not args.no_build or not os.path.exists(lib_path):
        lib_path = build()
    os.environ["CTRLV_HIP_LIB"] = lib_path           # the bf16 library of this process is the stamped variant
    import torch
    from ctrlv_amd import _lib, ops
    lib = _lib.load()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)

    def reader(unit):
        fn = getattr(lib, "ctrlv_debug_clock_" + unit)
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
        return fn

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g, device=dev) * scale).to(torch.bfloat16)

    N1, N2 = 50 * 2304, 50 * 576
    probes = []

    def gemm_probe(name, unit, M, N, K, taps=1, mode=0, geo=None, geglu=0, tile=0, r1=False):
        A, W = rnd(M, K), rnd(N, taps * K, scale=(taps * K) ** -0.5)
        bias = torch.randn(N, generator=g, device=dev)
        out = torch.empty(M, N // 2 if geglu else N, dtype=torch.bfloat16, device=dev)
        kw = dict(N=N, cin=K, taps=taps, mode=mode, bias=bias, geglu=geglu, tile=tile)
        if mode == 1:
            kw["conv"] = geo
        if mode == 2:
            kw["temporal"] = geo
        if r1:
            kw["R1"] = rnd(M, N)
        probes.append((name, unit, lambda: ops.gemm(A, W, out, **kw), 2.0 * M * N * taps * K))

    gemm_probe("pp 256x320 conv3x3 1920->640 @36x64 (tile 6)", "pp_m1", N1, 640, 1920, 9, 1, (36, 64, 36, 64, 1, 0), tile=6)
    gemm_probe("pp 256x320 GEGLU 640->5120 (tile 6)", "pp_m0", N1, 5120, 640, geglu=1, tile=6)
    gemm_probe("pp 256x256 GEGLU 1280->10240 (tile 5)", "pp_m0", N2, 10240, 1280, geglu=1, tile=5)
    gemm_probe("w16 256x256 GEGLU 640->5120 (tile 12)", "w16", N1, 5120, 640, geglu=1, tile=12)
    gemm_probe("w16 256x256 GEGLU 1280->10240 (tile 12)", "w16", N2, 10240, 1280, geglu=1, tile=12)
    gemm_probe("w16 256x320 conv3x3 1920->640 @36x64 (tile 13)", "w16", N1, 640, 1920, 9, 1, (36, 64, 36, 64, 1, 0), tile=13)
    gemm_probe("pp 256x320 qkv 320->960 @72x128 (tile 6)", "pp_m0", 50 * 9216, 960, 320, tile=6)
    # fused C = 320 feed-forward
    from ctrlv_amd import packing
    M0 = 50 * 9216
    with packing.element_dtype(torch.bfloat16):
        W1p, b1p = packing.pack_geglu(torch.randn(2560, 320) / 320 ** 0.5, torch.randn(2560) * 0.1)
        W2p = packing.pack_linear(torch.randn(320, 1280) / 1280 ** 0.5)
    w1f, w2f = ops.ff_fused_pack(W1p.to(dev), b1p.to(dev), W2p.to(dev))
    xff, r1ff, off = rnd(M0, 320), rnd(M0, 320), torch.empty(M0, 320, dtype=torch.bfloat16, device=dev)
    b2 = torch.randn(320, generator=g, device=dev)
    probes.append(("ff_fused C=320 +R1 @72x128", "ff_fused", lambda: ops.ff_fused(xff, w1f, w2f, off, bias=b2, R1=r1ff),
                   2.0 * M0 * 320 * (2560 + 1280)))
    # fused temporal self-attention block at C = 320 (2 clips x 25 frames x 72 x 128)
    with packing.element_dtype(torch.bfloat16):
        wqkv_t = packing.pack_qkv(*(torch.randn(320, 320) / 320 ** 0.5 for _ in range(3))).to(dev)
        wo_t = packing.pack_linear(torch.randn(320, 320) / 320 ** 0.5).to(dev)
    wf_t = ops.temporal_fused_pack(wqkv_t, wo_t)
    Mt = 2 * 25 * 9216
    xt, rt, ot = rnd(Mt, 320), rnd(Mt, 320), torch.empty(Mt, 320, dtype=torch.bfloat16, device=dev)
    probes.append(("temporal_fused C=320 +R1 @72x128", "temporal_fused",
                   lambda: ops.temporal_fused(xt, wf_t, ot, 2, 25, 9216, bias=b2, R1=rt),
                   2.0 * Mt * 320 * 1280 + 4.0 * 2 * 9216 * 5 * 25 * 25 * 64))
    # weight gradients of the cfg5 training step (csrc/wgrad_pp.hip): L1 3x3 conv 640 -> 640, L0 GEGLU projection
    Mw = 25 * 36 * 64
    aw, yw = rnd(Mw, 640), rnd(Mw, 640)
    dww = torch.zeros(640, 9 * 640, dtype=torch.float32, device=dev)
    probes.append(("wgrad_pp conv3x3 640->640 @36x64 (25 frames)", "wgrad_pp",
                   lambda: ops.gemm_wgrad(aw, yw, dww, N=640, cin=640, taps=9, mode=1, conv=(36, 64, 36, 64, 1, 0)),
                   2.0 * Mw * 640 * 9 * 640))
    Mg = 25 * 9216
    ag, yg = rnd(Mg, 320), rnd(Mg, 2560)
    dwg = torch.zeros(2560, 320, dtype=torch.float32, device=dev)
    probes.append(("wgrad_pp GEGLU 320->2560 @72x128 (25 frames)", "wgrad_pp",
                   lambda: ops.gemm_wgrad(ag, yg, dwg, N=2560, cin=320), 2.0 * Mg * 2560 * 320))
    # spatial attention at S = 9216 (the 64-rows-per-wave kernel), pre-scaled q
    qkv, ao = rnd(10 * 9216, 960), torch.empty(10 * 9216, 320, dtype=torch.bfloat16, device=dev)
    probes.append(("attn_spatial64 S=9216, 10 images x 5 heads", "attention",
                   lambda: ops.attention_spatial(qkv, ao, 10, 9216, 320, prescaled=True), 4.0 * 10 * 5 * 9216.0 * 9216 * 64))
    rows = []
    for name, unit, fn, flops in probes:
        rd = reader(unit)
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < args.seconds:          # >= 2 s of back-to-back launches: the clock has settled
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 2)()
        rd(buf, 1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        rd(buf, 1)
        ms = s.elapsed_time(e) / args.reps
        ghz = buf[0] / max(buf[1], 1) * 0.1
        rows.append(dict(kernel=name, clock_ghz=round(ghz, 3), ms=round(ms, 4), tflops=round(flops / ms / 1e9, 1)))
        print(f"{name:52s} in-kernel clock {ghz:5.3f} GHz   {ms:8.3f} ms   {flops / ms / 1e9:7.0f} TFLOP/s", flush=True)
    if args.json:
        json.dump(dict(_what="in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, diagnostic build -DCTRLV_CLOCK_STAMP, "
                             f">= {args.seconds} s of back-to-back launches on random bf16 data before the measured {args.reps}",
                       _build_id=_lib.source_build_id(), probes=rows), open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
