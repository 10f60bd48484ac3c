#!/usr/bin/env python
"""Diagnostic: a STAMPED build of the fused feed-forward (-DCTRLV_FF_STAMP) -> cycles per phase and per wave group."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "libctrlv_ffstamp.so")


def build():
    import __graft_entry__ as ge
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objs, procs = [], []
    for s in ("ff_fused.hip", "abi.hip"):
        o = os.path.join(ROOT, "gpurun_out", s + ".ffstamp.o")
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                       "-DCTRLV_FF_STAMP", "-c", os.path.join(ge.CSRC, s), "-o", o]))
        objs.append(o)
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)


def main():
    build()
    from ctrlv_amd import _lib, ops, packing
    lib = ctypes.CDLL(OUT)
    lib.ctrlv_ff_fused_ln.restype = ctypes.c_int
    DEV = "cuda:0"
    g = torch.Generator(device=DEV).manual_seed(0)
    M, C, I = 50 * 9216, 320, 1280
    r = lambda *s: torch.randn(*s, generator=g, device=DEV)
    w1p, b1p = packing.pack_geglu(r(2 * I, C) / C ** 0.5, r(2 * I))
    w2p = packing.pack_linear(r(C, I) / I ** 0.5)
    w1f, w2f = ops.ff_fused_pack(w1p, b1p.float().contiguous(), w2p)
    x, r1 = r(M, C).bfloat16(), r(M, C).bfloat16()
    out = torch.empty(M, C, dtype=torch.bfloat16, device=DEV)
    b2 = r(C)
    stamps = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=DEV)   # [workgroup][wave][GEMM1, GEGLU, DMA, B, barrier, slot loop]
    d = _lib.GemmDesc()
    d.out, d.bias, d.R1 = out.data_ptr(), b2.data_ptr(), r1.data_ptr()
    d.M, d.N, d.Cin, d.taps, d.mode = M, 320, 1280, 1, 0
    d.ldo, d.n_store, d.ldr1 = 320, 320, 320
    d.s_acc, d.s1, d.s2 = 1.0, 1.0, 0.0
    vp = ctypes.c_void_p
    st = vp(torch.cuda.current_stream().cuda_stream)
    # (ln_gamma = null: no LayerNorm; the stamped build writes its sums to the ln_V pointer)
    rc = lib.ctrlv_ff_fused_ln(vp(x.data_ptr()), 320, None, None, ctypes.c_float(0), vp(stamps.data_ptr()), 1, 1, 320,
                               vp(w1f.data_ptr()), vp(w2f.data_ptr()), ctypes.byref(d), st)
    torch.cuda.synchronize()
    print("rc", rc)
    s = stamps.view(-1, 8, 8).double()
    s = s[s[:, 0, 0] > 0]
    nslot = 40 * 15                                         # A segments per wave: 40 per 128-row tile x 15 tiles per workgroup
    print(f"{s.shape[0]} workgroups; cycles per A segment / DMA issue / B segment / slot barriers, per wave and slot PAIR;")
    print("last column: the whole slot loop of a tile / 40 (what a slot pair takes, the closing slots included)")
    print("            GEMM1 | GEGLU+h | DMA issue |   B   | barriers | (sum) | slot loop / 40")
    for w in range(8):
        v = [s[:, w, i].mean().item() / nslot for i in range(8)]
        print(f"  wave {w}: " + " ".join(f"{x_:8.0f}" for x_ in v[:5]) + f"   {sum(v[:5]):7.0f}   {v[5]:7.0f}" +
              f"   per tile: prologue {v[6] * 40:7.0f}  slot loop {v[5] * 40:8.0f}  epilogue (stores drained) {v[7] * 40:7.0f}")
    ms = []
    for _ in range(3):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        lib.ctrlv_ff_fused_ln(vp(x.data_ptr()), 320, None, None, ctypes.c_float(0), vp(stamps.data_ptr()), 1, 1, 320,
                              vp(w1f.data_ptr()), vp(w2f.data_ptr()), ctypes.byref(d), st)
        t1.record(); torch.cuda.synchronize()
        ms.append(t0.elapsed_time(t1))
    print("stamped launch: " + " ".join(f"{m_ * 1e3:.0f}" for m_ in ms) + " us")


if __name__ == "__main__":
    main()
