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
    stamps = torch.zeros(256 * 8 * 4, dtype=torch.int64, device=DEV)
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
    s = stamps.view(-1, 8, 4).double()
    s = s[s[:, 0, 0] > 0]
    nchunk = 80 * 8
    print(f"{s.shape[0]} workgroups; cycles per chunk (8 tiles x 80 chunks per workgroup):  GEMM1 | vmcnt wait | barrier | DMA issue | (sum)")
    for w in range(8):
        v = [s[:, w, i].mean().item() / nchunk for i in range(4)]
        print(f"  wave {w}: " + " ".join(f"{x_:7.0f}" for x_ in v) + f"   {sum(v):7.0f}")


if __name__ == "__main__":
    main()
