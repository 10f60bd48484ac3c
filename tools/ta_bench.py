#!/usr/bin/env python
"""Times the fused temporal self-attention block at C = 320 (csrc/temporal_fused.hip) against the three launches it replaces,
at M = 2 x 25 x 9216 rows with buffer sets in rotation (developer tool).  --stamp: builds a -DCTRLV_TA_STAMP variant of the
kernel ON THE GPU BOX and prints the cycle sums of its phases (the product library contains no stamp)."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ctrlv_amd import _lib, ops, packing  # noqa: E402

DEV = "cuda:0"
B, F, S, C = 2, 25, int(os.environ.get("TA_S", 9216)), 320
M = B * F * S
g = torch.Generator(device=DEV).manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g, device=DEV)      # noqa: E731
wqkv = packing.pack_qkv(r(C, C) / C ** 0.5, r(C, C) / C ** 0.5, r(C, C) / C ** 0.5)
wop, bo = packing.pack_linear(r(C, C) / C ** 0.5), r(C)
vt = r(B, C)
wf = ops.temporal_fused_pack(wqkv, wop)
NSET = 4
sets = [dict(x=r(M, C).bfloat16(), r1=r(M, C).bfloat16(), qkv=torch.empty(M, 3 * C, dtype=torch.bfloat16, device=DEV),
             a=torch.empty(M, C, dtype=torch.bfloat16, device=DEV), out=torch.empty(M, C, dtype=torch.bfloat16, device=DEV))
        for _ in range(NSET)]
kw = dict(bias=bo, V=vt, vmode=1, vdiv=F * S)


def three(b):
    ops.gemm(b["x"], wqkv, b["qkv"], N=3 * C, cin=C)
    ops.attention_temporal(b["qkv"], b["a"], B, F, S, C)
    ops.gemm(b["a"], wop, b["out"], N=C, cin=C, R1=b["r1"], **kw)


def fused(b):
    ops.temporal_fused(b["x"], wf, b["out"], B, F, S, R1=b["r1"], **kw)


gam, bet = r(C), r(C)
for b_ in sets:
    b_["tt"] = torch.empty(M, C, dtype=torch.bfloat16, device=DEV)


def ln_then_fused(b):          # the block's LayerNorm as a launch of its own (x = r1: the block normalises its residual input)
    ops.layernorm(b["r1"], gam, bet, 1e-5, b["tt"])
    ops.temporal_fused(b["tt"], wf, b["out"], B, F, S, R1=b["r1"], **kw)


def fused_ln(b):               # ... inside the kernel
    ops.temporal_fused(b["r1"], wf, b["out"], B, F, S, R1=b["r1"], ln=(gam, bet, 1e-5), **kw)


def timeit(name, fn):
    for b in sets:
        fn(b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        for b in sets:
            fn(b)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / (3 * NSET)
    fl = 2.0 * M * C * 4 * C + 4.0 * B * S * 5 * F * F * 64
    print(f"{name:14s} {ms * 1e3:8.1f} us   {fl / ms / 1e9:6.0f} TFLOP/s", flush=True)


if "--stamp" not in sys.argv and "--variant" not in sys.argv:
    for name, fn in (("three launches", three), ("fused", fused), ("LN + fused", ln_then_fused), ("fused incl. LN", fused_ln)) * 2:
        timeit(name, fn)
else:
    import __graft_entry__ as ge
    out = os.path.join(ROOT, "gpurun_out", "libctrlv_tastamp.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objs, procs = [], []
    for src in ("temporal_fused.hip", "abi.hip"):
        o = os.path.join(ROOT, "gpurun_out", src + ".tastamp.o")
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
                                       *(["-DCTRLV_TA_STAMP"] if "--stamp" in sys.argv else []), *os.environ.get("TA_FLAGS", "").split(),
                                       "-c", os.path.join(ge.CSRC, src), "-o", o]))
        objs.append(o)
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    lib = ctypes.CDLL(out)
    lib.ctrlv_temporal_fused.restype = ctypes.c_int
    if "--variant" in sys.argv:          # an unstamped variant build (TA_FLAGS), timed like the product kernel
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i, b in enumerate(sets):
            b["desc"] = ops._temporal_fused_desc(b["x"], wf, b["out"], B, F, S, R1=b["r1"], **kw)

        def var(b):
            lib.ctrlv_temporal_fused(ctypes.byref(b["desc"]), st)
        timeit("variant " + os.environ.get("TA_FLAGS", ""), var)
        timeit("variant " + os.environ.get("TA_FLAGS", ""), var)
        sys.exit(0)
    stamps = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=DEV)
    lib.ctrlv_temporal_fused_set_stamp(ctypes.c_void_p(stamps.data_ptr()))
    b = sets[0]
    d = ops._temporal_fused_desc(b["x"], wf, b["out"], B, F, S, R1=b["r1"], **kw)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.ctrlv_temporal_fused.restype = ctypes.c_int
    for _ in range(2):
        rc = lib.ctrlv_temporal_fused(ctypes.byref(d), st)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(4):
        lib.ctrlv_temporal_fused(ctypes.byref(d), st)
    ev1.record(); torch.cuda.synchronize()
    print("rc", rc, f"stamped launch {ev0.elapsed_time(ev1) / 4 * 1e3:.0f} us  flags {os.environ.get('TA_FLAGS', '')}")
    s = stamps.view(-1, 8, 8).double()
    s = s[s[:, 0, 0] > 0]
    rounds = (B * S + 7) // 8 / s.shape[0]
    print(f"{s.shape[0]} workgroups, {rounds:.1f} pixel groups each; cycles per pixel group and wave:")
    print("          total | barrier | DMA issue | chains | vmcnt wait | softmax+PV+pack | out-proj loads+epilogue | x issue")
    for w in range(8):
        print(f"  wave {w}: " + " ".join(f"{s[:, w, i].mean().item() / rounds:9.0f}" for i in range(8)))
