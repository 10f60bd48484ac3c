#!/usr/bin/env python
"""Diagnostic: builds a STAMPED copy of the ping-pong GEMM (-DCTRLV_PP_STAMP, s_memtime around every phase) into a
scratch library and prints where a persistent workgroup's cycles go, per layer shape.  Never part of the product."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "ctrlv_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "libctrlv_stamp.so")


def build():
    import __graft_entry__ as ge
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objs, procs = [], []
    for s in ge.HIP_SOURCES:
        if not (s.startswith("gemm") or s == "abi.hip"):
            continue
        o = os.path.join(ROOT, "gpurun_out", s + ".stamp.o")
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                       "-DCTRLV_PP_STAMP=" + os.environ.get("STAMP_MODE", "1"), "-c", os.path.join(CSRC, s), "-o", o]))
        objs.append(o)
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)


def main():
    dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    force_tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    build()
    from ctrlv_amd import _lib
    lib = ctypes.CDLL(OUT)
    lib.ctrlv_gemm.restype = ctypes.c_int
    lib.ctrlv_gemm.argtypes = [ctypes.POINTER(_lib.GemmDesc), ctypes.c_void_p]
    dev = "cuda:0"
    torch.zeros(1, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    N0, N1 = 50 * 9216, 50 * 2304
    shapes = [("L0 qkv 320->960", N0, 960, 320, 1, 0, None, 6, 0),
              ("L0 proj 320->320 +R1", N0, 320, 320, 1, 0, None, 6, 1),
              ("L0 ff.out 1280->320 +R1", N0, 320, 1280, 1, 0, None, 6, 1),
              ("L0 geglu 320->2560", N0, 2560, 320, 1, 0, None, 5, 0),
              ("L0 conv3x3 320->320", N0, 320, 320, 9, 1, (72, 128, 72, 128, 1, 0), 6, 0),
              ("L1 conv3x3 1920->640", N1, 640, 1920, 9, 1, (36, 64, 36, 64, 1, 0), 6, 0),
              ("L1 geglu 640->5120", N1, 5120, 640, 1, 0, None, 5, 0)]
    names9 = ["total", "1st half (reads+mfma)", "vmcnt wait", "lgkm+barrier", "2nd half (reads+mfma)", "lgkm wait", "-",
              "epilogue"]
    names = ["total", "L:ds_read issue", "L:dma issue", "L:vmcnt+lgkm wait", "L:barrier", "C:mfma+dma", "C:barrier",
             "epilogue"]
    if os.environ.get("STAMP_MODE") == "2":     # by position in the tile (sums over the tile's half-steps of that class)
        names = ["total", "L phase, half-step 0", "L phase, half-step 1", "L phase, half-step 2", "L phase, half-steps >= 3",
                 "C phase, half-step 0", "C phase, half-steps 1-2", "C phase, half-steps >= 3"]
    for name, M, N, K, taps, mode, geo, tile, r1 in shapes:
        A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, taps * K, generator=g, device=dev) / (taps * K) ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, generator=g, device=dev)
        geglu = 1 if "geglu" in name else 0
        out = torch.empty(M, N // 2 if geglu else N, dtype=torch.bfloat16, device=dev)
        R1 = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16) if r1 else None
        stamps = torch.zeros(256 * 16 * 10, dtype=torch.int64, device=dev)
        d = _lib.GemmDesc()
        d.A, d.W, d.out, d.bias = A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr()
        d.R1 = R1.data_ptr() if r1 else None
        d.V = stamps.data_ptr()           # stamp buffer: V is ignored by the epilogue when vmode == 0
        d.M, d.N, d.Cin, d.taps, d.lda, d.mode = M, N, K, taps, K, mode
        if geo:
            d.H, d.Wd, d.Ho, d.Wo, d.stride, d.up = geo
        d.ldo, d.n_store, d.ldr1, d.ldr2 = out.shape[1], out.shape[1], N, N
        d.s_acc, d.s1, d.s2 = 1.0, 1.0, 0.0
        if force_tile == 9 and mode != 0:
            continue
        d.geglu, d.tile = geglu, force_tile or tile
        nm = names9 if force_tile == 9 else names
        d.out_f32 = dbg
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(2):
            rc = lib.ctrlv_gemm(ctypes.byref(d), st)
            assert rc == 0, rc
        torch.cuda.synchronize()
        s = stamps.view(-1, 10).double()
        if os.environ.get("STAMP_BY_WAVE"):       # rows are (workgroup, wave): the same table per wave id
            sw = s.view(-1, 8, 10)
            sw = sw[sw[:, 0, 8] > 0]
            print(f"{name}: per wave id, cycles per half-step: " + " | ".join(nm[1:8]))
            for w in range(8):
                hs = sw[:, w, 8].mean().item()
                print(f"   wave {w}: " + " ".join(f"{sw[:, w, i].mean().item() / hs:7.0f}" for i in range(1, 8))
                      + f"   total/tile {sw[:, w, 0].mean().item() / sw[:, w, 9].mean().item():8.0f}")
        s = s[s[:, 8] > 0]
        tot = s[:, 0].mean().item()
        print(f"{name:28s} half-steps/wg {int(s[0, 8]):5d} tiles/wg {int(s[0, 9]):3d}  cycles/wave {tot:10.0f}")
        for i in range(1, 8):
            v = s[:, i].mean().item()
            print(f"      {nm[i]:26s} {v:10.0f}  {100 * v / tot:5.1f}%   per half-step {v / s[0, 8].item():7.0f}   per tile {v / s[0, 9].item():8.0f}")


if __name__ == "__main__":
    main()
