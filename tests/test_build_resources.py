"""Register-budget regression guard for the ping-pong GEMM (CPU only: hipcc cross-compiles gfx950).

The 256x320 tile runs at 252-256 VGPRs; a careless edit pushes hundreds of registers to scratch and every layer shape
drops ~20x (seen during development).  This test compiles the three instantiation units with
-Rpass-analysis=kernel-resource-usage and bounds spills / scratch for every kernel."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ctrlv_amd", "csrc")


def test_pingpong_gemm_register_budget():
    procs = []
    with tempfile.TemporaryDirectory() as td:
        for unit in ("gemm_pp_m0.hip", "gemm_pp_m1.hip", "gemm_pp_m2.hip"):
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "--cuda-device-only",
                   "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, unit), "-o", os.path.join(td, unit + ".o")]
            procs.append((unit, subprocess.Popen(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)))
        n_kernels = 0
        for unit, p in procs:
            err = p.communicate()[1]
            assert p.returncode == 0, err[-2000:]
            vg = [int(x) for x in re.findall(r"VGPRs: (\d+)", err)]
            sp = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", err)]
            sc = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", err)]
            occ = [int(x) for x in re.findall(r"Occupancy \[waves/SIMD\]: (\d+)", err)]
            n_kernels += len(vg)
            assert vg and max(vg) <= 256, (unit, vg)
            assert all(o >= 2 for o in occ), (unit, occ)          # two waves per SIMD: the schedule depends on it
            assert max(sp) <= 40 and max(sc) <= 128, (unit, sp, sc)
        assert n_kernels == 24
