#!/usr/bin/env python
"""Per-kernel register / spill / scratch / LDS report of the HIP units for both element types (CPU only: hipcc
cross-compiles gfx950).  usage: python tools/resource_report.py [unit.hip ...] [--f16] [-D...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def report(unit, defines):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage"] + g.EXTRA_FLAGS.get(unit, []) + defines + \
          [os.path.join(g.CSRC, unit), "-o", "/dev/null"]
    err = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True).stderr
    rows = []
    for blk in err.split("Function Name: ")[1:]:
        name = blk.split()[0]
        f = lambda pat: int(re.search(pat, blk).group(1))  # noqa: E731
        rows.append((name, f(r"VGPRs: (\d+)"), f(r"AGPRs: (\d+)"), f(r"VGPRs Spill: (\d+)"), f(r"ScratchSize \[bytes/lane\]: (\d+)"),
                     f(r"Occupancy \[waves/SIMD\]: (\d+)"), f(r"LDS Size \[bytes/block\]: (\d+)")))
    return rows


def main():
    args = sys.argv[1:]
    defines = [a for a in args if a.startswith("-D")]
    if "--f16" in args:
        defines.append("-DCTRLV_ELEM_F16=1")
    units = [a for a in args if a.endswith(".hip")] or [u for u in g.HIP_SOURCES if u != "abi.hip"]
    for u in units:
        for name, vg, ag, sp, sc, occ, lds in report(u, defines):
            short = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
            short = short.replace("(anonymous namespace)::", "").split("(")[0]
            print(f"{u:18s} {short:60s} vgpr {vg:3d} agpr {ag:3d} spill {sp:3d} scratch {sc:4d} occ {occ} lds {lds}")


if __name__ == "__main__":
    main()
