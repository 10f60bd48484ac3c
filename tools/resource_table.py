#!/usr/bin/env python
"""Per-kernel register / spill / scratch table of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage),
with the template arguments of gemm_pp_kernel<...> readable: python tools/resource_table.py gemm_pp_m1.hip [-DCTRLV_ELEM_F16=1]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    unit = sys.argv[1]
    defs = sys.argv[2:]
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    asm = f"/tmp/{unit}.{'_'.join(d.strip('-') for d in defs) or 'bf16'}.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", *g.EXTRA_FLAGS.get(unit, []), *defs,
           os.path.join(ROOT, "ctrlv_amd", "csrc", unit), "-o", asm]
    t = subprocess.run(cmd, capture_output=True, text=True).stderr
    names = re.findall(r"Function Name: (\S+)", t)
    vg = re.findall(r" VGPRs: (\d+)", t)
    ag = re.findall(r"AGPRs: (\d+)", t)
    sp = re.findall(r"VGPRs Spill: (\d+)", t)
    sc = re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", t)
    occ = re.findall(r"Occupancy \[waves/SIMD\]: (\d+)", t)
    for n, v, a, s, c, o in zip(names, vg, ag, sp, sc, occ):
        dn = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        m = re.search(r"(\w+)<(.*?)>\(", dn)
        print(f"{(m.group(1) + '<' + m.group(2) + '>') if m else dn[:70]:70s} VGPR {v:>3s} AGPR {a:>3s} spill {s:>3s} scratch {c:>4s} occ {o}")
    print("asm:", asm)


if __name__ == "__main__":
    main()
