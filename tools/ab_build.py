#!/usr/bin/env python
"""Developer aid: build variant copies of libctrlv_hip.so with extra -D flags (A/B in ONE gpurun session, same device).
usage: python tools/ab_build.py NAME -DFOO=1 ...   -> ctrlv_amd/lib/ab/libctrlv_NAME.so ; select with CTRLV_HIP_LIB=<path>"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(ROOT, "ctrlv_amd", "lib", "ab")      # in-tree: travels to the GPU box (git-ignored *.so / *.o)
os.makedirs(out_dir, exist_ok=True)
procs, objs = [], []
for s in g.HIP_SOURCES:
    o = os.path.join(out_dir, f"{s}.{name}.o")
    objs.append(o)
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *g.EXTRA_FLAGS.get(s, []),
                                   *flags, "-c",
                                   os.path.join(g.CSRC, s), "-o", o]))
assert all(p.wait() == 0 for p in procs)
lib = os.path.join(out_dir, f"libctrlv_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
