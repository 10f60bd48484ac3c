#!/usr/bin/env python
"""Static check for the VMEM store-data hazard LLVM does not cover on gfx950 (csrc/gemm_pp_kernel.h,
store_data_hazard_guard): a buffer_store_dwordx3/x4 whose soffset is an SGPR, followed within two wait states by a VALU
write of one of its data VGPRs.  LLVM's GCNHazardRecognizer exempts the SGPR-soffset form; the hardware does not (seen as
zero / raw-fp32 dwords in stored tiles).  usage: python tools/hazard_scan.py file.s ...   (hipcc -S --cuda-device-only)"""
import re
import sys


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(text, wait_states=2):
    """[(kernel, store line, clobbering line)] for every unguarded store in the ISA text."""
    lines = [ln.strip() for ln in text.split("\n")]
    kern, hits = "?", []
    for i, ln in enumerate(lines):
        if re.match(r"^_Z\S+:", ln):
            kern = ln.split(":")[0]
        m = re.match(r"buffer_store_dwordx[34] (v\[\d+:\d+\]), (\S+), s\[\d+:\d+\], (\S+)", ln)
        if not m or not m.group(3).startswith("s"):
            continue
        data = _regs(m.group(1))
        k, j = 0, i + 1
        while k < wait_states and j < len(lines):
            t = lines[j]
            j += 1
            if not t or t[0] in ";.":
                continue
            if t.startswith("s_nop"):
                k += int(t.split()[1]) + 1
                continue
            k += 1
            if t.startswith("v_") and _regs(t.split()[1].rstrip(",")) & data:
                hits.append((kern, ln, t))
    return hits


def scan_mfma_to_lds_store(text, wait_states=19):
    """[(kernel, mfma line, store line, distance)]: a ds_write whose DATA registers were written by a v_mfma fewer than
    `wait_states` issue slots before it, in program order within a basic block.  Round 5 writes the epilogue staging stores
    as asm statements (csrc/gemm_pp_kernel.h stg_write16): LLVM's hazard recognizer inserts the wait states of "XDL write
    VGPR -> LDS / VMEM read of it" (11 for the 8-pass bf16 MFMAs, 19 for 16 passes) for instructions it knows, not inside
    asm statements -- so the distance is checked here.  Every other instruction counts as one slot (s_nop n as n + 1)."""
    lines = [ln.strip() for ln in text.split("\n")]
    kern, hits = "?", []
    for i, ln in enumerate(lines):
        if re.match(r"^_Z\S+:", ln):
            kern = ln.split(":")[0]
        m = re.match(r"ds_write_b(?:32|64|96|128) v\d+, (v\[\d+:\d+\]|v\d+)", ln)
        if not m:
            continue
        data = _regs(m.group(1))
        k, j = 0, i - 1
        while k < wait_states and j >= 0:
            t = lines[j]
            j -= 1
            if not t or t[0] == ";":
                continue
            if t[0] == "." or re.match(r"^_Z\S+:", t):      # block / kernel boundary: predecessors unknown, stop
                break
            if t.startswith("s_nop"):
                k += int(t.split()[1]) + 1
                continue
            if t.startswith("v_mfma") and _regs(t.split()[1].rstrip(",")) & data:
                hits.append((kern, t, ln, k))
                break
            k += 1
    return hits


if __name__ == "__main__":
    total = 0
    for f in sys.argv[1:]:
        text = open(f).read()
        hits = scan(text)
        near = scan_mfma_to_lds_store(text)
        for kern, mf, st, dist in near:
            print(f"{f}: {kern[:80]}: {mf[:60]}  ->  {st}  ({dist} slots)")
        total += len(hits) + len(near)
        for kern, st, cl in hits:
            print(f"{f}: {kern[:80]}: {st}  ->  {cl}")
        print(f"{f}: {len(hits)} unguarded stores")
    sys.exit(1 if total else 0)
