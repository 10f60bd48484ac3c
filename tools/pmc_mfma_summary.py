#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc pass of SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA / SQ_BUSY_CU_CYCLES / GRBM_GUI_ACTIVE over one
eager bench step into per-kernel-family MATRIX-PIPE utilisation (north_star: "MFMA utilisation on attention against CDNA4
peak").

usage: python tools/pmc_mfma_summary.py <counter_collection.csv> <out.json> [clock_probe.json]

Units (MI355X_MICROARCH.md, cycle constants): SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over every SIMD of
the chip (32 per v_mfma_f32_32x32x16, 16 per v_mfma_f32_16x16x32 -- `cycles_per_mfma` in the output is the check);
GRBM_GUI_ACTIVE is summed over the 8 XCDs, so a dispatch ran GRBM_GUI_ACTIVE / 8 shader cycles.
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs)      (share of the pipe's cycles in use)
  clock_ghz = GRBM_GUI_ACTIVE / 8 / dispatch duration (reads high on dispatches under ~0.3 ms; the in-kernel clock of
              tools/clock_probe.py is the reference, merged in when its JSON is given)
Profiled passes run at a lower clock than unprofiled ones (the guide: 1.89-1.95 vs 2.02 GHz): ratios, not wall times.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

N_SIMD = 256 * 4


def family(name):
    m = re.search(r"gemm_pp_kernel<\s*\d+,\s*\d+,\s*\d+,\s*(\d)", name)
    if m:
        return {"0": "gemm_linear.pingpong", "1": "gemm_conv3x3.pingpong", "2": "gemm_conv_temporal.pingpong"}[m.group(1)]
    m = re.search(r"gemm_w16_kernel<\s*\d+,\s*(\d)", name)
    if m:
        return {"0": "gemm_linear.w16", "1": "gemm_conv3x3.w16", "2": "gemm_conv_temporal.w16"}[m.group(1)]
    if "ff_fused_kernel" in name:
        return "gemm_linear.ff_fused"
    if "temporal_fused_kernel" in name:
        return "gemm_temporal_block.fused"
    if "attn_spatial64" in name:
        return "attention_spatial.rows64"
    if "attn_spatial" in name:
        return "attention_spatial.rows32"
    if "attn_temporal" in name:
        return "attention_temporal"
    if "gemm_kernel" in name:
        return "gemm.two_stage"
    return None


def main():
    path, out = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    dur = defaultdict(dict)
    with open(path) as f:
        for r in csv.DictReader(f):
            fam = family(r["Kernel_Name"])
            if fam is None:
                continue
            acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
            did = r.get("Dispatch_Id") or r.get("Correlation_Id")
            disp[fam].add(did)
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur[fam][did] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from ctrlv_amd import _lib
    res = {"_build_id": _lib.source_build_id(),
           "_note": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); cycles_per_mfma = "
                    "SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA (32 for 32x32x16, 16 for 16x16x32: the unit check); clock_ghz = "
                    "GRBM_GUI_ACTIVE / 8 / dispatch time (profiled pass); one eager step of bench.py"}
    for fam in sorted(acc):
        a = acc[fam]
        gui = a.get("GRBM_GUI_ACTIVE", 0.0)
        busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        insts = a.get("SQ_INSTS_MFMA", 0.0)
        ns = sum(dur[fam].values())
        e = {"launches": len(disp[fam]), "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_INSTS_MFMA": insts,
             "SQ_BUSY_CU_CYCLES": a.get("SQ_BUSY_CU_CYCLES", 0.0), "GRBM_GUI_ACTIVE": gui}
        if gui > 0:
            e["mfma_busy"] = round(busy / (gui / 8.0 * N_SIMD), 4)
        if insts > 0:
            e["cycles_per_mfma"] = round(busy / insts, 2)
        if ns > 0 and gui > 0:
            e["ms"] = round(ns * 1e-6, 3)
            e["clock_ghz"] = round(gui / 8.0 / ns, 3)
        res[fam] = e
    if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
        res["in_kernel_clock"] = json.load(open(sys.argv[3]))
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        if not k.startswith("_") and k != "in_kernel_clock":
            print(f"{k:30s} launches {v['launches']:4d}  mfma_busy {v.get('mfma_busy', float('nan')):6.3f}  cycles/mfma "
                  f"{v.get('cycles_per_mfma', float('nan')):6.2f}  clock {v.get('clock_ghz', float('nan')):5.2f} GHz  {v.get('ms', 0):8.2f} ms")


if __name__ == "__main__":
    main()
