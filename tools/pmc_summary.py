#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs) into per-kernel HBM traffic.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [steps]
(steps = denoising steps the profiled command ran: per-step launch counts of each kernel set, so that bench.py can report
 the number of launches its `traffic` figure covers)

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports
exactly 1/2 of the bytes of a wide coalesced streaming read (128-B requests tallied at 64 B) -> doubled here;
WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Other access widths are uncalibrated (flagged in the output).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def family(name):
    m = re.search(r"gemm_pp_kernel<\s*\d+,\s*\d+,\s*\d+,\s*(\d)", name)
    if m:                      # 4th template argument = gather mode
        return {"0": "gemm_linear", "1": "gemm_conv3x3", "2": "gemm_conv_temporal"}[m.group(1)]
    if "ff_fused_kernel" in name:   # the fused C = 320 feed-forward: a member of the plain-GEMM family
        return "gemm_linear"
    if "ln_rows_kernel" in name:    # the rows-per-wave LayerNorm: same family as ln_kernel
        return "ln_kernel"
    if "attn_spatial" in name:      # 32- and 64-rows-per-wave variants
        return "attn_spatial_kernel"
    for k in ("gemm_kernel", "temporal_fused_kernel", "attn_temporal_kernel", "gn_stats_kernel", "gn_finalize_kernel", "gn_apply_kernel",
              "ln_kernel", "axpby_kernel", "im2col3x3_kernel", "cfg_euler_kernel"):
        if k in name:
            return k
    return None


def load(path, counter):
    per = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            fam = family(r["Kernel_Name"])
            if fam is None:
                continue
            per[fam][0] += 1
            per[fam][1] += float(r["Counter_Value"])
    return per


def main():
    fetch, write, out = sys.argv[1:4]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    fe, wr = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
    from ctrlv_amd import _lib
    res = {"_build_id": _lib.source_build_id(),     # bench.py reports `traffic` only for the library this came from
           "_note": "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024 / launches; FETCH doubled per the gfx950 "
                    "calibration for wide coalesced reads; includes Infinity-Cache hits (fabric-side counters)"}
    for fam in sorted(set(fe) | set(wr)):     # noqa: E501
        nf, vf = fe.get(fam, [0, 0.0])
        nw, vw = wr.get(fam, [0, 0.0])
        res[fam] = {
            "launches_fetch_pass": nf, "launches_write_pass": nw,
            "fetch_bytes_per_launch": 2.0 * vf * 1024 / max(nf, 1),
            "write_bytes_per_launch": vw * 1024 / max(nw, 1),
        }
        res[fam]["traffic_bytes_per_launch"] = res[fam]["fetch_bytes_per_launch"] + res[fam]["write_bytes_per_launch"]
        if steps > 0 and nf % steps == 0:
            res[fam]["launches_per_step"] = nf // steps
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        if not k.startswith("_"):
            print(f"{k:24s} launches {v['launches_fetch_pass']:5d}  fetch {v['fetch_bytes_per_launch'] / 1e6:10.1f} MB  "
                  f"write {v['write_bytes_per_launch'] / 1e6:10.1f} MB per launch")


if __name__ == "__main__":
    main()
