#!/usr/bin/env python
"""Diagnostic: idle time on the device during HIP-graph replay of the denoising step.
usage (GPU box): cd /tmp; rocprofv3 --kernel-trace --output-format csv -d <dir> -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
                 python tools/graph_gaps.py <dir>/p_kernel_trace.csv
Takes the kernels of one step (between two consecutive cfg_euler kernels; default: the last TIMED step of bench.py), merges their [start, end)
intervals and reports the busy union, the idle gaps and which kernels follow the large gaps."""
import csv
import sys
from collections import defaultdict


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "cfg_euler" in r[2]]
    if len(marks) < 2:
        print("no two cfg_euler kernels in the trace"); return
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2      # 1 = last step of the trace (bench.py: the eager instrumented one)
    a, b = marks[-back - 1], marks[-back]
    step = rows[a + 1:b + 1]
    t0, t1 = rows[a][1], rows[b][1]
    busy, gaps, cur_end, prev_name = 0, [], t0, rows[a][2]
    last_end_name = prev_name
    for s, e, n in step:
        if s > cur_end:
            gaps.append((s - cur_end, last_end_name, n))
            busy += 0
            cur_start = s
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
            last_end_name = n
    wall = t1 - t0
    print(f"step wall {wall / 1e6:.2f} ms, {len(step)} kernels, device busy (union) {busy / 1e6:.2f} ms, idle {(wall - busy) / 1e6:.2f} ms "
          f"in {len(gaps)} gaps; sum of kernel durations {sum(e - s for s, e, _ in step) / 1e6:.2f} ms")
    hist = defaultdict(lambda: [0, 0])
    for g, _, _ in gaps:
        k = 1 if g < 2000 else 2 if g < 5000 else 5 if g < 10000 else 10 if g < 20000 else 20
        hist[k][0] += 1; hist[k][1] += g
    for k in sorted(hist):
        print(f"   gaps {'<2' if k == 1 else '>=' + str(k)} us: {hist[k][0]:5d}, {hist[k][1] / 1e6:7.3f} ms")
    byk = defaultdict(lambda: [0, 0])
    for g, prev, nxt in gaps:
        key = nxt.split("(")[0][-60:]
        byk[key][0] += 1; byk[key][1] += g
    print("   idle before kernel (top 12):")
    for k, (n, t) in sorted(byk.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"      {t / 1e6:7.3f} ms {n:5d}  {k}")


if __name__ == "__main__":
    main(sys.argv[1])
