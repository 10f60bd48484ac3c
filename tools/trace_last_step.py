"""Per-kernel time of the LAST step of a rocprofv3 --kernel-trace CSV whose program marks its steps with torch.cuda._sleep
(tools/train_bench.py --mark-steps 1): everything between the last mark and the end of the trace, grouped by kernel name.
usage: python tools/trace_last_step.py p_kernel_trace.csv [out.csv]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "spin_kernel" in r["Kernel_Name"] or "sleep" in r["Kernel_Name"].lower()]
if not marks:
    raise SystemExit("no step marks in the trace")
step = rows[marks[-1] + 1:]
agg = collections.OrderedDict()
for r in step:
    n = r["Kernel_Name"]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(n, [0, 0])
    a[0] += d
    a[1] += 1
tot = sum(a[0] for a in agg.values())
span = int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])
out = sorted(agg.items(), key=lambda kv: -kv[1][0])
w = csv.writer(open(sys.argv[2], "w")) if len(sys.argv) > 2 else None
if w:
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
print(f"last step: {len(step)} kernels, kernel time {tot / 1e6:.2f} ms, span {span / 1e6:.2f} ms")
for n, (d, c) in out:
    if w:
        w.writerow([n, c, d, d // c, round(100.0 * d / tot, 3)])
for n, (d, c) in out[:70]:
    short = re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", n)[:110]
    print(f"{d / 1e6:8.3f} ms {c:6d} x {d / c / 1e3:9.1f} us  {short}")
