#!/usr/bin/env python
"""Per-shape delta of two tools/shape_table.py outputs (e.g. --dtype fp16 against --dtype fp16 --trunk fp16x2, same box):
usage: python tools/shape_delta.py A.txt B.txt [title] > profiles/rNN_split_vs_fp16_shapes.txt"""
import collections
import sys


def load(p):
    rows = collections.OrderedDict()
    for ln in open(p):
        f = ln.split()
        if len(f) < 13 or not f[1].isdigit():
            continue
        rows[tuple(f[:8])] = (int(f[8]), float(f[9]))
    return rows


a, b = load(sys.argv[1]), load(sys.argv[2])
title = sys.argv[3] if len(sys.argv) > 3 else f"{sys.argv[1]} vs {sys.argv[2]}"
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print(f"# Per-shape step time (ms per step, plan profiler, same box): {title}")
print("# columns: family M N K gg R V act | calls | A ms | B ms | delta ms | delta %")
print(f"# totals: A {ta:.2f} ms, B {tb:.2f} ms, delta {tb - ta:+.2f} ms ({100 * (tb - ta) / ta:+.1f} %)\n")
rows = sorted(((b.get(k, (0, 0.0))[1] - a.get(k, (0, 0.0))[1], k) for k in set(a) | set(b)), reverse=True)
fam = collections.OrderedDict()
for d, k in rows:
    f = fam.setdefault(k[0], [0.0, 0.0])
    f[0] += a.get(k, (0, 0.0))[1]
    f[1] += b.get(k, (0, 0.0))[1]
print("by family:")
for f, (ma, mb) in sorted(fam.items(), key=lambda kv: -(kv[1][1] - kv[1][0])):
    print(f"  {f:22s} {ma:8.2f} {mb:8.2f} {mb - ma:+7.2f}  {100 * (mb - ma) / ma if ma else 0:+6.1f} %")
print("\nby shape (sorted by delta):")
for d, k in rows:
    if abs(d) < 0.02:
        continue
    ma, mb = a.get(k, (0, 0.0))[1], b.get(k, (0, 0.0))[1]
    c = a.get(k, b.get(k))[0]
    print(f"  {k[0]:20s} {k[1]:>7s} {k[2]:>6s} {k[3]:>6s} {k[4]:>2s} {k[5]} {k[6]} {k[7]:>3s} | {c:4d} | {ma:7.2f} | {mb:7.2f} | {d:+6.2f} | "
          f"{100 * d / ma if ma else 0:+6.1f} %")
