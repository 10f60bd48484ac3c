#!/usr/bin/env python
"""Pretty-print a bench.py JSON line (developer aid)."""
import json
import sys

for l in open(sys.argv[1]).read().strip().splitlines():
    if l.startswith("{"):
        d = json.loads(l)
        print({k: v for k, v in d.items() if k in ("value", "ms_per_step", "step_mfma_frac", "kernel_ms_per_step", "finite", "n_gpus")})
        for k, v in sorted(d.get("rooflines", {}).items(), key=lambda kv: -kv[1]["ms_per_step"]):
            print(f"  {k:22s} {v['ms_per_step']:9.2f} ms {v['launches']:4d} launches {v['achieved']:8.1f} {v['unit']} frac {v['frac']}")
        print(" cpu_baseline:", d.get("cpu_baseline"))
