#!/bin/bash
# Developer aid (GPU box): L2 <-> fabric traffic of single GEMM shapes (TCC_EA0_RDREQ x 128 B, TCC_EA0_WRREQ x 64 B) next
# to their algorithmic bytes.  usage: tools/pmc_fetch.sh "<shape substring>" ...
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for SHAPE in "$@"; do
  OUT=$R/gpurun_out/pmcf; rm -rf $OUT; mkdir -p $OUT
  timeout 150 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p -o p -- python3 $R/tools/gemm_sweep.py --tiles 0 --only "$SHAPE" --reps 2 > $OUT/log.txt 2>&1
  python3 - "$SHAPE" <<PY
import csv,glob,collections,sys
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/p/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_pp_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in acc.items()}
print(f"{sys.argv[1]:32s} read {m.get('TCC_EA0_RDREQ_sum',0)*128/1e6:8.0f} MB  written {m.get('TCC_EA0_WRREQ_sum',0)*64/1e6:8.0f} MB   L2 hit rate {m.get('TCC_HIT_sum',0)/max(1,m.get('TCC_HIT_sum',0)+m.get('TCC_MISS_sum',0)):.3f}")
PY
done
