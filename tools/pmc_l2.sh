#!/bin/bash
# Developer aid (GPU box): L2 (TCC) / L1->L2 counters of one GEMM shape, one rocprofv3 --pmc pass per counter group.
# usage: tools/pmc_l2.sh "<shape substring>" <tile> [extra gemm_sweep args]
R=$(cd "$(dirname "$0")/.." && pwd)
SHAPE=${1:-"L0 qkv"}; TILE=${2:-6}; shift 2
OUT=$R/gpurun_out/pmcl2; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum" \
         "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum" \
         "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_NORMAL_WRITEBACK_sum" \
         "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/gemm_sweep.py --tiles $TILE --only "$SHAPE" --reps 2 "$@" > $OUT/log$i.txt 2>&1
done
cd $R
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmcl2/p*/*counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_pp_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(k, "n=%d"%len(v), "mean=%.5g"%(sum(v)/len(v)))
PY
