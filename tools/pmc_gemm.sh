#!/bin/bash
# Developer aid (GPU box): SQ / TA counters of one GEMM shape, one rocprofv3 --pmc pass per counter group (no tracing).
# usage: tools/pmc_gemm.sh "<shape substring>" <tile> [extra gemm_sweep args]
R=$(cd "$(dirname "$0")/.." && pwd)
SHAPE=${1:-"L1 conv3x3 1920"}; TILE=${2:-6}; shift 2
OUT=$R/gpurun_out/pmcg; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_INST_LDS" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum" \
         "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/gemm_sweep.py --tiles $TILE --only "$SHAPE" --reps 2 "$@" > $OUT/log$i.txt 2>&1
done
cd $R
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmcg/p*/*counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_pp_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(f.split("/")[2], k, "n=%d"%len(v), "mean=%.5g"%(sum(v)/len(v)))
PY
