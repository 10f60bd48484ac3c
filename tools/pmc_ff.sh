#!/bin/bash
# Developer aid (GPU box): SQ counters of the fused feed-forward kernel, one rocprofv3 --pmc pass per counter group.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/pmcff; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_WAVES"; do
  i=$((i+1))
  FF_SETS=2 timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/ff_bench.py > $OUT/log$i.txt 2>&1
done
cd $R
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmcff/p*/*counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "ff_fused_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(f.split("/")[2], k, "n=%d"%len(v), "mean=%.5g"%(sum(v)/len(v)))
PY
