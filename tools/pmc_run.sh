#!/bin/bash
# Developer aid (GPU box): SQ counters of the kernels matching a name filter, one rocprofv3 --pmc pass per group (no tracing).
# usage: tools/pmc_run.sh <kernel-name substring> <python script and args...>
R=$(cd "$(dirname "$0")/.." && pwd)
FILT=$1; shift
OUT=$R/gpurun_out/pmcr; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_MFMA" \
         "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o p -- python3 "$@" > $OUT/log$i.txt 2>&1
done
cd $R
python3 - "$FILT" <<PY
import csv,glob,collections,sys
filt=sys.argv[1]
for f in sorted(glob.glob("gpurun_out/pmcr/p*/*counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(f.split("/")[2], k, "n=%d"%len(v), "max=%.5g"%max(v), "mean=%.5g"%(sum(v)/len(v)))
PY
