#!/bin/bash
# GPU box: matrix-pipe utilisation per kernel family -- ONE rocprofv3 --pmc pass (counters only, no tracing) over an eager
# bench step, plus the in-kernel clock of the MFMA kernels from the diagnostic build (tools/clock_probe.py)
#   -> gpurun_out/<tag>/pmc_mfma_busy_summary.json (copy to profiles/rNN_pmc_mfma_busy_summary.json).
# usage: tools/pmc_mfma_pass.sh <tag>
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-round}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python3 tools/clock_probe.py --no-build --json $OUT/inkernel_clock.json > $OUT/inkernel_clock.txt 2>&1
tail -12 $OUT/inkernel_clock.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_mfma
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o p -- python3 $R/bench.py --steps 1 --warmup 1 --hip-graph 0 --no-cpu-baseline --no-profile-step --no-fp16-leg > $OUT/pmc_mfma.log 2>&1
echo "mfma pass: exit $?"
cd $R
python3 tools/pmc_mfma_summary.py $OUT/pmc_mfma/p_counter_collection.csv $OUT/pmc_mfma_busy_summary.json $OUT/inkernel_clock.json > $OUT/pmc_mfma_summary.txt 2>&1
cat $OUT/pmc_mfma_summary.txt
rm -rf $OUT/pmc_mfma
