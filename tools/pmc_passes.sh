#!/bin/bash
# GPU box: only the two rocprofv3 --pmc passes of tools/round_profile.sh (FETCH_SIZE / WRITE_SIZE, separate runs) and their
# summary -> gpurun_out/<tag>/pmc_hbm_traffic_summary.json.  usage: tools/pmc_passes.sh <tag>
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-round}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$C
  timeout 420 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- python3 $R/bench.py --steps 1 --warmup 1 --hip-graph 0 --no-cpu-baseline --no-profile-step --no-fp16-leg > $OUT/pmc_$C.log 2>&1
  echo "$C pass: exit $?"
done
cd $R
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE/p_counter_collection.csv $OUT/pmc_WRITE_SIZE/p_counter_collection.csv $OUT/pmc_hbm_traffic_summary.json 2 > $OUT/pmc_summary.txt 2>&1
tail -14 $OUT/pmc_summary.txt
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
