#!/bin/bash
# GPU box: the round's measurement set -> gpurun_out/<tag>/ (copy what should be judged into profiles/).
#   1. bench.py (HIP-graph replay, default flags)                       -> bench.json
#   2. rocprofv3 --kernel-trace --stats of an eager bench run           -> kernel_stats.csv
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes         -> pmc_hbm_traffic_summary.json (build-id stamped)
# usage: tools/round_profile.sh <tag> [extra bench args]
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-round}; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python3 bench.py --steps 8 --warmup 2 "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python3 $R/bench.py --steps 4 --warmup 2 --hip-graph 0 --no-cpu-baseline > $OUT/prof.log 2>&1
cp $OUT/prof/p_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- python3 $R/bench.py --steps 1 --warmup 1 --hip-graph 0 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE/p_counter_collection.csv $OUT/pmc_WRITE_SIZE/p_counter_collection.csv $OUT/pmc_hbm_traffic_summary.json > $OUT/pmc_summary.txt 2>&1
tail -12 $OUT/pmc_summary.txt
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/prof/p_kernel_trace.csv
