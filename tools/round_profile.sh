#!/bin/bash
# GPU box: the round's measurement set -> gpurun_out/<tag>/ (copy what should be judged into profiles/).
#   1. bench.py (HIP-graph replay, default flags)                       -> bench.json
#   2. rocprofv3 --kernel-trace --stats of an eager bench run           -> kernel_stats.csv
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes         -> pmc_hbm_traffic_summary.json (build-id stamped)
#   4. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... + in-kernel clock   -> pmc_mfma_busy_summary.json (tools/pmc_mfma_pass.sh)
#   5. the other configurations of the same build, incl. bench.py --dtype fp16 --trunk fp16x2 (north_star's tolerance)
#   6. the cfg5 training step per kernel (rocprofv3 kernel trace cut at the step marks) and every weight-gradient shape
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
  timeout 900 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- python3 $R/bench.py --steps 1 --warmup 1 --hip-graph 0 --no-cpu-baseline --no-profile-step > $OUT/pmc_$C.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE/p_counter_collection.csv $OUT/pmc_WRITE_SIZE/p_counter_collection.csv $OUT/pmc_hbm_traffic_summary.json 2 > $OUT/pmc_summary.txt 2>&1
tail -12 $OUT/pmc_summary.txt
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/prof/p_kernel_trace.csv
tools/pmc_mfma_pass.sh $TAG
# 5. the other configurations of the same build (no CPU leg unless noted): the fp16 element build with the split residual
#    trunk (WITH the CPU oracle leg: its parity object is the north_star tolerance claim), SVD UNet only (cfg2), the
#    reference's default 320x512 size, the cfg5 training step, the end-to-end clip latency through the pipeline
python3 bench.py --steps 8 --warmup 2 --dtype fp16 --trunk fp16x2 > $OUT/bench_fp16_split_trunk.json 2>> $OUT/bench.err
python3 bench.py --steps 8 --warmup 2 --workload svd_unet --no-cpu-baseline > $OUT/bench_svd_unet.json 2>> $OUT/bench.err
python3 bench.py --steps 12 --warmup 3 --height 320 --width 512 --no-cpu-baseline > $OUT/bench_320x512.json 2>> $OUT/bench.err
python3 tools/train_bench.py --steps 4 --warmup 2 > $OUT/train_step.json 2>> $OUT/bench.err
python3 tools/train_bench.py --steps 4 --warmup 2 --gradient-checkpointing 1 > $OUT/train_step_checkpointed.json 2>> $OUT/bench.err
python3 tools/pipeline_bench.py > $OUT/pipeline_clip_latency.json 2>> $OUT/bench.err
tail -1 $OUT/bench_svd_unet.json | cut -c1-160; tail -1 $OUT/bench_320x512.json | cut -c1-160; tail -1 $OUT/train_step.json | cut -c1-200; tail -2 $OUT/pipeline_clip_latency.json | cut -c1-300
# 6. training step per kernel; the weight-gradient shapes on the LDS-DMA kernel and on the register-staged one
python3 tools/wgrad_bench.py > $OUT/wgrad_bench.txt 2>&1
CTRLV_WGRAD_PP=0 python3 tools/wgrad_bench.py > $OUT/wgrad_bench_register_staged.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tprof -o p -- python3 $R/tools/train_bench.py --steps 2 --warmup 2 --mark-steps 1 > $OUT/tprof.log 2>&1
cd $R
python3 tools/trace_last_step.py $OUT/tprof/p_kernel_trace.csv $OUT/train_last_step_kernels.csv > $OUT/train_last_step.txt 2>&1
rm -rf $OUT/tprof
head -3 $OUT/train_last_step.txt
