mkdir -p gpurun_out/r3f
python tools/train_bench.py --steps 3 --warmup 1 > gpurun_out/r3f/train.json 2> gpurun_out/r3f/train.err
cat gpurun_out/r3f/train.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3f/prof -o p -- python3 $R/tools/train_bench.py --steps 2 --warmup 1 > $R/gpurun_out/r3f/prof.log 2>&1
cp $R/gpurun_out/r3f/prof/p_kernel_stats.csv $R/gpurun_out/r3f/train_kernel_stats.csv
rm -f $R/gpurun_out/r3f/prof/p_kernel_trace.csv
head -45 $R/gpurun_out/r3f/train_kernel_stats.csv | cut -c1-150
