set -x
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -15 > gpurun_out/r3b/test_ops.txt
cat gpurun_out/r3b/test_ops.txt
for r in 1 2; do
CTRLV_HIP_LIB=$PWD/ctrlv_amd/lib/libctrlv_old.so python tools/gemm_sweep.py --tiles 5,6 > gpurun_out/r3b/sweep_old_$r.txt 2>&1
python tools/gemm_sweep.py --tiles 5,6 > gpurun_out/r3b/sweep_new_$r.txt 2>&1
done
paste gpurun_out/r3b/sweep_old_1.txt gpurun_out/r3b/sweep_new_1.txt gpurun_out/r3b/sweep_old_2.txt gpurun_out/r3b/sweep_new_2.txt | cut -c1-50,85-100,135-150,185-200
CTRLV_HIP_LIB=$PWD/ctrlv_amd/lib/libctrlv_old.so python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3b/bench_old.json 2> gpurun_out/r3b/bench_old.err
python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3b/bench_new.json 2> gpurun_out/r3b/bench_new.err
tail -2 gpurun_out/r3b/bench_new.err
python tools/show_bench.py gpurun_out/r3b/bench_old.json
python tools/show_bench.py gpurun_out/r3b/bench_new.json
