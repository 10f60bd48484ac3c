set -x
mkdir -p gpurun_out/r3a
./tools/micro/mfma_shape 300000 > gpurun_out/r3a/mfma_shape.txt 2>&1
cat gpurun_out/r3a/mfma_shape.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3a/test_ops.txt
cat gpurun_out/r3a/test_ops.txt
python tools/attn_bench.py 2>&1 | grep spatial > gpurun_out/r3a/attn.txt
cat gpurun_out/r3a/attn.txt
python tools/gemm_sweep.py --only geglu --tiles 5,6 > gpurun_out/r3a/sweep_geglu.txt 2>&1
cat gpurun_out/r3a/sweep_geglu.txt
python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
tail -3 gpurun_out/r3a/bench.err
python tools/show_bench.py gpurun_out/r3a/bench.json 2>/dev/null || cat gpurun_out/r3a/bench.json
