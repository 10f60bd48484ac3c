mkdir -p gpurun_out/r2e
python -m pytest tests/test_fullwidth_gpu.py tests/test_fullsize_gpu.py -q 2>&1 | tail -15 > gpurun_out/r2e/pytest.log; tail -3 gpurun_out/r2e/pytest.log
for r in 128 256 512; do echo GN_ROWS=$r; CTRLV_GN_ROWS=$r python tools/gn_bench.py 2>&1 | grep -E "L0|L1 "; done > gpurun_out/r2e/gn.log 2>&1; cat gpurun_out/r2e/gn.log
for t in 1 2; do echo TPB=$t; CTRLV_ATTN_TPB=$t python tools/attn_bench.py 2>&1 | grep spatial; done > gpurun_out/r2e/attn.log 2>&1; cat gpurun_out/r2e/attn.log
CTRLV_ATTN_TPB=2 python -m pytest tests/test_ops_gpu.py -q -k attention_spatial 2>&1 | tail -3
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --height 320 --width 512 > gpurun_out/r2e/bench_320.json 2> gpurun_out/r2e/bench_320.err; tail -1 gpurun_out/r2e/bench_320.err
