for rep in 1 2; do
python tools/attn_bench.py 2>&1 | grep -v amdgpu | grep "L0\|L1"
CTRLV_HIP_LIB=$PWD/ctrlv_amd/lib/ab/libctrlv_head.so python tools/attn_bench.py 2>&1 | grep -v amdgpu | grep "L0\|L1"
done
