mkdir -p gpurun_out/r3d
for v in old cur x1 x2; do
  if [ $v = cur ]; then L=$PWD/ctrlv_amd/lib/libctrlv_hip.so; elif [ $v = old ]; then L=$PWD/ctrlv_amd/lib/libctrlv_old.so; else L=$PWD/ctrlv_amd/lib/ab/libctrlv_$v.so; fi
  echo "== $v"
  CTRLV_HIP_LIB=$L timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "gemm" 2>&1 | tail -2
  CTRLV_HIP_LIB=$L python tools/gemm_sweep.py --tiles 5,6  2>/dev/null | grep -v "L3\|L2 qkv\|shape"
done
