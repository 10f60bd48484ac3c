for v in "" win1048576 win33554432 win134217728 nostore; do
  if [ -n "$v" ]; then export CTRLV_HIP_LIB=$PWD/ctrlv_amd/lib/ab/libctrlv_$v.so; else unset CTRLV_HIP_LIB; fi
  echo "=== variant: ${v:-default}"
  python tools/gemm_sweep.py --only "L0" --tiles 6 --reps 10 2>&1 | grep -v "amdgpu.ids\|conv"
done
