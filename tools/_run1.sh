for v in tapmajor ""; do
  if [ -n "$v" ]; then export CTRLV_HIP_LIB=$PWD/ctrlv_amd/lib/ab/libctrlv_$v.so; else unset CTRLV_HIP_LIB; fi
  echo "=== variant: ${v:-new}"
  python tools/shape_table.py 2>/dev/null | grep "gemm_conv" | sort -k2,2n -k3,3n -k4,4n | awk '{printf "%s %7d %5d %6d R%d V%d  calls %3d  %7.2f ms  %6.0f TFLOP/s\n",$1,$2,$3,$4,$6,$7,$9,$10,$11}'
done
