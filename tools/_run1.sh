for rep in 1 2 3; do for v in "" attnxcd; do
  if [ -n "$v" ]; then export CTRLV_HIP_LIB=$PWD/ctrlv_amd/lib/ab/libctrlv_$v.so; else unset CTRLV_HIP_LIB; fi
  echo "=== variant: ${v:-default}"
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python tools/show_bench.py /dev/stdin | grep "value\|attention_spatial"
done; done
