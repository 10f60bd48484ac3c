for st in 0 2 4 6 0 4; do
  echo "== stagger $st"
  python tools/gemm_sweep.py --tiles 6 --dbg $((st*256)) 2>/dev/null | grep -v "L3\|shape\|L2 conv\|L1 conv"
done
