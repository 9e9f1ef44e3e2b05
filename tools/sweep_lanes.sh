for v in "A=1" "MI355R_LANES=8" "MI355R_BATCH=16" "MI355R_LANES=8 MI355R_BATCH=16" "MI355R_LANES=2" "MI355R_BATCH=64"; do
  echo "== $v"
  for c in C1 C3 C5; do
    f=128; [ $c = C3 ] && f=240; [ $c = C1 ] && f=64
    env $v timeout -k 10 120 python3 tools/bench_configs.py $f $c plain 2>/dev/null | tail -1 || exit 1
  done
done
