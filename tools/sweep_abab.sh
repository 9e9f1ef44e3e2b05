# frames/s of C1 (64 frames per step), C3 (240), C5 (128) and the helmet under two environment settings, A B A B:
# tools/sweep_abab.sh "<env A>" "<env B>"
for v in "$1" "$2" "$1" "$2"; do
  echo "== $v"
  for c in C1 C3 C5; do
    f=128; [ $c = C3 ] && f=240; [ $c = C1 ] && f=64
    env $v timeout -k 10 120 python3 tools/bench_configs.py $f $c plain 2>/dev/null | tail -1 || exit 1
  done
  env $v timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu --no-other 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('C2', round(r['value']))" || exit 1
done
