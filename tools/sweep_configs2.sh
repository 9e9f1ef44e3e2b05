# frames/s of the configurations under environment settings: tools/sweep_configs2.sh "<env>" "<env>" ...
for v in "$@"; do
  echo "== $v"
  for c in C1 C3 C4 C5 S1; do
    env $v timeout -k 10 120 python3 tools/bench_configs.py 32 $c plain 2>/dev/null | tail -1 || exit 1
  done
  env $v timeout -k 10 120 python3 bench.py --steps 5 --warmup 2 --no-cpu --no-other 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('C2', round(r['value']), {k: round(v*1000,1) for k,v in r['kernel_ms_per_frame'].items()})" || exit 1
done
