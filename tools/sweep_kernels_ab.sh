for v in "MI355R_LIB=$PWD/ab/libmi355r_old.so" "A=1" "MI355R_LIB=$PWD/ab/libmi355r_old.so" "A=1"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 --no-cpu --no-other 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value']), {k: round(v*1000,2) for k,v in r['kernel_ms_per_frame'].items()})" || exit 1
done
