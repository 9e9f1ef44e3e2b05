# frames per step x passes per batch x lanes on the helmet bench
for cfg in "128 32 4" "256 64 4" "256 32 8" "512 64 8" "128 64 2" "256 32 4"; do
  set -- $cfg
  MI355R_BATCH=$2 MI355R_LANES=$3 timeout -k 10 200 python bench.py --frames $1 --steps 6 --warmup 2 --no-cpu --no-other > gpurun_out/bench_tb.json || exit 1
  python - "$cfg" <<'PY'
import json, sys
r = json.load(open('gpurun_out/bench_tb.json'))
print(sys.argv[1], round(r['value']), 'fps', {k: round(v * 1000, 1) for k, v in r['kernel_ms_per_frame'].items()})
PY
done
