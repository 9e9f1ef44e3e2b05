# helmet bench under the strip kernel's variants
mkdir -p gpurun_out
for v in "MI355R_STRIP=2" "MI355R_STRIP=4" "MI355R_STRIP=8" "MI355R_STRIP=1"; do
  env $v timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --no-other > gpurun_out/bench_sweep.json || exit 1
  python - "$v" <<'PY'
import json, sys
r = json.load(open('gpurun_out/bench_sweep.json'))
print(sys.argv[1], round(r['value']), 'fps', {k: round(v * 1000, 1) for k, v in r['kernel_ms_per_frame'].items()})
PY
done
