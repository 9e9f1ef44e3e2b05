# A/B of environment variants on the helmet bench: tools/ab_env.sh "VAR=1" "VAR=2 OTHER=3" ...
mkdir -p gpurun_out
for v in "$@"; do
  env $v timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --no-other > gpurun_out/bench_ab.json || exit 1
  python - "$v" <<'PY'
import json, sys
r = json.load(open('gpurun_out/bench_ab.json'))
print(sys.argv[1], round(r['value']), 'fps', {k: round(v * 1000, 1) for k, v in r['kernel_ms_per_frame'].items()})
PY
done
