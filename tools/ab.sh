# A/B of library builds on the helmet bench: tools/ab.sh <lib.so or "-" for the tree's> ...
mkdir -p gpurun_out
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset MI355R_LIB; else export MI355R_LIB=$PWD/$lib; fi
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --no-other > gpurun_out/bench_ab.json || exit 1
  python - "$lib" <<'PY'
import json, sys
r = json.load(open('gpurun_out/bench_ab.json'))
print(sys.argv[1], round(r['value']), 'fps', {k: round(v * 1000, 1) for k, v in r['kernel_ms_per_frame'].items()})
PY
done
