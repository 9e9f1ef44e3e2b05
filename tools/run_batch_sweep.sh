set -e
mkdir -p gpurun_out
for cfg in "4 16 64" "2 32 64" "1 64 64" "4 32 128" "2 64 128" "4 64 256"; do
  set -- $cfg
  MI355R_LANES=$1 MI355R_BATCH=$2 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --frames $3 --no-cpu > gpurun_out/bench_sweep.json
  python -c "
import json
r=json.load(open('gpurun_out/bench_sweep.json'))
print('lanes=$1 batch=$2 frames=$3:', round(r['value']), 'fps', {k: round(v*1000,1) for k,v in r['kernel_ms_per_frame'].items()})"
done
