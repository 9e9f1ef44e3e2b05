set -e
mkdir -p gpurun_out
for cfg in "1 16 32" "2 16 32" "4 8 32" "2 32 64" "4 16 64" "4 32 128" "8 16 128"; do
  set -- $cfg
  MI355R_LANES=$1 MI355R_BATCH=$2 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --frames $3 --no-cpu > gpurun_out/bench_sweep.json
  python -c "
import json
r=json.load(open('gpurun_out/bench_sweep.json'))
print('lanes=$1 batch=$2 frames=$3:', round(r['value']), 'fps')"
done
