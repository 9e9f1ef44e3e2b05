set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/pytest_gpu.log
for cfg in "1 1" "1 4" "1 32" "4 1" "4 8" "2 16" "4 4"; do
  set -- $cfg
  echo "lanes=$1 batch=$2"
  MI355R_LANES=$1 MI355R_BATCH=$2 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/bench_$1_$2.json
  python -c "
import json,sys
r=json.load(open('gpurun_out/bench_$1_$2.json'))
print(round(r['value']), {k:round(v*1000,1) for k,v in r['kernel_ms_per_frame'].items()})"
done
bash tools/profile.sh r1c > /dev/null
head -8 gpurun_out/prof_r1c/summary.txt
