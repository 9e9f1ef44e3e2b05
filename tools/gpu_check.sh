# parity tests, then the default bench; stops at the first failure
set -e
mkdir -p gpurun_out
rm -f gpurun_out/mi355r_fatal.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread --durations=8 > gpurun_out/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/pytest_gpu.log; cat gpurun_out/mi355r_fatal.log 2>/dev/null; exit 1; }
tail -14 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --no-cpu > gpurun_out/bench_check.json
python - <<'PY'
import json
r = json.load(open('gpurun_out/bench_check.json'))
print(round(r['value']), 'fps', {k: round(v * 1000, 1) for k, v in r['kernel_ms_per_frame'].items()})
PY
