#!/bin/bash
# kernel-trace only: per-kernel medians of a short bench run -> gpurun_out/trace_<tag>/summary.txt
TAG=${1:-t}
OUT=gpurun_out/trace_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 3 --warmup 1 --frames 32 --no-cpu --no-other > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
python3 tools/summarize_profile.py "$OUT" 32 > "$OUT/summary.txt" 2>&1
grep -v "^    " "$OUT/summary.txt"
