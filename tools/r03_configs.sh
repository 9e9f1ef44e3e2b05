#!/bin/bash
# Kernel statistics of the BASELINE configurations other than the helmet, one profiler run each:
# tools/r03_configs.sh <tag>   -> gpurun_out/r03_<tag>/<config>/..., summary on stdout
set -u
TAG=${1:-base}
export TMPDIR=/tmp
for C in C1 C3 C4 C5; do
    OUT=gpurun_out/r03_$TAG/$C
    rm -rf "$OUT"; mkdir -p "$OUT"
    N=32
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 tools/bench_configs.py $N $C > "$OUT/log.txt" 2>&1 || { echo "$C failed"; tail -5 "$OUT/log.txt"; exit 1; }
    grep '^{' "$OUT/log.txt"
    python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print(f"  {r['Name'][:80]:80s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
done
