# per-kernel time of one configuration: tools/stats_config.sh C4 [frames]
set -u
C=$1; N=${2:-32}
OUT=gpurun_out/stats_$C
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 tools/bench_configs.py $N $C > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
