# kernel statistics of a command: tools/stats_cmd.sh <rows> <program> [arguments]
set -u
ROWS=$1; shift
OUT=gpurun_out/stats_cmd
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- "$@" > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
tail -2 "$OUT/log.txt"
python3 - "$OUT" "$ROWS" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
total = sum(float(r['TotalDurationNs']) for r in rows)
print(f"  kernels {sum(int(r['Calls']) for r in rows)} launches, {total/1e3:.0f} us in all")
for r in rows[:int(sys.argv[2])]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} min_us {float(r['MinNs'])/1e3:8.1f} pct {r['Percentage']}")
PY
