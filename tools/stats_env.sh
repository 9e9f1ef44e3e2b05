# kernel statistics of one configuration under an environment: tools/stats_env.sh C5 "<env assignments>" [rows]
set -u
C=$1; ENVS=$2; ROWS=${3:-8}
OUT=gpurun_out/stats_env
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
export $ENVS
if [ "$C" = C2 ]; then CMD="python3 bench.py --steps 3 --warmup 1 --frames 32 --no-cpu --no-other"; else CMD="python3 tools/bench_configs.py 32 $C plain"; fi
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- $CMD > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
echo "== $C $ENVS"
python3 - "$OUT" "$ROWS" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[2])]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} max_us {float(r['MaxNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
