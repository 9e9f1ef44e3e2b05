#!/bin/bash
# kernel trace of one BASELINE configuration: tools/trace_config.sh C1
OUT=gpurun_out/trace_$1
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/bench_configs.py 32 $1 > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
python3 tools/summarize_profile.py "$OUT" 1 > "$OUT/summary.txt" 2>&1
grep -v "^    " "$OUT/summary.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last frame's launches in order
tail = rows[-40:]
for r in tail:
    print(r["Kernel_Name"].split("(")[0][:40].ljust(42), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
PY
