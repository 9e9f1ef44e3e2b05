#!/bin/bash
# kernel trace of the viewer's loop (one frame drawn and synchronised at a time): per-kernel durations and the gaps between them
OUT=gpurun_out/trace_interactive
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 tools/interactive_rate.py "$@" > "$OUT/run.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/run.log"; exit 1; }
cat "$OUT/run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in csv.DictReader(open(f))))
rows = rows[len(rows) * 3 // 4:]          # the last loop: frames finished on the device only
dur, gap, n = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
for (s0, e0, _), (s1, e1, k) in zip(rows, rows[1:]):
    dur[k] += (e1 - s1) / 1e3; gap[k] += max(0, s1 - e0) / 1e3; n[k] += 1
frames = n[max(n, key=lambda k: ("strip" in k or "visibility" in k, n[k]))] or 1
print(f"per frame over {frames} frames (us): kernel time, idle time in front of the kernel")
for k in sorted(dur, key=lambda k: -dur[k]):
    print(f"  {k:60s} {dur[k] / frames:7.1f} {gap[k] / frames:7.1f}   x{n[k] / frames:.1f}")
print(f"  total kernel {sum(dur.values()) / frames:.1f} us, idle {sum(gap.values()) / frames:.1f} us")
PY
