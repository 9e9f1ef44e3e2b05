#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: how much of the wall time between the first and the last launch of the later
half of the run had at least one kernel running, and the sum of kernel durations per kernel name there.
usage: busy_share.py <directory with *kernel_trace.csv below it> [frames in the later half]"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows]
rows.sort()
half = rows[len(rows) // 2:]
t0, t1 = half[0][0], max(e for _, e, _ in half)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in half:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"later half: {len(half)} launches, wall {(t1 - t0) / 1e3:.0f} us, some kernel running {busy / 1e3:.0f} us ({busy / (t1 - t0):.2f})")
per = collections.Counter()
for s, e, k in half:
    per[k] += e - s
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 0
for k, v in per.most_common(12):
    print(f"  {k[:48]:50s} {v / 1e3:10.0f} us" + (f"  {v / 1e3 / frames:7.2f} us/frame" if frames else ""))
