#!/bin/bash
# raster time of the configurations against the small-triangle settings: tools/sweep_micro.sh
for m in 0 128 256 512 1024; do
  echo "== MI355R_MICRO_MIN_LIST=$m"
  MI355R_MICRO_MIN_LIST=$m timeout -k 10 300 python tools/bench_configs.py 32 2>/dev/null | python3 -c '
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("  ", d["config"][:12], d["fps"], d["kernel_us_per_frame"]["raster"])
' || exit 1
done
