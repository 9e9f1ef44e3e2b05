#!/bin/bash
# raster time of the configurations against the small-triangle threshold: tools/sweep_micro.sh
for a in 0 8 16 32 64 128; do
  echo "== MI355R_MICRO_AREA=$a"
  MI355R_MICRO_AREA=$a timeout -k 10 300 python tools/bench_configs.py 16 2>/dev/null | python3 -c '
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l); print("  ", d["config"][:12], d["fps"], d["kernel_us_per_frame"]["raster"])
' || exit 1
done
