#!/usr/bin/env python3
"""One line per configuration of a tools/bench_configs.py log: frames/s, launch sequences per frame, raster and total kernel us per frame."""
import json
import sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        k = d["kernel_us_per_frame"]
        print(d["config"][:14], d["fps"], d["launch_sequences_per_frame"], "raster", k["raster"], "all", round(sum(k.values()), 1))
