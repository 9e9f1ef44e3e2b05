"""One frame per flush (the reference viewer's loop): frames/s to device-finished and to colour-on-host.
tools/interactive.py [scene] [frames]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from renderer_amd.binding import Pipeline, GPU_LIB

scene = sys.argv[1] if len(sys.argv) > 1 else "helmet"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 240
gpu = Pipeline(GPU_LIB)
print(json.dumps(bench.interactive_rates(gpu, scene, frames)))
