#!/usr/bin/env python3
"""Wall time and peak resident memory of loading a scene and rendering its first frame (which uploads
every image): tools/load_cost.py [scene]"""
import os, resource, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from renderer_amd.binding import Pipeline, GPU_LIB
name = sys.argv[1] if len(sys.argv) > 1 else "helmet"
gpu = Pipeline(GPU_LIB)
gpu.lib.mi355r_init(0)
fb = gpu.framebuffer(64, 64)
gpu.read(fb)                                    # the device is up, the kernels are loaded
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
t0 = time.perf_counter()
scene = gpu.open_scene(name)
t1 = time.perf_counter()
gpu.draw(scene, fb, gpu.default_view())
gpu.read(fb)
t2 = time.perf_counter()
rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
print(f"{name}: open {1e3 * (t1 - t0):.0f} ms, first frame (uploads) {1e3 * (t2 - t1):.0f} ms, peak RSS {rss0:.0f} -> {rss1:.0f} MiB")
