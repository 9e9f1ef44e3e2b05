#!/usr/bin/env python3
"""Host cost of the reference's per-face submission (graphics_draw_triangle once per face): time to record a
frame, time to flush it, frames per second of record + flush + wait.  usage: per_triangle_cost.py [scene] [frames]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from renderer_amd.binding import Pipeline, GPU_LIB, SCENES
gpu = Pipeline(GPU_LIB)
name = sys.argv[1] if len(sys.argv) > 1 else "helmet"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
like = {"kgirl_blend.scn": "kgirl"}.get(name)
scene = gpu.open_scene(os.path.join(SCENES, name) if name.endswith(".scn") else name, like)
faces = gpu.scene_info(scene)["faces"]
gpu.lib.rzs_set_per_triangle(1)
fbs = [gpu.framebuffer(1920, 1080) for _ in range(B)]
views = [gpu.orbit_view(i, 64) for i in range(B)]
for rep in range(4):
    t0 = time.perf_counter()
    for fb, v in zip(fbs, views):
        gpu.draw(scene, fb, v)
    t1 = time.perf_counter()
    gpu.lib.graphics_flush()
    t2 = time.perf_counter()
    gpu.read(fbs[-1])
    t3 = time.perf_counter()
    print(f"{name}: {faces} faces; record {(t1-t0)/B*1e6:.0f} us/frame ({(t1-t0)/B/faces*1e9:.1f} ns/face), "
          f"flush {(t2-t1)/B*1e6:.0f} us/frame, wait {(t3-t2)/B*1e6:.0f} us/frame, {B/(t3-t0):.0f} frames/s")
gpu.lib.rzs_set_per_triangle(0)
