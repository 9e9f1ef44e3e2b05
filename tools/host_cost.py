#!/usr/bin/env python3
"""How much of a bench step is host work: time until the submitting calls return vs until the GPU is done."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from renderer_amd import orbit
from renderer_amd.binding import Pipeline, GPU_LIB, SCENES
gpu = Pipeline(GPU_LIB)
# usage: host_cost.py [scene or .scn under scenes/] [catalogue name for its root transform] [width height] [frames per step]
name = sys.argv[1] if len(sys.argv) > 1 else "helmet"
like = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
scene = gpu.open_scene(os.path.join(SCENES, name) if name.endswith(".scn") else name, like)
B = int(sys.argv[5]) if len(sys.argv) > 5 else 32
fbs = [gpu.framebuffer(W, H) for _ in range(B)]
views = [gpu.orbit_view(i, 2 * B) for i in range(B)]
for _ in range(3):
    orbit.render_views(gpu, scene, fbs, views)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); sub = 0.0
    for _ in range(10):
        a = time.perf_counter()
        orbit.render_views(gpu, scene, fbs, views)
        sub += time.perf_counter() - a
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"submit {sub/(10*B)*1e6:.1f} us/frame, total {tot/(10*B)*1e6:.1f} us/frame, {10*B/tot:.0f} fps")
# recording alone (no flush): the scene layer + draw recording
t0 = time.perf_counter()
for fb, v in zip(fbs, views):
    gpu.draw(scene, fb, v)
rec = time.perf_counter() - t0
t0 = time.perf_counter()
gpu.lib.graphics_flush()
fl = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"record {rec/B*1e6:.1f} us/frame, flush {fl/B*1e6:.1f} us/frame")
