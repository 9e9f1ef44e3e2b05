#!/usr/bin/env python3
"""The viewer's loop: draw ONE frame, make its colour buffer valid on the host, repeat (no batching across frames).
usage: tools/interactive_rate.py [scene] [catalogue name|-] [width height]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from renderer_amd.binding import Pipeline, GPU_LIB, SCENES
gpu = Pipeline(GPU_LIB)
name = sys.argv[1] if len(sys.argv) > 1 else "helmet"
like = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
scene = gpu.open_scene(os.path.join(SCENES, name) if name.endswith(".scn") else name, like)
fb = gpu.framebuffer(W, H)
views = [gpu.orbit_view(i, 240) for i in range(240)]
for v in views[:20]:
    gpu.draw(scene, fb, v); gpu.lib.framebuffer_sync(fb)
gpu.lib.framebuffer_sync_color.argtypes = gpu.lib.framebuffer_sync.argtypes
for what, sync in (("colour and depth on the host (framebuffer_sync)", lambda: gpu.lib.framebuffer_sync(fb)),
                   ("colour on the host (framebuffer_sync_color)", lambda: gpu.lib.framebuffer_sync_color(fb)),
                   ("frame finished on the device (framebuffer_read_device of nothing)", lambda: gpu.lib.framebuffer_read_device(fb, None, None))):
    t0 = time.perf_counter()
    for v in views:
        gpu.draw(scene, fb, v)
        sync()
    dt = time.perf_counter() - t0
    print(f"{name} {W}x{H}: {len(views) / dt:.0f} frames/s, {dt / len(views) * 1e3:.3f} ms per frame, {what}")
