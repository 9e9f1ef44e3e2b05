#!/usr/bin/env python3
"""What a rank of `bench.py --gpus N` does per step apart from the collective: render 128 views, then copy the
frames as RGB8 into the device buffer the gather sends.  Frames per second with and without the copy."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from renderer_amd import orbit
from renderer_amd.binding import Pipeline, GPU_LIB
gpu = Pipeline(GPU_LIB)
scene = gpu.open_scene("helmet")
B, W, H = 128, 1920, 1080
fbs = [gpu.framebuffer(W, H) for _ in range(B)]
views = [gpu.orbit_view(i, B) for i in range(B)]
out = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda")
for copy in (False, True, False, True):
    for _ in range(2):
        orbit.render_views(gpu, scene, fbs, views)
    torch.cuda.synchronize(); gpu.lib.framebuffer_read_device(fbs[-1], None, None)
    t0 = time.perf_counter()
    for _ in range(8):
        orbit.render_views(gpu, scene, fbs, views)
        if copy:
            orbit.frames_to_tensor(gpu, fbs, B, out)
    for fb in fbs[-4:]:
        gpu.lib.framebuffer_read_device(fb, None, None)
    dt = time.perf_counter() - t0
    print(f"{'render + copy into the gather buffer' if copy else 'render only':40s} {8 * B / dt:8.0f} frames/s  {dt / 8 * 1e3:.2f} ms per step")
