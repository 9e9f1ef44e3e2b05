#!/usr/bin/env python3
"""What a rank of `bench.py --gpus N` does per step apart from the collective: render 128 views, then copy the
frames as RGB8 into the device buffer the gather sends.  Frames per second without the copy, with the copy
waited for at once, and with the copy in two halves (launched by this step, waited for by the next: what
orbit.FrameGather does with frames in device memory)."""
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
import ctypes as C
handles = (type(fbs[0]) * B)(*fbs)
outs = [out, torch.empty_like(out)]
for mode in ("render only", "render + copy, waited for at once", "render + copy in two halves", "render only",
             "render + copy, waited for at once", "render + copy in two halves"):
    for _ in range(2):
        orbit.render_views(gpu, scene, fbs, views)
    torch.cuda.synchronize(); gpu.lib.framebuffer_read_device(fbs[-1], None, None)
    t0 = time.perf_counter()
    for step in range(8):
        orbit.render_views(gpu, scene, fbs, views)
        if mode.endswith("at once"):
            orbit.frames_to_tensor(gpu, fbs, B, out)
        elif mode.endswith("halves"):
            gpu.lib.framebuffers_read_device_rgb_end()
            gpu.lib.framebuffers_read_device_rgb_begin(handles, B, C.c_void_p(outs[step % 2].data_ptr()), out[0].numel())
    gpu.lib.framebuffers_read_device_rgb_end()
    for fb in fbs[-4:]:
        gpu.lib.framebuffer_read_device(fb, None, None)
    dt = time.perf_counter() - t0
    print(f"{mode:40s} {8 * B / dt:8.0f} frames/s  {dt / 8 * 1e3:.2f} ms per step")
