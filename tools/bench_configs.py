#!/usr/bin/env python3
"""Throughput of the other BASELINE configurations (not the contract bench): orbit batches of each
scene through the GPU library, frames/s and per-kernel time.
usage: tools/bench_configs.py [frames] [label prefix, e.g. C4] [plain]
plain: the rendering loop only (what a profiler run wants: no read-back leg, no event-timed leg)."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from renderer_amd import orbit                                     # noqa: E402
from renderer_amd.binding import Pipeline, GPU_LIB, SCENES          # noqa: E402

from renderer_amd.configs import CONFIGS                           # noqa: E402


def main():
    import torch
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    only = sys.argv[2] if len(sys.argv) > 2 else None
    plain = len(sys.argv) > 3 and sys.argv[3] == "plain"
    gpu = Pipeline(GPU_LIB)
    for label, name, like, (w, h), _ in CONFIGS:
        if only and not label.startswith(only):
            continue
        scene = gpu.open_scene(name, like)
        fbs = [gpu.framebuffer(w, h) for _ in range(frames)]
        views = [gpu.orbit_view(i, 240) for i in range(frames)]
        for _ in range(2):
            orbit.render_views(gpu, scene, fbs, views)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 5
        for _ in range(steps):
            orbit.render_views(gpu, scene, fbs, views)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if plain:
            print(json.dumps({"config": label, "fps": round(steps * frames / dt, 1)}))
            for fb in fbs:
                gpu.release(fb)
            gpu.close_scene(scene)
            continue
        # the same with every frame copied to host memory (colour + depth over PCIe)
        t0 = time.perf_counter()
        for _ in range(steps):
            orbit.render_views(gpu, scene, fbs, views)
            for fb in fbs:
                gpu.lib.framebuffer_sync(fb)
        dt_host = time.perf_counter() - t0
        gpu.lib.mi355r_enable_timing(1)
        orbit.render_views(gpu, scene, fbs, views)
        passes, sequences, ms = gpu.kernel_times()
        gpu.lib.mi355r_enable_timing(0)
        print(json.dumps({"config": label, "fps": round(steps * frames / dt, 1), "fps_with_readback": round(steps * frames / dt_host, 1), "passes_per_frame": passes / frames, "launch_sequences_per_frame": round(sequences / frames, 2),
                          "kernel_us_per_frame": {k: round(v / frames * 1e3, 1) for k, v in ms.items()}}))
        for fb in fbs:
            gpu.release(fb)
        gpu.close_scene(scene)


if __name__ == "__main__":
    main()
