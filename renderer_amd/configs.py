"""The BASELINE configurations other than the one bench.py's headline value is quoted on, as orbit
workloads of the GPU library: (label, scene or .scn path, catalogue entry for the root transform,
framebuffer size, views per step).  Shared by bench.py (`other_configs`) and tools/bench_configs.py."""
import os
import time

from . import orbit
from .binding import SCENES

CONFIGS = [
    ("C1 azura + shadow 800x600", os.path.join(SCENES, "azura_shadow.scn"), "azura", (800, 600), 64),
    ("C2 helmet 1080p", "helmet", None, (1920, 1080), 128),
    ("C3 kgirl skinned + blend + shadow 1080p, 240-view orbit", os.path.join(SCENES, "kgirl_blend.scn"), "kgirl", (1920, 1080), 240),
    ("C4 dieselpunk + 2048^2 shadow 4K", os.path.join(SCENES, "dieselpunk_shadow.scn"), "dieselpunk", (3840, 2160), 32),
    ("C5 buster animated 1080p", "buster", None, (1920, 1080), 128),
    # procedural meshes and seeded textures only (tools/make_synthetic.py): no third-party asset involved
    ("S1 synthetic blinn + shadow + cutoff + blend 1080p", os.path.join(SCENES, "synthetic_blinn.scn"), None, (1920, 1080), 32),
]


def time_config(gpu, config, steps=3, warmup=1, frames=None, per_triangle=False, sync=None, measure=False):
    """Frames/s of one configuration: `frames` orbit views (animation time index / 30 s) recorded and
    flushed per step, every view into its own framebuffer, scene resident on the device.
    per_triangle: through the reference's own per-face loop (program_get_attribs +
    graphics_draw_triangle) instead of graphics_draw_mesh.
    measure: also one untimed step with the device-side workload counters on and two steps with HIP events
    around every launch (`measured`: counters, summed kernel milliseconds, frames, passes, launch sequences)."""
    label, name, like, (w, h), default_frames = config
    frames = frames or default_frames
    if sync is None:
        import torch
        sync = torch.cuda.synchronize
    scene = gpu.open_scene(name, like)
    fbs = [gpu.framebuffer(w, h) for _ in range(frames)]
    views = [gpu.orbit_view(i, max(frames, 240)) for i in range(frames)]
    gpu.lib.rzs_set_per_triangle(1 if per_triangle else 0)
    try:
        for _ in range(warmup):
            orbit.render_views(gpu, scene, fbs, views)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            orbit.render_views(gpu, scene, fbs, views)
        sync()
        dt = time.perf_counter() - t0
    finally:
        gpu.lib.rzs_set_per_triangle(0)
    info = gpu.scene_info(scene)
    measured = None
    if measure:
        gpu.lib.mi355r_reset_counters()
        gpu.lib.mi355r_enable_counters(1)
        orbit.render_views(gpu, scene, fbs, views)
        counters = gpu.counters()
        gpu.lib.mi355r_enable_counters(0)
        gpu.lib.mi355r_enable_timing(1)
        for _ in range(2):
            orbit.render_views(gpu, scene, fbs, views)
        passes, sequences, ms = gpu.kernel_times()
        gpu.lib.mi355r_enable_timing(0)
        measured = {"counters": counters, "kernel_ms": ms, "frames": 2 * frames, "passes": passes, "sequences": sequences,
                    "shadow_side": info["shadow_w"], "raster_symbol": gpu.lib.mi355r_last_raster_kernel().decode()}
    for fb in fbs:
        gpu.release(fb)
    gpu.close_scene(scene)
    return {"workload": label, "frames_per_step": frames, "steps": steps, "ms_per_step": dt / steps * 1e3,
            "fps": steps * frames / dt, "triangles_per_frame": info["faces"] + (12 if info["skybox"] else 0),
            "width": w, "height": h, "submission": "per triangle (graphics_draw_triangle)" if per_triangle else "per mesh (graphics_draw_mesh)",
            **({"measured": measured} if measure else {})}
