#!/usr/bin/env python3
"""Benchmark of the hot path: 'pbr helmet' (metallic-roughness + IBL + blurred
skybox) at 1920x1080 on MI355X, rendered through the C ABI of
renderer_amd/libmi355r.so.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one batch of FRAMES orbit views of the scene (camera on a circle of
radius 1.5 around the model, animation time index/30 s), every frame a full
pass of the pipeline (vertex -> setup/clip -> bin -> raster+shade) into its
own framebuffer; scene data (meshes, textures, IBL maps) is resident in HBM
before the timed region.  With N > 1 ranks every rank renders its own batch
(frames are independent: weak scaling) and the finished RGBA8 frames are
gathered on rank 0 over RCCL inside the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WIDTH, HEIGHT = 1920, 1080
SCENE = "helmet"
ORBIT = 64                  # views on the orbit circle
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E, spec


def algorithmic_bytes(counters, frames, faces):
    """Algorithmic HBM bytes of the rendered frames: 240 B per input triangle
    (3 x 80-byte vertices), 8 B per pixel (RGBA8 + f32 depth written once),
    and per shaded fragment the texels its shader samples at the reference's
    precision (16-byte float4): 9 for a pbr fragment (6 material maps, 2 cube
    maps, BRDF LUT), 1 for a skybox fragment."""
    shaded, skybox = counters[2], counters[6]
    pbr = shaded - skybox
    return frames * (240 * faces + 8 * WIDTH * HEIGHT) + 144 * pbr + 16 * skybox


def cpu_baseline(frames_wanted):
    """The CPU restatement (oracle/, built with the reference's own flags
    -O3 -flto -ffast-math), single thread, on a bounded sample of the same
    workload.  Reported beside the GPU number; never part of `value`."""
    from renderer_amd.binding import Pipeline
    lib = os.path.join(REPO, "oracle", "liboracle_fast.so")
    if not os.path.exists(lib):
        return None
    cpu = Pipeline(lib)
    scene = cpu.open_scene(SCENE)
    fb = cpu.framebuffer(WIDTH, HEIGHT)
    cpu.draw(scene, fb, cpu.orbit_view(0, ORBIT))           # warm-up, untimed
    t0 = time.perf_counter()
    done = 0
    for i in range(frames_wanted):
        cpu.draw(scene, fb, cpu.orbit_view(i * (ORBIT // frames_wanted), ORBIT))
        done += 1
        if time.perf_counter() - t0 > 30.0:
            break
    dt = time.perf_counter() - t0
    cpu.release(fb)
    cpu.close_scene(scene)
    return {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{done} orbit views of pbr helmet at {WIDTH}x{HEIGHT}, oracle/liboracle_fast.so "
                      f"(gcc -O3 -flto -ffast-math), {os.cpu_count()} host cores available"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=32, help="frames per step and rank")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from renderer_amd.binding import Pipeline, GPU_LIB

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    gpu = Pipeline(GPU_LIB)
    gpu.lib.mi355r_init(local)
    scene = gpu.open_scene(SCENE)
    info = gpu.scene_info(scene)
    faces = info["faces"] + (12 if info["skybox"] else 0)
    B = args.frames
    fbs = [gpu.framebuffer(WIDTH, HEIGHT) for _ in range(B)]
    # per-rank slice of the orbit: rank r renders views r*B .. r*B+B-1 (mod ORBIT)
    views = [gpu.orbit_view((rank * B + i) % ORBIT, ORBIT) for i in range(B)]
    frame_bytes = WIDTH * HEIGHT * 4
    out = torch.empty((B, HEIGHT, WIDTH, 4), dtype=torch.uint8, device="cuda") if world > 1 else None
    gathered = [torch.empty_like(out) for _ in range(world)] if (world > 1 and rank == 0) else None

    def step():
        for i in range(B):
            gpu.draw(scene, fbs[i], views[i])
        gpu.lib.graphics_flush()
        if world > 1:
            for i in range(B):
                gpu.lib.framebuffer_read_device(fbs[i], C.c_void_p(out.data_ptr() + i * frame_bytes), None)
            dist.gather(out, gathered, dst=0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        total_frames = args.steps * B * world
        fps = total_frames / elapsed

        # workload counters (untimed pass with device-side counting on)
        gpu.lib.mi355r_reset_counters()
        gpu.lib.mi355r_enable_counters(1)
        step() if world == 1 else [gpu.draw(scene, fbs[i], views[i]) for i in range(B)]
        counters = gpu.counters()
        gpu.lib.mi355r_enable_counters(0)
        shaded_per_frame = counters[2] / B

        # dominant kernel (raster + shade): HIP events on the pipeline's stream
        gpu.lib.mi355r_enable_timing(1)
        for _ in range(max(1, min(args.steps, 5))):
            for i in range(B):
                gpu.draw(scene, fbs[i], views[i])
            gpu.lib.graphics_flush()
        passes, ms = gpu.kernel_times()
        gpu.lib.mi355r_enable_timing(0)
        names = ["vertex", "setup", "bin", "raster", "shade"]
        per_frame_ms = {n: ms[i] / passes for i, n in enumerate(names)}
        # the frame's algorithmic bytes, attributed to the kernel that moves them:
        # vertex stream -> vertex kernel; depth plane -> raster; colour plane + texels -> shade
        # (when a pass shades inside the raster kernel, raster carries the shade share too)
        frame_bytes_total = algorithmic_bytes(counters, B, faces) / B
        share = {"vertex": 240.0 * faces, "raster": 4.0 * WIDTH * HEIGHT, "setup": 0.0, "bin": 0.0}
        share["shade"] = frame_bytes_total - share["vertex"] - share["raster"]
        if per_frame_ms["shade"] < 1e-6:
            share["raster"] += share["shade"]
            share["shade"] = 0.0
        dominant = max(names, key=lambda n: per_frame_ms[n])
        launch_ms = per_frame_ms[dominant]
        bytes_per_launch = share[dominant]
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        pipeline_ms = sum(per_frame_ms.values())
        frame_achieved = frame_bytes_total / (pipeline_ms * 1e-3) / 1e9

        result = {
            "metric": "fps @1080p 'pbr helmet' (orbit batch)", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "reference scene data (assets/helmet, assets/workshop)",
            "config": {"workload": f"pbr helmet + blurred skybox, {WIDTH}x{HEIGHT}, {B} orbit views per step and GPU "
                                   f"(eye 1.5*(sin a, 0.3, cos a), {ORBIT}-view circle)",
                       "frames_per_step": B * world, "triangles_per_frame": faces,
                       "gather": "rccl gather of RGBA8 frames to rank 0" if world > 1 else "none"},
            "ms_per_frame": elapsed / total_frames * 1e3,
            "shaded_fragments_per_frame": shaded_per_frame,
            "shaded_mfrags_per_s": shaded_per_frame * fps / 1e6,
            "kernel_ms_per_frame": per_frame_ms,
            "roofline": {"bound": "hbm", "kernel": "rz_%s_kernel" % dominant, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "launch_ms": launch_ms},
            "frame_roofline": {"bound": "hbm", "achieved": frame_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": frame_achieved / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_frame": frame_bytes_total,
                               "kernel_ms_per_frame_total": pipeline_ms},
        }
        if world == 1 and not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(48)

    for fb in fbs:
        gpu.release(fb)
    gpu.close_scene(scene)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
