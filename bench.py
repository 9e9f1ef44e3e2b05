#!/usr/bin/env python3
"""Benchmark of the hot path: 'pbr helmet' (metallic-roughness + IBL + blurred
skybox) at 1920x1080 on MI355X, rendered through the C ABI of
renderer_amd/libmi355r.so.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one batch of FRAMES orbit views of the scene (camera on a circle of
radius 1.5 around the model, animation time index/30 s), every frame a full
pass of the pipeline (vertex -> setup/clip -> bin -> raster -> shade) into its
own framebuffer; scene data (meshes, textures, IBL maps) is resident in HBM
before the timed region.  With N > 1 ranks every rank renders its own block of
the orbit (frames are independent: weak scaling, no collective while
rendering) and the finished RGBA8 frames are gathered on rank 0 over RCCL
inside the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WIDTH, HEIGHT = 1920, 1080
SCENE = "helmet"
ORBIT = 128                 # views on the orbit circle
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E
GATHER_CHANNELS = 3         # N > 1: frames travel as RGB8 (a draw never writes the alpha byte, graphics.c:466-468)
KERNEL_SYMBOL = {"vertex": "rz_vertex_kernel", "setup": "rz_clip_kernel", "bin": "rz_bin_fill_kernel",
                 "raster": "rz_strip_kernel<4, false>", "shade_skybox": "rz_sky_kernel<2>",
                 "shade_blinn": "rz_shade_list_kernel<0, true, true>", "shade_pbr": "rz_shade_list_kernel<1, true, true>"}


def reference_frame_bytes(counters, frames, faces):
    """Algorithmic HBM bytes of one frame as the reference's work defines them: 240 B per input
    triangle (3 x 80-byte vertices), 8 B per pixel (RGBA8 + f32 depth written once), and per
    fragment the reference shades (its draw_fragment calls) the texels its shader samples at the
    reference's precision (16-byte float4): 9 for a pbr fragment (6 material maps, 2 cube maps,
    BRDF LUT), 1 for a skybox fragment."""
    shaded, skybox = counters[2] / frames, counters[6] / frames
    return 240 * faces + 8 * WIDTH * HEIGHT + 144 * (shaded - skybox) + 16 * skybox


def kernel_frame_bytes(counters, frames, faces):
    """The same figures attributed to the kernel that moves them, per frame, with the fragments
    each shade kernel really shades (only fragments that stay visible): vertex stream -> vertex
    kernel; depth plane + the plane of fragment references it hands to the shade kernels (4 B per
    pixel each) -> raster; texels + the colour bytes of its pixels -> shade kernels."""
    shaded, skybox = counters[8] / frames, counters[9] / frames
    return {"vertex": 240.0 * faces, "setup": 0.0, "bin": 0.0, "raster": 8.0 * WIDTH * HEIGHT,
            "shade_skybox": (16 + 4) * skybox, "shade_blinn": 0.0, "shade_pbr": (144 + 4) * (shaded - skybox)}


def cpu_baseline(frames_wanted):
    """The CPU restatement (oracle/, built with the reference's own flags -O3 -flto -ffast-math),
    single thread, on a bounded sample of the same workload.  Reported beside the GPU number;
    never part of `value`."""
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


def measured_counters(symbol, frames_per_launch, key):
    """A per-launch figure of `symbol` from the committed PMC passes (tools/profile.sh ->
    profiles/traffic.json), if they were taken with the same number of frames per launch; else None."""
    path = os.path.join(REPO, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    if t.get("frames_per_launch") != frames_per_launch or symbol not in t.get("kernels", {}):
        return None
    return t["kernels"][symbol].get(key)


def calibrate_split(gpu, orbit, scene, B, world, rank, device, dist, torch):
    """Frames per peer and step for the gather-aware split, from the rate at which rank 0 renders frames and
    copies them out and the rate at which the peers' frames arrive when all of them send at once (both
    measured here, on B frames)."""
    fbs = [gpu.framebuffer(WIDTH, HEIGHT) for _ in range(B)]
    views = [gpu.orbit_view(i % ORBIT, ORBIT) for i in range(B)]
    trial = orbit.FrameGather(B, HEIGHT, WIDTH, device, rank, world, depth=1, channels=GATHER_CHANNELS)

    def fence():
        trial.drain()
        dist.barrier()
        torch.cuda.synchronize()
    orbit.render_views(gpu, scene, fbs, views)
    trial.submit(gpu, fbs, B)                   # first use of the scene, the links and the buffers
    fence()
    # rank 0's own frames as the timed loop produces them: recorded, rendered, then copied into the buffer it keeps them in
    own = torch.empty((B, HEIGHT, WIDTH, GATHER_CHANNELS), dtype=torch.uint8, device=device)
    t0 = time.perf_counter()
    for _ in range(2):
        orbit.render_views(gpu, scene, fbs, views)
        orbit.frames_to_tensor(gpu, fbs, B, own)
    torch.cuda.synchronize()
    render_s = (time.perf_counter() - t0) / 2
    del own
    fence()
    t0 = time.perf_counter()
    trial.submit(gpu, fbs, B)
    fence()
    gather_s = time.perf_counter() - t0
    rates = torch.tensor([render_s, gather_s], dtype=torch.float64, device=device)
    dist.broadcast(rates, src=0)                # rank 0's clock decides for everybody
    render_s, gather_s = float(rates[0]), float(rates[1])
    for fb in fbs:
        gpu.release(fb)
    del trial
    # rank 0 renders R frames in R * render_s / B, a peer delivers P frames in P * gather_s / B: equal times
    if not (render_s > 0 and gather_s > 0):
        return B, None
    peer = orbit.split_for(render_s, gather_s, B, world)
    return peer, {"render_ms_per_frame": render_s / B * 1e3, "delivery_ms_per_frame": gather_s / B * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=128, help="frames per step and rank (default: one 128-view orbit)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (gloo: CPU rehearsal of the gather)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-other", action="store_true", help="skip the other BASELINE configurations (other_configs)")
    ap.add_argument("--even", action="store_true", help="N > 1: every rank renders the same number of frames (no gather-aware split)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from renderer_amd import orbit
    from renderer_amd.binding import Pipeline, GPU_LIB

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    local = local % torch.cuda.device_count()       # a rehearsal may put several ranks on one GPU
    torch.cuda.set_device(local)
    on_device = args.backend == "nccl"
    if world > 1:
        if on_device:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    gpu = Pipeline(GPU_LIB)
    gpu.lib.mi355r_init(local)
    scene = gpu.open_scene(SCENE)
    info = gpu.scene_info(scene)
    faces = info["faces"] + (12 if info["skybox"] else 0)
    B = args.frames
    device = "cuda" if on_device else "cpu"
    # weak scaling: a step is B * world frames.  N > 1: a peer's frames have to cross its xGMI link to
    # rank 0, rank 0's own do not, and a GPU renders several times as many frames as a link carries --
    # so the split is gather-aware: every peer renders `peer_frames` of the step's views, rank 0 the rest,
    # sized from two measurements taken here so that rendering on rank 0 and delivery from the peers take
    # the same time (peer_frames == B is the even split)
    peer_frames, split = B, None
    if world > 1 and not args.even:
        peer_frames, split = calibrate_split(gpu, orbit, scene, B, world, rank, device, dist, torch)
    first, last = orbit.shard_views_rooted(B * world, rank, world, peer_frames)
    mine = last - first
    fbs = [gpu.framebuffer(WIDTH, HEIGHT) for _ in range(mine)]
    views = [gpu.orbit_view(i % ORBIT, ORBIT) for i in range(first, last)]
    gather = orbit.FrameGather(peer_frames, HEIGHT, WIDTH, device, rank, world, root_frames=B * world - peer_frames * (world - 1),
                               channels=GATHER_CHANNELS)

    def step(with_gather=True):
        orbit.render_views(gpu, scene, fbs, views)
        if with_gather:
            gather.submit(gpu, fbs, mine)   # N > 1: this step's transfer overlaps the next step's rendering

    def fence():
        gather.drain()                      # every frame submitted so far has arrived on rank 0
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(steps, with_gather=True):
        """`steps` steps between two fences; the slowest rank's time."""
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(with_gather)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_device else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        step()
    elapsed = timed(args.steps)
    # for N > 1 also the rendering alone, outside the reported value: it separates what the GPUs
    # render from what rank 0 can take in over its xGMI links
    render_only = timed(args.steps, with_gather=False) if world > 1 else None

    result = None
    if rank == 0:
        total_frames = args.steps * B * world
        fps = total_frames / elapsed

        # workload counters (untimed pass with device-side counting on)
        gpu.lib.mi355r_reset_counters()
        gpu.lib.mi355r_enable_counters(1)
        orbit.render_views(gpu, scene, fbs, views)
        counters = gpu.counters()
        gpu.lib.mi355r_enable_counters(0)
        shaded_per_frame = counters[2] / mine        # (rank 0's own frames: `mine` of them)

        # per-kernel device time: HIP events on the pipeline's own stream around every launch
        gpu.lib.mi355r_enable_timing(1)
        for _ in range(max(1, min(args.steps, 5))):
            orbit.render_views(gpu, scene, fbs, views)
        passes, sequences, ms = gpu.kernel_times()
        gpu.lib.mi355r_enable_timing(0)
        frames_per_launch = passes / sequences
        per_frame_ms = {k: v / passes for k, v in ms.items()}
        share = kernel_frame_bytes(counters, mine, faces)
        dominant = max(per_frame_ms, key=lambda k: per_frame_ms[k])
        launch_ms = per_frame_ms[dominant] * frames_per_launch
        bytes_per_launch = share[dominant] * frames_per_launch
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        pipeline_ms = sum(per_frame_ms.values())
        frame_bytes = reference_frame_bytes(counters, mine, faces)
        frame_achieved = frame_bytes / (pipeline_ms * 1e-3) / 1e9

        result = {
            "metric": "fps @1080p 'pbr helmet' (orbit batch)", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "reference scene data (assets/helmet, assets/workshop)",
            "config": {"workload": f"pbr helmet + blurred skybox, {WIDTH}x{HEIGHT}, {B} orbit views per step and GPU "
                                   f"(eye 1.5*(sin a, 0.3, cos a), {ORBIT}-view circle)",
                       "frames_per_step": B * world, "triangles_per_frame": faces,
                       "gather": ("%s gather of RGB8 frames to rank 0, overlapped with the next step's rendering"
                                  % ("rccl" if on_device else args.backend)) if world > 1 else "none",
                       "split": ({"frames_on_rank_0": mine, "frames_per_peer": peer_frames, "calibration": split}
                                 if world > 1 else None)},
            "ms_per_frame": elapsed / total_frames * 1e3,
            "render_only_fps": (total_frames / render_only) if render_only else None,
            "gathered_gb_per_s": (args.steps * peer_frames * (world - 1) * WIDTH * HEIGHT * GATHER_CHANNELS / elapsed / 1e9) if world > 1 else None,
            "shaded_fragments_per_frame": shaded_per_frame,
            "shaded_mfrags_per_s": shaded_per_frame * fps / 1e6,
            "fragments_shaded_on_device_per_frame": counters[8] / mine,
            "kernel_ms_per_frame": per_frame_ms,
            "roofline": {"bound": "hbm", "kernel": KERNEL_SYMBOL[dominant], "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_counters(KERNEL_SYMBOL[dominant], frames_per_launch, "hbm_bytes_per_launch"),
                         # what limits it: the kernel is far from the HBM roof; its waves spend their time issuing
                         # instructions (vector and scalar in about equal numbers) and waiting for dependent loads
                         "limiter": "instruction issue + load latency, not HBM",
                         "valu_issue_frac": measured_counters(KERNEL_SYMBOL[dominant], frames_per_launch, "valu_issue_frac"),
                         "salu_issue_frac": measured_counters(KERNEL_SYMBOL[dominant], frames_per_launch, "salu_issue_frac"),
                         "algorithmic_bytes_per_launch": bytes_per_launch, "launch_ms": launch_ms,
                         "frames_per_launch": frames_per_launch},
            "frame_roofline": {"bound": "hbm", "achieved": frame_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": frame_achieved / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_frame": frame_bytes,
                               "kernel_ms_per_frame_total": pipeline_ms},
        }
        if world > 1:
            result["config"]["ranks_seen"] = dist.get_world_size()
            result["config"]["backend"] = dist.get_backend()
        if world == 1 and not args.no_other:
            # the other BASELINE configurations, a few steps each, outside the timed region of `value`
            from renderer_amd.configs import CONFIGS, time_config
            for fb in fbs:
                gpu.release(fb)
            fbs = []
            others = {}
            for config in CONFIGS:
                if config[0].startswith("C2"):
                    continue
                r = time_config(gpu, config, steps=3, warmup=1)
                others[config[0].split()[0]] = r
            # the compatible slow path: the reference's own per-face loops (one graphics_draw_triangle per face)
            for config in CONFIGS:
                if config[0].startswith(("C2", "C3")):
                    r = time_config(gpu, config, steps=2, warmup=1, frames=32, per_triangle=True)
                    others[config[0].split()[0] + "_per_triangle"] = r
            # ... and the reference's OWN host (its scene factories, draw_model loops and frame driver, compiled from its
            # sources in the build container, linked against the library: oracle/Makefile, target refhost) as a child
            # process, when that binary travelled with the tree: frames per second of its frame loop at 1920x1080
            host = os.path.join(REPO, "oracle", "_ref", "refhost_mi355r")
            if os.path.exists(host):
                import subprocess
                for label, scene_name in (("C2_reference_host", "pbr:helmet"), ("C3_reference_host", "blinn:kgirl")):
                    try:
                        out = subprocess.run([host, "--scene", scene_name, "--assets", os.path.join(REPO, "assets"), "--size", str(WIDTH), str(HEIGHT),
                                              "--bench", "100"], capture_output=True, text=True, timeout=120, check=True).stdout
                        r = json.loads(out.strip().splitlines()[-1])
                        others[label] = {"workload": f"{scene_name} {WIDTH}x{HEIGHT}, one frame after the other into one framebuffer", "fps": 1e3 / r["median_ms"],
                                         "ms_per_step": r["median_ms"], "frames_per_step": 1, "steps": r["frames"],
                                         "submission": "the reference's own host: per face (graphics_draw_triangle), median frame time of its loop"}
                    except Exception as e:      # the figure is an extra: the bench line does not depend on it
                        others[label] = {"error": str(e)[:200]}
            result["other_configs"] = others
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
