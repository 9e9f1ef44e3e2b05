#!/usr/bin/env python3
"""Benchmark of the hot path through the C ABI of renderer_amd/libmi355r.so.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

N = 1 (the headline): 'pbr helmet' (metallic-roughness + IBL + blurred skybox) at 1920x1080.  One step =
one batch of 128 orbit views (camera on a circle of radius 1.5 around the model, animation time index/30 s),
every frame a full pass of the pipeline (vertex -> setup/clip -> bin -> raster -> shade) into its own
framebuffer; scene data (meshes, textures, IBL maps) is resident in HBM before the timed region and the
frames stay there (`value`); `value_with_copy_out` is the same loop with every frame also copied out as RGB8
into a device buffer, which is what every rank of an N > 1 run does before the gather.

N > 1 (the batch-render job): 8192 camera views of the animated 'pbr buster' at 1920x1080, frame-parallel.
A step is 128 x N consecutive views of the 8192-view orbit (view i: angle 2 pi i / 8192, animation time
i / 30 s; the next step goes on where this one ended), every rank renders its own block of them (frames are
independent: weak scaling, no collective while rendering) and the finished RGB8 frames are gathered on rank 0
over RCCL inside the timed region, through a ring of two slots of 128 frames per rank, so that a step's
transfer overlaps the next step's rendering.  `--scene helmet` runs the N = 1 workload on N GPUs instead.
The one-GPU figure to compare the N > 1 values with is `other_configs.C5_orbit_8192.fps_with_copy_out` of the
N = 1 line (same scene, same views, same copy-out, no link to cross).

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
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E
GATHER_CHANNELS = 3         # N > 1: frames travel as RGB8 (a draw never writes the alpha byte, graphics.c:466-468)
JOB_VIEWS = 8192            # views of the batch-render job (N > 1, and other_configs.C5_orbit_8192)
KERNEL_SYMBOL = {"vertex": "rz_vertex_kernel", "setup": "rz_clip_kernel", "bin": "rz_bin_fill_kernel",
                 "raster": "rz_strip_kernel<2>", "shade_skybox": "rz_sky_kernel<2>",
                 "shade_blinn": "rz_shade_list_kernel<0, true, true>", "shade_pbr": "rz_shade_list_kernel<1, true, true>"}


def reference_frame_bytes(counters, frames, faces, width, height, family, shadow_side):
    """Algorithmic HBM bytes of one frame as the reference's work defines them: 240 B per input triangle
    (3 x 80-byte vertices), 8 B per pixel (RGBA8 + f32 depth written once), per fragment the reference
    shades (its draw_fragment calls) the texels its shader samples at the reference's precision (16-byte
    float4) -- 9 for a pbr fragment (6 material maps, 2 cube maps, BRDF LUT), 1 for a blinn fragment (the
    scenes here bind a diffuse map only), 1 for a skybox fragment, 1 more for the shadow map where there is
    one -- and 8 B per shadow-map texel (4 B depth written, 4 B read back)."""
    shaded, skybox = counters[2] / frames, counters[6] / frames
    per_fragment = (144 if family == "pbr" else 16) + (16 if shadow_side else 0)
    return 240 * faces + 8 * width * height + per_fragment * (shaded - skybox) + 16 * skybox + 8 * shadow_side * shadow_side


def kernel_frame_bytes(counters, frames, faces, width, height, family, shadow_side):
    """The same figures attributed to the kernel that moves them, per frame, with the fragments each
    shade kernel really shades (only fragments that stay visible): vertex stream -> vertex kernel; depth
    plane + the plane of fragment references it hands to the shade kernels (4 B per pixel each, the shadow
    map's depth plane with them) -> raster; texels + the colour bytes of its pixels -> shade kernels."""
    shaded, skybox = counters[8] / frames, counters[9] / frames
    per_fragment = (144 if family == "pbr" else 16) + (16 if shadow_side else 0) + 4
    model = per_fragment * (shaded - skybox)
    return {"vertex": 240.0 * faces * (2 if shadow_side else 1), "setup": 0.0, "bin": 0.0,
            "raster": 8.0 * width * height + 4.0 * shadow_side * shadow_side,
            "shade_skybox": (16 + 4) * skybox, "shade_blinn": model if family == "blinn" else 0.0,
            "shade_pbr": model if family == "pbr" else 0.0}


def cpu_baseline(frames_wanted, scene_name, orbit_views):
    """The CPU restatement (oracle/, built with the reference's own flags -O3 -flto -ffast-math),
    single thread, on a bounded sample of the same workload.  Reported beside the GPU number;
    never part of `value`."""
    from renderer_amd.binding import Pipeline
    lib = os.path.join(REPO, "oracle", "liboracle_fast.so")
    if not os.path.exists(lib):
        return None
    cpu = Pipeline(lib)
    scene = cpu.open_scene(scene_name)
    fb = cpu.framebuffer(WIDTH, HEIGHT)
    cpu.draw(scene, fb, cpu.orbit_view(0, orbit_views))           # warm-up, untimed
    t0 = time.perf_counter()
    done = 0
    for i in range(frames_wanted):
        cpu.draw(scene, fb, cpu.orbit_view(i * (orbit_views // frames_wanted), orbit_views))
        done += 1
        if time.perf_counter() - t0 > 30.0:
            break
    dt = time.perf_counter() - t0
    cpu.release(fb)
    cpu.close_scene(scene)
    return {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{done} orbit views of pbr {scene_name} at {WIDTH}x{HEIGHT}, oracle/liboracle_fast.so "
                      f"(gcc -O3 -flto -ffast-math), {os.cpu_count()} host cores available"}


def committed_counters(label, symbol):
    """The figures of kernel `symbol` from the committed PMC passes of workload `label`
    (tools/profile.sh -> profiles/traffic_<label>.json); {} when there are none."""
    path = os.path.join(REPO, "profiles", f"traffic_{label}.json")
    if not os.path.exists(path):
        return {}
    return json.load(open(path)).get("kernels", {}).get(symbol, {})


RASTER_KERNELS = ("rz_strip_kernel", "rz_strip_team_kernel", "rz_raster_list_kernel", "rz_blend_visibility_kernel")


def raster_symbol(label, last_launched):
    """The kernel that stands for the timed stage "raster" of workload `label`: of the rasterising kernels in the
    committed profile the one that took the most time there in all (a step may run several: strips for the opaque
    passes, the ordered kernel for the blended ones); without a profile the one the library launched last."""
    path = os.path.join(REPO, "profiles", f"traffic_{label}.json")
    if os.path.exists(path):
        kernels = json.load(open(path)).get("kernels", {})
        mine = {k: v.get("steady_avg_us", 0.0) * v.get("launches", 0) for k, v in kernels.items() if k.startswith(RASTER_KERNELS)}
        mine = {k: t for k, t in mine.items() if t == t and t > 0}
        if mine:
            return max(mine, key=lambda k: mine[k])
    return last_launched


def roofline_of(label, ms, frames, sequences, share, symbols=KERNEL_SYMBOL):
    """The roofline object of the dominant kernel (by measured time).  ms: summed milliseconds per kernel over
    `frames` frames in `sequences` launch sequences (HIP events around every launch); share: algorithmic bytes
    per frame attributed to each kernel.  `achieved` / `frac` follow the algorithmic bytes (reference precision)
    over the live kernel time.  Beside them what the hardware counters of the committed profile of this
    workload say (profiles/traffic_<label>.json, each fraction over the kernel's own duration in that profile):
    HBM traffic as measured (FETCH_SIZE + WRITE_SIZE; the x2 the gfx950 guide prescribes for wide coalesced
    reads is NOT applied here -- the shade kernels' reads are 4-byte gathers -- `traffic_frac_x2` carries the
    doubled figure as an upper bound), the share of the vector ALU's issue slots used, and the share of cycles
    the texture data path was busy.  `bound` names the highest of those three: the roof the kernel is actually
    under; without committed counters it stays "hbm"."""
    dominant = max(ms, key=lambda k: ms[k])
    launch_ms = ms[dominant] / sequences
    bytes_per_launch = share[dominant] * frames / sequences
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    pmc = committed_counters(label, symbols[dominant])
    fracs = {"hbm": pmc.get("traffic_frac"), "valu": pmc.get("valu_issue_frac"), "td": pmc.get("td_busy_frac")}
    known = {k: v for k, v in fracs.items() if v is not None}
    bound = max(known, key=lambda k: known[k]) if known else "hbm"
    return dominant, {
        "bound": bound, "kernel": symbols[dominant], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": (pmc["fetch_size_bytes_raw"] + pmc["write_size_bytes"]) if "fetch_size_bytes_raw" in pmc else None,
        "bound_frac": known.get(bound), "algorithmic_frac": achieved / HBM_PEAK_GBS, "traffic_frac": fracs["hbm"],
        "traffic_frac_x2": pmc.get("traffic_frac_x2"),
        "valu_issue_frac": fracs["valu"], "salu_issue_frac": pmc.get("salu_issue_frac"), "td_busy_frac": fracs["td"],
        "counters_from": f"profiles/traffic_{label}.json" if pmc else None,
        "algorithmic_bytes_per_launch": bytes_per_launch, "launch_ms": launch_ms, "frames_per_launch": frames / sequences}


def measure_kernels(gpu, orbit, scene, fbs, views, repeats):
    """(workload counters of one untimed step with device-side counting on, passes, launch sequences,
    summed kernel milliseconds) for the views as they are submitted in the timed loop."""
    gpu.lib.mi355r_reset_counters()
    gpu.lib.mi355r_enable_counters(1)
    orbit.render_views(gpu, scene, fbs, views)
    counters = gpu.counters()
    gpu.lib.mi355r_enable_counters(0)
    # per-kernel device time: HIP events on the pipeline's own stream around every launch
    gpu.lib.mi355r_enable_timing(1)
    for _ in range(repeats):
        orbit.render_views(gpu, scene, fbs, views)
    passes, sequences, ms = gpu.kernel_times()
    gpu.lib.mi355r_enable_timing(0)
    return counters, passes, sequences, ms


def calibrate_split(gpu, orbit, scene, B, world, rank, device, dist, torch, views):
    """Frames per peer and step for the gather-aware split, from the rate at which rank 0 renders frames and
    copies them out and the rate at which the peers' frames arrive when all of them send at once (both
    measured here, on B frames)."""
    fbs = [gpu.framebuffer(WIDTH, HEIGHT) for _ in range(B)]
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
    return peer, {"render_ms_per_frame": render_s / B * 1e3, "delivery_ms_per_frame": gather_s / B * 1e3,
                  # what bounds the job whatever the split: rank 0 takes in (world - 1) * B frames per gather_s over
                  # its links while it renders B frames per render_s itself
                  "gather_ceiling_fps": B / render_s + (world - 1) * B / gather_s}


class CopyOut:
    """What a rank of an N > 1 run does with a step's frames apart from the collective: the RGB8 copies
    into one of two device buffers are launched behind the step's rendering and waited for when the next
    step has been submitted (framebuffers_read_device_rgb_begin / _end)."""

    def __init__(self, torch, frames):
        self.buffers = [torch.empty((frames, HEIGHT, WIDTH, GATHER_CHANNELS), dtype=torch.uint8, device="cuda") for _ in range(2)]
        self.count = 0

    def submit(self, gpu, fbs):
        target = self.buffers[self.count % 2]
        self.count += 1
        handles = (type(fbs[0]) * len(fbs))(*fbs)
        gpu.lib.framebuffers_read_device_rgb_end()
        gpu.lib.framebuffers_read_device_rgb_begin(handles, len(fbs), C.c_void_p(target.data_ptr()), target[0].numel())

    def drain(self, gpu):
        gpu.lib.framebuffers_read_device_rgb_end()


def interactive_rates(gpu, scene_name, frames=240):
    """The reference viewer's loop (test_helper.c:171-255: draw, present, repeat) on this library: ONE frame
    per flush, nothing to batch.  Frames per second to a frame finished on the device, and with its colour
    buffer on the host (what a blit needs)."""
    scene = gpu.open_scene(scene_name)
    fb = gpu.framebuffer(WIDTH, HEIGHT)
    gpu.lib.framebuffer_sync_color.argtypes = gpu.lib.framebuffer_sync.argtypes
    views = [gpu.orbit_view(i, frames) for i in range(frames)]
    for v in views[:20]:
        gpu.draw(scene, fb, v)
        gpu.lib.framebuffer_sync(fb)
    out = {"workload": f"pbr {scene_name} {WIDTH}x{HEIGHT}, one frame per flush into one framebuffer, {frames} frames"}
    for key, sync in (("fps_device_finished", lambda: gpu.lib.framebuffer_read_device(fb, None, None)),
                      ("fps_colour_on_host", lambda: gpu.lib.framebuffer_sync_color(fb))):
        t0 = time.perf_counter()
        for v in views:
            gpu.draw(scene, fb, v)
            sync()
        dt = time.perf_counter() - t0
        out[key] = frames / dt
        out[key.replace("fps", "ms_per_frame")] = dt / frames * 1e3
    gpu.release(fb)
    gpu.close_scene(scene)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=128, help="frames per step and rank (default: 128)")
    ap.add_argument("--scene", default=None, help="helmet (default for N = 1) or buster (default for N > 1: the 8192-view batch-render job)")
    ap.add_argument("--views", type=int, default=None, help="views of the whole orbit (default: 128 for helmet, 8192 for buster)")
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

    scene_name = args.scene or ("helmet" if world == 1 else "buster")
    orbit_views = args.views or (JOB_VIEWS if scene_name == "buster" else 128)
    gpu = Pipeline(GPU_LIB)
    gpu.lib.mi355r_init(local)
    scene = gpu.open_scene(scene_name)
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
        peer_frames, split = calibrate_split(gpu, orbit, scene, B, world, rank, device, dist, torch,
                                             [gpu.orbit_view(i % orbit_views, orbit_views) for i in range(B)])
    first, last = orbit.shard_views_rooted(B * world, rank, world, peer_frames)
    mine = last - first
    fbs = [gpu.framebuffer(WIDTH, HEIGHT) for _ in range(mine)]
    # step s of the job (warm-up steps included) shows views [s * B * world, (s + 1) * B * world) of the orbit,
    # this rank its block of them; with orbit_views == B * world every step shows the same circle
    step_views = {}

    def views_of(s):
        if s not in step_views:
            step_views[s] = [gpu.orbit_view((s * B * world + i) % orbit_views, orbit_views) for i in range(first, last)]
        return step_views[s]
    gather = orbit.FrameGather(peer_frames, HEIGHT, WIDTH, device, rank, world, root_frames=B * world - peer_frames * (world - 1),
                               channels=GATHER_CHANNELS)
    copy_out = CopyOut(torch, mine) if world == 1 else None
    counter = [0]

    def step(with_gather=True, with_copy_out=False):
        orbit.render_views(gpu, scene, fbs, views_of(counter[0]))
        counter[0] += 1
        if with_gather:
            gather.submit(gpu, fbs, mine)   # N > 1: this step's transfer overlaps the next step's rendering
        if with_copy_out:
            copy_out.submit(gpu, fbs)

    def fence():
        gather.drain()                      # every frame submitted so far has arrived on rank 0
        if copy_out:
            copy_out.drain(gpu)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(steps, **how):
        """`steps` steps between two fences; the slowest rank's time."""
        for s in range(counter[0], counter[0] + steps):
            views_of(s)                     # the views' parameters are made outside the timed region
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(**how)
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
    # for N = 1 also with the RGB8 copy-out every rank of an N > 1 run pays
    copied_out = timed(args.steps, with_copy_out=True) if world == 1 else None

    result = None
    if rank == 0:
        total_frames = args.steps * B * world
        fps = total_frames / elapsed
        views = views_of(0)
        counters, passes, sequences, ms = measure_kernels(gpu, orbit, scene, fbs, views, max(1, min(args.steps, 5)))
        shaded_per_frame = counters[2] / mine        # (rank 0's own frames: `mine` of them)
        timed_frames = mine * max(1, min(args.steps, 5))
        per_frame_ms = {k: v / timed_frames for k, v in ms.items()}
        shadow_side = info["shadow_w"]
        share = kernel_frame_bytes(counters, mine, faces, WIDTH, HEIGHT, "pbr", shadow_side)
        label = "C2" if scene_name == "helmet" else "C5"
        symbols = dict(KERNEL_SYMBOL, raster=raster_symbol(label, gpu.lib.mi355r_last_raster_kernel().decode()))
        dominant, roofline = roofline_of(label, ms, timed_frames, sequences, share, symbols)
        pipeline_ms = sum(per_frame_ms.values())
        frame_bytes = reference_frame_bytes(counters, mine, faces, WIDTH, HEIGHT, "pbr", shadow_side)
        frame_achieved = frame_bytes / (pipeline_ms * 1e-3) / 1e9

        what = (f"pbr helmet + blurred skybox, {WIDTH}x{HEIGHT}, {B} orbit views per step and GPU "
                f"(eye 1.5*(sin a, 0.3, cos a), {orbit_views}-view circle)") if scene_name == "helmet" else \
               (f"batch render: {orbit_views} views of animated pbr {scene_name}, {WIDTH}x{HEIGHT}, {B} consecutive views per step and GPU "
                f"(view i: eye 1.5*(sin a, 0.3, cos a), a = 2 pi i / {orbit_views}, animation time i / 30 s)")
        result = {
            "metric": f"fps @1080p 'pbr {scene_name}' (orbit batch)", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": f"reference scene data (assets/{scene_name})",
            "config": {"workload": what,
                       "frames_per_step": B * world, "triangles_per_frame": faces,
                       "value_includes": ("rendering, RGB8 copy-out and the gather of every frame on rank 0 "
                                          "(one GPU, like for like: other_configs.C5_orbit_8192.fps_with_copy_out of the N = 1 line)")
                                         if world > 1 else "rendering; the frames stay in HBM (value_with_copy_out: plus the RGB8 copy-out an N > 1 rank pays)",
                       "frame_ring": f"2 slots of {peer_frames} RGB8 frames per rank" if world > 1 else None,
                       "gather": ("%s gather of RGB8 frames to rank 0, overlapped with the next step's rendering"
                                  % ("rccl" if on_device else args.backend)) if world > 1 else "none",
                       "split": ({"frames_on_rank_0": mine, "frames_per_peer": peer_frames, "calibration": split}
                                 if world > 1 else None)},
            "ms_per_frame": elapsed / total_frames * 1e3,
            "value_with_copy_out": (total_frames / copied_out) if copied_out else None,
            "render_only_fps": (total_frames / render_only) if render_only else None,
            "gather_ceiling_fps": split["gather_ceiling_fps"] if split else None,
            "gathered_gb_per_s": (args.steps * peer_frames * (world - 1) * WIDTH * HEIGHT * GATHER_CHANNELS / elapsed / 1e9) if world > 1 else None,
            "shaded_fragments_per_frame": shaded_per_frame,
            "shaded_mfrags_per_s": shaded_per_frame * fps / 1e6,
            "fragments_shaded_on_device_per_frame": counters[8] / mine,
            "kernel_ms_per_frame": per_frame_ms,
            "roofline": roofline,
            "frame_roofline": {"bound": "hbm", "achieved": frame_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": frame_achieved / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_frame": frame_bytes,
                               "kernel_ms_per_frame_total": pipeline_ms},
        }
        if world > 1:
            result["config"]["ranks_seen"] = dist.get_world_size()
            result["config"]["backend"] = dist.get_backend()
        if world == 1 and not args.no_other:
            result["other_configs"] = other_configs(gpu, orbit, torch, fbs)
            fbs = []
        if world == 1 and not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(48, scene_name, orbit_views)

    for fb in fbs:
        gpu.release(fb)
    gpu.close_scene(scene)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def other_configs(gpu, orbit, torch, fbs):
    """The other BASELINE configurations, a few steps each, outside the timed region of `value`: frames/s,
    per-kernel time and a roofline object each; the 8192-view batch-render job on this one GPU; the
    frame-at-a-time loop; the reference's own per-face submission."""
    from renderer_amd.configs import CONFIGS, time_config
    for fb in fbs:
        gpu.release(fb)
    others = {}
    for config in CONFIGS:
        if config[0].startswith("C2"):
            continue
        r = time_config(gpu, config, steps=5, warmup=2, measure=True)
        label = config[0].split()[0]
        m = r.pop("measured")
        family = "pbr" if label in ("C4", "C5") else "blinn"
        share = kernel_frame_bytes(m["counters"], r["frames_per_step"], r["triangles_per_frame"], r["width"], r["height"], family, m["shadow_side"])
        symbols = dict(KERNEL_SYMBOL, raster=raster_symbol(label, m["raster_symbol"]))
        _, r["roofline"] = roofline_of(label, m["kernel_ms"], m["frames"], m["sequences"], share, symbols)
        r["kernel_ms_per_frame"] = {k: v / m["frames"] for k, v in m["kernel_ms"].items()}
        r["passes_per_frame"] = m["passes"] / m["frames"]
        frame_bytes = reference_frame_bytes(m["counters"], r["frames_per_step"], r["triangles_per_frame"], r["width"], r["height"], family, m["shadow_side"])
        r["frame_roofline"] = {"algorithmic_bytes_per_frame": frame_bytes, "achieved": frame_bytes * r["fps"] / 1e9, "unit": "GB/s",
                               "peak": HBM_PEAK_GBS, "frac": frame_bytes * r["fps"] / 1e9 / HBM_PEAK_GBS,
                               "note": "the reference-defined bytes of a frame x frames/s of the whole step (host included)"}
        others[label] = r
    others["C5_orbit_8192"] = batch_job(gpu, orbit, torch)
    others["C2_interactive"] = interactive_rates(gpu, "helmet")
    # the compatible slow path: the reference's own per-face loops (one graphics_draw_triangle per face)
    for config in CONFIGS:
        if config[0].startswith(("C2", "C3")):
            r = time_config(gpu, config, steps=2, warmup=1, frames=32, per_triangle=True)
            others[config[0].split()[0] + "_per_triangle"] = r
    # ... and the reference's OWN host (its scene factories, draw_model loops and frame driver) as a child process.  The
    # binary is built from the reference's host sources in the build container (oracle/Makefile, target refhost) and
    # linked against this library; it is not tracked, and where it did not travel with the tree the leg says so
    host = os.path.join(REPO, "oracle", "_ref", "refhost_mi355r")
    for label, scene_name in (("C2_reference_host", "pbr:helmet"), ("C3_reference_host", "blinn:kgirl")):
        if not os.path.exists(host):
            others[label] = {"skipped": "oracle/_ref/refhost_mi355r is not in this tree (it is built only where the reference's sources are)",
                             "built_from": "reference host sources"}
            continue
        import subprocess
        try:
            out = subprocess.run([host, "--scene", scene_name, "--assets", os.path.join(REPO, "assets"), "--size", str(WIDTH), str(HEIGHT),
                                  "--bench", "100"], capture_output=True, text=True, timeout=120, check=True).stdout
            r = json.loads(out.strip().splitlines()[-1])
            others[label] = {"workload": f"{scene_name} {WIDTH}x{HEIGHT}, one frame after the other into one framebuffer", "fps": 1e3 / r["median_ms"],
                             "ms_per_step": r["median_ms"], "frames_per_step": 1, "steps": r["frames"], "built_from": "reference host sources",
                             "submission": "the reference's own host: per face (graphics_draw_triangle), median frame time of its loop"}
        except Exception as e:      # the figure is an extra: the bench line does not depend on it
            others[label] = {"error": str(e)[:200], "built_from": "reference host sources"}
    return others


def batch_job(gpu, orbit, torch, frames=128):
    """BASELINE configs[4] on ONE GPU: all 8192 views of the animated buster orbit, 128 consecutive views per
    step (64 steps, every view with its own animation time), rendered only and with every frame copied out as
    RGB8 through the two-slot ring -- the figure the N > 1 values of this bench compare with."""
    scene = gpu.open_scene("buster")
    fbs = [gpu.framebuffer(WIDTH, HEIGHT) for _ in range(frames)]
    steps = JOB_VIEWS // frames
    views = [[gpu.orbit_view(s * frames + i, JOB_VIEWS) for i in range(frames)] for s in range(steps)]
    copy_out = CopyOut(torch, frames)
    out = {"workload": f"batch render: {JOB_VIEWS} views of animated pbr buster, {WIDTH}x{HEIGHT}, {frames} consecutive views per step, {steps} steps",
           "frames_per_step": frames, "steps": steps}
    for key, with_copy in (("fps", False), ("fps_with_copy_out", True)):
        for v in views[:2]:
            orbit.render_views(gpu, scene, fbs, v)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for v in views:
            orbit.render_views(gpu, scene, fbs, v)
            if with_copy:
                copy_out.submit(gpu, fbs)
        copy_out.drain(gpu)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[key] = JOB_VIEWS / dt
        out[key.replace("fps", "seconds")] = dt
    for fb in fbs:
        gpu.release(fb)
    gpu.close_scene(scene)
    return out


if __name__ == "__main__":
    main()
