"""Per-wave probe of the strip kernel (library built with -DRZ_PROBE, named by MI355R_LIB): tools/probe_strip.py <C1|C5|...> [frames per flush]"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from renderer_amd import orbit
from renderer_amd.binding import Pipeline, GPU_LIB
from renderer_amd.configs import CONFIGS
import torch
only = sys.argv[1]
gpu = Pipeline(GPU_LIB)
for label, name, like, (w, h), _ in CONFIGS:
    if not label.startswith(only): continue
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    scene = gpu.open_scene(name, like)
    fbs = [gpu.framebuffer(w, h) for _ in range(frames)]
    views = [gpu.orbit_view(i, 240) for i in range(frames)]
    for _ in range(3):
        orbit.render_views(gpu, scene, fbs, views)
    torch.cuda.synchronize()
    W = 12
    buf = np.zeros((1 << 19, W), np.uint64)
    gpu.lib.mi355r_debug_probe(None, 0)
    orbit.render_views(gpu, scene, fbs, views)
    n = gpu.lib.mi355r_debug_probe(buf.ctypes.data_as(C.c_void_p), 1 << 19)
    r = buf[:n].astype(np.int64)
    print(label, "waves recorded", n)
    t = r[:, 7] / 100.0     # us (100 MHz)
    start = (r[:, 11] - r[:, 11].min()) / 100.0
    print("launch span us", (start + t).max(), "sum wave-us", t.sum(), "mean", t.mean(), "p50", np.median(t), "p99", np.percentile(t, 99), "max", t.max())
    for lo, hi in ((0, 64), (64, 256), (256, 1024), (1024, 4096), (4096, 1 << 30)):
        m = (r[:, 2] >= lo) & (r[:, 2] < hi)
        if m.sum() == 0: continue
        print(f"  list {lo:5d}-{hi:<10d} waves {m.sum():7d} wave-us sum {t[m].sum():10.0f} mean {t[m].mean():8.2f} max {t[m].max():8.1f} groups/wave {r[m,3].mean():6.1f} survivors/wave {r[m,4].mean():7.1f} cutoff {r[m,5].mean():7.1f} small {r[m,10].mean():7.1f} batches {r[m,6].mean():5.1f} ordered cyc {r[m,8].mean():9.0f} ring cyc {r[m,9].mean():9.0f}")
    top = np.argsort(-t)[:12]
    for i in top:
        print("   top:", dict(pas=int(r[i,0]), bin=int(r[i,1]), count=int(r[i,2]), groups=int(r[i,3]), surv=int(r[i,4]), cut=int(r[i,5]), small=int(r[i,10]), batches=int(r[i,6]), us=round(t[i],1), start=round(start[i],1), ordered_cyc=int(r[i,8]), ring_cyc=int(r[i,9])))
    # how busy is the device over time: waves alive per 10us slice
    end = start + t
    for s0 in range(0, int(end.max()) + 1, 20):
        alive = ((start < s0 + 20) & (end > s0)).sum()
        print(f"   t={s0:4d}us alive {alive}")
