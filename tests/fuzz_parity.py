#!/usr/bin/env python3
"""Random views of every scene through the GPU library and the CPU oracle, compared bit for bit:
tests/fuzz_parity.py [iterations] [seed] [largest width].  Sizes are ragged on purpose (bins, tiles and sky regions cut by
the frame's edges), cameras go inside the models and close to the sky box (clipping), times are random
(animation), and every few views a frame is rendered twice into the same framebuffer (pending clears
against a used buffer, the strip height that follows the last frame).  FUZZ_PER_TRIANGLE=1: the GPU library is driven
face by face (graphics_draw_triangle, the reference's draw_model loops), so that models come back under other views
and times and their vertex streams are drawn from the copies remembered on the device."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))     # test infrastructure: the oracle is the checker here
from conftest import ORACLE_LIB                                   # noqa: E402
from renderer_amd.binding import GPU_LIB, SCENES, Pipeline         # noqa: E402
from renderer_amd.configs import CONFIGS                            # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    widest = int(sys.argv[3]) if len(sys.argv) > 3 else 520
    gpu, oracle = Pipeline(GPU_LIB), Pipeline(ORACLE_LIB)
    if os.environ.get("FUZZ_PER_TRIANGLE") == "1":
        gpu.lib.rzs_set_per_triangle(1)
    scenes = [(name, like) for _, name, like, _, _ in CONFIGS] + [("craftsman", None), ("drone", None)]
    opened = {}
    bad = 0
    for it in range(n):
        name, like = scenes[int(rng.integers(len(scenes)))]
        if (name, like) not in opened:
            try:
                opened[(name, like)] = (gpu.open_scene(name, like), oracle.open_scene(name, like))
            except Exception as e:                                  # a scene this tree does not carry
                print("skip", name, e)
                opened[(name, like)] = None
        if opened[(name, like)] is None:
            continue
        gs, os_ = opened[(name, like)]
        w, h = int(rng.integers(33, widest)), int(rng.integers(17, max(18, widest * 9 // 16)))
        radius = float(rng.choice([0.05, 0.3, 0.8, 1.5, 3.0]))
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        eye = [float(x) for x in direction * radius]
        target = [float(x) for x in rng.uniform(-0.2, 0.2, 3)]
        when = float(rng.uniform(0, 3))
        views = [p.make_view(eye, target, when, -1) for p in (gpu, oracle)]
        want = oracle.render(os_, w, h, views[1])
        fb = gpu.framebuffer(w, h)
        repeats = 2 if it % 4 == 0 else 1
        for r in range(repeats):
            gpu.draw(gs, fb, views[0])
            got = gpu.read(fb)
            diff = int((np.asarray(got[0]) != np.asarray(want[0])).sum()) + int((np.asarray(got[1]).view(np.uint32) != np.asarray(want[1]).view(np.uint32)).sum())
            if diff:
                bad += 1
                print(f"MISMATCH {name} {w}x{h} eye {eye} target {target} time {when} repeat {r}: {diff} values differ", flush=True)
        gpu.release(fb)
        if it % 10 == 9:
            print(f"{it + 1} views, {bad} mismatches", flush=True)
    print("FUZZ", n, "views", bad, "mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
