#!/usr/bin/env python3
"""Random views through the UNMODIFIED reference (oracle/_ref/refdump, built where the reference's sources are) and the
CPU restatement, compared bit for bit: tests/fuzz_reference.py [iterations] [seed].  CPU only; needs /root/reference."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))
from conftest import ORACLE_LIB, render_case                      # noqa: E402
from make_goldens import run_case                                   # noqa: E402
from renderer_amd.binding import Pipeline                           # noqa: E402

SCENES = [("helmet", "pbr:helmet"), ("kgirl", "blinn:kgirl"), ("buster", "pbr:buster"), ("azura", "blinn:azura"),
          ("craftsman", "blinn:craftsman"), ("drone", "pbr:drone"), ("dieselpunk", "pbr:dieselpunk"), ("phoenix", "blinn:phoenix"),
          ("azura_shadow.scn", "blinn:azura"), ("kgirl_blend.scn", "blinn:kgirl"), ("dieselpunk_shadow.scn", "pbr:dieselpunk"),
          ("helmet_cutoff.scn", "pbr:helmet"), ("synthetic_blinn.scn", "identity")]
LIKE = {"azura_shadow.scn": "azura", "kgirl_blend.scn": "kgirl", "dieselpunk_shadow.scn": "dieselpunk", "helmet_cutoff.scn": "helmet"}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    oracle = Pipeline(ORACLE_LIB)
    bad = 0
    for it in range(n):
        scene, ref = SCENES[int(rng.integers(len(SCENES)))]
        direction = rng.normal(size=3)
        direction /= np.linalg.norm(direction)
        radius = float(rng.choice([0.05, 0.3, 0.8, 1.5, 3.0]))
        case = {"name": "random", "scene": scene, "ref": ref, "size": [int(rng.integers(33, 200)), int(rng.integers(17, 120))],
                "eye": [float(np.float32(x)) for x in direction * radius],
                "target": [float(np.float32(x)) for x in rng.uniform(-0.2, 0.2, 3)],
                "time": float(np.float32(rng.uniform(0, 3)))}
        if scene in LIKE:
            case["like"] = LIKE[scene]
        if ref.startswith("pbr") and rng.integers(4) == 0:
            case["layer"] = int(rng.integers(0, 6))
        want = run_case(case, os.path.join(REPO, "assets"))
        got = render_case(oracle, case)
        diff = int((got[0] != want[0]).sum()) + int((got[1].view(np.uint32) != want[1].view(np.uint32)).sum())
        if diff:
            bad += 1
            print("MISMATCH", case, diff, "values differ", flush=True)
        if it % 10 == 9:
            print(f"{it + 1} views, {bad} mismatches", flush=True)
    print("FUZZ-REFERENCE", n, "views", bad, "mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
