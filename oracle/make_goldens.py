#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- golden-vector generator (runs only where the
reference is present).

Renders every case of tests/golden/cases.json with the UNMODIFIED reference
(oracle/_ref/refdump, strict-IEEE build, see oracle/Makefile) and stores the
colour and depth buffers as tests/golden/<name>.npz.  The fixtures are data:
inputs (scene, size, camera, time) and the reference's outputs."""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLDEN = os.path.join(REPO, "tests", "golden")
REFDUMP = os.path.join(HERE, "_ref", "refdump")


def run_case(case, assets, binary=None, env=None):
    """Renders one case with a build of ref_driver.c (default: the unmodified reference, strict build)."""
    w, h = case["size"]
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as tmp:
        out = tmp.name
    cmd = [binary or REFDUMP, "--scene", case["ref"], "--assets", assets, "--size", str(w), str(h), "--out", out]
    if case["scene"].endswith(".scn"):
        cmd += ["--scn", os.path.join(REPO, "scenes", case["scene"])]
    if "eye" in case:
        cmd += ["--eye"] + [repr(float(x)) for x in case["eye"]]
    if "target" in case:
        cmd += ["--target"] + [repr(float(x)) for x in case["target"]]
    if "time" in case:
        cmd += ["--time", repr(float(case["time"]))]
    if "layer" in case:
        cmd += ["--layer", str(case["layer"])]
    for op in case.get("overlay", []):        # 2-D overlays over the finished frame, in order (ref_driver.c)
        cmd += ["--" + op[0]] + [x if isinstance(x, str) else repr(float(x)) for x in op[1:]]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, env=env, timeout=300)
    raw = open(out, "rb").read()
    os.unlink(out)
    magic, rw, rh, _ = struct.unpack("4i", raw[:16])
    assert magic == 0x31475A52 and (rw, rh) == (w, h)
    color = np.frombuffer(raw, np.uint8, w * h * 4, 16).reshape(h, w, 4)
    depth = np.frombuffer(raw, np.float32, w * h, 16 + w * h * 4).reshape(h, w)
    return color, depth


def main():
    # the reference reads the very same asset files this repo ships under assets/
    assets = os.path.join(REPO, "assets")
    cases = json.load(open(os.path.join(GOLDEN, "cases.json")))
    only = set(sys.argv[1:])
    for case in cases:
        if only and case["name"] not in only:
            continue
        color, depth = run_case(case, assets)
        path = os.path.join(GOLDEN, case["name"] + ".npz")
        np.savez_compressed(path, color=color, depth=depth)
        drawn = int((depth < 1).sum())
        print(f"{case['name']:22s} {color.shape[1]}x{color.shape[0]}  covered px {drawn:7d}  "
              f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
