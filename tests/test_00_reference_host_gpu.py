"""The reference's own host code driving the GPU library (GPU box).

oracle/_ref/refhost_mi355r is built in the CPU container from the reference's sources where they lie
(oracle/Makefile, target refhost): everything except core/graphics.c, core/texture.c and core/draw2d.c, i.e. its scene
factories, .scn parser, caches, loaders, the shaders' update_model / draw_model loops (one
graphics_draw_triangle call per face) and tests/test_helper.c's frame driver, all unchanged, plus our
headless driver, linked against renderer_amd/libmi355r.so.  The binary travels to the GPU box; the
reference's sources do not.  Each case runs it as a child process and compares the frame it dumps with
the golden frame the unmodified reference rendered: identical depth and coverage, colour within the
bar of tests/test_gpu_parity.py.

The file sorts first on purpose: the child is started before this process has touched the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

from __graft_entry__ import compare_frames
from conftest import REPO, golden_cases, load_golden

pytestmark = pytest.mark.gpu

EXE = os.path.join(REPO, "oracle", "_ref", "refhost_mi355r")


def gpu_in_use_by_this_process():
    try:
        return any(os.path.realpath(os.path.join("/proc/self/fd", f)).startswith("/dev/kfd") for f in os.listdir("/proc/self/fd"))
    except OSError:
        return False


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/refhost_mi355r was not built (needs the reference's sources)")
@pytest.mark.parametrize("name", ["helmet_default", "kgirl_anim", "azura_shadow", "kgirl_blend", "helmet_inside", "buster_anim",
                                  "helmet_overlay"])
def test_reference_host_drives_the_gpu_library(name):
    if gpu_in_use_by_this_process():
        pytest.skip("run this file first (or alone): its child processes are started before the pipeline is loaded here")
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_goldens import run_case
    case = next(c for c in golden_cases() if c["name"] == name)
    got = run_case(case, os.path.join(REPO, "assets"), binary=EXE)
    want = load_golden(name)
    n = got[1].size
    any_diff, big_diff, coverage, bad_depth, any_depth = compare_frames(got, want)
    print(f"{name}: colour!= {any_diff} (>1LSB {big_diff}) coverage {coverage} depth!= {any_depth} of {n}")
    assert coverage == 0 and any_depth == 0
    assert any_diff <= max(2, n * 2e-4) and big_diff <= n * 1e-3
