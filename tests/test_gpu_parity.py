"""GPU parity (the tests proper): the HIP pipeline, called through the C ABI,
against (1) the golden frames rendered by the unmodified reference and (2) the
CPU oracle on the same inputs.

Bar (BASELINE.json north_star): depth within 1e-5 relative, 8-bit colour
within 1 LSB on >= 99.9 % of pixels, coverage differing on <= 0.01 % of pixels.
The pipeline keeps the reference's f32 expression order with FMA contraction
off, so the expected result is much tighter: identical coverage and depth, and
colour differing only where the device's double-precision pow() rounds
differently from glibc's."""
import numpy as np
import pytest

from __graft_entry__ import compare_frames
from conftest import golden_cases, load_golden, render_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_gpu_matches_reference_golden(gpu, case):
    want = load_golden(case["name"])
    got = render_case(gpu, case)
    n = got[1].size
    any_diff, big_diff, coverage, bad_depth, any_depth = compare_frames(got, want)
    print(f"{case['name']}: colour!= {any_diff} (>1LSB {big_diff}) coverage {coverage} "
          f"depth>1e-5 {bad_depth} depth!= {any_depth} of {n}")
    # north-star tolerance
    assert coverage <= max(1, n * 1e-4)
    assert bad_depth <= coverage
    assert big_diff <= n * 1e-3
    # what this implementation actually achieves
    assert coverage == 0 and any_depth == 0
    assert any_diff <= max(2, n * 2e-4)


@pytest.mark.parametrize("name,like,size,index", [
    ("helmet", None, (640, 360), 5),
    ("azura", None, (400, 300), 11),
    ("craftsman", None, (400, 300), 2),
    ("kgirl_blend.scn", "kgirl", (512, 288), 7),
])
def test_gpu_matches_oracle_on_orbit_views(gpu, oracle, name, like, size, index):
    import os
    from renderer_amd.binding import SCENES
    frames = []
    for pipe in (gpu, oracle):
        scene = pipe.open_scene(os.path.join(SCENES, name) if name.endswith(".scn") else name, like)
        frames.append(pipe.render(scene, size[0], size[1], pipe.orbit_view(index, 16)))
        pipe.close_scene(scene)
    n = size[0] * size[1]
    any_diff, big_diff, coverage, bad_depth, any_depth = compare_frames(frames[0], frames[1])
    print(f"{name}@{index}: colour!= {any_diff} (>1LSB {big_diff}) coverage {coverage} depth!= {any_depth} of {n}")
    assert coverage == 0 and any_depth == 0
    assert any_diff <= max(2, n * 2e-4) and big_diff <= n * 1e-3
