"""Pins the CPU oracle (oracle/) to the reference itself: every golden frame
under tests/golden was rendered by the unmodified reference, compiled strictly
(oracle/Makefile `ref`, oracle/make_goldens.py); the oracle must reproduce each
one BIT FOR BIT -- colour bytes and depth floats -- through its own loaders,
scene layer, pipeline, samplers and shaders."""
import numpy as np
import pytest

from conftest import golden_cases, load_golden, render_case


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c["name"])
def test_oracle_reproduces_reference_frame(oracle, case):
    want_color, want_depth = load_golden(case["name"])
    color, depth = render_case(oracle, case)
    assert color.shape == want_color.shape
    bad_depth = int((depth.view(np.uint32) != want_depth.view(np.uint32)).sum())
    bad_color = int((color != want_color).any(axis=2).sum())
    assert bad_depth == 0, f"{bad_depth} depth values differ"
    assert bad_color == 0, f"{bad_color} pixels differ"
