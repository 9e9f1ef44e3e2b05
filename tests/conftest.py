import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
ORACLE_LIB = os.path.join(REPO, "oracle", "liboracle.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the library's last words before an abort: pytest's capture of stderr dies with the process
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    os.environ.setdefault("MI355R_FATAL_LOG", os.path.join(out, "mi355r_fatal.log"))


def golden_cases():
    return json.load(open(os.path.join(GOLDEN, "cases.json")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return z["color"], z["depth"]


def render_case(pipe, case):
    """Renders one tests/golden/cases.json entry through `pipe` (GPU library or oracle)."""
    from renderer_amd.binding import SCENES
    if case["scene"].endswith(".scn"):
        scene = pipe.open_scene(os.path.join(SCENES, case["scene"]), case.get("like"))
    else:
        scene = pipe.open_scene(case["scene"])
    view = pipe.make_view(case.get("eye"), case.get("target"), case.get("time", 0.0), case.get("layer", -1))
    w, h = case["size"]
    color, depth = pipe.render(scene, w, h, view, case.get("overlay", ()))
    pipe.close_scene(scene)
    return color, depth


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): the checker, never the thing under test on GPU runs."""
    from renderer_amd.binding import Pipeline
    if not os.path.exists(ORACLE_LIB):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "oracle"], check=True)
    return Pipeline(ORACLE_LIB)


@pytest.fixture(scope="session")
def gpu():
    """The product: renderer_amd/libmi355r.so through its C ABI.  No fallback."""
    from renderer_amd.binding import Pipeline, GPU_LIB
    try:
        # tests that hand the library device tensors need torch's HIP runtime up BEFORE the library's: brought up
        # second it finds no device ("No HIP GPUs are available"), whichever test happens to run first
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    pipe = Pipeline(GPU_LIB)
    assert pipe.backend.startswith("hip"), pipe.backend
    return pipe
