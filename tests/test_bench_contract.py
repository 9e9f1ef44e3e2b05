"""bench.py's one JSON line: the keys and types the driver reads, on a short run (GPU box)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "32", "--no-other"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for key, kind in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                      ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                      ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(out[key], kind), (key, out.get(key))
    assert out["vs_baseline"] is None and out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and "workload" in out["config"]
    assert out["value"] > 500                                   # frames/s at 1920x1080, the target's order of magnitude at least
    assert abs(out["value"] * out["ms_per_step"] / 1e3 - 32) < 1e-6 * 32        # value = frames of a step / its time
    roof = out["roofline"]
    # `bound` names the roof the committed hardware counters put the kernel under (HBM traffic, vector issue, texture data path)
    assert roof["bound"] in ("hbm", "mfma", "valu", "td") and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0 < roof["frac"] < 1
    assert roof["algorithmic_frac"] == roof["frac"] and roof["kernel"]
    assert out["value_with_copy_out"] > 500 and "value_includes" in out["config"]
    cpu = out["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] == 1 and cpu["value"] > 0 and cpu["sample"]
