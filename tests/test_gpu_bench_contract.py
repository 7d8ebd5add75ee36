"""GPU: bench.py honours the driver's contract -- exactly ONE line on stdout, a JSON object with the required keys, the
roofline and cpu_baseline objects, values consistent with each other (a child process: bench.py owns its process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                        # nothing but the JSON line on stdout
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run("--steps", "2", "--warmup", "1", "--train-steps", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["unit"] == "Mpixels/s"
    assert "workload" in d["config"] and "configs[2]" in d["config"]["workload"] and "model" not in d["config"]
    px = 8 * 512 * 512
    assert abs(d["value"] - px / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]          # value = pixels / step time
    # `roofline` = the kernel family that took more of the timed region (tree-context pair or lifting), the other one beside it
    fam = {("k_conv3" in r_["kernel"]): r_ for r_ in (d["roofline"], d["roofline_second"])}
    assert set(fam) == {True, False}
    for r in fam.values():
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches"):
            assert k in r, k
        assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.05 < r["frac"] < 1.0
    assert fam[True]["launches"] == 3 * d["steps"] and fam[False]["launches"] == 32 * d["steps"]
    assert d["roofline"]["avg_launch_ms"] * d["roofline"]["launches"] >= d["roofline_second"]["avg_launch_ms"] * d["roofline_second"]["launches"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mpixels/s" and c["sample"]
    assert d["value"] > 20 * c["value"]                   # sanity: the GPU path is not the CPU path
    assert d["train"]["ms_per_step"] > d["ms_per_step"] and d["train"]["loss"] > 0


def test_bench_other_config_is_labelled_truthfully():
    d = _run("--config", "1", "--steps", "1", "--warmup", "1", "--train-steps", "0", "--no-cpu-baseline", "--no-hbm-kernels")
    assert "configs[1]" in d["config"]["workload"] and "factorized" in d["config"]["workload"]
    assert d["config"]["per_gpu_images"] == 16 and d["config"]["image_hw"] == [256, 256]
    assert d["cpu_baseline"] is None
