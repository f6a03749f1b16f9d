"""bench.py keeps its contract: ONE JSON line on stdout with the fields the driver reads, for one and two lanes, on a
small instance of the workload (2 frames of 256x320, 2 faces per frame, 2,000-row gallery, full-size networks)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "2", "--faces", "2", "--gallery", "2000",
           "--height", "256", "--width", "320", "--cpu-frames", "0", *extra]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]                 # exactly one JSON line on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("lanes", [2, 1])
def test_bench_line_contract(lanes):
    d = _run("--lanes", str(lanes))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["unit"] == "faces/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f16"
    # value is whole-job throughput of exactly `steps` steps: faces = steps x batch x faces-per-frame
    assert abs(d["value"] - 3 * 2 * 2 / (d["ms_per_step"] * 3 / 1e3)) <= 0.01 * d["value"] + 0.1
    cfg, rf = d["config"], d["roofline"]
    assert cfg["lanes"] == lanes and "workload" in cfg and cfg["batch_frames"] == 2 and cfg["gallery"] == 2000
    assert (cfg["one_batch_at_a_time"] is None) == (lanes == 1)
    assert cfg["host_to_host"]["ms_per_step"] > 0 and cfg["threshold_mode"]["steps"] == 3
    assert (cfg["host_to_host_lanes"] is None) == (lanes == 1) and (cfg["threshold_mode_lanes"] is None) == (lanes == 1)
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_source_sha256_16", "measured_on"):
        assert key in rf, key
    assert rf["bound"] == "mfma" and rf["peak"] == 2500.0 and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert sum(cfg["stage_ms_per_step"].values()) > 0
    assert "cpu_baseline" not in d                              # --cpu-frames 0
