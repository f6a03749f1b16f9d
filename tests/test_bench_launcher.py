"""`python bench.py --gpus N` as the driver types it (no launcher, WORLD_SIZE unset): bench.py must start its own N ranks
before touching the GPU.  Driven here with N = 2 on the gloo backend (`--rehearse-cpu`: the multi-rank plumbing only -
launcher, rendezvous on 127.0.0.1, stream -> rank split, gallery shard all-gather, consistency check, rank 0's JSON line)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    return r


def test_bench_gpus2_starts_its_own_ranks_gloo():
    r = _run(["--gpus", "2", "--rehearse-cpu", "--gallery", "3001"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    cfg = out["config"]
    assert cfg["rccl_ranks"] == 2 and cfg["backend"] == "gloo" and cfg["ranks_seen"] == [0, 1]
    assert cfg["streams_of_rank"] == [[0], [1]] and "started its own ranks" in cfg["launcher"]
    assert out["value"] is None and out["scaling"] == "weak"
    # the gathered matrix is the one a single process builds
    sys.path.insert(0, ROOT)
    import frp_amd_loader  # noqa: F401
    import torch
    import bench
    from frp_amd import dist as fdist
    full = fdist.normalize_rows_f16(bench.gallery_rows(3001, 0, 3001))
    assert cfg["gathered_gallery_checksum"] == fdist.gallery_checksum(torch.from_numpy(full))
    # a flipped bit or two swapped rows change it
    bad = full.copy()
    bad[[5, 6]] = bad[[6, 5]]
    assert fdist.gallery_checksum(torch.from_numpy(bad)) != cfg["gathered_gallery_checksum"]


def test_bench_launcher_is_a_noop_for_one_gpu_and_under_a_launcher(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench._self_launch_if_needed(["--gpus", "1", "--steps", "3"]) is None
    assert bench._self_launch_if_needed(["--steps", "3"]) is None
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert bench._self_launch_if_needed(["--gpus", "2"]) is None      # already one of the ranks


def test_bench_refuses_a_world_that_does_not_match_gpus():
    r = _run(["--gpus", "2", "--rehearse-cpu"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr
