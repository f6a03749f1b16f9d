#!/usr/bin/env python3
"""A/B of the load-time fusions and head tiles inside the pipeline: bench.py's one-batch-at-a-time stage times with
FRP_NO_KCONCAT=1 (shortcut convs as separate launches) against the default, alternating, same process family.
    python tools/fuse_probe.py [rounds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(env_extra, extra=()):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--cpu-frames", "0", "--pcie-steps", "0",
                          "--threshold-steps", "0", "--lanes", "1", *extra], env=env, capture_output=True, text=True).stdout
    d = json.loads(out.strip().split("\n")[-1])
    s = d["config"]["stage_ms_per_step"]
    return d["ms_per_step"], s["det_conv"], s["emb_conv"], d["roofline"]["achieved"]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for _ in range(rounds):
        for name, env in (("separate shortcut launches", {"FRP_NO_KCONCAT": "1"}), ("K-concat (default)", {})):
            ms, det, emb, tf = run(env)
            print(f"{name:28s} ms/step {ms:7.3f}  det {det:6.3f}  emb {emb:6.3f}  conv TF {tf:6.1f}", flush=True)


if __name__ == "__main__":
    main()
