#!/usr/bin/env python3
"""Which detector layer is not deterministic?  Prefix runs (`Engine.det_prefix(n)`: the program up to op n - 1 on the resident
frames) repeated REPS times for every prefix length in turn, the last op's output compared bit for bit with the first run.
The first prefix length that shows a mismatch names the kernel; the mismatch is localised in that op's output.
    python tools/det_bisect.py [B=4] [REPS=600] [family: direct|wino] [first_n last_n]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native, netspec  # noqa: E402
from conftest import get_raw_and_blob  # noqa: E402
from test_gpu_pipeline import _frames  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    fam = sys.argv[3] if len(sys.argv) > 3 else "direct"
    layers = netspec.detector_layers((1, 2, 2, 2))
    lo = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    hi = int(sys.argv[5]) if len(sys.argv) > 5 else len(layers)
    if fam == "direct":
        os.environ["FRP_NO_WINO"] = "1"
    rng = np.random.default_rng(77)
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    fr = _frames(rng, B, 1080, 1920)
    eng = native.Engine(0)
    eng.load_weights(blob)
    eng.upload_frames(fr)
    for n in range(lo, hi + 1):
        name = layers[n - 1].name
        first = eng.det_prefix(n)
        bad = 0
        for r in range(reps):
            got = eng.det_prefix(n)
            if not np.array_equal(first.view(np.uint16), got.view(np.uint16)):
                bad += 1
                idx = np.argwhere(first.view(np.uint16) != got.view(np.uint16))
                i, y, x, c = idx.T
                d = np.abs(first.astype(np.float32) - got.astype(np.float32))
                print(f"  prefix {n} ({name}, out {first.shape}) rep {r}: {len(idx)} elements differ, max |diff| {d.max():.4f}; images {sorted(set(i.tolist()))}, "
                      f"rows {y.min()}..{y.max()}, cols {x.min()}..{x.max()}, channels {c.min()}..{c.max()} ({len(set(c.tolist()))} distinct); "
                      f"flat pixel index {int((i[0] * first.shape[1] + y.min()) * first.shape[2] + x[y == y.min()].min())}", flush=True)
        print(f"prefix {n:2d} {name:28s} out {first.shape}: {bad} of {reps} runs differed", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
