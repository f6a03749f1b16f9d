#!/usr/bin/env python3
"""Run-to-run determinism of the detector at camera size, per kernel family: the same seeded frames through `Engine.detect`
REPS times (optionally while a second handle keeps other kernels running on another stream, which shifts every DMA's
landing time), head maps compared BIT FOR BIT with the first run.  A mismatch convicts the family it happened in (the
cross-family test cannot) and is localised: map, image, row / column range, channels, the 8 x 30 tiles it touches.

    python tools/det_determinism.py [B=4] [REPS=40] [noise=1]
"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402
from conftest import get_raw_and_blob  # noqa: E402
from test_gpu_pipeline import _frames  # noqa: E402


def describe(first, got, lv):
    a, b = first.view(np.uint16), got.view(np.uint16)
    idx = np.argwhere(a != b)
    d = np.abs(first.astype(np.float32) - got.astype(np.float32))
    n, y, x, c = idx.T
    stride = (8, 16, 32)[lv]
    return (f"map {lv} (stride {stride}): {len(idx)} elements differ, max |diff| {d.max():.4f} (scale {np.abs(first.astype(np.float32)).max():.2f}); "
            f"images {sorted(set(n.tolist()))}, rows {y.min()}..{y.max()}, cols {x.min()}..{x.max()}, channels {c.min()}..{c.max()}")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    noise_on = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
    rng = np.random.default_rng(77)
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    fr = _frames(rng, B, 1080, 1920)
    eng = native.Engine(0)
    other = native.Engine(0)
    stop = threading.Event()
    xo = rng.standard_normal((9, 56, 56, 64)).astype(np.float16)
    wo = (rng.standard_normal((64, 3, 3, 64)) / 24).astype(np.float16)
    bo = np.zeros(64, np.float32)

    def noise():
        while not stop.is_set():
            other.conv2d(xo, wo, bo, flags=0x40000)
    th = threading.Thread(target=noise, daemon=True)
    if noise_on:
        th.start()
    bad = 0
    try:
        for fam, env in (("direct (FRP_NO_WINO=1)", "1"), ("winograd 2-D tiles", None)):
            if env:
                os.environ["FRP_NO_WINO"] = env
            else:
                os.environ.pop("FRP_NO_WINO", None)
            eng.load_weights(blob)
            eng.reset_counters()
            eng.detect(fr, max_faces=16, det_thresh=0.5)
            first = [h.copy() for h in eng.head_maps()]
            n_bad = 0
            for r in range(reps):
                eng.detect(fr, max_faces=16, det_thresh=0.5)
                got = eng.head_maps()
                for lv in range(3):
                    if not np.array_equal(first[lv].view(np.uint16), got[lv].view(np.uint16)):
                        n_bad += 1
                        print(f"[{fam}] rep {r}: " + describe(first[lv], got[lv], lv), flush=True)
            print(f"[{fam}] B={B}: {reps} repetitions, {n_bad} head maps differed from the first run", flush=True)
            bad += n_bad
    finally:
        stop.set()
        if noise_on:
            th.join()
        other.close()
        eng.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
