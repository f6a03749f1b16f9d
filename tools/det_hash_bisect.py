#!/usr/bin/env python3
"""Which op of a FULL detector pass differs from run to run?  With `Engine.det_hashes()` on, every pass hashes each op's output
right behind the op; a pass whose head maps differ from the first pass's names the first op whose hash moved.
    python tools/det_hash_bisect.py [B=4] [REPS=3000] [direct|wino]"""
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
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    fam = sys.argv[3] if len(sys.argv) > 3 else "direct"
    if fam == "direct":
        os.environ["FRP_NO_WINO"] = "1"
    layers = netspec.detector_layers((1, 2, 2, 2))
    rng = np.random.default_rng(77)
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    fr = _frames(rng, B, 1080, 1920)
    eng = native.Engine(0)
    eng.load_weights(blob)
    eng.upload_frames(fr)
    eng.det_hashes(True, fetch=False)
    eng.detect_resident((1080, 1920), max_faces=16, det_thresh=0.5)
    first = eng.det_hashes()
    print("hashes of the first pass:", " ".join(f"{i + 1}:{int(v) & 0xffff:04x}" for i, v in enumerate(first[:len(layers)])), flush=True)
    bad = 0
    for r in range(reps):
        eng.detect_resident((1080, 1920), max_faces=16, det_thresh=0.5)
        got = eng.det_hashes()
        d = [i for i in range(len(layers)) if first[i] != got[i]]
        if d:
            bad += 1
            print(f"rep {r}: first differing op {d[0] + 1} ({layers[d[0]].name}); differing ops {[i + 1 for i in d]}", flush=True)
    print(f"[{fam}] B={B}: {reps} passes, {bad} differed", flush=True)
    eng.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
