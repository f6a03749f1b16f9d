#!/usr/bin/env python3
"""The bench's step repeated N times on resident frames: every output compared bit for bit with the first step's (the long form of
tests/test_gpu_pipeline.py::test_headline_step_is_bit_reproducible).   python tools/step_determinism.py [N=2000] [B=32]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402
from conftest import get_raw_and_blob  # noqa: E402
from test_gpu_pipeline import _frames  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
raw, blob = get_raw_and_blob((1, 2, 2, 2), (3, 13, 30, 3))
eng = native.Engine(0)
eng.load_weights(blob)
rng = np.random.default_rng(31)
eng.upload_frames(_frames(rng, B, 1080, 1920))
eng.gallery_set(rng.standard_normal((100_000, 512)).astype(np.float32))
eng.det_hashes(True, fetch=False)
eng.process_resident(10, flags=native.FLAG_FORCED_K)
first = eng.fetch_results()
h0 = eng.det_hashes()
bad = 0
for r in range(N):
    eng.process_resident(10, flags=native.FLAG_FORCED_K)
    got = eng.fetch_results()
    hr = eng.det_hashes()
    d = [i + 1 for i in range(64) if h0[i] != hr[i]]
    keys = [k for k in ("boxes", "kps", "scores", "counts", "emb", "match_idx", "match_cos") if not np.array_equal(first[k], got[k])]
    if d or keys:
        bad += 1
        where = ""
        if "emb" in keys:
            w = np.argwhere((first["emb"] != got["emb"]).any(-1))
            where = f"; embeddings of (frame, face) {w[:6].tolist()} ({len(w)} in all)"
        print(f"step {r}: detector ops {d[:4]}{'...' if len(d) > 4 else ''} outputs {keys}{where}", flush=True)
print(f"B={B}: {N} steps, {bad} differed from the first", flush=True)
sys.exit(1 if bad else 0)
