#!/usr/bin/env python3
"""A few detector passes on resident frames (subject of rocprofv3 --pmc runs on the stem / decode kernels).  args: B iters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frp_amd_loader  # noqa
from frp_amd import native, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
it = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = native.Engine(0, max_batch=B, max_faces=10, max_h=1080, max_w=1920)
eng.load_weights(weights.pack_blob(weights.make_synthetic_raw(7, emb_blocks=(1, 1, 1, 1)), emb_blocks=(1, 1, 1, 1)))
eng.upload_frames(np.random.default_rng(1).integers(0, 256, (B, 1080, 1920, 3), dtype=np.uint8))
for _ in range(it):
    eng.detect_resident((1080, 1920), max_faces=10, flags=1)
