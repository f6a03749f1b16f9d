#!/usr/bin/env python3
"""A few B-frame / K-face calls on resident 1080p frames (the subject of a rocprofv3 kernel trace of the small-batch path).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_b1 -o kt -- python3 tools/b1_run.py [B K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frp_amd_loader  # noqa
from frp_amd import native, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = native.Engine(0, max_batch=32, max_faces=10, max_h=1080, max_w=1920)
eng.load_weights(weights.pack_blob(weights.make_synthetic_raw(7)))
eng.gallery_set(np.random.default_rng(0).standard_normal((100000, 512)).astype(np.float32))
eng.upload_frames(np.random.default_rng(B).integers(0, 256, (B, 1080, 1920, 3), dtype=np.uint8))
for _ in range(5):
    eng.process_resident(K, flags=1)
    eng.fetch_results()
