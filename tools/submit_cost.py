#!/usr/bin/env python3
"""Host time of ONE frp_process_resident call (the ~145 launches of a headline step are queued asynchronously; the call returns when they are
queued): launch by launch (FRP_NO_GRAPH=1) against replay from captured graphs.    [FRP_NO_GRAPH=1] python tools/submit_cost.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native, weights  # noqa: E402

eng = native.Engine(0)
eng.load_weights(weights.pack_blob(weights.make_synthetic_raw(7)))
rng = np.random.default_rng(0)
eng.upload_frames(rng.integers(0, 256, size=(32, 1080, 1920, 3), dtype=np.uint8))
G = rng.standard_normal((100_000, 512)).astype(np.float32)
eng.gallery_set(G)
for _ in range(3):
    eng.process_resident(10, flags=native.FLAG_FORCED_K)
    eng.fetch_results()
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    eng.process_resident(10, flags=native.FLAG_FORCED_K)
    ts.append(time.perf_counter() - t0)
    eng.fetch_results()
print(f"FRP_NO_GRAPH={os.environ.get('FRP_NO_GRAPH', '-')}: process_resident returns after {np.median(ts) * 1e3:.3f} ms (median of 20; min {min(ts) * 1e3:.3f}, max {max(ts) * 1e3:.3f}); "
      f"graph replays {eng.graph_replays()}")
eng.close()
