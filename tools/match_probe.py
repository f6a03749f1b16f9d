#!/usr/bin/env python3
"""Gallery match (top-1), wall time per frp_match call and identity of results: persistent running-best kernel vs the per-tile kernel
(FRP_MATCH_V1=1), N = 100k / 1M rows, M = 32 / 320 / 512 queries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frp_amd_loader  # noqa
from frp_amd import native
eng = native.Engine(0)
rng = np.random.default_rng(0)
for N in (100_000, 1_000_000):
    eng.gallery_set(rng.standard_normal((N, 512)).astype(np.float32))
    for M in (32, 320, 512):
        q = rng.standard_normal((M, 512)).astype(np.float32)
        res = {}
        for v1 in (False, True):
            if v1:
                os.environ["FRP_MATCH_V1"] = "1"
            else:
                os.environ.pop("FRP_MATCH_V1", None)
            idx, cos = eng.match(q)
            t = time.perf_counter()
            for _ in range(20):
                eng.match(q)
            res[v1] = ((time.perf_counter() - t) / 20 * 1e3, idx, cos)       # wall ms per call (incl. query upload / result fetch)
        same = np.array_equal(res[False][1], res[True][1]) and np.array_equal(res[False][2], res[True][2])
        a, b = res[False][0], res[True][0]
        print(f"N={N:8d} M={M:3d}: running-best {a*1e3:7.1f} us per call | per-tile {b*1e3:7.1f} us per call | identical results {same}")
