#!/usr/bin/env python3
"""The matcher's kernels alone, for a rocprofv3 kernel trace: 30 top-1 passes of M queries over an N-row gallery.
    rocprofv3 --kernel-trace --stats -d gpurun_out/r5/match_kt -o kt -- python3 tools/match_kernel_probe.py [N=100000] [M=320]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frp_amd_loader  # noqa: F401
from frp_amd import native

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 320
eng = native.Engine(0)
rng = np.random.default_rng(0)
eng.gallery_set(rng.standard_normal((N, 512)).astype(np.float32))
q = rng.standard_normal((M, 512)).astype(np.float32)
first = eng.match(q)
for _ in range(30):
    got = eng.match(q)
    assert np.array_equal(first[0], got[0]) and np.array_equal(first[1], got[1])
G = eng.gallery_get(0, N).astype(np.float64)
qq = q.astype(np.float64)
qq /= np.linalg.norm(qq, axis=1, keepdims=True)
print("top-1 equals float64 argmax:", bool(np.array_equal((qq @ G.T).argmax(1), first[0])))
