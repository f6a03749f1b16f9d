#!/usr/bin/env python3
"""Does running two independent pipelines concurrently on one GPU (two handles = two streams)
raise aggregate throughput?  If yes, overlapping independent launches is worth building in."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frp_amd_loader  # noqa
from frp_amd import native, weights
import bench

blob = weights.pack_blob(weights.make_synthetic_raw(7))
G = bench.gallery_rows(100000, 0, 100000)
K = 10


def make(B, seed):
    e = native.Engine(0, max_batch=B, max_faces=K)
    e.load_weights(blob)
    e.gallery_set(G)
    e.upload_frames(bench.synth_frames(B, 1080, 1920, K, seed))
    for _ in range(2):
        e.process_resident(K, flags=1)
    e.synchronize()
    return e


def run(engs, steps):
    def work(e):
        for _ in range(steps):
            e.process_resident(K, flags=1)
        e.synchronize()
    ts = [threading.Thread(target=work, args=(e,)) for e in engs]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    return time.perf_counter() - t0


one = make(32, 1)
dt = run([one], 10)
print(f"1 engine  x B=32: {10*32/dt:8.1f} frames/s")
one.close()
two = [make(16, 2), make(16, 3)]
dt = run(two, 20)
print(f"2 engines x B=16: {2*20*16/dt:8.1f} frames/s")
[e.close() for e in two]
two = [make(32, 4), make(32, 5)]
dt = run(two, 10)
print(f"2 engines x B=32: {2*10*32/dt:8.1f} frames/s")
