#!/usr/bin/env python3
"""Does the chip gain from two independent pipelines in flight (two handles = two streams)?  One host thread per lane vs
one thread that waits for the lanes in a fixed order (the latter loses the overlap whenever the hardware scheduler
favours one queue for a while - see lanes.py).
One handle at B=32 vs two handles at B=16 / B=32 each, forced K=10, R100, 100k gallery; faces/s over all handles."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frp_amd_loader  # noqa
from frp_amd import native, weights

blob = weights.pack_blob(weights.make_synthetic_raw(7))
G = np.random.default_rng(0).standard_normal((100000, 512)).astype(np.float32)


def make(B, seed):
    e = native.Engine(0, max_batch=B, max_faces=10, max_h=1080, max_w=1920)
    e.load_weights(blob)
    e.gallery_set(G)
    e.upload_frames(np.random.default_rng(seed).integers(0, 256, (B, 1080, 1920, 3), dtype=np.uint8))
    for _ in range(2):
        e.process_resident(10, flags=1); e.fetch_results()
    return e


def run(engs, B, steps):
    def loop(e):
        for _ in range(steps):
            e.process_resident(10, flags=1)
            e.fetch_results()
    th = [threading.Thread(target=loop, args=(e,)) for e in engs]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t
    return len(engs) * steps * B * 10 / dt, dt / steps * 1e3


one = make(32, 1)
print("1 x B=32: %.0f faces/s  (%.2f ms per step)" % run([one], 32, 20))
two16 = [make(16, 2), make(16, 3)]
print("2 x B=16: %.0f faces/s  (%.2f ms per round)" % run(two16, 16, 20))
print("1 x B=16: %.0f faces/s  (%.2f ms per step)" % run(two16[:1], 16, 20))
two32 = [one, make(32, 4)]
print("2 x B=32: %.0f faces/s  (%.2f ms per round)" % run(two32, 32, 20))
print("1 x B=32: %.0f faces/s  (%.2f ms per step)" % run([one], 32, 20))


def run_pipelined(engs, B, steps):
    """one host thread, len(engs) batches in flight: submit step s on lane s % L, then fetch the oldest one"""
    L = len(engs)
    t = time.perf_counter()
    for s in range(steps):
        engs[s % L].process_resident(10, flags=1)
        if s >= L - 1:
            engs[(s - (L - 1)) % L].fetch_results()
    for s in range(steps - (L - 1), steps):
        engs[s % L].fetch_results()
    dt = time.perf_counter() - t
    return steps * B * 10 / dt, dt / steps * 1e3


three32 = two32 + [make(32, 5)]
for L in (1, 2, 3):
    print("single thread, %d lanes x B=32: %.0f faces/s  (%.2f ms per step)" % ((L,) + run_pipelined(three32[:L], 32, 30)))
for L in (1, 2):
    print("single thread, %d lanes x B=32: %.0f faces/s  (%.2f ms per step)" % ((L,) + run_pipelined(three32[:L], 32, 30)))
