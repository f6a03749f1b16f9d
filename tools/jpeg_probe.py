#!/usr/bin/env python3
"""JPEG ingest: a batch of 1080p baseline JPEG stills into the engine's frame buffer, (a) by PIL on the host into page-locked
staging + upload (ingest.StagedIngest's host path, what the reference's upload routes do plus the copy), (b) by
frp_upload_jpeg_async (entropy decoding on host threads, the rest on the GPU's copy stream).  ms per batch and frames/s, best of 3.
    python tools/jpeg_probe.py [B] [quality]"""
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
import frp_amd_loader  # noqa: E402,F401
import bench  # noqa: E402
from frp_amd import native  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 90
RST = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # restart interval in MCU rows (0: none - the host entropy decoder; > 0: the
                                                             # device's, unless FRP_JPEG_HOST_HUFFMAN=1 is set)
frames = bench.synth_frames(B, 1080, 1920, 10, 77)
jpegs = []
for f in frames:
    b = io.BytesIO()
    Image.fromarray(f[..., ::-1]).save(b, "JPEG", quality=Q, **({"restart_marker_rows": RST} if RST else {}))
    jpegs.append(b.getvalue())
print(f"{B} x 1080p JPEG stills, quality {Q}, restart interval {RST} MCU rows, FRP_JPEG_HOST_HUFFMAN={os.environ.get('FRP_JPEG_HOST_HUFFMAN', '')}: {sum(map(len, jpegs)) / B / 1e3:.0f} kB each; host threads available: {len(os.sched_getaffinity(0))}")
eng = native.Engine(0, max_batch=B, max_faces=10, max_h=1080, max_w=1920)
stage = eng.host_frames(B, 1080, 1920)


def host_path():
    for i, j in enumerate(jpegs):
        with Image.open(io.BytesIO(j)) as im:
            np.copyto(stage[i], np.asarray(im.convert("RGB")))
    eng.upload_frames_async(stage)
    eng.swap_frames()
    eng.synchronize()


def host_path_threads():
    from concurrent.futures import ThreadPoolExecutor

    def one(i):
        with Image.open(io.BytesIO(jpegs[i])) as im:
            np.copyto(stage[i], np.asarray(im.convert("RGB")))
    with ThreadPoolExecutor(16) as ex:
        list(ex.map(one, range(B)))
    eng.upload_frames_async(stage)
    eng.swap_frames()
    eng.synchronize()


def device_path():
    eng.upload_jpeg_async(jpegs)
    eng.swap_frames()
    eng.synchronize()


def device_path_pipelined(n=6):
    """n batches back to back, one wait at the end: the host decodes batch t+1 (second staging buffer) while the copy and the
    kernels of batch t run - the shape of StagedIngest / lanes in steady state"""
    for _ in range(n):
        eng.upload_jpeg_async(jpegs)
        eng.swap_frames()
    eng.synchronize()


for name, fn in (("PIL on one host thread + upload", host_path), ("PIL on 16 host threads + upload", host_path_threads),
                 ("frp_upload_jpeg_async (host entropy decode + device pixels)", device_path)):
    fn()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    print(f"  {name:62s} {best * 1e3:8.1f} ms per batch  {B / best:8.0f} frames/s")
device_path_pipelined()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    device_path_pipelined(6)
    best = min(best, (time.perf_counter() - t0) / 6)
print(f"  {'frp_upload_jpeg_async, 6 batches in a row, one wait at the end':62s} {best * 1e3:8.1f} ms per batch  {B / best:8.0f} frames/s")
import threading
t0 = time.perf_counter()
ths = [threading.Thread(target=lambda lo=lo: [native.jpeg_coefficients(j) for j in jpegs[lo::16]]) for lo in range(16)]
[t.start() for t in ths]
[t.join() for t in ths]
print(f"  {'(host entropy decode alone, 16 threads, through ctypes)':62s} {(time.perf_counter() - t0) * 1e3:8.1f} ms per batch")
