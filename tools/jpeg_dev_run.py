#!/usr/bin/env python3
"""A few batches of restart-interval 1080p JPEGs through frp_upload_jpeg_async (subject of a rocprofv3 kernel trace of the
device entropy decoder).   rocprofv3 --kernel-trace --stats -d out -o kt -- python3 tools/jpeg_dev_run.py [B] [quality] [rows]"""
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
R = int(sys.argv[3]) if len(sys.argv) > 3 else 1            # restart interval: > 0 MCU rows, < 0 that many MCUs
os.environ.setdefault("FRP_JPEG_DEVICE_HUFFMAN", "1")
frames = bench.synth_frames(B, 1080, 1920, 10, 77)
jpegs = []
for f in frames:
    b = io.BytesIO()
    Image.fromarray(f[..., ::-1]).save(b, "JPEG", quality=Q, **({"restart_marker_rows": R} if R > 0 else {"restart_marker_blocks": -R}))
    jpegs.append(b.getvalue())
eng = native.Engine(0, max_batch=B, max_faces=10, max_h=1080, max_w=1920)
for _ in range(6):
    t0 = time.perf_counter()
    eng.upload_jpeg_async(jpegs)
    t1 = time.perf_counter()
    eng.swap_frames()
    eng.synchronize()
    print(f"upload_jpeg_async returned after {(t1 - t0) * 1e3:.2f} ms, batch resident after {(time.perf_counter() - t0) * 1e3:.2f} ms; device-decoded batches: {eng.jpeg_device_batches()}")
