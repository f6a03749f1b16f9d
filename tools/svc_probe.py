#!/usr/bin/env python3
"""FaceService.process_stream against the engine-level loops it wraps, threshold mode, two lanes, over a LONG stream: arrival time
of every result, so that the pipeline's fill (first uploads not overlapped) and drain can be told from its steady state.
    python tools/svc_probe.py [batches]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frp_amd_loader  # noqa: E402,F401
import bench  # noqa: E402
from frp_amd import native, weights  # noqa: E402
from frp_amd.face_service import FaceService  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B, K, N = 32, 10, 100000
H, W = 1080, 1920
lanes = [native.Engine(0, max_batch=B, max_faces=K, max_h=H, max_w=W) for _ in range(2)]
blob = weights.pack_blob(weights.make_synthetic_raw(7))
rows = bench.gallery_rows(N, 0, N)
for e in lanes:
    e.load_weights(blob)
    e.gallery_set(rows)
frames = bench.synth_frames(B, H, W, K, 1234)
probe = lanes[0].detect(frames, max_faces=64, det_thresh=1e-6, nms_iou=0.4)
kth = np.sort(probe["scores"], axis=1)[:, ::-1][:, K - 1]
thr = float(np.clip(np.median(kth[kth > 0]), 1e-4, 0.9999))

# engine level, resident frames (bench.py: threshold_mode_lanes)
for e in lanes:
    e.upload_frames(frames)
    e.process_resident(K, det_thresh=thr, nms_iou=0.4, flags=0)
    e.fetch_results()
cnt = {"next": 0, "faces": 0}
lock = threading.Lock()


def loop(i, pinned=None):
    j = 0
    while True:
        with lock:
            if cnt["next"] >= n:
                return
            cnt["next"] += 1
        lanes[i].process_resident(K, det_thresh=thr, nms_iou=0.4, flags=0)
        if pinned is not None:
            lanes[i].upload_frames_async(pinned[i][j & 1])
        r = lanes[i].fetch_results()
        if pinned is not None:
            lanes[i].swap_frames()
        j += 1
        with lock:
            cnt["faces"] += int(r["counts"].sum())


def timed(target, *a):
    cnt["next"] = cnt["faces"] = 0
    ths = [threading.Thread(target=target, args=(i,) + a) for i in range(2)]
    t0 = time.perf_counter()
    [t.start() for t in ths]
    [t.join() for t in ths]
    for e in lanes:
        e.synchronize()
    dt = time.perf_counter() - t0
    return cnt["faces"] / dt, dt / n * 1e3


f, ms = timed(loop)
print(f"engine, resident frames, 2 lanes:            {f:9.0f} faces/s  {ms:7.3f} ms per batch")
pinned = [[e.host_frames(B, H, W) for _ in range(2)] for e in lanes]
for pp in pinned:
    for p in pp:
        p[...] = frames
for i, e in enumerate(lanes):
    e.upload_frames_async(pinned[i][0])
    e.swap_frames()
f, ms = timed(loop, pinned)
print(f"engine, host frames every batch, 2 lanes:    {f:9.0f} faces/s  {ms:7.3f} ms per batch")

svc = FaceService(engine=lanes[0], second_engine=lanes[1])
svc.ENCODINGS.adopt_device([f"id{i:07d}" for i in range(N)])
bufs = [svc.frame_buffer(B, H, W) for _ in range(8)]
for b in bufs:
    b[...] = frames
for _ in svc.process_stream((bufs[i % 8] for i in range(4)), max_faces=K, det_thresh=thr):
    pass
stamps, faces = [], []
t0 = time.perf_counter()
for per_frame in svc.process_stream((bufs[i % 8] for i in range(n)), max_faces=K, det_thresh=thr):
    faces.append(sum(len(x) for x in per_frame))
    stamps.append(time.perf_counter() - t0)
tot = sum(faces)
print(f"FaceService.process_stream, whole stream:    {tot / stamps[-1]:9.0f} faces/s  {stamps[-1] / n * 1e3:7.3f} ms per batch  (first result after {stamps[0] * 1e3:.1f} ms)")
w = 6
steady = (stamps[-1] - stamps[w - 1]) / (n - w)
print(f"FaceService.process_stream, after {w} results:  {sum(faces[w:]) / (stamps[-1] - stamps[w - 1]):9.0f} faces/s  {steady * 1e3:7.3f} ms per batch")
