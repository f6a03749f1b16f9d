#!/usr/bin/env python3
"""A few threshold-mode passes (score threshold + NMS, ragged face counts: the reference's loop) on one lane, resident frames:
the subject of `rocprofv3 --kernel-trace --stats` when the embedder's kernels at ~186 faces are to be looked at.  args: steps"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frp_amd_loader  # noqa: E402,F401
import bench  # noqa: E402
from frp_amd import native, weights  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B, K, N = 32, 10, 100000
eng = native.Engine(0, max_batch=B, max_faces=K, max_h=1080, max_w=1920)
eng.load_weights(weights.pack_blob(weights.make_synthetic_raw(7)))
eng.gallery_set(bench.gallery_rows(N, 0, N))
frames = bench.synth_frames(B, 1080, 1920, K, 1234)
probe = eng.detect(frames, max_faces=64, det_thresh=1e-6, nms_iou=0.4)
kth = np.sort(probe["scores"], axis=1)[:, ::-1][:, K - 1]
thr = float(np.clip(np.median(kth[kth > 0]), 1e-4, 0.9999))
eng.upload_frames(frames)
import time
for i in range(steps + 1):
    if i == 1:
        eng.synchronize()
        t0 = time.perf_counter()
    eng.process_resident(K, det_thresh=thr, nms_iou=0.4, flags=0)
    r = eng.fetch_results()
eng.synchronize()
print(f"{int(r['counts'].sum())} faces per step, {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step", file=sys.stderr)
