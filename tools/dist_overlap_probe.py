#!/usr/bin/env python3
"""Why does the overlapped host-to-host loop lose its overlap when torch / RCCL live in the process (round-1
rehearsal: 20.65 vs 16.4 ms per step)?  One condition per process:
    python tools/dist_overlap_probe.py {plain|import|cuda|pg|allgather}
prints ms per step of (a) the resident loop, (b) the H2D copy alone, (c) the overlapped host-to-host loop."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native, weights  # noqa: E402
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
LATE = "late" in sys.argv          # torch / RCCL work AFTER the engine (its streams, its first launches) exists, as bench.py does
PROFILE = "profile" in sys.argv


def torch_side():
    global dist
    if mode != "plain":
        import torch
        if mode in ("cuda", "pg", "allgather"):
            torch.cuda.set_device(0)
            torch.zeros(8, device="cuda")
        if mode in ("pg", "allgather", "barrier", "gallery"):
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        if mode in ("allgather", "barrier", "gallery"):
            x = torch.ones(1 << 20, device="cuda")
            out = torch.empty(1 << 20, device="cuda")
            dist.all_gather_into_tensor(out, x)
            torch.cuda.synchronize()
        if mode == "barrier":
            dist.barrier()
            t_ = torch.tensor([1.0], device="cuda")
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            torch.cuda.synchronize()



if not LATE:
    torch_side()
B, K, H, W = 32, 10, 1080, 1920
eng = native.Engine(0, max_batch=B, max_faces=K, max_h=H, max_w=W, profile=PROFILE)
eng.load_weights(weights.pack_blob(weights.make_synthetic_raw(7)))
if mode == "gallery" and not LATE:
    from frp_amd import dist as fdist
    fdist.allgather_gallery_into_engine(eng, 100000, lambda first, cnt: bench.gallery_rows(100000, first, cnt), 0)
else:
    eng.gallery_set(bench.gallery_rows(10000, 0, 10000))
if LATE:
    torch_side()
frames = bench.synth_frames(B, H, W, K, 1)
eng.upload_frames(frames)
for _ in range(2):
    eng.process_resident(K, flags=1)
eng.synchronize()
t = time.perf_counter()
for _ in range(8):
    eng.process_resident(K, flags=1)
eng.synchronize()
res_ms = (time.perf_counter() - t) / 8 * 1e3
stage = [eng.host_frames(B, H, W) for _ in range(2)]
for s in stage:
    s[...] = frames
eng.upload_frames_async(stage[0]); eng.swap_frames(); eng.synchronize()
t = time.perf_counter()
for i in range(8):
    eng.upload_frames_async(stage[i & 1]); eng.swap_frames(); eng.synchronize()
copy_ms = (time.perf_counter() - t) / 8 * 1e3
eng.upload_frames_async(stage[1])
t = time.perf_counter()
for i in range(8):
    eng.upload_frames_async(stage[i & 1])
    eng.process_resident(K, flags=1)
    if "fetch" in sys.argv:
        eng.fetch_results()
    eng.swap_frames()
eng.synchronize()
ovl_ms = (time.perf_counter() - t) / 8 * 1e3
print(f"{mode:10s} resident {res_ms:6.2f} ms   copy alone {copy_ms:5.2f} ms ({B*H*W*3/copy_ms/1e6:5.1f} GB/s)   overlapped loop {ovl_ms:6.2f} ms")
