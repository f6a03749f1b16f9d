#!/usr/bin/env python3
"""Headline benchmark: faces/sec (detect+embed+match) on 1080p frames vs a 100k-identity gallery.

One process per GPU (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`),
one camera stream per GPU (weak scaling, no data-path collective); the gallery shards are
all-gathered once over RCCL at setup.  Frames are resident in HBM when the
timed region starts.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time



def _self_launch_if_needed(argv):
    """`python bench.py --gpus N` typed without a launcher (WORLD_SIZE unset, N > 1): start the N ranks as FRESH child
    processes - `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`, rendezvous on 127.0.0.1 -
    before this process has imported anything that could touch the GPU (a process that has initialised HIP must not
    exec or fork GPU children), hand rank 0's JSON line through on stdout and exit with the children's code.  The
    reference's fan-out this replaces: one task per camera on a ThreadPoolExecutor(4),
    backend/app/routes/camera.py:30,277-279,304-305.  Under a launcher (WORLD_SIZE set) and for N = 1 this is a no-op."""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["FRP_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.exit(subprocess.run(cmd, env=env).returncode)


if __name__ == "__main__":
    _self_launch_if_needed(sys.argv[1:])

import numpy as np  # noqa: E402

# two lanes per GPU next to torch + RCCL need more than the default 4 hardware queues (see native.py); before any HIP call
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native, netspec, weights  # noqa: E402

LANE_SETTLE_STEPS = int(os.environ.get("FRP_BENCH_SETTLE", "20"))      # untimed two-lane steps in front of the timed region (see run_steps' caller)
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def _pmc_summary():
    """The newest committed PMC summary (profiles/r<N>/pmc_traffic.json)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "pmc_traffic.json")),
                   key=lambda p: int(os.path.basename(os.path.dirname(p))[1:]))
    return os.path.relpath(found[-1], ROOT) if found else os.path.join("profiles", "pmc_traffic.json")


PMC_SUMMARY = _pmc_summary()


def pmc_conv_traffic_per_launch(launches_per_step):
    """HBM-side bytes per conv launch from the committed rocprofv3 --pmc passes of this same command
    (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, collected in separate passes as the microarch guide
    prescribes; tools/pmc_summarise.py).  The summary records a hash of the native sources it was measured
    on: None (not a stale number) when the kernels have changed since, or when the summary is absent."""
    try:
        with open(os.path.join(ROOT, PMC_SUMMARY)) as f:
            t = json.load(f)
        if t.get("kernel_source_sha256_16") != native.kernel_source_hash():
            return None
        return round(t["conv_traffic_bytes_per_step"] / max(1, launches_per_step))
    except Exception:
        return None


def synth_frames(B, H, W, K, seed):
    """low-amplitude noise background (mean 110, sigma 12) + K planted face blobs."""
    rng = np.random.default_rng(seed)
    frames = np.empty((B, H, W, 3), dtype=np.uint8)
    tmpl = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366], [41.5493, 92.3655], [70.7299, 92.2041]]) / 112.0
    for b in range(B):
        f = rng.normal(110.0, 12.0, size=(H, W, 3)).astype(np.float32)
        r2 = np.random.default_rng(seed * 1000 + b)
        for _ in range(K):
            s = r2.uniform(80, 200)
            x0, y0 = r2.uniform(0, W - s), r2.uniform(0, H - s)
            yy, xx = np.mgrid[0:int(s), 0:int(s)]
            m = ((xx / s - 0.5) / 0.38) ** 2 + ((yy / s - 0.5) / 0.48) ** 2 < 1.0
            patch = f[int(y0):int(y0) + int(s), int(x0):int(x0) + int(s)]
            patch[m] += 60.0
            for (tx, ty) in tmpl:
                cx, cy = int(tx * s), int(ty * s)
                patch[max(0, cy - 3):cy + 3, max(0, cx - 3):cx + 3] -= 70.0
        frames[b] = np.clip(f, 0, 255).astype(np.uint8)
    return frames


def gallery_rows(n_total, first, count, seed=42):
    """rows [first, first+count) of default_rng(42).standard_normal((N,512)), unit-normalised.
    Generated blockwise so every rank can produce its own shard without the whole matrix."""
    out = np.empty((count, 512), dtype=np.float32)
    blk = 4096
    for b0 in range((first // blk) * blk, first + count, blk):
        rows = np.random.default_rng([seed, b0 // blk]).standard_normal((blk, 512)).astype(np.float32)
        lo, hi = max(b0, first), min(b0 + blk, first + count)
        out[lo - first:hi - first] = rows[lo - b0:hi - b0]
    out /= np.linalg.norm(out, axis=1, keepdims=True)
    return out


def cpu_baseline(raw, frames, K, n_gallery, sample_frames):
    """Bounded CPU sample of the same workload on this box's host cores: the fp32 oracle
    network (torch-CPU, all cores) + the reference's own per-face compare loop
    (oracle/plumbing.py restating face_service.py:395-443 + camera.py:243-259, 1 thread)."""
    import torch
    from oracle import network as onet
    from oracle import plumbing
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("FRP_CPU_THREADS", "16")))   # a 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    fr = frames[:sample_frames]
    H, W = fr.shape[1:3]
    canvas = ((H + 31) // 32 * 32, (W + 31) // 32 * 32)
    t0 = time.perf_counter()
    res = onet.process_frames(raw, fr, None, canvas, score_thresh=0.0, nms_iou=2.0, max_faces=K)
    t_net = time.perf_counter() - t0
    n_faces = sum(len(r["emb"]) for r in res)
    # reference plumbing at the same gallery size, timed on 8 faces and scaled
    t_face = plumbing.time_reference_plumbing(n_gallery, 512, 8)
    total = t_net + t_face * n_faces
    return {"value": round(n_faces / total, 3), "unit": "faces/s", "cores": cores, "kind": "port",
            "sample": f"{sample_frames}x1080p frames, {n_faces} faces: fp32 torch-CPU oracle network {t_net:.2f}s "
                      f"({cores} threads) + reference compare_faces loop at N={n_gallery} {t_face:.3f}s/face (1 thread); "
                      "the reference's literal dlib path is not installable offline"}


def run_config4(args, json_fd):
    """BASELINE config 4 on one GPU (the largest single-GPU configuration): a 4K camera stream, detection pyramid
    {1, .5, .25}, 1M-identity gallery.  One step = B resident 3840x2160 frames: 3 x (resize + detector + decode/NMS,
    threshold mode), cross-scale merge on the host (pyramid.merge_scales), align from the full-resolution frames,
    embed, match, host results out.  A side line (never the headline `value` the driver records)."""
    from frp_amd import pyramid
    B, K, N, H, W = args.batch if args.batch != 32 else 4, args.faces, 1_000_000, 2160, 3840
    scales = (1.0, 0.5, 0.25)
    raw = weights.make_synthetic_raw(7)
    L = max(1, args.lanes)
    blob4 = weights.pack_blob(raw)
    g4 = gallery_rows(N, 0, N)
    frames = synth_frames(B, H, W, K, 4321)
    lanes = []
    for _ in range(L):                          # independent handles: the host-side merge of one batch runs under the
        e_ = native.Engine(0, max_batch=B, max_faces=K, max_h=H, max_w=W, profile=(L == 1))     # other batch's kernels
        e_.load_weights(blob4)
        e_.gallery_set(g4)
        e_.upload_frames(frames)
        lanes.append(e_)
    del g4
    eng = lanes[0]
    # synthetic weights: calibrate the score threshold so that about K faces per frame survive the merge
    probe = eng.detect_resident((H, W), max_faces=64, det_thresh=1e-6, nms_iou=0.4)
    kth = np.sort(probe["scores"], axis=1)[:, ::-1][:, K - 1]
    thr = float(np.clip(np.median(kth[kth > 0]) if np.any(kth > 0) else 0.5, 1e-4, 0.9999))

    def step(e_):
        per = []
        for sc in scales:
            hw = pyramid.scaled_size(H, W, sc)
            per.append((hw, e_.detect_resident(hw, max_faces=64, det_thresh=thr, nms_iou=0.4)))
        boxes, kps, scores, counts = pyramid.merge_scales(per, (H, W), K, 0.4)
        return e_.finish_faces(boxes, kps, scores, counts, K)

    import threading
    faces_of = [0] * L

    def lane_loop(i, counter, n_steps):
        with lanes[i].sequence():
            while True:
                with counter["lock"]:
                    if counter["next"] >= n_steps:
                        return
                    counter["next"] += 1
                faces_of[i] += int(step(lanes[i])["counts"].sum())

    lane_errors = []

    def guarded_loop(i, counter, n_steps):        # a lane thread's exception must fail the bench
        try:
            lane_loop(i, counter, n_steps)
        except BaseException as ex:                # noqa: BLE001
            lane_errors.append(ex)

    def run_steps(n_steps):
        counter = {"next": 0, "lock": threading.Lock()}
        th = [threading.Thread(target=guarded_loop, args=(i, counter, n_steps)) for i in range(L)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        if lane_errors:
            raise lane_errors[0]

    # the stage times come from one batch at a time on lane 0 (kernels of two lanes share the chip otherwise)
    single = None
    if L > 1:
        eng.set_profile(True)
        with eng.sequence():
            step(eng)
            eng.reset_counters()
            eng.synchronize()
            t_s = time.perf_counter()
            n_s = sum(int(step(eng)["counts"].sum()) for _ in range(args.steps))
            eng.synchronize()
            dt_s = time.perf_counter() - t_s
        single = {"faces_per_s": round(n_s / dt_s, 1), "ms_per_step": round(dt_s / args.steps * 1e3, 3)}
        ctr = eng.counters()
        eng.set_profile(False)
    run_steps(max(args.warmup, LANE_SETTLE_STEPS) if L > 1 else args.warmup)      # (see the headline workload: a steady two-lane state)
    for e_ in lanes:
        e_.synchronize()
        if L == 1:
            e_.reset_counters()
    faces_of = [0] * L
    t0 = time.perf_counter()
    run_steps(args.steps)
    for e_ in lanes:
        e_.synchronize()
    dt = time.perf_counter() - t0
    n_faces = sum(faces_of)
    if L == 1:
        ctr = eng.counters()
    conv_ms = ctr["ms_det_conv"] + ctr["ms_emb_conv"]
    conv_flops = ctr["det_conv_flops"] + ctr["emb_conv_flops"]
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    match_gbs = ctr["match_bytes"] / (ctr["ms_match"] * 1e-3) / 1e9 if ctr["ms_match"] > 0 else None
    line = {
        "metric": "faces/sec (detect+embed+match) on 4K pyramid @ 1M gallery (BASELINE config 4, per GPU)",
        "value": round(n_faces / dt, 2), "unit": "faces/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"{B}x{H}x{W} BGR frames resident in HBM per step, pyramid scales {scales}, threshold mode "
                               f"(det_thresh {thr:.4f} calibrated to ~{K} faces/frame, NMS 0.4, host merge across scales), "
                               f"{N}-identity fp16 gallery, host results out every step",
                   "lanes": L, "one_batch_at_a_time": single,
                   "frames_per_s": round(args.steps * B / dt, 2), "faces_per_frame_mean": round(n_faces / (args.steps * B), 2),
                   "gflop_per_frame_detect_all_scales": round(ctr["det_conv_flops"] / max(1, args.steps * B) / 1e9, 1),
                   "stage_ms_per_step": {k[3:]: round(ctr[k] / args.steps, 3) for k in
                                         ("ms_preprocess", "ms_det_conv", "ms_decode", "ms_align", "ms_emb_conv", "ms_l2norm", "ms_match")}},
        "roofline": {"bound": "mfma", "kernel": "conv family (detector at 3 scales + embedder)", "achieved": round(achieved, 2),
                     "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                     "kernel_source_sha256_16": native.kernel_source_hash(),
                     "match_kernel": {"bound": "hbm", "achieved": round(match_gbs, 1) if match_gbs else None, "peak": HBM_PEAK_GBS,
                                      "unit": "GB/s", "frac": round(match_gbs / HBM_PEAK_GBS, 4) if match_gbs else None,
                                      "bytes_per_launch": N * 512 * 2}},
    }
    os.write(json_fd, (json.dumps(line) + "\n").encode())
    for e_ in lanes:
        e_.close()


def run_rehearsal_cpu(args, json_fd, rank, world):
    """--rehearse-cpu: the multi-rank plumbing of the headline command without a GPU (gloo): process group from the
    launcher's environment, stream -> rank split, this rank's gallery shard, the one collective (all-gather of the shards
    into the full matrix), the cross-rank consistency check and rank 0's JSON line.  Nothing is measured: `value` is null."""
    import torch
    import torch.distributed as dist
    from frp_amd import dist as fdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N = args.gallery
    first, cnt = fdist.shard_range(N, rank, world)
    shard = fdist.normalize_rows_f16(gallery_rows(N, first, cnt)) if cnt else np.zeros((0, 512), np.float16)
    full = fdist.allgather_gallery(shard, N)
    gallery_sum = fdist.gallery_checksum(full)
    sums = [None] * world
    dist.all_gather_object(sums, gallery_sum)
    assert all(s_ == sums[0] for s_ in sums), f"gathered galleries differ across ranks: {sums}"
    t = torch.tensor([1.0 + rank])
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        line = {"metric": "faces/sec (detect+embed+match) on 1080p @ 100k gallery", "value": None, "unit": "faces/s",
                "n_gpus": 0, "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f16", "data": "synthetic",
                "config": {"workload": "CPU rehearsal of the multi-rank plumbing (gloo, no GPU, nothing measured)",
                           "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "gallery": N,
                           "streams": world, "streams_of_rank": [fdist.streams_of_rank(world, r, world) for r in range(world)],
                           "ranks_seen": [int(x.item()) - 1 for x in every], "gathered_gallery_checksum": gallery_sum,
                           "launcher": ("bench.py started its own ranks (torch.distributed.run, 127.0.0.1)"
                                        if os.environ.get("FRP_BENCH_SELF_LAUNCHED") == "1" else "external torch.distributed.run")}}
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config2", choices=["config2", "config4", "config5"],
                    help="config2 (default, the headline): 32 x 1080p, 100k gallery; config4: 4K pyramid, 1M gallery; "
                         "config5: 2 x 720p streams mixed into one batch per GPU, fp8 embedder (fp8 MFMA)")
    ap.add_argument("--weights", default="fp16", choices=["fp16", "fp8-mfma"],
                    help="fp8-mfma: the embedder's stage 2-4 3x3 convs run on E4M3 activations and weights (BASELINE config 5)")
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults = the driver's own command: a timed region pays ~6 ms for starting from a drained GPU - the first pair of two-lane steps
    # takes 26-29 ms instead of 22.5, FRP_BENCH_TRACE=1 shows it -, which is 6 % of a 10-step region and 3 % of a 20-step one)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--faces", type=int, default=10)
    ap.add_argument("--gallery", type=int, default=100000)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--lanes", type=int, default=2,
                    help="batches in flight per GPU in the timed region (one C-ABI handle = one stream each; 1 = strictly one "
                         "batch at a time)")
    ap.add_argument("--cpu-frames", type=int, default=6, help="frames in the CPU baseline sample (0 = skip); the default is ~10-15 s of CPU work")
    ap.add_argument("--pcie-steps", type=int, default=-1,
                    help="steps of the host-to-host side measurement (-1 = as many as --steps, 0 = skip)")
    ap.add_argument("--jpeg-steps", type=int, default=-1,
                    help="steps of the JPEG-bytes-in side measurement on the lanes (-1 = as many as --steps, 0 = skip)")
    ap.add_argument("--threshold-steps", type=int, default=-1,
                    help="steps of the threshold-mode (NMS on, ragged face counts) side measurement (-1 = --steps, 0 = skip)")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="no GPU: run only the multi-rank plumbing of this command (launcher, rendezvous, stream -> rank split, gallery "
                         "shard all-gather, JSON line) on the gloo backend; `value` is null.  What tests/test_bench_launcher.py drives")
    args = ap.parse_args()

    # Everything except the final JSON line goes to stderr: RCCL prints a version banner on the
    # stdout file descriptor when its communicator is created.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.workload == "config4":
        run_config4(args, json_fd)
        return
    if args.workload == "config5":               # 16 streams over 8 GPUs = 2 per GPU, their frames mixed into one batch
        args.height, args.width, args.weights = 720, 1280, "fp8-mfma"
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
            sys.exit(2)
    if args.rehearse_cpu:
        run_rehearsal_cpu(args, json_fd, rank, world)
        return
    dist = None
    # FRP_FORCE_DIST=1 exercises the RCCL path (process group, gallery all-gather, device hand-off)
    # even with one rank -- used to rehearse the multi-GPU code on a one-GPU box
    use_dist = world > 1 or os.environ.get("FRP_FORCE_DIST") == "1"
    if use_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    B, K, N, H, W = args.batch, args.faces, args.gallery, args.height, args.width
    raw = weights.make_synthetic_raw(7)
    blob = weights.pack_blob(raw, weight_format=args.weights)
    # lanes: independent handles on this GPU (private stream, buffers, weights, gallery copy each).  Lane 0 is also the
    # engine of the single-lane measurements (per-kernel roofline, stage times, side lines).
    L = max(1, args.lanes)
    lanes = [native.Engine(local_rank, max_batch=B, max_faces=K, max_h=H, max_w=W, profile=True) for _ in range(L)]
    eng = lanes[0]
    for e_ in lanes:
        e_.load_weights(blob)

    # gallery: each rank owns rows [r*N/R, (r+1)*N/R); one RCCL all-gather replicates it (setup only)
    if use_dist:
        import torch
        from frp_amd import dist as fdist
        # the collective runs on the LIBRARY's own RCCL communicator (frp_dist_* / frp_gallery_allgather: no torch tensor on the data
        # plane; torch.distributed carries the 128-byte id and the timing reductions).  FRP_DIST_TORCH=1, or a failed native
        # initialisation, takes the torch.distributed collective into the same reserved snapshot - the line says which one ran.
        gather_path = "native RCCL (frp_gallery_allgather)"
        try:
            if os.environ.get("FRP_DIST_TORCH") == "1":
                raise RuntimeError("FRP_DIST_TORCH=1")
            fdist.native_allgather_gallery(lanes, N, lambda first, cnt: gallery_rows(N, first, cnt), dist.get_rank(), dist.get_world_size(),
                                           fdist.share_id_over_torch)
        except Exception as ex_:       # (every rank takes the same branch: the id exchange and the init are collectives that fail together)
            gather_path = f"torch.distributed all_gather_into_tensor ({type(ex_).__name__}: {ex_})"
            fdist.allgather_gallery_into_engine(lanes, N, lambda first, cnt: gallery_rows(N, first, cnt), local_rank)
        # every rank must now hold the same matrix: checksum of the committed snapshot (device memory), compared across ranks
        g_view = torch.as_tensor(fdist._DevicePtr(eng.gallery_device_ptr(), N, 512), device=torch.device("cuda", local_rank))
        gallery_sum = fdist.gallery_checksum(g_view)
        sums = [None] * dist.get_world_size()
        dist.all_gather_object(sums, gallery_sum)
        assert all(s_ == sums[0] for s_ in sums), f"gathered galleries differ across ranks: {sums}"
        rccl_ranks = dist.get_world_size()
    else:
        g_ = gallery_rows(N, 0, N)
        for e_ in lanes:
            e_.gallery_set(g_)
        del g_
    assert all(e_.gallery_size() == N for e_ in lanes)

    if args.workload == "config5":     # 16 synthetic streams over 8 GPUs: this rank's two (stream s -> rank s mod R), mixed
        from frp_amd import dist as fdist_, mixer as mx          # frame by frame into one batch by the stream mixer
        my_streams = fdist_.streams_of_rank(2 * max(1, world), rank, max(1, world))
        caps = {s_: mx.SyntheticStream(synth_frames(B // 2, H, W, K, 1234 + s_)) for s_ in my_streams}
        mixer_ = mx.StreamMixer(caps, batch=B, buffers=[eng.host_frames(B, H, W)])
        frames, meta_ = next(iter(mixer_))
        mixer_.close()
        assert [m_[0] for m_ in meta_] == [my_streams[i_ % 2] for i_ in range(B)]
        frames = frames.copy()
    else:
        frames = synth_frames(B, H, W, K, 1234 + rank)
    flags = native.FLAG_FORCED_K
    for e_ in lanes:
        e_.upload_frames(frames)       # inputs resident in HBM (one copy per lane) before the timed region
        for _ in range(args.warmup):
            e_.process_resident(K, flags=flags)
        e_.synchronize()
        e_.reset_counters()
        e_.set_profile(L == 1)         # stage timers only where one stream owns the chip (their intervals overlap otherwise)

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
        for e_ in lanes:
            e_.synchronize()

    # Per-kernel measurement (before the timed region): K steps one batch at a time on lane 0 with the stage timers on
    # (HIP events on the stream the kernels run on).  With one lane the timed region itself is that measurement.
    single = None
    if L > 1:
        eng.set_profile(True)
        eng.process_resident(K, flags=flags)      # (first use of the timers: untimed)
        eng.fetch_results()
        eng.reset_counters()
        eng.synchronize()
        t_s = time.perf_counter()
        for _ in range(args.steps):
            eng.process_resident(K, flags=flags)
            res = eng.fetch_results()
        eng.synchronize()
        dt_s = time.perf_counter() - t_s
        single = {"faces_per_s": round(args.steps * B * K / dt_s, 1), "ms_per_step": round(dt_s / args.steps * 1e3, 3)}
        ctr = eng.counters()
        eng.set_profile(False)

    # Timed region (contract: inputs resident in HBM when it starts): K steps, each ONE pass of the hot path over the
    # resident batch with its results brought back to host memory (boxes, landmarks, scores, counts, 512-d embeddings,
    # match ids and cosines - what the reference's loop hands on, routes/camera.py:243-259).
    # With L lanes there are L batches in flight: one host thread per lane (ctypes drops the GIL in the C calls) takes the
    # next of the K steps from a shared counter, submits it on its lane and fetches its results - every one of the K
    # steps is submitted AND fetched between the brackets.  (Fetching in a fixed round-robin order from ONE thread lost
    # the overlap whenever the hardware scheduler favoured one queue for a while: the other lane's batch finished late,
    # the favoured lane sat idle until its turn came.)
    import threading
    last = [None] * L

    lane_errors = []

    def guarded(fn):
        """a lane thread's exception must fail the bench, not silently drop the steps that thread had taken"""
        def run(*a):
            try:
                fn(*a)
            except BaseException as ex:      # noqa: BLE001
                lane_errors.append(ex)
        return run

    def join_lanes(th):
        for x in th:
            x.start()
        for x in th:
            x.join()
        if lane_errors:
            raise lane_errors[0]

    step_trace = [] if os.environ.get("FRP_BENCH_TRACE") else None      # (lane, step taken, submitted, fetched) in seconds, to stderr

    def lane_loop(i, counter, n_steps):
        while True:
            with counter["lock"]:
                if counter["next"] >= n_steps:
                    return
                mine = counter["next"]
                counter["next"] += 1
            ta = time.perf_counter()
            lanes[i].process_resident(K, flags=flags)
            tb = time.perf_counter()
            last[i] = lanes[i].fetch_results()
            if step_trace is not None:
                step_trace.append((i, mine, ta, tb, time.perf_counter()))

    def run_steps(n_steps):
        counter = {"next": 0, "lock": threading.Lock()}
        if L == 1:
            lane_loop(0, counter, n_steps)
            return
        join_lanes([threading.Thread(target=guarded(lane_loop), args=(i, counter, n_steps)) for i in range(L)])

    if L > 1:
        # the lanes' own warm-up: steps submitted exactly like the timed ones - W of them, and never fewer than LANE_SETTLE_STEPS: the
        # first ~10 two-lane steps after the one-batch-at-a-time phase run up to 7 % slower than the steady state (measured: at W = 5
        # four runs on one box gave 26.4 / 27.0 / 28.3 / 28.4 k faces/s, at W = 20 28.3 / 28.4 / 28.4 / 28.5 k, with the same per-kernel
        # figures) - `value` is a steady-state rate, so the region starts when the state is steady.  Reported as config.lane_warmup_steps.
        run_steps(max(args.warmup, LANE_SETTLE_STEPS))
    barrier()
    import gc
    gc_was = gc.isenabled()
    if not os.environ.get("FRP_BENCH_GC"):          # (as timeit does: a collection of the interpreter's heap inside a 0.2 s region is not the pipeline's time)
        gc.collect()
        gc.disable()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    if step_trace is not None:
        for (i_, m_, ta_, tb_, tc_) in sorted(step_trace, key=lambda r_: r_[2]):
            if ta_ >= t0:
                print(f"trace lane {i_} step {m_:3d}: taken +{(ta_ - t0) * 1e3:8.2f} ms, submitted after {(tb_ - ta_) * 1e3:6.2f}, results after {(tc_ - ta_) * 1e3:7.2f}", file=sys.stderr)
        print(f"trace region {dt * 1e3:.2f} ms", file=sys.stderr)
    per_rank = None
    if dist is not None:
        import torch
        t = torch.tensor([dt], device="cuda")
        every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(every, t)
        per_rank = [round(args.steps * B * K / float(x.item()), 1) for x in every]     # each rank's own clock
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # The same with HOST frames every step (never `value`): each lane's thread uploads its batch from page-locked memory
    # on its own stream (the copy runs under the other lane's kernels), processes it and fetches the results.
    h2h_lanes = None
    if L > 1 and args.pcie_steps != 0 and rank == 0:
        n_h = args.steps if args.pcie_steps < 0 else args.pcie_steps
        pinned = [[e_.host_frames(B, H, W) for _ in range(2)] for e_ in lanes]
        for pp_ in pinned:
            for p_ in pp_:
                p_[...] = frames
        for i_, e_ in enumerate(lanes):              # prime each lane's overlapped-ingest loop (as in the one-lane side line)
            e_.upload_frames_async(pinned[i_][0])
            e_.swap_frames()
            e_.upload_frames_async(pinned[i_][1])

        def lane_loop_h2h(i, counter):
            j = 0
            while True:
                with counter["lock"]:
                    if counter["next"] >= n_h:
                        return
                    counter["next"] += 1
                lanes[i].process_resident(K, flags=flags)
                last[i] = lanes[i].fetch_results()
                lanes[i].swap_frames()                               # the batch uploaded during this step becomes resident
                lanes[i].upload_frames_async(pinned[i][j & 1])       # next batch: copy stream, under both lanes' kernels
                j += 1

        for e_ in lanes:
            e_.synchronize()
        counter = {"next": 0, "lock": threading.Lock()}
        t_h = time.perf_counter()
        join_lanes([threading.Thread(target=guarded(lane_loop_h2h), args=(i, counter)) for i in range(L)])
        for e_ in lanes:
            e_.synchronize()
        dt_h = time.perf_counter() - t_h
        h2h_lanes = {"faces_per_s": round(n_h * B * K / dt_h, 1), "ms_per_step": round(dt_h / n_h * 1e3, 3), "steps": n_h,
                     "mode": f"{L} lanes, each running the overlapped-ingest loop (page-locked host frames, H2D on the lane's copy "
                             "stream under the kernels of both lanes, host results out every step)"}
    # JPEG bytes in (never `value`): every step hands the lane a batch of baseline JPEG stills of the same frames
    # (what routes/face.py:177-185 receives); frp_upload_jpeg_async decodes the bit streams on host threads while the lane's
    # kernels of the current batch run, the pixels are produced on the copy stream.  Detection runs on the DECODED frames
    # (lossy: not the same pixels as the resident batch), forced K as everywhere in this bench.
    jpeg_lanes = None
    if L > 1 and args.jpeg_steps != 0 and rank == 0:
        try:
            import io
            from PIL import Image
            n_j = args.steps if args.jpeg_steps < 0 else args.jpeg_steps
            jpegs = []
            for f_ in frames:
                b_ = io.BytesIO()
                Image.fromarray(np.ascontiguousarray(f_[..., ::-1])).save(b_, "JPEG", quality=90)
                jpegs.append(b_.getvalue())
            for e_ in lanes:
                e_.upload_jpeg_async(jpegs)
                e_.swap_frames()
                e_.process_resident(K, flags=flags)
                e_.fetch_results()

            def lane_loop_jpeg(i, counter):
                while True:
                    with counter["lock"]:
                        if counter["next"] >= n_j:
                            return
                        counter["next"] += 1
                    lanes[i].process_resident(K, flags=flags)
                    lanes[i].upload_jpeg_async(jpegs)                    # next batch: host entropy decode under this batch's kernels
                    last[i] = lanes[i].fetch_results()
                    lanes[i].swap_frames()

            for e_ in lanes:
                e_.upload_jpeg_async(jpegs)
                e_.swap_frames()
                e_.synchronize()
            counter = {"next": 0, "lock": threading.Lock()}
            t_j = time.perf_counter()
            join_lanes([threading.Thread(target=guarded(lane_loop_jpeg), args=(i, counter)) for i in range(L)])
            for e_ in lanes:
                e_.synchronize()
            dt_j = time.perf_counter() - t_j
            jpeg_lanes = {"faces_per_s": round(n_j * B * K / dt_j, 1), "frames_per_s": round(n_j * B / dt_j, 1),
                          "ms_per_step": round(dt_j / n_j * 1e3, 3), "steps": n_j,
                          "jpeg_bytes_per_frame": int(sum(map(len, jpegs)) / len(jpegs)),
                          "mode": f"{L} lanes; baseline JPEG bytes in (quality 90, 4:2:0), bit streams decoded on <= 16 host threads per "
                                  "lane under the lane's kernels, IDCT / upsampling / colour on the copy stream, host results out every step"}
            for e_ in lanes:
                e_.upload_frames(frames)
        except ImportError:
            jpeg_lanes = None
    # Threshold mode on the lanes (never `value`): score threshold + NMS + ragged face counts, the path the reference's
    # loop runs; one lane's mid-pipeline host round trip (the 4-byte face count) hides under the other lane's kernels.
    thr_lanes = None
    if L > 1 and args.threshold_steps != 0 and rank == 0:
        n_t = args.steps if args.threshold_steps < 0 else args.threshold_steps
        probe = lanes[0].detect(frames, max_faces=64, det_thresh=1e-6, nms_iou=0.4)
        kth = np.sort(probe["scores"], axis=1)[:, ::-1][:, min(K, 63) - 1]
        dthr = float(np.clip(np.median(kth[kth > 0]) if np.any(kth > 0) else 0.5, 1e-4, 0.9999))
        faces_l = [0] * L
        for e_ in lanes:
            e_.upload_frames(frames)
            e_.process_resident(K, det_thresh=dthr, nms_iou=0.4, flags=0)
            e_.fetch_results()

        def lane_loop_thr(i, counter):
            while True:
                with counter["lock"]:
                    if counter["next"] >= n_t:
                        return
                    counter["next"] += 1
                lanes[i].process_resident(K, det_thresh=dthr, nms_iou=0.4, flags=0)
                faces_l[i] += int(lanes[i].fetch_results()["counts"].sum())

        counter = {"next": 0, "lock": threading.Lock()}
        t_t = time.perf_counter()
        join_lanes([threading.Thread(target=guarded(lane_loop_thr), args=(i, counter)) for i in range(L)])
        dt_t = time.perf_counter() - t_t
        thr_lanes = {"faces_per_s": round(sum(faces_l) / dt_t, 1), "frames_per_s": round(n_t * B / dt_t, 1),
                     "ms_per_step": round(dt_t / n_t * 1e3, 3), "steps": n_t, "det_thresh": round(dthr, 6), "nms_iou": 0.4,
                     "mode": f"{L} lanes, resident frames, score threshold + NMS + ragged face counts, host results out every step"}
        for e_ in lanes:
            e_.upload_frames(frames)
    # The boundary the reference's callers use (never `value`): FaceService.process_stream from HOST frames (page-locked
    # capture buffers, overlapped upload per lane) to the per-frame lists of per-face dicts the route hands on
    # (routes/camera.py:243-259) - name lookup, distance, bucket and threshold included - on the same two lanes, in
    # threshold mode (the service API has no forced-K switch).  Compare with threshold_mode_lanes (engine level, resident).
    svc_line = None
    if L > 1 and args.threshold_steps != 0 and rank == 0 and thr_lanes is not None:
        from frp_amd.face_service import FaceService
        n_s = args.steps if args.threshold_steps < 0 else args.threshold_steps
        svc = FaceService(engine=lanes[0], second_engine=lanes[1])
        svc.ENCODINGS.adopt_device([f"id{i:07d}" for i in range(N)])
        # capture buffers: page-locked (FaceService.frame_buffer), 3 x lanes + 2 in rotation as a capture thread would fill them
        NBUF = 3 * L + 2
        bufs = [svc.frame_buffer(B, H, W) for _ in range(NBUF)]
        for b_ in bufs:
            b_[...] = frames
        for _ in svc.process_stream((bufs[i_ % NBUF] for i_ in range(2)), max_faces=K, det_thresh=thr_lanes["det_thresh"]):
            pass
        # a stream long enough that the pipeline's fill (the first uploads have nothing to hide under: the first result arrives
        # after ~2 steps) is not a tenth of what is timed; both the whole stream and its steady part are reported
        n_s = max(n_s, 40)
        t_v = time.perf_counter()
        n_faces_v = n_match_v = 0
        arrivals, faces_at = [], []
        for per_frame in svc.process_stream((bufs[i_ % NBUF] for i_ in range(n_s)), max_faces=K, det_thresh=thr_lanes["det_thresh"]):
            for faces in per_frame:
                n_faces_v += len(faces)
                n_match_v += sum(1 for f_ in faces if f_["target"] is not None)
            arrivals.append(time.perf_counter() - t_v)
            faces_at.append(n_faces_v)
        dt_v = time.perf_counter() - t_v
        assert n_match_v == n_faces_v > 0
        w_ = 3 * L
        steady = (faces_at[-1] - faces_at[w_ - 1]) / max(1e-9, arrivals[-1] - arrivals[w_ - 1])
        svc_line = {"faces_per_s": round(n_faces_v / dt_v, 1), "frames_per_s": round(n_s * B / dt_v, 1),
                    "ms_per_step": round(dt_v / n_s * 1e3, 3), "steps": n_s,
                    "first_result_ms": round(arrivals[0] * 1e3, 2),
                    "steady_faces_per_s": round(steady, 1),
                    "fraction_of_engine_threshold_mode_lanes": round((n_faces_v / dt_v) / max(1e-9, thr_lanes["faces_per_s"]), 3),
                    "steady_fraction_of_engine_threshold_mode_lanes": round(steady / max(1e-9, thr_lanes["faces_per_s"]), 3),
                    "mode": "FaceService.process_stream: host frames in page-locked capture buffers (FaceService.frame_buffer; a lane uploads its next batch on its copy stream under the running batch's kernels "
                            "and builds the previous batch's dicts there too), list of per-face dicts out "
                            "(target name, distance, cosine, confidence bucket, match flag, bbox, kps, score, 512-d embedding), 2 lanes, threshold mode; "
                            f"whole stream of {n_s} batches from a cold pipeline; steady = after the first {w_} results"}
        for e_ in lanes:
            e_.upload_frames(frames)
    done_ = [r_ for r_ in last if r_ is not None]
    assert done_ and all(np.all(r_["counts"] == K) for r_ in done_)
    res = done_[0]
    if L == 1:
        ctr = eng.counters()
    for e_ in lanes[1:]:
        e_.close()
    lanes = lanes[:1]

    # PCIe-inclusive rate (never `value`): the same batch handed over as HOST frames every step.  Two page-locked
    # staging buffers; the copy of step t+1 runs on the library's copy stream while step t is processed.
    if args.pcie_steps < 0:
        args.pcie_steps = args.steps
    if args.threshold_steps < 0:
        args.threshold_steps = args.steps
    eng.set_profile(False)      # the side loops below run as a caller would: no per-stage events, no forced sync per call
    pcie = None
    if args.pcie_steps > 0 and rank == 0:
        stage = [eng.host_frames(B, H, W) for _ in range(2)]
        for s_ in stage:
            s_[...] = frames
        eng.upload_frames_async(stage[0])
        eng.swap_frames()
        eng.upload_frames_async(stage[1])
        eng.process_resident(K, flags=flags)             # warm-up of the overlapped loop
        eng.fetch_results()
        eng.swap_frames()
        t1 = time.perf_counter()
        for i in range(args.pcie_steps):
            eng.upload_frames_async(stage[i & 1])
            eng.process_resident(K, flags=flags)
            r2 = eng.fetch_results()                     # host results of every step
            eng.swap_frames()
        eng.synchronize()
        dt2 = time.perf_counter() - t1
        assert np.all(r2["counts"] == K)
        pcie = {"faces_per_s": round(args.pcie_steps * B * K / dt2, 1), "ms_per_step": round(dt2 / args.pcie_steps * 1e3, 3),
                "steps": args.pcie_steps,
                "mode": "SURVEY 8(d) end to end: host u8 frames in (page-locked, H2D on a copy stream overlapped with the previous "
                        "step), host results out every step"}

    # Threshold mode (never `value`): the path the reference's loop runs (camera.py:232-259) - score threshold, NMS, a
    # ragged number of faces per frame, and therefore ONE host round trip in the middle of the pipeline for the 4-byte
    # face count (frp_api.cpp:run_faces).  With synthetic weights the scores mean nothing, so the threshold is
    # calibrated on this batch: the K-th highest NMS survivor of the median frame.
    thr = None
    if args.threshold_steps > 0 and rank == 0:
        probe = eng.detect(frames, max_faces=64, det_thresh=1e-6, nms_iou=0.4)
        kth = np.sort(probe["scores"], axis=1)[:, ::-1][:, min(K, 63) - 1]
        det_thresh = float(np.clip(np.median(kth[kth > 0]) if np.any(kth > 0) else 0.5, 1e-4, 0.9999))
        eng.upload_frames(frames)
        eng.process_resident(K, det_thresh=det_thresh, nms_iou=0.4, flags=0)
        eng.fetch_results()
        t2 = time.perf_counter()
        n_faces = 0
        for _ in range(args.threshold_steps):
            eng.process_resident(K, det_thresh=det_thresh, nms_iou=0.4, flags=0)
            r3 = eng.fetch_results()
            n_faces += int(r3["counts"].sum())
        eng.synchronize()
        dt3 = time.perf_counter() - t2
        cnt = r3["counts"]
        thr = {"faces_per_s": round(n_faces / dt3, 1), "frames_per_s": round(args.threshold_steps * B / dt3, 1),
               "ms_per_step": round(dt3 / args.threshold_steps * 1e3, 3), "steps": args.threshold_steps,
               "det_thresh": round(det_thresh, 6), "nms_iou": 0.4,
               "faces_per_frame": {"min": int(cnt.min()), "mean": round(float(cnt.mean()), 2), "max": int(cnt.max())},
               "mode": "resident frames, score threshold + NMS + ragged face counts (one mid-pipeline host sync for the face "
                       "count), host results out every step"}

    if rank == 0:
        faces_total = world * args.steps * B * K
        conv_ms = ctr["ms_det_conv"] + ctr["ms_emb_conv"]
        conv_flops = ctr["det_conv_flops"] + ctr["emb_conv_flops"]
        launches = ctr["det_conv_launches"] + ctr["emb_conv_launches"]
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        f8 = args.weights == "fp8-mfma"
        out = {
            "metric": ("faces/sec (detect+embed+match) on 720p mixed streams @ 100k gallery, fp8 embedder (BASELINE config 5, per GPU)"
                       if args.workload == "config5" else "faces/sec (detect+embed+match) on 1080p @ 100k gallery"),
            "value": round(faces_total / dt, 2),
            "unit": "faces/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f8" if f8 else "f16", "data": "synthetic",
            "config": {"workload": f"{B}x{H}x{W} BGR frames per GPU per step resident in HBM"
                                   + (" (2 camera streams interleaved)" if args.workload == "config5" else "") +
                                   f", host results out every step, "
                                   + (f"{L} batches in flight per GPU (one handle = one stream each, used round-robin), " if L > 1 else "") +
                                   f"forced K={K} faces/frame, "
                                   f"{N}-identity fp16 gallery, FRPDet detector fp16 + ArcFace IResNet-100 "
                                   + ("with E4M3 activations and weights on the fp8 MFMA for the 3x3 stride-1 convs of stages 2-4 "
                                      "(residual stream, stage 1, strided convs, FC: fp16)" if f8 else "fp16") + " (synthetic seeded weights)",
                       "frames_per_s": round(world * args.steps * B / dt, 2),
                       "batch_frames": B, "faces_per_frame": K, "gallery": N, "streams": world,
                       "lanes": L, "lane_warmup_steps": (max(args.warmup, LANE_SETTLE_STEPS) if L > 1 else args.warmup), "one_batch_at_a_time": single,
                       "host_to_host": pcie, "host_to_host_lanes": h2h_lanes, "jpeg_to_host_lanes": jpeg_lanes,
                       "threshold_mode": thr, "threshold_mode_lanes": thr_lanes, "service_api": svc_line,
                       "gflop_per_frame_detect": round(ctr["det_conv_flops"] / max(1, ctr["frames"]) / 1e9, 2),
                       "gflop_per_face_embed": round(ctr["emb_conv_flops"] / max(1, ctr["faces"]) / 1e9, 3),
                       "stage_ms_per_step": {k[3:]: round(ctr[k] / args.steps, 3) for k in
                                             ("ms_preprocess", "ms_det_conv", "ms_decode", "ms_align", "ms_emb_conv",
                                              "ms_l2norm", "ms_match")}},
            "roofline": {"bound": "mfma", "kernel": "conv3x3_lean_kernel + conv3x3_wino2_kernel + conv3x3_c64_kernel + conv_mfma_kernel + stem12_u8_kernel + emb_stem_kernel (conv family, all detector+embedder launches; FLOPs = the direct convolution's algorithmic count, also for the Winograd launches)",
                         "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                         "traffic": pmc_conv_traffic_per_launch(launches // max(1, args.steps)),
                         "traffic_source": PMC_SUMMARY + " (same native sources: hash-checked)",
                         "kernel_source_sha256_16": native.kernel_source_hash(),
                         "launches_per_step": launches // max(1, args.steps),
                         "avg_launch_us": round(conv_ms * 1e3 / max(1, launches), 2),
                         "algorithmic_gflop_per_step": round(conv_flops / args.steps / 1e9, 1),
                         "measured_on": ("the timed region" if L == 1 else
                                         "lane 0 running the same steps one batch at a time just before the timed region (config.one_batch_at_a_time): with "
                                         f"{L} batches in flight kernels of different streams share the chip and their durations overlap"),
                         "chip_level_timed_region": None if L == 1 else {
                             "achieved": round(conv_flops / args.steps / (dt / args.steps) / 1e12, 2),
                             "frac": round(conv_flops / args.steps / (dt / args.steps) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                             "note": "conv FLOPs of a step / wall time of a step in the timed region (charges the conv family "
                                     "for the decode / align / match time as well: a lower bound)"},
                         "match_hbm_GBs": round(ctr["match_bytes"] / (ctr["ms_match"] * 1e-3) / 1e9, 1) if ctr["ms_match"] > 0 else None},
        }
        if f8:      # the embedder family against the fp8 matrix peak (its fp16 launches included: a lower bound)
            MFMA_PEAK_FP8 = 5000.0
            emb_tf = ctr["emb_conv_flops"] / (ctr["ms_emb_conv"] * 1e-3) / 1e12 if ctr["ms_emb_conv"] > 0 else 0.0
            out["roofline_fp8_embedder"] = {
                "bound": "mfma", "kernel": "embedder conv family (conv3x3_lean_kernel<F8> on E4M3 operands + its fp16 launches)",
                "achieved": round(emb_tf, 2), "peak": MFMA_PEAK_FP8, "unit": "TFLOP/s", "frac": round(emb_tf / MFMA_PEAK_FP8, 4),
                "fp8_share_of_embedder_flops": round(ctr["f8_conv_flops"] / max(1.0, ctr["emb_conv_flops"]), 4),
                "fp8_launches_per_step": ctr["f8_conv_launches"] // max(1, args.steps)}
        if dist is not None:       # (single-process runs keep the line byte-compatible: no extra keys)
            out["config"]["rccl_ranks"] = rccl_ranks
            out["config"]["gallery_allgather"] = gather_path
            out["config"]["faces_per_s_per_rank"] = per_rank
            out["config"]["gathered_gallery_checksum"] = gallery_sum
            out["config"]["launcher"] = ("bench.py started its own ranks (torch.distributed.run, 127.0.0.1)"
                                         if os.environ.get("FRP_BENCH_SELF_LAUNCHED") == "1" else "external torch.distributed.run")
        if world == 1 and args.cpu_frames > 0:
            out["cpu_baseline"] = cpu_baseline(raw, frames, K, N, args.cpu_frames)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for e_ in lanes:
        e_.close()


if __name__ == "__main__":
    main()
