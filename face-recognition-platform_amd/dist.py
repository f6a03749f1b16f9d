"""Multi-GPU sharding of the hot path: independent camera streams, one per
GPU, and ONE collective -- an all-gather of the gallery shards at load/update time
(RCCL over xGMI on GPUs, gloo in the CPU tests).  The reference has no distributed code
(single process, ThreadPoolExecutor(4) over cameras: backend/app/routes/camera.py:30,304-305)."""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """rows [first, first+count) owned by `rank`: contiguous, sizes differ by at most 1."""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def stream_to_rank(stream_id: int, world: int) -> int:
    """camera stream s -> GPU s mod R (config 3: 8 streams / 8 GPUs; config 5: 16 -> 2 per GPU)."""
    return stream_id % world


def streams_of_rank(n_streams: int, rank: int, world: int) -> List[int]:
    return [s for s in range(n_streams) if stream_to_rank(s, world) == rank]


def normalize_rows_f16(rows: np.ndarray) -> np.ndarray:
    r = rows.astype(np.float32)
    n = np.linalg.norm(r, axis=1, keepdims=True)
    n[n == 0] = 1.0
    return (r / n).astype(np.float16)


def allgather_gallery(shard_f16, n_total: int, device=None):
    """All-gather unit fp16 shards [count_r, 512] (ragged by <=1 row) into the full
    [n_total, 512] tensor on every rank.  torch.distributed must be initialised
    (backend nccl == RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    max_rows = (n_total + world - 1) // world
    t = torch.as_tensor(shard_f16)
    if device is not None:
        t = t.to(device)
    pad = torch.zeros((max_rows, t.shape[1]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    out = torch.empty((world * max_rows, t.shape[1]), dtype=t.dtype, device=t.device)
    if dist.get_backend() == "gloo":
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        out = torch.cat(parts, 0)
    else:
        dist.all_gather_into_tensor(out, pad)
    # drop the per-rank padding
    keep = []
    for r in range(world):
        _, cnt = shard_range(n_total, r, world)
        keep.append(out[r * max_rows: r * max_rows + cnt])
    return torch.cat(keep, 0).contiguous()


def allgather_gallery_into_engine(engine, n_total: int, make_rows: Callable[[int, int], np.ndarray], local_rank: int):
    """Rank r builds rows [first, first+count) on the host, uploads them as unit fp16,
    all-gathers over RCCL and hands the device matrix to the engine (frp_gallery_set_device)."""
    import torch
    import torch.distributed as dist
    first, cnt = shard_range(n_total, dist.get_rank(), dist.get_world_size())
    shard = normalize_rows_f16(make_rows(first, cnt))
    full = allgather_gallery(shard, n_total, device=torch.device("cuda", local_rank))
    torch.cuda.synchronize()
    for e in (engine if isinstance(engine, (list, tuple)) else [engine]):     # every lane of this GPU gets its own copy
        e.gallery_set_device(full.data_ptr(), n_total)
    return full.shape[0]


def broadcast_names(names: Sequence[str], src: int = 0) -> List[str]:
    """The host-side name table travels as a Python object broadcast (not on the data path)."""
    import torch.distributed as dist
    box = [list(names) if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]
