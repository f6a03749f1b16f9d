"""Multi-GPU sharding of the hot path: independent camera streams, one per
GPU, and ONE collective -- an all-gather of the gallery shards at load/update time
(RCCL over xGMI on GPUs, gloo in the CPU tests).  The reference has no distributed code
(single process, ThreadPoolExecutor(4) over cameras: backend/app/routes/camera.py:30,304-305)."""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """rows [first, first+count) owned by `rank`: contiguous blocks of ceil(N / R) rows, the last ranks take what is left
    (possibly nothing).  Equal block offsets are what lets the all-gather land every shard at its final position of the
    full matrix - no padding between shards, no compaction copy afterwards."""
    block = (n_total + world - 1) // world if world > 0 else 0
    first = min(rank * block, n_total)
    return first, min(block, n_total - first)


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


class _DevicePtr:
    """a raw device allocation as something torch.as_tensor accepts (__cuda_array_interface__, version 2)"""

    def __init__(self, ptr: int, rows: int, cols: int):
        self.__cuda_array_interface__ = {"shape": (rows, cols), "typestr": "<f2", "data": (ptr, False), "version": 2, "strides": None}


def allgather_gallery(shard_f16, n_total: int, device=None, out=None):
    """All-gather unit fp16 shards (rank r holds rows shard_range(n_total, r, R)) into the full matrix on every rank.
    `out`: a [R * ceil(N/R), 512] fp16 tensor to gather INTO (the engine's reserved snapshot); allocated when None.
    Every shard lands at its final row offset r * ceil(N/R), so rows [0, n_total) of `out` ARE the gallery: the only
    staging is this rank's own shard padded to the block size.  torch.distributed must be initialised (backend nccl ==
    RCCL on GPUs, gloo on CPU).  -> out[:n_total] (a view)"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    block = (n_total + world - 1) // world
    t = torch.as_tensor(shard_f16)
    if device is not None:
        t = t.to(device)
    if t.shape[0] != block:                                  # the last ranks' short (or empty) shards
        pad = torch.zeros((block, t.shape[1]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        t = pad
    if out is None:
        out = torch.empty((world * block, t.shape[1]), dtype=t.dtype, device=t.device)
    if out.shape[0] != world * block or out.dtype != t.dtype or not out.is_contiguous():
        raise ValueError("all-gather target must be a contiguous [R * ceil(N/R), 512] fp16 tensor")
    if dist.get_backend() == "gloo":
        dist.all_gather(list(out.chunk(world, 0)), t.contiguous())       # chunks are row-block views of `out`
    else:
        dist.all_gather_into_tensor(out, t.contiguous())
    return out[:n_total]


def gallery_checksum(mat) -> int:
    """position-weighted checksum of a unit fp16 matrix (torch tensor, host or device): the sum of its bit patterns
    weighted by position modulo 2^61 - equal on every rank iff the all-gather left the same bytes in the same rows"""
    import torch
    v = mat.contiguous().view(torch.int16).to(torch.int64).reshape(-1) & 0xFFFF
    w = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 8191) + 1
    return int(((v * w).sum() % ((1 << 61) - 1)).item())


def allgather_gallery_into_engine(engine, n_total: int, make_rows: Callable[[int, int], np.ndarray], local_rank: int, gallery=None):
    """Rank r builds rows shard_range(n_total, r, R) on the host, uploads them as unit fp16 and all-gathers over RCCL
    STRAIGHT INTO the snapshot the first engine reserved (frp_gallery_reserve / frp_gallery_commit): the matrix exists
    once per handle, not three more times in torch tensors.  Further lanes of this GPU copy theirs from the first.
    `gallery`: the gallery.Gallery over these engines, if there is one - its exclusive lock is held from reserve to commit, so
    an enrolment on another thread waits instead of being refused by the library (which rejects every other gallery update
    while a reservation is pending: RCCL is writing into it)."""
    if gallery is not None:
        with gallery.locked():
            return allgather_gallery_into_engine(engine, n_total, make_rows, local_rank)
    import torch
    import torch.distributed as dist
    engines = list(engine) if isinstance(engine, (list, tuple)) else [engine]
    world = dist.get_world_size()
    first, cnt = shard_range(n_total, dist.get_rank(), world)
    shard = normalize_rows_f16(make_rows(first, cnt)) if cnt else np.zeros((0, 512), np.float16)
    block = (n_total + world - 1) // world
    dev = torch.device("cuda", local_rank)
    ptr = engines[0].gallery_reserve(world * block)
    try:
        out = torch.as_tensor(_DevicePtr(ptr, world * block, 512), device=dev)
        allgather_gallery(shard, n_total, device=dev, out=out)
        torch.cuda.synchronize()
    except BaseException:
        engines[0].gallery_cancel()
        raise
    engines[0].gallery_commit(n_total)
    for e in engines[1:]:                                    # every lane of this GPU matches against its own copy
        e.gallery_set_device(engines[0].gallery_device_ptr(), n_total)
    return n_total


def native_allgather_gallery(engine, n_total: int, make_rows: Callable[[int, int], np.ndarray], rank: int, world: int,
                             share_id: Callable[[bytes | None], bytes], gallery=None) -> int:
    """The same collective on the LIBRARY's own RCCL communicator (include/frp.h: frp_dist_unique_id / frp_dist_init /
    frp_gallery_allgather): no torch tensor, no torch collective on the data plane - the launcher only carries 128 bytes.
    `share_id(id_or_None) -> id`: the control channel; rank 0 calls it with a fresh id, the others with None, all get the id back
    (`share_id_over_torch` below, an MPI bcast, a file on shared storage ...).  Rank r builds rows shard_range(n_total, r, world) on
    the host; the first engine gathers straight into a reserved snapshot and commits it; further lanes of this GPU copy theirs
    from it.  The communicator stays with the engine (later watch-list reloads gather again without a new id)."""
    if gallery is not None:
        with gallery.locked():
            return native_allgather_gallery(engine, n_total, make_rows, rank, world, share_id)
    from . import native
    engines = list(engine) if isinstance(engine, (list, tuple)) else [engine]
    e0 = engines[0]
    if getattr(e0, "_dist", None) is None:
        uid = share_id(native.Engine.dist_unique_id() if rank == 0 else None)
        e0.dist_init(uid, rank, world)
    first, cnt = shard_range(n_total, rank, world)
    rows = np.ascontiguousarray(make_rows(first, cnt), dtype=np.float32) if cnt else np.zeros((0, 512), np.float32)
    e0.gallery_allgather(rows, n_total)
    for e in engines[1:]:                                    # every lane of this GPU matches against its own copy
        e.gallery_set_device(e0.gallery_device_ptr(), n_total)
    return n_total


def share_id_over_torch(payload, src: int = 0) -> bytes:
    """control channel for `native_allgather_gallery` when torch.distributed is the launcher: an object broadcast (host side)"""
    import torch.distributed as dist
    box = [payload if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def broadcast_names(names: Sequence[str], src: int = 0) -> List[str]:
    """The host-side name table travels as a Python object broadcast (not on the data path)."""
    import torch.distributed as dist
    box = [list(names) if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


class GalleryReplicator:
    """Enrolment and deletion at run time on a multi-GPU node (reference: ENCODINGS[name] = ... / del ENCODINGS[name] on the
    one process it has, backend/app/services/face_service.py:374,522).  Every rank holds the whole watch list; an update
    made on one rank must reach all of them, and in the SAME ORDER everywhere, so that every rank's device matrix, row
    numbering and name table stay identical (swap-remove makes the layout order-dependent).

    `store` / `delete` only queue the update on the calling rank.  `sync()` is a collective every rank calls at the same
    point of its loop (once per batch or per poll): the queued updates of all ranks are exchanged (all_gather_object: a
    name, an op code and the 512 fp32 values per update - a few KB, off the frame path) and applied in (rank, queue
    order) on every rank through `Gallery.put / remove`, i.e. under the gallery's exclusive lock and into every lane's
    copy.  An update is therefore visible one sync later - on all ranks at once.  `generation` counts applied rounds that
    carried at least one update and is identical on every rank."""

    def __init__(self, gallery, group=None):
        import threading
        self.gallery = gallery
        self.group = group
        self._lock = threading.Lock()
        self._pending: List[tuple] = []
        self.generation = 0
        self.applied = 0

    def store(self, name: str, emb) -> None:
        e = np.asarray(emb, dtype=np.float32).reshape(-1)
        if e.shape[0] != 512:
            raise ValueError("embedding must have 512 values")
        with self._lock:
            self._pending.append(("put", str(name), e.tobytes()))

    def delete(self, name: str) -> None:
        with self._lock:
            self._pending.append(("del", str(name), b""))

    def sync(self) -> int:
        """collective: exchange and apply the queued updates of every rank.  -> number of updates applied (all ranks)"""
        import torch.distributed as dist
        with self._lock:
            mine, self._pending = self._pending, []
        world = dist.get_world_size(self.group)
        box = [None] * world
        dist.all_gather_object(box, mine, group=self.group)
        n = 0
        for r in range(world):
            for op, name, payload in box[r]:
                if op == "put":
                    self.gallery.put(name, np.frombuffer(payload, dtype=np.float32))
                else:
                    self.gallery.remove(name)
                n += 1
        if n:
            self.generation += 1
            self.applied += n
        return n
