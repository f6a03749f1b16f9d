"""Multi-GPU path on CPU: world_size-2 gloo run of the gallery shard all-gather and the
stream -> rank partitioning (the only distributed pieces of the hot path)."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frp_amd_loader  # noqa: E402,F401  (spawned workers re-import this module without conftest)

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from frp_amd import dist as fdist


def test_shard_ranges_partition_exactly():
    for n in (0, 1, 7, 100000, 100003):
        for w in (1, 2, 3, 8):
            spans = [fdist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert [fdist.stream_to_rank(s, 8) for s in range(16)] == list(range(8)) * 2        # config 5: 2 streams per GPU
    assert fdist.streams_of_rank(16, 3, 8) == [3, 11] and fdist.streams_of_rank(8, 7, 8) == [7]


def _worker(rank, world, port, n_total, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import frp_amd_loader  # noqa: F401
    from frp_amd import dist as fd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full_ref = np.random.default_rng(42).standard_normal((n_total, 512)).astype(np.float32)
    first, cnt = fd.shard_range(n_total, rank, world)
    shard = fd.normalize_rows_f16(full_ref[first:first + cnt])       # each rank only touches its own rows
    full = fd.allgather_gallery(shard, n_total)
    names = fd.broadcast_names([f"id_{i}" for i in range(n_total)] if rank == 0 else [], src=0)
    np.save(os.path.join(out_dir, f"full_{rank}.npy"), full.numpy())
    with open(os.path.join(out_dir, f"names_{rank}.txt"), "w") as f:
        f.write("\n".join(names))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [1001, 64])
def test_allgather_gallery_world2_gloo(tmp_path, n_total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    ref = fdist.normalize_rows_f16(np.random.default_rng(42).standard_normal((n_total, 512)).astype(np.float32))
    for r in range(2):
        got = np.load(tmp_path / f"full_{r}.npy")
        assert got.shape == (n_total, 512) and got.dtype == np.float16
        assert np.array_equal(got, ref)                              # every rank holds the identical full matrix
        assert (tmp_path / f"names_{r}.txt").read_text().split("\n") == [f"id_{i}" for i in range(n_total)]
