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
            block = -(-n // w)
            assert all(f == min(r * block, n) for r, (f, _) in enumerate(spans))     # shard r starts at r * ceil(n / w): the
            assert all(c == block for _, c in spans[:-1] if c and _ + c < n)         # all-gather lands it in its final place
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
    assert full.shape[0] == n_total and full._base is not None and full._base.shape[0] == world * (-(-n_total // world))
    # gather INTO a caller-provided matrix (what the engine's reserved snapshot is on a GPU)
    target = torch.full((world * (-(-n_total // world)), 512), 7.0, dtype=torch.float16)
    view = fd.allgather_gallery(shard, n_total, out=target)
    assert view.data_ptr() == target.data_ptr() and torch.equal(view, full)
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


def _replica_worker(rank, world, port, out_dir):
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import frp_amd_loader  # noqa: F401
    from fake_engine import FakeEngine
    from frp_amd import dist as fd
    from frp_amd.face_service import FaceService
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["FACE_BACKUP_DIR"] = os.path.join(out_dir, f"backups_{rank}")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n0 = 37
    rng = np.random.default_rng(5)
    base = rng.standard_normal((n0, 512)).astype(np.float32)
    extra = {k: rng.standard_normal(512).astype(np.float32) for k in ("zed", "amy", "id_5", "bob")}
    queries = rng.standard_normal((6, 512)).astype(np.float32)
    first, cnt = fd.shard_range(n0, rank, world)
    full = fd.allgather_gallery(fd.normalize_rows_f16(base[first:first + cnt]), n0).float().numpy()
    names = fd.broadcast_names([f"id_{i}" for i in range(n0)] if rank == 0 else [], src=0)
    lanes = [FakeEngine(), FakeEngine()]                   # two lanes per rank: every update must reach both copies
    svc = FaceService(engine=lanes[0], second_engine=lanes[1])
    svc.ENCODINGS.set_bulk(names, full)
    rep = fd.GalleryReplicator(svc.ENCODINGS)
    log = []

    def snapshot(step):
        G = svc.ENCODINGS
        mats = [e.gallery_get() for e in lanes]
        assert np.array_equal(mats[0], mats[1])
        res = [[(m["target"], round(m["distance"], 12)) for m in svc.compare_faces(q)][:5] for q in queries]
        log.append({"step": step, "gen": rep.generation, "names": G.names(), "rows": [G.row_of(n) for n in G.names()],
                    "matrix_sum": float(mats[0].sum()), "matrix": mats[0].tolist() if step == 5 else None, "top": res})

    for step in range(6):                                  # the stream loop: one sync per batch, on every rank
        if step == 2 and rank == 1:
            rep.store("zed", extra["zed"])                 # enrolled on rank 1 mid-stream
            assert "zed" not in svc.ENCODINGS              # ... visible nowhere before the next sync
        if step == 2 and rank == 0:
            rep.delete("id_3")
            rep.store("id_5", extra["id_5"])               # an update of an existing identity
        if step == 4 and rank == 0:
            rep.store("amy", extra["amy"])
            rep.store("bob", extra["bob"])
            rep.delete("bob")
        applied = rep.sync()
        assert applied == {2: 3, 4: 3}.get(step, 0)
        snapshot(step)
    assert rep.generation == 2 and rep.applied == 6
    with open(os.path.join(out_dir, f"replica_{rank}.json"), "w") as f:
        json.dump(log, f)
    dist.barrier()
    dist.destroy_process_group()


def test_gallery_updates_reach_every_rank_in_the_same_order(tmp_path):
    """run-time enrolment / deletion on a 2-rank job (gloo): updates queued on either rank are applied by the next
    collective sync on BOTH, in the same order - identical name tables, row numbering, device matrices (both lanes of
    each rank) and compare_faces results after every batch; nothing is visible before the sync that carries it"""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_replica_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    logs = [json.load(open(tmp_path / f"replica_{r}.json")) for r in range(2)]
    assert logs[0] == logs[1]
    by_step = {e["step"]: e for e in logs[0]}
    assert "zed" not in by_step[1]["names"] and "zed" in by_step[2]["names"] and "id_3" not in by_step[2]["names"]
    assert "amy" in by_step[4]["names"] and "bob" not in by_step[4]["names"] and len(by_step[5]["names"]) == 38
    assert by_step[1]["gen"] == 0 and by_step[2]["gen"] == 1 and by_step[5]["gen"] == 2
    assert by_step[1]["matrix_sum"] != by_step[2]["matrix_sum"]
