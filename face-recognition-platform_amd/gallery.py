"""Host name table over the device-resident gallery matrix.

Replaces `state.ENCODINGS: Dict[str, list]` (backend/app/state.py:78): the embeddings live
in HBM as unit fp16 rows [N x 512] (libfrp.so, frp_gallery_*), this class keeps the
insertion-ordered name -> row mapping the reference's dict provides (compare_faces builds
its result in dict order, face_service.py:403,414; the duplicate scan stops at the first
hit in dict order, :353-364) and a read-only Mapping view for callers that only look at
`len()` / `.keys()` / `in` (main.py:186, face_service.py:511,654).
"""
from __future__ import annotations

import threading
from collections.abc import Mapping
from typing import Dict, Iterator, List, Optional

import numpy as np


class _RWLock:
    """Writers (gallery updates, and every reader that takes `Gallery.locked()`) are exclusive and re-entrant per thread;
    shared readers (`Gallery.reading()`: the streaming lanes, which resolve device rows to names for several batches at
    once) run together.  Waiting writers hold back new readers."""

    def __init__(self):
        self._cv = threading.Condition(threading.Lock())
        self._readers = 0
        self._writer: Optional[int] = None
        self._depth = 0
        self._waiting_writers = 0

    # exclusive side: usable as `with lock:`
    def acquire(self):
        me = threading.get_ident()
        with self._cv:
            if self._writer == me:
                self._depth += 1
                return True
            self._waiting_writers += 1
            while self._readers > 0 or self._writer is not None:
                self._cv.wait()
            self._waiting_writers -= 1
            self._writer, self._depth = me, 1
            return True

    def release(self):
        with self._cv:
            self._depth -= 1
            if self._depth == 0:
                self._writer = None
                self._cv.notify_all()

    __enter__ = acquire

    def __exit__(self, *exc):
        self.release()

    # shared side
    def acquire_read(self):
        me = threading.get_ident()
        with self._cv:
            if self._writer == me:              # a reader inside its own exclusive section
                self._depth += 1
                return
            while self._writer is not None or self._waiting_writers > 0:
                self._cv.wait()
            self._readers += 1

    def release_read(self):
        me = threading.get_ident()
        with self._cv:
            if self._writer == me:
                self._depth -= 1
                return
            self._readers -= 1
            if self._readers == 0:
                self._cv.notify_all()


class _Reading:
    def __init__(self, lock: _RWLock):
        self._l = lock

    def __enter__(self):
        self._l.acquire_read()

    def __exit__(self, *exc):
        self._l.release_read()


class Gallery(Mapping):
    def __init__(self, engine_getter):
        self._eng = engine_getter          # callable -> native.Engine (created lazily)
        self._mirrors: List = []           # further engines holding a copy of the matrix (the second lane of a GPU)
        self._rows: Dict[str, int] = {}    # insertion-ordered: name -> device row
        self._names: List[Optional[str]] = []   # device row -> name
        self._lock = _RWLock()
        self.exact = False                 # the first engine keeps float64 rows as enrolled (enable_exact)
        self.dim = 0                       # width of the rows as enrolled (the reference's encodings are 128-d; device rows are padded to 512)

    def add_mirror(self, engine) -> None:
        """A second engine that receives every update from now on (two batches in flight on one GPU: each lane matches
        against its own copy).  Only while the gallery is empty: both copies are then built from the same fp32 rows by
        the same kernel and stay bit-identical."""
        with self._lock:
            if self._names:
                raise ValueError("mirrors must be added before the gallery is filled")
            self._mirrors.append(engine)

    def _engines(self):
        return [self._eng()] + list(self._mirrors)

    def reading(self):
        """Shared access for a reader that resolves device rows to names: like `locked()` it keeps updates out between
        its device call and its lookup, but several such readers (the lanes of a GPU) may be inside together.
        Not re-entrant, and a reader must not ask for `locked()` (or anything that takes it: put / remove / `G[name]`)
        while inside: the exclusive side waits for all readers, itself included."""
        return _Reading(self._lock)

    # ---- Mapping view (values are fetched from the device on demand)
    def __len__(self) -> int:
        return len(self._rows)

    def __iter__(self) -> Iterator[str]:
        return iter(list(self._rows.keys()))

    def __contains__(self, name) -> bool:
        return name in self._rows

    def __getitem__(self, name: str) -> list:
        """the stored row: as enrolled (float64) when the engine keeps exact rows (FaceService's default), else the unit fp16 row"""
        with self._lock:
            row = self._rows[name]
            eng = self._eng()
            if self.exact and hasattr(eng, "gallery_get_exact"):
                v = eng.gallery_get_exact(row, 1)[0]
                return v[: self.dim].tolist() if self.dim else v.tolist()
            return eng.gallery_get(row, 1)[0].astype(np.float64).tolist()

    def enable_exact(self) -> bool:
        """Switch the first engine to exact rows (frp_gallery_exact): float64 copies of the rows as enrolled, the operands of the
        REST-style compat path (compare_faces / find_k_nearest / batch_compare_faces / duplicate scan / cluster_faces; the streaming
        lanes keep matching on the unit fp16 rows, and their mirrors hold no exact copy).  -> whether the engine supports it"""
        with self._lock:
            eng = self._eng()
            if not hasattr(eng, "gallery_exact"):
                return False
            eng.gallery_exact(True)
            self.exact = True
            return True

    def locked(self):
        """The lock every update takes.  A reader that resolves device row indices to names (match ->
        name_of_row / rows_of) holds it across the device call AND the lookup: `remove` is a swap-remove, so a
        delete between the two would attribute a face to the identity that was moved into that row (the
        reference builds names and matrix in one call, face_service.py:403-411, and cannot mis-attribute)."""
        return self._lock

    # ---- updates
    def names(self) -> List[str]:
        return list(self._rows.keys())

    def row_of(self, name: str) -> int:
        return self._rows[name]

    def name_of_row(self, row: int) -> Optional[str]:
        return self._names[row] if 0 <= row < len(self._names) else None

    def rows_of(self, names: List[str]) -> np.ndarray:
        return np.fromiter((self._rows[n] for n in names), dtype=np.int64, count=len(names))

    def put(self, name: str, emb: np.ndarray) -> bool:
        """insert or overwrite; returns True if the name already existed."""
        with self._lock:
            e = np.asarray(emb)
            e = e.astype(np.float64 if e.dtype == np.float64 else np.float32, copy=False).reshape(-1)   # (float64 stays: exact rows)
            if not self._names:
                self.dim = int(e.shape[0])
            engines = self._engines()
            if name in self._rows:
                row = self._rows[name]
                old = None
                for i, eng in enumerate(engines):
                    try:
                        if i == 0 and len(engines) > 1:
                            old = eng.gallery_get(row, 1)[0].astype(np.float32)      # the stored unit row, for the rollback
                        eng.gallery_update_row(row, e)
                    except Exception:
                        for done in engines[:i]:            # a later copy failed: the copies must not differ by a row
                            done.gallery_update_row(row, old)
                        raise
                return True
            row = len(self._names)
            for i, eng in enumerate(engines):
                try:
                    eng.gallery_update_row(row, e)         # row == size appends
                except Exception:
                    for done in engines[:i]:
                        done.gallery_remove_row(row)
                    raise
            self._rows[name] = row
            self._names.append(name)
            return False

    def remove(self, name: str) -> bool:
        with self._lock:
            if name not in self._rows:
                return False
            row = self._rows[name]
            last = len(self._names) - 1
            engines = self._engines()
            try:
                for eng in engines:
                    eng.gallery_remove_row(row)        # device: last row moves into `row`
            except Exception:
                # a copy failed half way (the swap already happened on the earlier ones): the removal is abandoned and
                # every copy is brought back to the rows the name table still lists
                self._resync_from(engines)
                raise
            self._rows.pop(name)
            if row != last:
                moved = self._names[last]
                self._names[row] = moved
                self._rows[moved] = row                # dict position (insertion order) is unchanged
            self._names.pop()
            return True

    def _resync_from(self, engines) -> None:
        """after a failed multi-copy update: make every copy equal to the name table again (rows the table still
        lists, read back from whichever copy still has `len(names)` rows)"""
        n = len(self._names)
        src = next((e for e in engines if e.gallery_size() == n), None)
        if src is None:
            return
        rows = src.gallery_get(0, n).astype(np.float32) if n else np.zeros((0, 512), np.float32)
        for e in engines:
            if e is not src:
                e.gallery_set(rows)

    def set_bulk(self, names: List[str], emb: np.ndarray):
        """replace the whole gallery (startup load / all-gathered watchlist)."""
        with self._lock:
            if len(set(names)) != len(names) or len(names) != len(emb):
                raise ValueError("names must be unique and match the rows")
            for eng in self._engines():
                eng.gallery_set(np.asarray(emb))
            self._rows = {n: i for i, n in enumerate(names)}
            self._names = list(names)
            self.dim = int(np.asarray(emb).shape[1]) if len(names) else 0

    def adopt_device(self, names: List[str]):
        """name table for a matrix that was installed with frp_gallery_set_device."""
        with self._lock:
            if any(eng.gallery_size() != len(names) for eng in self._engines()):
                raise ValueError("name table does not match the device gallery")
            self._rows = {n: i for i, n in enumerate(names)}
            self._names = list(names)

    def clear(self):
        with self._lock:
            for eng in self._engines():
                eng.gallery_set(np.zeros((0, 512), np.float32))
            self._rows.clear()
            self._names.clear()
