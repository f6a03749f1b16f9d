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


class Gallery(Mapping):
    def __init__(self, engine_getter):
        self._eng = engine_getter          # callable -> native.Engine (created lazily)
        self._rows: Dict[str, int] = {}    # insertion-ordered: name -> device row
        self._names: List[Optional[str]] = []   # device row -> name
        self._lock = threading.RLock()

    # ---- Mapping view (values are fetched from the device on demand)
    def __len__(self) -> int:
        return len(self._rows)

    def __iter__(self) -> Iterator[str]:
        return iter(list(self._rows.keys()))

    def __contains__(self, name) -> bool:
        return name in self._rows

    def __getitem__(self, name: str) -> list:
        with self._lock:
            row = self._rows[name]
            return self._eng().gallery_get(row, 1)[0].astype(np.float64).tolist()

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
            e = np.asarray(emb, dtype=np.float32).reshape(-1)
            if name in self._rows:
                self._eng().gallery_update_row(self._rows[name], e)
                return True
            row = len(self._names)
            self._eng().gallery_update_row(row, e)     # row == size appends
            self._rows[name] = row
            self._names.append(name)
            return False

    def remove(self, name: str) -> bool:
        with self._lock:
            if name not in self._rows:
                return False
            row = self._rows.pop(name)
            last = len(self._names) - 1
            self._eng().gallery_remove_row(row)        # device: last row moves into `row`
            if row != last:
                moved = self._names[last]
                self._names[row] = moved
                self._rows[moved] = row                # dict position (insertion order) is unchanged
            self._names.pop()
            return True

    def set_bulk(self, names: List[str], emb: np.ndarray):
        """replace the whole gallery (startup load / all-gathered watchlist)."""
        with self._lock:
            if len(set(names)) != len(names) or len(names) != len(emb):
                raise ValueError("names must be unique and match the rows")
            self._eng().gallery_set(np.asarray(emb))
            self._rows = {n: i for i, n in enumerate(names)}
            self._names = list(names)

    def adopt_device(self, names: List[str]):
        """name table for a matrix that was installed with frp_gallery_set_device."""
        with self._lock:
            if self._eng().gallery_size() != len(names):
                raise ValueError("name table does not match the device gallery")
            self._rows = {n: i for i, n in enumerate(names)}
            self._names = list(names)

    def clear(self):
        with self._lock:
            self._eng().gallery_set(np.zeros((0, 512), np.float32))
            self._rows.clear()
            self._names.clear()
