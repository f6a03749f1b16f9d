"""Test double for native.Engine (tests only): the gallery/match half in float64 numpy so the
HOST logic of FaceService (ordering, dict shapes, buckets, name-table bookkeeping) can be
checked on a box without a GPU.  Not importable from the product."""
import numpy as np


class FakeEngine:
    def __init__(self):
        self.G = np.zeros((0, 512), np.float64)
        self.X = None                      # exact rows as enrolled (gallery_exact), or None
        self.canned = None

    def gallery_exact(self, on=True):
        self.X = (self.G.copy() if on else None)

    def gallery_distances(self, q):
        q = np.asarray(q, dtype=np.float64).reshape(-1, self.X.shape[1] if len(self.X) else np.asarray(q).shape[-1])
        return np.stack([np.linalg.norm(self.X - qq, axis=1) if len(self.X) else np.zeros(0) for qq in q])

    def gallery_get_exact(self, first=0, n=None):
        n = len(self.X) - first if n is None else n
        return self.X[first:first + n].copy()

    def gallery_set(self, emb):
        e = np.asarray(emb, dtype=np.float64).reshape(-1, emb.shape[1] if emb.ndim == 2 else 512)
        n = np.linalg.norm(e, axis=1, keepdims=True)
        n[n == 0] = 1
        self.G = e / n
        if self.X is not None:
            self.X = e.copy()

    def gallery_size(self):
        return len(self.G)

    def gallery_update_row(self, row, emb):
        raw = np.asarray(emb, dtype=np.float64).reshape(1, -1)
        e = raw / max(np.linalg.norm(raw), 1e-300)
        if self.G.shape[1] != e.shape[1]:
            assert len(self.G) == 0
            self.G = np.zeros((0, e.shape[1]))
            if self.X is not None:
                self.X = np.zeros((0, e.shape[1]))
        if row == len(self.G):
            self.G = np.concatenate([self.G, e])
            if self.X is not None:
                self.X = np.concatenate([self.X, raw])
        else:
            self.G[row] = e
            if self.X is not None:
                self.X[row] = raw

    def gallery_remove_row(self, row):
        last = len(self.G) - 1
        if row != last:
            self.G[row] = self.G[last]
            if self.X is not None:
                self.X[row] = self.X[last]
        self.G = self.G[:last]
        if self.X is not None:
            self.X = self.X[:last]

    def gallery_get(self, first=0, n=None):
        n = len(self.G) - first if n is None else n
        return self.G[first:first + n].copy()

    def match_scores(self, q):
        q = np.asarray(q, dtype=np.float64).reshape(-1, self.G.shape[1])
        q = q / np.linalg.norm(q, axis=1, keepdims=True)
        return q @ self.G.T

    def match(self, q, topk=1):
        S = self.match_scores(q)
        idx = np.argsort(-S, axis=1, kind="stable")[:, :topk]
        cos = np.take_along_axis(S, idx, axis=1)
        if topk > S.shape[1]:
            pad = topk - S.shape[1]
            idx = np.concatenate([idx, np.full((len(S), pad), -1)], axis=1)
            cos = np.concatenate([cos, np.full((len(S), pad), -2.0)], axis=1)
        return (idx[:, 0], cos[:, 0]) if topk == 1 else (idx.astype(np.int32), cos)

    def process_frames(self, frames, max_faces=10, det_thresh=0.5, nms_iou=0.4, flags=0):
        return self.canned

    def counters(self):
        return {}
