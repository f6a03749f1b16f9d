"""Caller loop against golden vectors produced by the reference's own
process_camera_sync / record_detection (tests/golden/make_camera_golden.py)."""
import json
import os
from datetime import datetime, timedelta

import numpy as np
import pytest

from fake_engine import FakeEngine
from frp_amd import camera_loop
from frp_amd.face_service import FaceService

HERE = os.path.dirname(os.path.abspath(__file__))


class Cap:
    def __init__(self, n_frames, opened=True, reopen_ok=False):
        self.frames = [np.zeros((8, 8, 3), np.uint8) for _ in range(n_frames)]
        self.opened, self.reopen_ok, self.reads = opened, reopen_ok, 0

    def isOpened(self):
        return self.opened

    def open(self, src):
        self.opened = self.reopen_ok
        return self.opened

    def read(self):
        self.reads += 1
        return (True, self.frames.pop(0)) if self.frames else (False, None)


class CannedEngine(FakeEngine):
    """detections are the scenario's canned faces (as the golden generator canned them for the reference)"""

    def __init__(self):
        super().__init__()
        self.locs, self.encs = [], []

    def process_frames(self, frames, max_faces=10, det_thresh=0.5, nms_iou=0.4, flags=0):
        B = frames.shape[0]
        n = min(len(self.locs), max_faces)
        K = max_faces
        out = dict(boxes=np.zeros((B, K, 4), np.float32), kps=np.zeros((B, K, 5, 2), np.float32), scores=np.zeros((B, K), np.float32),
                   counts=np.full((B,), n, np.int32), emb=np.zeros((B, K, 512), np.float32),
                   match_idx=np.full((B, K), -1, np.int32), match_cos=np.full((B, K), -1, np.float32))
        for k in range(n):
            t, r, b, l = self.locs[k]
            out["boxes"][:, k] = [l, t, r, b]
            out["emb"][:, k] = self.encs[k]
        if n and self.gallery_size() and not (flags & 4):
            S = self.match_scores(out["emb"][0, :n])
            out["match_idx"][:, :n] = S.argmax(1)
            out["match_cos"][:, :n] = S.max(1)
        return out


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(HERE, "golden", "camera_golden.json")))


def _service(golden):
    eng = CannedEngine()
    fs = FaceService(engine=eng)
    for n, g in zip(golden["names"], golden["gallery"]):
        fs.store_face(n, np.array(g))
    return fs, eng


def test_process_camera_sync_matches_reference(golden):
    fs, eng = _service(golden)
    for sc in golden["scenarios"]:
        eng.locs, eng.encs = [tuple(l) for l in sc["locations"]], [np.array(e, np.float32) for e in sc["encodings"]]
        fs.tolerance = sc["tolerance"]
        cap = None if sc["cap"] is None else Cap(sc["n_frames"], **sc["cap"])
        got = camera_loop.process_camera_sync(7, cap, sc["config"], service=fs, metadata={})
        exp = sc["result"]
        assert len(got) == len(exp), sc["name"]
        for g, e in zip(got, exp):
            assert list(g.keys()) == list(e.keys()) and g["camera_id"] == e["camera_id"] and g["target"] == e["target"], sc["name"]
            assert g["confidence"] == e["confidence"] and abs(g["distance"] - e["distance"]) < 1e-6, sc["name"]
        if cap is not None:
            assert cap.reads == sc["reads"], sc["name"]           # same number of frames consumed


def test_scan_cameras_batches_and_keeps_camera_order(golden):
    fs, eng = _service(golden)
    sc = golden["scenarios"][0]
    eng.locs, eng.encs = [tuple(l) for l in sc["locations"]], [np.array(e, np.float32) for e in sc["encodings"]]
    fs.tolerance = 0.6
    cams = {3: Cap(1), 9: None, 1: Cap(1), 4: Cap(0)}
    calls = []
    orig = eng.process_frames
    eng.process_frames = lambda frames, **k: (calls.append(frames.shape[0]), orig(frames, **k))[1]
    got = camera_loop.scan_cameras(cams, None, service=fs, metadata={})
    assert calls == [2]                                           # one device batch for the two live cameras
    exp = sc["result"]
    assert [g["camera_id"] for g in got] == [3] * len(exp) + [1] * len(exp)
    assert [g["target"] for g in got] == [e["target"] for e in exp] * 2


def test_cooldown_matches_reference_tracking(golden):
    cd = camera_loop.DetectionCooldown(golden["cooldown_seconds"])
    t0 = datetime(2025, 1, 1, 12, 0, 0)
    for step in golden["tracking"]:
        r = cd.record_detection(step["person"], step["camera_id"], 0.3, timestamp=t0 + timedelta(seconds=step["dt"]))
        assert {k: r.get(k) for k in step["result"]} == step["result"], step
