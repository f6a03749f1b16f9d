#!/usr/bin/env python3
"""Golden vectors for the caller loop from the reference's own code:
backend/app/routes/camera.py::process_camera_sync (:171-272) and
backend/app/services/tracking_service.py::record_detection cooldown (:94-134).

Runs only in the build container.  Absent third-party modules (cv2, face_recognition, dotenv,
Mongo) are inert `sys.modules` stubs; `face_recognition.face_locations/face_encodings` return
the CANNED detections listed in each scenario (the detector/embedder are not what is pinned
here -- the loop logic around them is), `face_distance` is its published one-liner.
Output: tests/golden/camera_golden.json (scenario inputs + the reference's outputs)."""
import json
import logging
import os
import sys
import tempfile
import types
from datetime import datetime, timedelta

import numpy as np

REF = "/root/reference/backend"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "camera_golden.json")
CANNED = {"locations": [], "encodings": []}


def install():
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_BGR2RGB = 4
    cv2.cvtColor = lambda f, code: f[..., ::-1]
    cv2.VideoCapture = object
    cv2.IMWRITE_JPEG_QUALITY = 1
    sys.modules["cv2"] = cv2
    fr = types.ModuleType("face_recognition")
    fr.face_locations = lambda img, *a, **k: list(CANNED["locations"])
    fr.face_encodings = lambda img, locs=None, *a, **k: [np.array(CANNED["encodings"][CANNED["locations"].index(l)]) for l in locs]

    def face_distance(encs, q):
        if len(encs) == 0:
            return np.empty((0))
        return np.linalg.norm(np.asarray(encs) - q, axis=1)

    fr.face_distance = face_distance
    sys.modules["face_recognition"] = fr
    dotenv = types.ModuleType("dotenv")
    dotenv.load_dotenv = lambda *a, **k: None
    sys.modules["dotenv"] = dotenv
    app = types.ModuleType("app")
    app.__path__ = [os.path.join(REF, "app")]
    sys.modules["app"] = app
    state = types.ModuleType("app.state")
    state.ENCODINGS = {}
    state.CAMERAS = {}
    state.CAMERA_METADATA = {}
    state.PERSON_LOCATIONS = {}
    state.init_cameras = lambda ids: None
    sys.modules["app.state"] = state
    utils = types.ModuleType("app.utils")
    utils.__path__ = [os.path.join(REF, "app", "utils")]
    sys.modules["app.utils"] = utils
    db = types.ModuleType("app.utils.db")
    for n in ("store_embedding", "retrieve_embedding", "save_detection_to_db", "log_alert", "save_watchlist_db", "load_watchlist_db",
              "save_geofence_db", "load_geofence_db", "get_alerts_from_db", "load_geofences_db", "save_geofences_db"):
        setattr(db, n, lambda *a, **k: True)

    class _C:
        def delete_one(self, q):
            return types.SimpleNamespace(deleted_count=0)

        def find_one(self, *a, **k):
            return None

    db.faces_collection = _C()
    db.__getattr__ = lambda name: (lambda *a, **k: None)
    sys.modules["app.utils.db"] = db
    lg = types.ModuleType("app.utils.logger")
    lg.get_logger = lambda name=None: logging.getLogger(name or "ref")
    sys.modules["app.utils.logger"] = lg
    services = types.ModuleType("app.services")
    services.__path__ = [os.path.join(REF, "app", "services")]
    sys.modules["app.services"] = services
    alert = types.ModuleType("app.services.alert_service")
    alert.alert_service = types.SimpleNamespace(generate_alert=lambda **k: None, get_alerts=lambda limit=50: [])
    sys.modules["app.services.alert_service"] = alert
    routes = types.ModuleType("app.routes")
    routes.__path__ = [os.path.join(REF, "app", "routes")]
    sys.modules["app.routes"] = routes
    return state


class FakeCap:
    def __init__(self, frames, opened=True, reopen_ok=False):
        self.frames = list(frames)
        self.opened = opened
        self.reopen_ok = reopen_ok
        self.reads = 0

    def isOpened(self):
        return self.opened

    def open(self, src):
        self.opened = self.reopen_ok
        return self.opened

    def read(self):
        self.reads += 1
        if not self.frames:
            return False, None
        return True, self.frames.pop(0)


def main():
    os.chdir(tempfile.mkdtemp(prefix="frp_cam_golden_"))
    state = install()
    import importlib
    cam = importlib.import_module("app.routes.camera")
    fsmod = sys.modules["app.services.face_service"]
    trk = importlib.import_module("app.services.tracking_service")

    rng = np.random.default_rng(11)
    D = 512
    names = [f"wl_{i}" for i in range(6)]
    G = rng.standard_normal((6, D))
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    G[5] = G[2] + 0.02 * rng.standard_normal(D)       # near-duplicate identity: two targets can match one face
    G[5] /= np.linalg.norm(G[5])
    for n, g in zip(names, G):
        state.ENCODINGS[n] = g.tolist()

    def face(i, noise):
        v = G[i] + noise * rng.standard_normal(D) / np.sqrt(D)
        return (v / np.linalg.norm(v)).tolist()

    frame = np.zeros((8, 8, 3), np.uint8)
    scenarios = []

    def run(name, locs, encs, cap_kwargs, n_frames, config, tol=0.6, meta=None):
        CANNED["locations"], CANNED["encodings"] = [tuple(l) for l in locs], encs
        fsmod.face_service.tolerance = tol
        state.CAMERA_METADATA.clear()
        if meta:
            state.CAMERA_METADATA[7] = meta
        cap = None if cap_kwargs is None else FakeCap([frame.copy() for _ in range(n_frames)], **cap_kwargs)
        out = cam.process_camera_sync(7, cap, config)
        scenarios.append({"name": name, "locations": [list(l) for l in locs], "encodings": encs, "cap": cap_kwargs,
                          "n_frames": n_frames, "config": config, "tolerance": tol, "reads": None if cap is None else cap.reads,
                          "result": out})

    locs3 = [(10, 50, 60, 5), (20, 120, 90, 70), (5, 200, 40, 160)]
    encs3 = [face(0, 0.3), face(2, 0.2), face(4, 2.5)]
    run("three_faces_default", locs3, encs3, {}, 1, None)
    run("max_faces_2", locs3, encs3, {}, 1, {"confidence_threshold": 0.6, "frame_skip": 1, "max_faces": 2})
    run("threshold_tighter_than_tolerance", locs3, encs3, {}, 1, {"confidence_threshold": 0.3, "frame_skip": 1, "max_faces": 10})
    run("tolerance_tighter_than_threshold", locs3, encs3, {}, 1, {"confidence_threshold": 0.9, "frame_skip": 1, "max_faces": 10}, tol=0.25)
    run("frame_skip_3", locs3[:1], encs3[:1], {}, 3, {"confidence_threshold": 0.6, "frame_skip": 3, "max_faces": 10})
    run("frame_skip_short_read", locs3[:1], encs3[:1], {}, 1, {"confidence_threshold": 0.6, "frame_skip": 2, "max_faces": 10})
    run("no_faces", [], [], {}, 1, None)
    run("cap_none", locs3, encs3, None, 0, None)
    run("cap_closed_reopen_fails", locs3, encs3, {"opened": False, "reopen_ok": False}, 1, None)
    run("cap_closed_reopen_ok", locs3[:2], encs3[:2], {"opened": False, "reopen_ok": True}, 1, None)

    # tracking cooldown (tracking_service.py:122-134)
    state.CAMERA_METADATA.clear()
    ts = trk.TrackingService() if hasattr(trk, "TrackingService") else trk.tracking_service
    t0 = datetime(2025, 1, 1, 12, 0, 0)
    cooldown_s = int(ts.cooldown.total_seconds())
    seq = [("alice", 1, 0), ("alice", 1, cooldown_s - 1), ("alice", 2, cooldown_s - 1), ("bob", 1, 1),
           ("alice", 1, cooldown_s + 1), ("alice", 2, 2 * cooldown_s + 5), ("alice", "x", 5)]
    track = []
    for person, cam_id, dt in seq:
        r = ts.record_detection(person, cam_id, 0.3, timestamp=t0 + timedelta(seconds=dt))
        track.append({"person": person, "camera_id": cam_id, "dt": dt,
                      "result": {k: r.get(k) for k in ("recorded", "is_new_location", "previous_location", "duplicate", "message") if k in r}})
    json.dump({"names": names, "gallery": G.tolist(), "scenarios": scenarios, "cooldown_seconds": cooldown_s, "tracking": track},
              open(OUT, "w"))
    print("wrote", OUT, len(scenarios), "scenarios;", [(s["name"], len(s["result"])) for s in scenarios])
    print(track)


if __name__ == "__main__":
    main()
