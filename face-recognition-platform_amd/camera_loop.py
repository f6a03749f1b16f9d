"""The caller loop around the hot path, restated on `process_frames`.

Mirrors backend/app/routes/camera.py::process_camera_sync (:171-272) -- null/closed capture
handling and reopen (:176-200), `frame_skip` reads (:204-213), `fps_limit` (:216-221), the
`max_faces` cap (:233-235), the `match and distance <= confidence_threshold` filter (:246-256),
per-camera performance counters (:262-267) -- and the (person, camera) cooldown de-dup of
backend/app/services/tracking_service.py::record_detection (:122-134) that gates alerts in
`camera_alerts` (camera.py:310-341).  The three library calls + O(N) Python loop per face of
the reference become one device pass per frame batch; `scan_cameras` batches one frame from
every camera (the reference fans them out over ThreadPoolExecutor(4), camera.py:30,304-305).
"""
from __future__ import annotations

import logging
import threading
import time
from datetime import datetime, timedelta
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

logger = logging.getLogger(__name__)

CAMERA_METADATA: Dict[Any, dict] = {}           # camera_id -> {name, geo, source, fps_limit} (state.py:95)
camera_performance: Dict[Any, Dict[str, Any]] = {}
_performance_lock = threading.RLock()


def _config(config: Optional[Dict[str, Any]]) -> Tuple[float, int, int]:
    if not config:
        return 0.6, 1, 10
    return config.get("confidence_threshold", 0.6), max(1, config.get("frame_skip", 1)), config.get("max_faces", 10)


def grab_frame(cam_id, cap, frame_skip: int, metadata: Optional[Dict[Any, dict]] = None):
    """camera.py:176-213: None capture -> None; closed capture -> one reopen attempt from the
    stored source; read `frame_skip` frames and keep the last; any failed read -> None."""
    metadata = CAMERA_METADATA if metadata is None else metadata
    if cap is None:
        logger.warning("Camera %s has null capture object - skipping", cam_id)
        return None
    if not cap.isOpened():
        logger.warning("Camera %s is not opened, attempting reconnect", cam_id)
        try:
            source = metadata.get(cam_id, {}).get("source", cam_id)
            try:
                source = int(source)
            except Exception:
                pass
            cap.open(source)
            if not cap.isOpened():
                logger.error("Failed to reconnect camera %s", cam_id)
                return None
        except Exception as err:
            logger.error("Reconnection failed for camera %s: %s", cam_id, err)
            return None
    frame = None
    for _ in range(frame_skip):
        ret, candidate = cap.read()
        if not ret:
            return None
        frame = candidate
    return frame


def _detections_of(cam_id, faces: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
    out = []
    for f in faces:
        for m in f.get("matches", []):
            out.append({"camera_id": cam_id, "target": m["target"], "distance": m["distance"], "confidence": m["confidence"]})
    return out


def _account(cam_id, seconds: float):
    with _performance_lock:
        perf = camera_performance.setdefault(cam_id, {"total_frames": 0, "total_time": 0.0, "avg_fps": 0.0})
        perf["total_frames"] += 1
        perf["total_time"] += seconds
        perf["avg_fps"] = (perf["total_frames"] / perf["total_time"]) if perf["total_time"] > 0 else 0.0


def process_camera_sync(cam_id, cap, config: Optional[Dict[str, Any]] = None, service=None,
                        metadata: Optional[Dict[Any, dict]] = None) -> List[Dict[str, Any]]:
    """One camera, one frame: -> [{camera_id, target, distance, confidence}] exactly as the
    reference's loop would list them (face order = detector order, per face ascending distance)."""
    if service is None:
        from .face_service import face_service as service
    metadata = CAMERA_METADATA if metadata is None else metadata
    results: List[Dict[str, Any]] = []
    start = time.time()
    try:
        threshold, frame_skip, max_faces = _config(config)
        frame = grab_frame(cam_id, cap, frame_skip, metadata)
        if frame is None:
            return results
        fps_limit = metadata.get(cam_id, {}).get("fps_limit")
        if fps_limit:
            elapsed = time.time() - start
            if elapsed < 1.0 / float(fps_limit):
                time.sleep(1.0 / float(fps_limit) - elapsed)
        faces = service.process_frames(np.asarray(frame)[None], max_faces=max_faces, threshold=threshold, all_matches=True)[0]
        results = _detections_of(cam_id, faces)
        _account(cam_id, time.time() - start)
    except Exception as e:  # the reference never lets the loop raise (camera.py:269-270)
        logger.error("Error processing camera %s: %s", cam_id, e, exc_info=True)
    return results


def scan_cameras(cameras: Dict[Any, Any], config: Optional[Dict[str, Any]] = None, service=None,
                 metadata: Optional[Dict[Any, dict]] = None) -> List[Dict[str, Any]]:
    """`camera_alerts` fan-out (camera.py:304-306) as ONE device batch per frame size: one frame per
    camera, results flattened in camera order."""
    if service is None:
        from .face_service import face_service as service
    metadata = CAMERA_METADATA if metadata is None else metadata
    threshold, frame_skip, max_faces = _config(config)
    start = time.time()
    grabbed = []
    for cam_id, cap in cameras.items():
        try:
            frame = grab_frame(cam_id, cap, frame_skip, metadata)
        except Exception as e:
            logger.error("Error reading camera %s: %s", cam_id, e)
            frame = None
        if frame is not None:
            grabbed.append((cam_id, np.asarray(frame)))
    per_cam: Dict[Any, List[Dict[str, Any]]] = {}
    by_shape: Dict[Tuple[int, ...], List[int]] = {}
    for i, (_, f) in enumerate(grabbed):
        by_shape.setdefault(f.shape, []).append(i)
    for idxs in by_shape.values():
        try:
            batch = np.stack([grabbed[i][1] for i in idxs])
            faces = service.process_frames(batch, max_faces=max_faces, threshold=threshold, all_matches=True)
            for i, fl in zip(idxs, faces):
                per_cam[grabbed[i][0]] = _detections_of(grabbed[i][0], fl)
        except Exception as e:
            logger.error("Error processing camera batch: %s", e, exc_info=True)
    dt = time.time() - start
    out: List[Dict[str, Any]] = []
    for cam_id, _ in grabbed:
        _account(cam_id, dt / max(1, len(grabbed)))
        out.extend(per_cam.get(cam_id, []))
    return out


class DetectionCooldown:
    """(person, camera) cooldown + location bookkeeping of TrackingService.record_detection
    (tracking_service.py:107-134,140-141,164-170): the gate `camera_alerts` applies before it
    raises an alert (camera.py:321-326)."""

    def __init__(self, cooldown_seconds: int = 10):
        self.cooldown = timedelta(seconds=cooldown_seconds)
        self.last_detection: Dict[Tuple[str, int], datetime] = {}
        self.current_locations: Dict[str, int] = {}
        self._lock = threading.RLock()

    def record_detection(self, person_name: str, camera_id, distance: float, timestamp: Optional[datetime] = None) -> Dict[str, Any]:
        if timestamp is None:
            timestamp = datetime.now()
        with self._lock:
            try:
                camera_id = int(camera_id)
            except Exception:
                return {"recorded": False, "message": "Invalid camera_id"}
            key = (person_name, camera_id)
            if key in self.last_detection and timestamp - self.last_detection[key] < self.cooldown:
                return {"recorded": False, "is_new_location": False, "previous_location": None, "duplicate": True,
                        "message": f"Duplicate detection (cooldown: {int(self.cooldown.total_seconds())}s)"}
            previous = self.current_locations.get(person_name)
            self.current_locations[person_name] = camera_id
            self.last_detection[key] = timestamp
            return {"recorded": True, "is_new_location": previous != camera_id, "previous_location": previous,
                    "duplicate": False, "message": "Detection recorded successfully"}
