"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of the *plumbing* half of the reference hot path: the Python
logic of backend/app/services/face_service.py and the per-face filter loop of
backend/app/routes/camera.py, with the gallery kept exactly as the reference
keeps it (a dict name -> list[float], rebuilt into an ndarray on every call).

Pinned: tests/test_oracle_plumbing.py checks every function here against
tests/golden/plumbing_golden.{json,npz}, which were produced by the reference's
own module (tests/golden/make_plumbing_golden.py, run in the build container).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.
"""
from __future__ import annotations

import math
import time
from collections import deque
from datetime import datetime
from typing import Any, Dict, List, Optional

import numpy as np


def face_distance(face_encodings, face_to_compare):
    """face_recognition 1.3.0 `face_distance` (third-party, not vendored in the
    reference; call sites face_service.py:357,410,465,576,599): Euclidean norm
    per row, empty in -> empty out."""
    if len(face_encodings) == 0:
        return np.empty((0))
    return np.linalg.norm(np.asarray(face_encodings) - face_to_compare, axis=1)


def confidence_level(distance: float) -> str:
    """face_service.py:486-492."""
    if distance < 0.4:
        return "high"
    elif distance < 0.6:
        return "medium"
    return "low"


def calibrate_confidence(distance: float) -> float:
    """face_service.py:497-506."""
    x = max(0.0, min(1.0, 1.0 - distance))
    k = 12.0
    calibrated = 100.0 / (1.0 + np.exp(-k * (x - 0.5)))
    return round(float(calibrated), 2)


class PlumbingOracle:
    """State + methods of FaceService that touch the gallery (face_service.py:51-82)."""

    def __init__(self, tolerance: float = 0.6):
        self.tolerance = tolerance
        self.ENCODINGS: Dict[str, list] = {}  # state.py:78
        self._comparison_history = deque(maxlen=5000)
        self.total_comparisons = 0

    # face_service.py:395-443
    def compare_faces(self, test_encoding, target_names: Optional[List[str]] = None,
                      return_distances: bool = True) -> List[Dict[str, Any]]:
        E = self.ENCODINGS
        targets = list(E.keys()) if target_names is None else [t for t in target_names if t in E]
        if not targets:
            return []
        stored = np.array([E[t] for t in targets])  # :409 per-call rebuild
        distances = face_distance(stored, test_encoding)  # :410
        matches = distances <= self.tolerance  # :411
        results = []
        for i, target in enumerate(targets):  # :414-429
            distance = float(distances[i])
            is_match = bool(matches[i])
            item = {"target": target, "match": is_match}
            if return_distances:
                item["distance"] = distance
                item["confidence"] = confidence_level(distance)
                item["confidence_score"] = calibrate_confidence(distance)
            results.append(item)
            self._comparison_history.append(
                {"distance": distance, "match": is_match, "timestamp": datetime.now().isoformat()})
        if return_distances:
            results.sort(key=lambda x: x.get("distance", 1.0))  # :432 (stable)
        self.total_comparisons += len(targets)
        return results

    # face_service.py:448-481
    def batch_compare_faces(self, test_encodings, target_names=None):
        E = self.ENCODINGS
        targets = list(E.keys()) if target_names is None else [t for t in target_names if t in E]
        if not targets:
            return [[] for _ in test_encodings]
        stored = np.array([E[t] for t in targets])
        out = []
        for q in test_encodings:
            distances = face_distance(stored, q)
            matches = distances <= self.tolerance
            res = []
            for i, target in enumerate(targets):
                if matches[i]:
                    res.append({"target": target, "match": True, "distance": float(distances[i]),
                                "confidence": confidence_level(distances[i])})
            res.sort(key=lambda x: x["distance"])
            out.append(res)
        return out

    # face_service.py:590-612
    def find_k_nearest(self, test_encoding, k: int = 5):
        E = self.ENCODINGS
        if len(E) == 0:
            return []
        targets = list(E.keys())
        enc = np.array([E[t] for t in targets])
        distances = face_distance(enc, test_encoding)
        k = min(k, len(distances))
        idx = np.argpartition(distances, k - 1)[:k]
        idx = idx[np.argsort(distances[idx])]
        return [{"target": targets[int(i)], "distance": float(distances[int(i)]),
                 "confidence": confidence_level(float(distances[int(i)])),
                 "confidence_score": calibrate_confidence(float(distances[int(i)]))} for i in idx]

    # face_service.py:552-585
    def cluster_faces(self, distance_threshold: float = 0.6):
        E = self.ENCODINGS
        if len(E) < 2:
            return {"cluster_0": list(E.keys())}
        targets = list(E.keys())
        enc = np.array([E[t] for t in targets])
        clusters, cid, assigned = {}, 0, set()
        for i, target in enumerate(targets):
            if target in assigned:
                continue
            members = [target]
            assigned.add(target)
            for j, other in enumerate(targets):
                if other in assigned or i == j:
                    continue
                dist = float(face_distance([enc[i]], enc[j])[0])
                if dist <= distance_threshold:
                    members.append(other)
                    assigned.add(other)
            clusters[f"cluster_{cid}"] = members
            cid += 1
        return clusters

    # face_service.py:349-364 (duplicate scan part of store_face) + :366,:374
    def store_face(self, target_name: str, encoding) -> Dict[str, Any]:
        E = self.ENCODINGS
        enc_list = encoding.tolist() if isinstance(encoding, np.ndarray) else encoding
        is_dup, similar = False, None
        for existing, e in E.items():
            if existing == target_name:
                continue
            d = float(face_distance([np.array(e)], encoding)[0])
            if d < 0.3:
                is_dup, similar = True, existing
                break
        already = target_name in E
        E[target_name] = enc_list
        message = f"Face {'updated' if already else 'stored'} successfully for '{target_name}'"
        if is_dup:
            message += f" (Warning: Similar to '{similar}')"
        return {"success": True, "message": message, "is_duplicate": is_dup,
                "similar_to": similar if is_dup else None, "was_update": already}


def quality_geometry(image_shape, face_location):
    """Geometry terms of assess_face_quality (face_service.py:251-275, :299-329) with
    blur/lighting at the reference's own fallback value 50.0 (:282-284, :295-297)."""
    top, right, bottom, left = face_location
    height, width = image_shape[:2]
    fw = max(1, right - left)
    fh = max(1, bottom - top)
    size_ratio = float(fw * fh) / float(width * height) if width * height > 0 else 0.0
    size_score = min(100.0, (size_ratio / 0.25) * 100.0)
    cx, cy = (left + right) / 2.0, (top + bottom) / 2.0
    dist = np.sqrt(((cx - width / 2.0) / width) ** 2 + ((cy - height / 2.0) / height) ** 2) if width and height else 0.0
    position_score = max(0.0, (1.0 - dist) * 100.0)
    aspect_ratio = min(fw, fh) / max(fw, fh)
    aspect_score = aspect_ratio * 100.0
    blur_score = lighting_score = 50.0
    overall = size_score * 0.25 + position_score * 0.2 + aspect_score * 0.2 + blur_score * 0.2 + lighting_score * 0.15
    issues = []
    if size_ratio < 0.05:
        issues.append("Face too small - move closer or crop image")
    if size_ratio > 0.8:
        issues.append("Face too large - image should show some background")
    if dist > 0.4:
        issues.append("Face not centered - adjust framing")
    if aspect_ratio < 0.75:
        issues.append("Face appears distorted or at extreme angle")
    return {"score": round(overall, 2), "size_score": round(size_score, 2),
            "position_score": round(position_score, 2), "aspect_score": round(aspect_score, 2),
            "blur_score": round(blur_score, 2), "lighting_score": round(lighting_score, 2), "issues": issues}


def camera_filter_loop(oracle: PlumbingOracle, cam_id, face_encodings, confidence_threshold: float = 0.6):
    """routes/camera.py:243-259: per face, full-gallery compare then the Python filter."""
    results = []
    for enc in face_encodings:
        matches = oracle.compare_faces(enc, target_names=None, return_distances=True)
        for match in matches:
            distance = match.get("distance", 0.0)
            if match.get("match") and distance <= confidence_threshold:
                results.append({"camera_id": cam_id, "target": match.get("target"),
                                "distance": distance, "confidence": match.get("confidence", 1.0)})
    return results


def time_reference_plumbing(N: int, D: int, n_faces: int, seed: int = 42) -> float:
    """cpu_baseline helper: seconds per face of the reference's compare loop at gallery size N."""
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((N, D))
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    o = PlumbingOracle()
    for i in range(N):
        o.ENCODINGS[f"id_{i:07d}"] = G[i].tolist()
    qs = [G[(7 * i) % N] for i in range(n_faces)]
    t0 = time.perf_counter()
    camera_filter_loop(o, 0, qs)
    return (time.perf_counter() - t0) / max(1, n_faces)
