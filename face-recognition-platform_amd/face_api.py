"""`face_recognition`-shaped lower API on the HIP engine, for the reference call sites that
bypass FaceService: routes/camera.py:232 (`face_locations(rgb)`), :237 (`face_encodings(rgb,
locations)`), face_service.py:139 (`load_image_file`), :410 (`face_distance`).

Drop-in: `import frp_amd.face_api as face_recognition`.
Detection and embedding are one fused device pass; `face_locations` keeps the landmarks and
embeddings of the image it just processed (per thread) so the `face_encodings` call that
always follows it in the reference (camera.py:232-237, face_service.py:156-179) costs nothing.
"""
from __future__ import annotations

import threading
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .face_service import box_to_location, face_service, load_image_file  # noqa: F401  (re-export)

_tls = threading.local()


def _fingerprint(img: np.ndarray):
    return (id(img), img.shape, int(img[::max(1, img.shape[0] // 7), ::max(1, img.shape[1] // 7)].sum()))


def _run(img: np.ndarray, max_faces: int = 64):
    out = face_service._detect_and_embed(img[None], rgb=True, max_faces=max_faces)
    n = int(out["counts"][0])
    h, w = img.shape[:2]
    locs = [box_to_location(out["boxes"][0, k], h, w) for k in range(n)]
    _tls.last = (_fingerprint(img), locs, out["boxes"][0, :n].copy(), out["kps"][0, :n].copy(), out["emb"][0, :n].copy())
    return _tls.last


def face_locations(img: np.ndarray, number_of_times_to_upsample: int = 1, model: str = "hog") -> List[Tuple[int, int, int, int]]:
    """-> [(top, right, bottom, left)] clipped to the image, detector order (score descending)."""
    return list(_run(img)[1])


def _iou(a, b) -> float:
    t, r, bo, l = a
    t2, r2, bo2, l2 = b
    iw, ih = min(r, r2) - max(l, l2), min(bo, bo2) - max(t, t2)
    if iw <= 0 or ih <= 0:
        return 0.0
    inter = iw * ih
    return inter / float((r - l) * (bo - t) + (r2 - l2) * (bo2 - t2) - inter)


def face_encodings(face_image: np.ndarray, known_face_locations: Optional[Sequence[Tuple[int, int, int, int]]] = None,
                   num_jitters: int = 1, model: str = "small") -> List[np.ndarray]:
    """-> one unit 512-d float64 vector per location (reference: 128-d dlib descriptors)."""
    last = getattr(_tls, "last", None)
    if last is None or last[0] != _fingerprint(face_image):
        last = _run(face_image)
    _, locs, boxes, kps, emb = last
    if known_face_locations is None:
        return [e.astype(np.float64) for e in emb]
    out = []
    for loc in known_face_locations:
        loc = tuple(int(v) for v in loc)
        if loc in locs:
            out.append(emb[locs.index(loc)].astype(np.float64))
            continue
        # a box the detector did not produce: use the best-overlapping detection's landmarks, or
        # the 5-point template scaled into the box, and embed that face alone
        best = max(range(len(locs)), key=lambda i: _iou(loc, locs[i]), default=None)
        if best is not None and _iou(loc, locs[best]) >= 0.3:
            out.append(emb[best].astype(np.float64))
            continue
        t, r, b, l = loc
        tmpl = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366], [41.5493, 92.3655], [70.7299, 92.2041]]) / 112.0
        k = tmpl * [r - l, b - t] + [l, t]
        from . import native
        e = face_service._eng().embed_faces(face_image, k[None].astype(np.float32), flags=native.FLAG_RGB)[0]
        out.append(e.astype(np.float64))
    return out


def face_distance(face_encodings_, face_to_compare) -> np.ndarray:
    """Euclidean distance per row, empty in -> empty out (face_recognition 1.3.0).  Host-side
    helper for the 1-vs-few call sites (face_service.py:357,576); the gallery-sized compare
    goes through FaceService.compare_faces -> the device matcher."""
    if len(face_encodings_) == 0:
        return np.empty((0))
    return np.linalg.norm(np.asarray(face_encodings_) - face_to_compare, axis=1)


def compare_faces(known_face_encodings, face_encoding_to_check, tolerance: float = 0.6) -> List[bool]:
    return list(face_distance(known_face_encodings, face_encoding_to_check) <= tolerance)
