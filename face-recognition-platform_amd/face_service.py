"""Drop-in `FaceService` for backend/app/services/face_service.py on MI355X.

Same public surface as the reference class (methods, argument meaning, result
dict keys and their order, messages, "never raise" error convention); the
arithmetic runs in libfrp.so (HIP, gfx950) through `native.Engine`:

  reference call                                         -> here
  face_recognition.load_image_file      (:139)           -> PIL decode to RGB (host)
  face_recognition.face_locations       (:156)           -> Engine.process_frames (detector + decode/NMS)
  face_recognition.face_encodings       (:179)           -> same call (5-point warp + IResNet-100), 512-d
  np.array([ENCODINGS[t] ...]) rebuild  (:409,461,558,595)-> device-resident fp16 gallery (gallery.Gallery)
  face_recognition.face_distance        (:410,465,599)   -> Engine.match_scores / match (cosine on MFMA);
                                                            distance = sqrt(max(0, 2 - 2 cos))
Embeddings are unit vectors, so the Euclidean `distance` field, the 0.4 / 0.6
buckets (:486-492), `tolerance` (:43,411) and the sigmoid score (:497-506) keep
their meaning.  There is NO CPU fallback: without libfrp.so or a GPU the compute
methods report failure through the reference's own error convention.

New streaming entry points (fill the dead probe list at
backend/app/services/async_task_manager.py:125): process_frames / process_frame.
"""
from __future__ import annotations

import json
import logging
import os
import threading
import time
from collections import deque
from datetime import datetime
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import lanes, native
from .gallery import Gallery
from .mjpeg import JpegBatch

logger = logging.getLogger(__name__)

DEFAULT_TOLERANCE = float(os.getenv("FACE_TOLERANCE", "0.6"))         # face_service.py:43
DEFAULT_MODEL = os.getenv("FACE_MODEL", "hog")                        # :44 (kept as a label only)
CACHE_TTL_SECONDS = int(os.getenv("FACE_CACHE_TTL", "3600"))          # :45
BATCH_WORKERS = int(os.getenv("FACE_BATCH_WORKERS", "4"))             # :46 (batching replaces the thread pool)
BACKUP_DIR = Path(os.getenv("FACE_BACKUP_DIR", "data/backups"))       # :47
DET_THRESH = float(os.getenv("FRP_DET_THRESH", "0.5"))
NMS_IOU = float(os.getenv("FRP_NMS_IOU", "0.4"))
CLUSTER_TILE = 64          # candidate seeds scored per gallery pass in cluster_faces
MAX_FACES_ENCODE = int(os.getenv("FRP_MAX_FACES", "64"))

_METRIC_KEYS = ("total_encodings", "total_comparisons", "cache_hits", "cache_misses",
                "cumulative_encoding_time", "cumulative_comparison_time", "failed_encodings")


def cos_to_distance(cos) -> np.ndarray:
    return np.sqrt(np.maximum(0.0, 2.0 - 2.0 * np.asarray(cos, dtype=np.float64)))


def confidence_level(distance: float) -> str:                         # :486-492
    return "high" if distance < 0.4 else ("medium" if distance < 0.6 else "low")


def calibrate_confidence(distance: float) -> float:                   # :497-506
    x = max(0.0, min(1.0, 1.0 - distance))
    return round(float(100.0 / (1.0 + np.exp(-12.0 * (x - 0.5)))), 2)


def box_to_location(box, h: int, w: int) -> Tuple[int, int, int, int]:
    """(x1,y1,x2,y2) float -> (top,right,bottom,left) ints clipped to the image, the
    face_recognition css order the reference passes around (face_service.py:252,
    routes/face.py:199-217); truncation as deepfake_utils.py:153 does with insightface boxes."""
    x1, y1, x2, y2 = (int(v) for v in box)
    return max(y1, 0), min(x2, w), min(y2, h), max(x1, 0)


def load_image_file(path: str) -> np.ndarray:
    """face_recognition.load_image_file: PIL decode -> RGB u8 [H,W,3]."""
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("RGB"))


class NullStorage:
    """Persistence hook standing in for app.utils.db (Mongo + Fernet: out of scope)."""

    def store_embedding(self, target: str, embedding: List[float]) -> bool:
        return True

    def delete(self, target: str) -> int:
        return 0

    def has(self, target: str) -> bool:
        return True

    def ping(self) -> None:
        return None


class FaceService:
    def __init__(self, engine: Optional[Any] = None, storage: Optional[Any] = None, device: Optional[int] = None,
                 weights_blob: Optional[bytes] = None, second_engine: Optional[Any] = None):
        self.tolerance = DEFAULT_TOLERANCE
        self.model = DEFAULT_MODEL
        self._engine = engine
        self._engine_lock = threading.Lock()
        self._device = int(os.getenv("FRP_DEVICE", "0")) if device is None else device
        self._weights_blob = weights_blob
        self._storage = storage or NullStorage()
        self.ENCODINGS = Gallery(self._eng)
        # REST-style compat calls (compare_faces, find_k_nearest, batch_compare_faces, the duplicate scan, cluster_faces) score
        # against float64 copies of the rows AS ENROLLED (frp_gallery_exact / frp_gallery_distances): the reference's own
        # arithmetic (face_service.py:409-410), distances equal to 1e-6 - also for an exact copy (0.0), for d == tolerance, for
        # rows that are not unit vectors and for 128-d rows.  FRP_EXACT_COMPAT=0: cosines of the unit fp16 rows instead
        # (4 KB per identity less device memory; distances good to ~2e-3, 0.03 near 0).
        self._exact = os.getenv("FRP_EXACT_COMPAT", "1") != "0"
        if engine is not None and self._exact and hasattr(engine, "gallery_exact"):
            engine.gallery_exact(True)
            self.ENCODINGS.exact = True
        # process_stream keeps two batches in flight on the GPU (lanes.py): a second handle with its own copy of the
        # gallery, fed by every update from the start (Gallery.add_mirror).  Created on first use, or injected (tests).
        self._engine2 = second_engine
        if second_engine is not None:
            self.ENCODINGS.add_mirror(second_engine)
        self._encoding_cache: Dict[str, Dict[str, Any]] = {}
        self._cache_ttl = CACHE_TTL_SECONDS
        self._cache_lock = threading.RLock()
        self._quality_history = deque(maxlen=1000)
        self._metrics_lock = threading.RLock()
        self._metrics = {k: (0.0 if k.startswith("cumulative") else 0) for k in _METRIC_KEYS}
        self._comparison_history = deque(maxlen=5000)
        self._last_detection = threading.local()
        logger.info("FaceService initialized (model=%s, tolerance=%.3f)", self.model, self.tolerance)

    # ------------------------------------------------------------------ engine
    def _eng(self):
        """The HIP engine, created on first use (import must work on a box without a GPU;
        every compute call then fails loudly through the error convention)."""
        if self._engine is None:
            with self._engine_lock:
                if self._engine is None:
                    eng = native.Engine(self._device)
                    blob = self._weights_blob
                    if blob is None:
                        path = os.getenv("FRP_WEIGHTS")
                        if path:
                            with open(path, "rb") as f:
                                blob = f.read()
                        else:
                            from . import weights
                            logger.warning("FRP_WEIGHTS not set: using seeded synthetic weights (no pretrained pack offline)")
                            blob = weights.synthetic_blob()
                    eng.load_weights(blob)
                    self._blob_loaded = blob
                    if self._exact:
                        eng.gallery_exact(True)
                        self.ENCODINGS.exact = True
                    self._engine = eng
        return self._engine

    def _eng2(self):
        """The second lane's engine (same device, same weights).  It can only join while the gallery is empty - both
        copies of the matrix are then built from the same rows; afterwards process_stream runs on one lane."""
        if self._engine2 is None:
            eng = self._eng()
            with self._engine_lock:
                if self._engine2 is None:
                    if len(self.ENCODINGS) > 0 or getattr(self, "_blob_loaded", None) is None:
                        return None
                    e2 = native.Engine(self._device)
                    try:
                        e2.load_weights(self._blob_loaded)
                        self.ENCODINGS.add_mirror(e2)      # raises if an enrolment slipped in since the check above
                    except Exception:
                        e2.close()
                        return None
                    self._engine2 = e2
        return self._engine2

    def enable_second_lane(self) -> bool:
        """Create the second lane now (call before the watch list is loaded).  -> whether process_stream will overlap"""
        return self._eng2() is not None

    def _bump(self, key: str, by=1):
        with self._metrics_lock:
            self._metrics[key] += by

    # ------------------------------------------------------------------ encode (face_service.py:87-219)
    def _detect_and_embed(self, images: np.ndarray, rgb: bool = True, max_faces: int = MAX_FACES_ENCODE, match: bool = False):
        flags = (native.FLAG_RGB if rgb else 0) | (0 if match else native.FLAG_NO_MATCH)
        return self._eng().process_frames(images, max_faces=max_faces, det_thresh=DET_THRESH, nms_iou=NMS_IOU, flags=flags)

    def encode_face(self, image_path_or_array, return_locations: bool = False) -> Dict[str, Any]:
        start = time.time()

        def failure(msg):
            return {"success": False, "face_count": 0, "encodings": [], "message": msg,
                    "processing_time": time.time() - start}

        try:
            is_path = isinstance(image_path_or_array, str)
            if is_path:
                cached = self._get_from_cache(image_path_or_array)
                if cached:
                    self._bump("cache_hits")
                    result = {"success": True, "face_count": len(cached.get("encodings", [])),
                              "encodings": cached.get("encodings", []), "message": "Retrieved from cache",
                              "cached": True, "processing_time": time.time() - start}
                    if return_locations:
                        result["locations"] = cached.get("locations", [])
                    return result
                self._bump("cache_misses")
                image = load_image_file(image_path_or_array)
            elif isinstance(image_path_or_array, np.ndarray):
                image = image_path_or_array
            else:
                return failure("Invalid input type")

            t_enc = time.time()
            out = self._detect_and_embed(image[None] if image.ndim == 3 else image)
            n = int(out["counts"][0])
            if n == 0:
                self._bump("failed_encodings")
                return failure("No faces detected in image")
            h, w = image.shape[:2]
            locations = [box_to_location(out["boxes"][0, k], h, w) for k in range(n)]
            encodings = [out["emb"][0, k].astype(np.float64) for k in range(n)]
            enc_time = time.time() - t_enc
            with self._metrics_lock:
                self._metrics["total_encodings"] += n
                self._metrics["cumulative_encoding_time"] += enc_time
            if is_path:
                self._add_to_cache(image_path_or_array, {"encodings": encodings, "locations": locations})
            total = time.time() - start
            logger.info("Encoded %d face(s) in %.3fs (io+proc=%.3fs)", n, total, enc_time)
            result = {"success": True, "face_count": n, "encodings": encodings,
                      "message": f"Successfully encoded {n} face(s)", "processing_time": total}
            if return_locations:
                result["locations"] = locations
            return result
        except Exception as e:  # never raise across the API (:209-219)
            logger.exception("Error encoding face: %s", e)
            self._bump("failed_encodings")
            return failure(f"Error encoding face: {str(e)}")

    def batch_encode_faces(self, image_paths: List[str], max_workers: int = BATCH_WORKERS) -> List[Dict[str, Any]]:
        """:224-246.  The reference fans paths out over a 4-thread pool (one detect+embed per image,
        results in completion order); here uncached images of equal size are stacked into ONE device
        batch (chunks of `max_workers * 8` frames) and results come back in input order."""
        start = time.time()
        results: List[Optional[Dict[str, Any]]] = [None] * len(image_paths)
        pending: Dict[Tuple[int, ...], List[Tuple[int, np.ndarray]]] = {}
        for i, path in enumerate(image_paths):
            try:
                cached = self._get_from_cache(path) if isinstance(path, str) else None
                if cached is not None or not isinstance(path, str):
                    results[i] = self.encode_face(path)          # cache hit / invalid input: the single-image path
                    continue
                self._bump("cache_misses")
                img = load_image_file(path)
                pending.setdefault(img.shape, []).append((i, img))
            except Exception as e:
                logger.exception("Batch encode failed for %s: %s", path, e)
                self._bump("failed_encodings")
                results[i] = {"success": False, "face_count": 0, "encodings": [], "message": f"Error encoding face: {str(e)}",
                              "processing_time": time.time() - start}
        chunk = max(1, max_workers) * 8
        for shape, items in pending.items():
            for c0 in range(0, len(items), chunk):
                part = items[c0:c0 + chunk]
                t0 = time.time()
                try:
                    out = self._detect_and_embed(np.stack([im for _, im in part]))
                except Exception as e:
                    logger.exception("Batch encode failed: %s", e)
                    for i, _ in part:
                        self._bump("failed_encodings")
                        results[i] = {"success": False, "face_count": 0, "encodings": [], "message": f"Error encoding face: {str(e)}",
                                      "processing_time": time.time() - start}
                    continue
                dt = time.time() - t0
                for bi, (i, img) in enumerate(part):
                    n = int(out["counts"][bi])
                    if n == 0:
                        self._bump("failed_encodings")
                        results[i] = {"success": False, "face_count": 0, "encodings": [], "message": "No faces detected in image",
                                      "processing_time": time.time() - start}
                        continue
                    h, w = img.shape[:2]
                    locs = [box_to_location(out["boxes"][bi, k], h, w) for k in range(n)]
                    encs = [out["emb"][bi, k].astype(np.float64) for k in range(n)]
                    with self._metrics_lock:
                        self._metrics["total_encodings"] += n
                        self._metrics["cumulative_encoding_time"] += dt / len(part)
                    self._add_to_cache(image_paths[i], {"encodings": encs, "locations": locs})
                    results[i] = {"success": True, "face_count": n, "encodings": encs,
                                  "message": f"Successfully encoded {n} face(s)", "processing_time": time.time() - start}
        for i, path in enumerate(image_paths):
            results[i]["image_path"] = path
        return results  # type: ignore[return-value]

    # ------------------------------------------------------------------ quality (:251-339), host arithmetic
    @staticmethod
    def _gray(rgb: np.ndarray) -> np.ndarray:
        # cv2.COLOR_RGB2GRAY fixed-point form: (R*4899 + G*9617 + B*1868 + 8192) >> 14
        r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
        return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)

    @staticmethod
    def _laplacian_var(gray: np.ndarray) -> float:
        # cv2.Laplacian(gray, CV_64F): 3x3 [[0,1,0],[1,-4,1],[0,1,0]], BORDER_REFLECT_101
        g = np.pad(gray.astype(np.float64), 1, mode="reflect")
        lap = g[:-2, 1:-1] + g[2:, 1:-1] + g[1:-1, :-2] + g[1:-1, 2:] - 4.0 * g[1:-1, 1:-1]
        return float(lap.var())

    def assess_face_quality(self, image, face_location: Tuple[int, int, int, int]) -> Dict[str, Any]:
        top, right, bottom, left = face_location
        height, width = image.shape[:2]
        fw, fh = max(1, right - left), max(1, bottom - top)
        size_ratio = float(fw * fh) / float(width * height) if width * height > 0 else 0.0
        size_score = min(100.0, (size_ratio / 0.25) * 100.0)
        cx, cy = (left + right) / 2.0, (top + bottom) / 2.0
        off = np.sqrt(((cx - width / 2.0) / width) ** 2 + ((cy - height / 2.0) / height) ** 2) if width and height else 0.0
        position_score = max(0.0, (1.0 - off) * 100.0)
        aspect_ratio = min(fw, fh) / max(fw, fh)
        aspect_score = aspect_ratio * 100.0
        try:
            crop = image[top:bottom, left:right]
            if crop.size == 0 or crop.ndim != 3:
                raise ValueError("empty crop")
            gray = self._gray(crop)
            blur_score = min(100.0, (self._laplacian_var(gray) / 500.0) * 100.0)
            brightness = 100.0 - abs(float(np.mean(gray)) - 128.0) / 128.0 * 100.0
            contrast = min(100.0, (float(np.std(gray)) / 50.0) * 100.0)
            lighting_score = (brightness + contrast) / 2.0
        except Exception as err:
            logger.debug("Blur/lighting analysis error: %s", err)
            blur_score = lighting_score = 50.0
        overall = size_score * 0.25 + position_score * 0.2 + aspect_score * 0.2 + blur_score * 0.2 + lighting_score * 0.15
        issues: List[str] = []
        if size_ratio < 0.05:
            issues.append("Face too small - move closer or crop image")
        if size_ratio > 0.8:
            issues.append("Face too large - image should show some background")
        if off > 0.4:
            issues.append("Face not centered - adjust framing")
        if aspect_ratio < 0.75:
            issues.append("Face appears distorted or at extreme angle")
        if blur_score < 40:
            issues.append("Image is blurry - use better focus or steady camera")
        if lighting_score < 40:
            issues.append("Poor lighting - improve lighting conditions")
        self._quality_history.append({"timestamp": datetime.now().isoformat(), "score": overall,
                                      "blur_score": blur_score, "lighting_score": lighting_score})
        return {"score": round(overall, 2), "size_score": round(size_score, 2), "position_score": round(position_score, 2),
                "aspect_score": round(aspect_score, 2), "blur_score": round(blur_score, 2),
                "lighting_score": round(lighting_score, 2), "issues": issues}

    # ------------------------------------------------------------------ gallery distances
    def _distances(self, encoding, names: Optional[List[str]] = None) -> Tuple[List[str], np.ndarray]:
        """names (dict order) and their distances to `encoding`: one device pass over the gallery."""
        G = self.ENCODINGS
        with G.locked():           # names, device rows and the score row belong to ONE gallery state
            targets = G.names() if names is None else [t for t in names if t in G]
            if not targets:
                return [], np.zeros((0,))
            if G.exact:            # float64 Euclidean distances to the rows as enrolled (face_recognition.face_distance, :410)
                d = self._eng().gallery_distances(np.asarray(encoding, dtype=np.float64).reshape(1, -1))[0]
                return targets, d[G.rows_of(targets)]
            q = np.asarray(encoding, dtype=np.float32).reshape(1, -1)
            cos = self._eng().match_scores(q)[0]
            return targets, cos_to_distance(cos[G.rows_of(targets)])

    # ------------------------------------------------------------------ store / delete (:344-390, :517-547)
    def store_face(self, target_name: str, encoding: np.ndarray) -> Dict[str, Any]:
        """:344-390.  The exclusive gallery lock (which stalls every streaming lane) is held for the duplicate scan and for
        the insert only; the storage write (DB round trip) and the on-disk backup run outside it, in the reference's order:
        scan -> store_embedding -> ENCODINGS[name] = ... -> backup."""
        try:
            enc = np.asarray(encoding, dtype=np.float64).reshape(-1)
            is_dup, similar = False, None
            with self.ENCODINGS.locked():          # scan and `already` against ONE gallery state
                if len(self.ENCODINGS):
                    names, d = self._distances(enc)
                    for n, dist in zip(names, d):          # first hit in dict order, as :353-364
                        if n != target_name and dist < 0.3:
                            is_dup, similar = True, n
                            logger.warning("Potential duplicate: %s ~ %s (distance=%.3f)", target_name, n, dist)
                            break
            if not self._storage.store_embedding(target_name, enc.tolist()):
                return {"success": False, "message": "Failed to store in database", "is_duplicate": is_dup}
            already = self.ENCODINGS.put(target_name, enc)     # exclusive inside; reaches every lane's copy
            try:
                self._backup_encoding_atomic(target_name, enc.tolist())
            except Exception as be:
                logger.warning("Backup failed for %s: %s", target_name, be)
            message = f"Face {'updated' if already else 'stored'} successfully for '{target_name}'"
            if is_dup:
                message += f" (Warning: Similar to '{similar}')"
            return {"success": True, "message": message, "is_duplicate": is_dup,
                    "similar_to": similar if is_dup else None, "was_update": already}
        except Exception as e:
            logger.exception("Error storing face %s: %s", target_name, e)
            return {"success": False, "message": f"Error storing face: {str(e)}", "is_duplicate": False}

    def delete_face(self, target_name: str) -> Dict[str, Any]:
        try:
            removed_mem = self.ENCODINGS.remove(target_name)
            self._remove_from_cache(target_name)
            removed_db = self._storage.delete(target_name) > 0
            try:
                bp = BACKUP_DIR / f"{target_name}_backup.json"
                if bp.exists():
                    bp.unlink()
            except Exception:
                logger.debug("Failed to remove backup for %s (non-fatal)", target_name)
            if removed_mem or removed_db:
                return {"success": True, "message": f"Face '{target_name}' deleted successfully",
                        "removed_from_memory": removed_mem, "removed_from_db": removed_db}
            return {"success": False, "message": f"Face '{target_name}' not found in database or memory"}
        except Exception as e:
            logger.exception("Error deleting face %s: %s", target_name, e)
            return {"success": False, "message": f"Error deleting face: {str(e)}"}

    def get_all_targets(self) -> List[str]:
        return self.ENCODINGS.names()

    # ------------------------------------------------------------------ compare (:395-443)
    def compare_faces(self, test_encoding: np.ndarray, target_names: Optional[List[str]] = None,
                      return_distances: bool = True) -> List[Dict[str, Any]]:
        start = time.time()
        try:
            targets, distances = self._distances(test_encoding, target_names)
            if not targets:
                logger.warning("No targets available for comparison")
                return []
            tol = self.tolerance
            now = datetime.now().isoformat
            results: List[Dict[str, Any]] = []
            for target, d in zip(targets, distances.tolist()):
                is_match = d <= tol
                item = {"target": target, "match": is_match}
                if return_distances:
                    item["distance"] = d
                    item["confidence"] = confidence_level(d)
                    item["confidence_score"] = calibrate_confidence(d)
                results.append(item)
                self._comparison_history.append({"distance": d, "match": is_match, "timestamp": now()})
            if return_distances:
                results.sort(key=lambda x: x.get("distance", 1.0))
            with self._metrics_lock:
                self._metrics["total_comparisons"] += len(targets)
                self._metrics["cumulative_comparison_time"] += time.time() - start
            return results
        except Exception as e:
            logger.exception("Error comparing faces: %s", e)
            return []

    def batch_compare_faces(self, test_encodings: List[np.ndarray], target_names: Optional[List[str]] = None):
        """:448-481 -- one device GEMM for all queries instead of a Python loop over them."""
        G = self.ENCODINGS
        if len(test_encodings) == 0:
            return []
        try:
            with G.locked():
                targets = G.names() if target_names is None else [t for t in target_names if t in G]
                if not targets:
                    return [[] for _ in test_encodings]
                if G.exact:
                    Q = np.stack([np.asarray(q, dtype=np.float64).reshape(-1) for q in test_encodings])
                    D = self._eng().gallery_distances(Q)[:, G.rows_of(targets)]
                else:
                    Q = np.stack([np.asarray(q, dtype=np.float32).reshape(-1) for q in test_encodings])
                    D = cos_to_distance(self._eng().match_scores(Q)[:, G.rows_of(targets)])
        except Exception as e:
            logger.exception("Error in batch comparison: %s", e)
            return [[] for _ in test_encodings]
        out = []
        for d in D:
            res = [{"target": targets[i], "match": True, "distance": float(d[i]), "confidence": confidence_level(d[i])}
                   for i in np.nonzero(d <= self.tolerance)[0]]
            res.sort(key=lambda x: x["distance"])
            out.append(res)
        return out

    def _get_confidence_level(self, distance: float) -> str:
        return confidence_level(distance)

    def _calibrate_confidence(self, distance: float) -> float:
        return calibrate_confidence(distance)

    # ------------------------------------------------------------------ clustering / k-NN (:552-612)
    def cluster_faces(self, distance_threshold: float = 0.6) -> Dict[str, List[str]]:
        G = self.ENCODINGS
        with G.locked():
            if len(G) < 2:
                return {"cluster_0": G.names()}
            names = G.names()
            rows = G.rows_of(names)
            emb = self._eng().gallery_get_exact()[rows] if G.exact else self._eng().gallery_get().astype(np.float32)[rows]
            clusters: Dict[str, List[str]] = {}
            assigned = np.zeros(len(names), dtype=bool)
            cid = 0
            # N x N on the device: the seeds of one greedy sweep are not known up front (a member never becomes
            # a seed), so score rows are produced in tiles of CLUSTER_TILE candidate seeds per gallery pass
            # (one frp_match_scores GEMM each) instead of one pass per seed
            tile_rows: Dict[int, int] = {}
            tile = None
            for i, name in enumerate(names):           # greedy seed order = dict order, as the reference
                if assigned[i]:
                    continue
                if i not in tile_rows:                 # next tile: the first CLUSTER_TILE unassigned names from i on
                    cand = [j for j in range(i, len(names)) if not assigned[j]][:CLUSTER_TILE]
                    tile_rows = {j: t for t, j in enumerate(cand)}
                    tile = (self._eng().gallery_distances(emb[cand])[:, rows] if G.exact
                            else cos_to_distance(self._eng().match_scores(emb[cand])[:, rows]))
                d = tile[tile_rows[i]]
                take = (~assigned) & (d <= distance_threshold)
                take[i] = True
                members = [name] + [names[j] for j in np.nonzero(take)[0] if j != i]
                assigned |= take
                clusters[f"cluster_{cid}"] = members
                cid += 1
            return clusters

    def find_k_nearest(self, test_encoding: np.ndarray, k: int = 5) -> List[Dict[str, Any]]:
        """face_service.py:590-612 with the distance pass AND the k-selection on the device (ties: the
        earlier gallery row first; the reference's argpartition leaves tie order unspecified)."""
        n = len(self.ENCODINGS)
        if n == 0:
            return []
        k = min(int(k), n)
        if k < 1:
            return []
        if self.ENCODINGS.exact or k > native.MAX_TOPK:   # exact rows (or k beyond the device limit): full distance row + host selection
            targets, distances = self._distances(test_encoding)
            idx = np.argsort(distances, kind="stable")[:k]
            pairs = [(targets[int(i)], float(distances[int(i)])) for i in idx]
        else:
            q = np.asarray(test_encoding, dtype=np.float32).reshape(1, -1)
            with self.ENCODINGS.locked():      # rows -> names against the gallery state the match saw
                rows, cos = self._eng().match(q, topk=k)
                rows, cos = np.atleast_2d(rows)[0], np.atleast_2d(cos)[0]
                pairs = [(self.ENCODINGS.name_of_row(int(r)), float(cos_to_distance(float(c)))) for r, c in zip(rows, cos) if r >= 0]
        return [{"target": t, "distance": d, "confidence": confidence_level(d),
                 "confidence_score": calibrate_confidence(d)} for t, d in pairs]

    # ------------------------------------------------------------------ streaming entry points (new)
    def process_frames(self, frames_bgr: np.ndarray, max_faces: int = 10, threshold: Optional[float] = None,
                       det_thresh: Optional[float] = None, all_matches: bool = False) -> List[List[Dict[str, Any]]]:
        """The live-loop body of routes/camera.py:225-259 for a batch of BGR frames, with the
        per-face compare + filter reduced to a fused device top-1: per frame a list of
        {bbox, kps, score, embedding, target, distance, cosine, confidence, match}.
        all_matches=True additionally lists EVERY enrolled target within both the service
        tolerance and `threshold` (the reference's exact loop semantics, camera.py:246-256: a face
        can hit several near-duplicate identities), ascending by distance, under "matches"."""
        return self._process_frames_on(self._eng(), self.ENCODINGS.locked(), frames_bgr, max_faces, threshold, det_thresh, all_matches)

    def _process_frames_on(self, eng, guard, frames_bgr, max_faces, threshold, det_thresh, all_matches, take_next=None, staged=None):
        """`take_next` / `staged` (process_stream on a real engine): the overlapped-ingest form - this batch goes to the device
        through the engine's staging buffer (already there when the previous call staged it), the batch the lane will get
        next is claimed and its upload started on the copy stream while this batch's kernels run."""
        tol = self.tolerance if threshold is None else min(self.tolerance, threshold)   # camera.py:250
        G = self.ENCODINGS
        overlapped = take_next is not None and staged is not None
        # The device returns gallery ROW indices; store/delete move rows (swap-remove).  The guard (exclusive for
        # process_frames, shared between the lanes of process_stream) is held over the device call and the
        # row -> name snapshot, so a concurrent delete can neither mis-attribute a face to the identity that was
        # moved into its row nor shrink the table under the lookup.
        if overlapped:
            key = id(frames_bgr)
            if staged.pop("key", None) != key:                         # not staged by the previous call: stage it now
                self._stage_on(eng, frames_bgr)
            eng.swap_frames()
        with guard:
            have_gallery = len(G) > 0
            dt = DET_THRESH if det_thresh is None else det_thresh
            fl = 0 if have_gallery else native.FLAG_NO_MATCH
            if overlapped:
                eng.process_resident(max_faces, det_thresh=dt, nms_iou=NMS_IOU, flags=fl)       # asynchronous
                nxt = take_next()
                if nxt is not None:
                    self._stage_on(eng, nxt)                           # copy stream: under this batch's kernels
                    staged["key"] = id(nxt)
                idle = getattr(take_next, "idle", None)
                if idle is not None:
                    idle()                                             # the PREVIOUS batch's result dicts, under this batch's kernels
                out = eng.fetch_results()                              # waits for the pass
            elif isinstance(frames_bgr, JpegBatch) and hasattr(eng, "upload_jpeg_async"):
                with eng.sequence():
                    self._stage_on(eng, frames_bgr)
                    eng.swap_frames()
                    eng.process_resident(max_faces, det_thresh=dt, nms_iou=NMS_IOU, flags=fl)
                    out = eng.fetch_results()
            else:
                if isinstance(frames_bgr, JpegBatch):
                    frames_bgr = frames_bgr.decode()                   # an engine without the device decoder (tests' FakeEngine)
                out = eng.process_frames(frames_bgr, max_faces=max_faces, det_thresh=dt, nms_iou=NMS_IOU, flags=fl)
            n_gallery = len(G)
            row_names = {int(r): G.name_of_row(int(r)) for r in np.unique(out["match_idx"]) if r >= 0}
            all_d = names = None
            if all_matches and have_gallery and int(out["counts"].sum()) > 0:
                Q = np.concatenate([out["emb"][b, :int(c)] for b, c in enumerate(out["counts"])])
                names = G.names()
                all_d = cos_to_distance(eng.match_scores(Q)[:, G.rows_of(names)])
        def build():
            # One pass of array arithmetic for the whole batch (distance, bucket, threshold), ONE tolist() per field; the per-face
            # work left is building the dict (the reference builds N of them per face, camera.py:243-259).
            counts = np.asarray(out["counts"], dtype=np.int64)
            K = out["match_idx"].shape[1]
            live = np.arange(K)[None, :] < counts[:, None]                    # [B, K]: slots that hold a face
            rows = out["match_idx"][live].astype(np.int64)
            hit = rows >= 0
            cos = out["match_cos"][live].astype(np.float64)
            dist = cos_to_distance(cos)
            conf = np.where(dist < 0.4, "high", np.where(dist < 0.6, "medium", "low"))         # confidence_level, vectorised
            is_match = hit & (dist <= tol)
            boxes, kps, scores = out["boxes"][live].tolist(), out["kps"][live].tolist(), out["scores"][live].tolist()
            emb = out["emb"][live]                                            # [n, 512]: one gather, rows handed out as views
            rows_l, hit_l, cos_l, dist_l, conf_l, match_l = rows.tolist(), hit.tolist(), cos.tolist(), dist.tolist(), conf.tolist(), is_match.tolist()
            result = []
            f_idx = 0
            for c in counts.tolist():
                faces = []
                for _ in range(c):
                    i = f_idx
                    h_ = hit_l[i]
                    face = {"bbox": boxes[i], "kps": kps[i], "score": scores[i], "embedding": emb[i],
                            "target": row_names.get(rows_l[i]) if h_ else None,
                            "distance": dist_l[i] if h_ else None, "cosine": cos_l[i] if h_ else None,
                            "confidence": conf_l[i] if h_ else None, "match": match_l[i]}
                    if all_matches:
                        hits = []
                        if all_d is not None:
                            dd = all_d[i]
                            order = np.argsort(dd, kind="stable")          # compare_faces sorts ascending (:432)
                            hits = [{"target": names[j], "distance": float(dd[j]), "confidence": confidence_level(float(dd[j]))}
                                    for j in order if dd[j] <= tol]
                        face["matches"] = hits
                    faces.append(face)
                    f_idx += 1
                result.append(faces)
            n_total = f_idx
            with self._metrics_lock:
                self._metrics["total_encodings"] += n_total
                self._metrics["total_comparisons"] += n_total * n_gallery
            return result

        # process_stream: the dicts of this batch are built by the lane's thread while its NEXT batch is on the device
        # (lanes.Deferred); everything they need was taken above, under the guard.
        return lanes.Deferred(build) if overlapped else build()

    @staticmethod
    def _stage_on(eng, batch) -> None:
        """start moving a batch into the lane's staging frame buffer: pixel arrays by DMA, encoded batches (mjpeg.JpegBatch)
        through the engine's JPEG path - bit streams decoded here on host threads, pixels produced on the copy stream -
        unless the device decoder does not cover the batch (progressive frames, mixed sampling): then PIL decodes it"""
        if isinstance(batch, JpegBatch):
            from . import ingest
            jp = ingest.device_jpeg_batch(batch, len(batch), batch.hw)
            if jp is not None:
                eng.upload_jpeg_async(jp)
                return
            batch = batch.decode()
        eng.upload_frames_async(batch)

    def process_stream(self, batches, max_faces: int = 10, threshold: Optional[float] = None,
                       det_thresh: Optional[float] = None, all_matches: bool = False):
        """process_frames for a stream of batches with TWO batches in flight on the GPU (lanes.py: a second handle and
        host thread; +7...13 % faces/s).  Yields one process_frames result per batch, in order.  Enrolment and deletion
        may run concurrently: they wait for the batches inside their device call and reach both gallery copies under
        the same lock, so every result is the one process_frames would have given for SOME gallery state between the
        batch's submission and its delivery.  Falls back to one lane when the second one cannot join (see _eng2)."""
        engines = [self._eng()]
        e2 = self._eng2()
        if e2 is not None:
            engines.append(e2)
        def on(eng):
            if not all(hasattr(eng, m) for m in ("upload_frames_async", "swap_frames", "process_resident", "fetch_results", "sequence")):
                return lambda frames, take_next: self._process_frames_on(eng, self.ENCODINGS.reading(), frames, max_faces, threshold,
                                                                         det_thresh, all_matches)
            staged = {}                         # which batch sits in this lane's staging buffer

            def fn(frames, take_next):
                with eng.sequence():            # upload -> swap -> process -> fetch of one lane is one uninterrupted sequence
                    return self._process_frames_on(eng, self.ENCODINGS.reading(), frames, max_faces, threshold, det_thresh, all_matches,
                                                   take_next, staged)
            return fn
        return lanes.run_ordered(batches, [on(e) for e in engines], prefetch=True)

    def frame_buffer(self, batch: int, height: int, width: int) -> np.ndarray:
        """A page-locked u8 [batch, height, width, 3] array (owned by the engine, freed with it) for the capture / decode
        threads to write frames into.  process_frames / process_stream accept any array; one that lives in page-locked
        memory is copied to the GPU by DMA at PCIe rate, an ordinary (pageable) numpy array goes through the runtime's
        staging copies first - at 32 x 1080p (199 MB per batch) that is the difference between the engine's rate and
        ~0.8 of it (bench.py: config.service_api).  Reference: frames come out of cv2.VideoCapture.read() as fresh
        pageable arrays (routes/camera.py:204-209); a capture loop on this service reads into these buffers instead."""
        return self._eng().host_frames(batch, height, width)

    def process_frame(self, frame_bgr_or_path, metadata: Optional[Dict[str, Any]] = None):
        if isinstance(frame_bgr_or_path, str):
            frame = np.ascontiguousarray(load_image_file(frame_bgr_or_path)[..., ::-1])
        else:
            frame = frame_bgr_or_path
        cfg = metadata or {}
        return self.process_frames(frame[None], max_faces=int(cfg.get("max_faces", 10)),
                                   threshold=cfg.get("confidence_threshold"))[0]

    # ------------------------------------------------------------------ statistics / housekeeping (:617-763)
    def get_quality_statistics(self) -> Dict[str, Any]:
        if not self._quality_history:
            return {"total_assessments": 0, "average_score": 0, "average_blur_score": 0, "average_lighting_score": 0}
        s = [q["score"] for q in self._quality_history]
        bl = [q["blur_score"] for q in self._quality_history]
        li = [q["lighting_score"] for q in self._quality_history]
        return {"total_assessments": len(s), "average_score": round(float(np.mean(s)), 2),
                "min_score": round(float(np.min(s)), 2), "max_score": round(float(np.max(s)), 2),
                "average_blur_score": round(float(np.mean(bl)), 2), "average_lighting_score": round(float(np.mean(li)), 2),
                "std_deviation": round(float(np.std(s)), 2)}

    def get_performance_metrics(self) -> Dict[str, Any]:
        with self._metrics_lock:
            m = dict(self._metrics)
        enc, cmp_ = m["total_encodings"], m["total_comparisons"]
        req = m["cache_hits"] + m["cache_misses"]
        attempts = enc + m["failed_encodings"]
        return {**m,
                "average_encoding_time": round(m["cumulative_encoding_time"] / enc, 4) if enc else 0.0,
                "average_comparison_time": round(m["cumulative_comparison_time"] / cmp_, 4) if cmp_ else 0.0,
                "cache_hit_rate": round(m["cache_hits"] / req * 100.0, 2) if req else 0.0,
                "encoding_success_rate": round(enc / attempts * 100.0, 2) if attempts else 100.0,
                "cache_size": len(self._encoding_cache), "total_faces_stored": len(self.ENCODINGS),
                "comparison_history_size": len(self._comparison_history)}

    def get_device_counters(self) -> Dict[str, Any]:
        """per-stage HIP-event times and algorithmic work from the native library (frp_get_counters)."""
        return self._eng().counters()

    def clear_cache(self) -> Dict[str, Any]:
        with self._cache_lock:
            n = len(self._encoding_cache)
            self._encoding_cache.clear()
        return {"success": True, "cleared_entries": n}

    def optimize_storage(self) -> Dict[str, Any]:
        cleaned = self._clean_cache()
        synced = 0
        for target in self.ENCODINGS.names():
            try:
                if not self._storage.has(target):
                    self._storage.store_embedding(target, self.ENCODINGS[target])
                    synced += 1
            except Exception as e:
                logger.debug("Sync failed for %s: %s", target, e)
        return {"success": True, "cache_entries_cleaned": cleaned, "database_synced": synced,
                "current_cache_size": len(self._encoding_cache), "total_faces": len(self.ENCODINGS)}

    def reset_metrics(self) -> Dict[str, Any]:
        with self._metrics_lock:
            old = dict(self._metrics)
            self._metrics = {k: (0.0 if k.startswith("cumulative") else 0) for k in _METRIC_KEYS}
        return {"success": True, "previous_metrics": old}

    def health_check(self) -> Dict[str, Any]:
        health = {"status": "healthy", "issues": [], "warnings": []}
        if len(self.ENCODINGS) == 0:
            health["warnings"].append("No faces enrolled in system")
        if len(self._encoding_cache) > 1000:
            health["warnings"].append(f"Large cache size: {len(self._encoding_cache)}")
        with self._metrics_lock:
            if self._metrics.get("failed_encodings", 0) > 100:
                health["warnings"].append(f"High failure rate: {self._metrics['failed_encodings']}")
        try:
            self._storage.ping()
        except Exception as db_err:
            health["status"] = "degraded"
            health["issues"].append(f"Database connectivity issue: {db_err}")
        if len(health["issues"]) > 2:
            health["status"] = "unhealthy"
        return health

    # ------------------------------------------------------------------ cache / backup helpers (:699-741)
    def _add_to_cache(self, key: str, data: Dict[str, Any]):
        with self._cache_lock:
            self._encoding_cache[key] = {"data": data, "timestamp": datetime.now()}

    def _get_from_cache(self, key: str) -> Optional[Dict[str, Any]]:
        with self._cache_lock:
            entry = self._encoding_cache.get(key)
            if not entry:
                return None
            if (datetime.now() - entry["timestamp"]).total_seconds() > self._cache_ttl:
                del self._encoding_cache[key]
                return None
            return entry["data"]

    def _remove_from_cache(self, key: str):
        with self._cache_lock:
            self._encoding_cache.pop(key, None)

    def _clean_cache(self) -> int:
        with self._cache_lock:
            now = datetime.now()
            dead = [k for k, v in self._encoding_cache.items() if (now - v["timestamp"]).total_seconds() > self._cache_ttl]
            for k in dead:
                del self._encoding_cache[k]
            return len(dead)

    def _backup_encoding_atomic(self, target_name: str, encoding: List[float]):
        BACKUP_DIR.mkdir(parents=True, exist_ok=True)
        final = BACKUP_DIR / f"{target_name}_backup.json"
        tmp = BACKUP_DIR / f"{target_name}_backup.json.tmp"
        with open(tmp, "w", encoding="utf-8") as f:
            json.dump({"target": target_name, "encoding": encoding, "timestamp": datetime.now().isoformat(), "version": 1},
                      f, indent=2)
            f.flush()
            os.fsync(f.fileno())
        tmp.replace(final)


# Module singleton, as face_service.py:769.  The HIP engine behind it is created on first use.
face_service = FaceService()

# Module-level hooks probed by services/async_task_manager.py:125
process_frames = face_service.process_frames
process_frame = face_service.process_frame
process_image = face_service.process_frame
recognize = face_service.process_frame
search_face = face_service.find_k_nearest
find_matches = face_service.compare_faces
