#!/usr/bin/env python3
"""Generate plumbing golden vectors from the reference's own FaceService logic.

Runs ONLY in the build container (needs /root/reference, which never travels to
the GPU box).  The reference module is imported from where it lies; the absent
third-party modules it pulls in at import time (cv2, socketio, dotenv, pymongo
database) are replaced by inert `sys.modules` stubs, and `face_recognition` by
its published one-line definition of `face_distance`
(face_recognition 1.3.0 api.py: `np.linalg.norm(face_encodings - face_to_compare, axis=1)`,
empty input -> empty output).  Everything that is recorded below is computed by
the reference's own Python code in backend/app/services/face_service.py
(compare_faces :395-443, batch_compare_faces :448-481, _get_confidence_level
:486-492, _calibrate_confidence :497-506, cluster_faces :552-585,
find_k_nearest :590-612, assess_face_quality geometry terms :251-275,
store_face duplicate scan :349-364).

Output: tests/golden/plumbing_golden.npz + plumbing_golden.json (inputs and
expected outputs only -- data, no reference source).
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference/backend"
OUT_DIR = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    cv2 = types.ModuleType("cv2")

    def _boom(*a, **k):
        raise RuntimeError("cv2 stub")

    cv2.cvtColor = _boom
    cv2.Laplacian = _boom
    cv2.COLOR_RGB2GRAY = 7
    cv2.CV_64F = 6
    cv2.VideoCapture = object
    sys.modules["cv2"] = cv2

    fr = types.ModuleType("face_recognition")

    def face_distance(face_encodings, face_to_compare):
        if len(face_encodings) == 0:
            return np.empty((0))
        return np.linalg.norm(np.asarray(face_encodings) - face_to_compare, axis=1)

    fr.face_distance = face_distance
    sys.modules["face_recognition"] = fr

    # app.state: only ENCODINGS is used by face_service
    app = types.ModuleType("app")
    app.__path__ = [os.path.join(REF, "app")]
    sys.modules["app"] = app
    state = types.ModuleType("app.state")
    state.ENCODINGS = {}
    sys.modules["app.state"] = state
    utils = types.ModuleType("app.utils")
    utils.__path__ = [os.path.join(REF, "app", "utils")]
    sys.modules["app.utils"] = utils
    db = types.ModuleType("app.utils.db")
    db.store_embedding = lambda target, emb: True
    db.retrieve_embedding = lambda target: []

    class _Coll:
        def delete_one(self, q):
            return types.SimpleNamespace(deleted_count=0)

        def find_one(self, *a, **k):
            return None

    db.faces_collection = _Coll()
    sys.modules["app.utils.db"] = db
    lg = types.ModuleType("app.utils.logger")
    import logging

    lg.get_logger = lambda name=None: logging.getLogger(name or "ref")
    sys.modules["app.utils.logger"] = lg
    services = types.ModuleType("app.services")
    services.__path__ = [os.path.join(REF, "app", "services")]
    sys.modules["app.services"] = services
    return state


def main():
    scratch = tempfile.mkdtemp(prefix="frp_golden_")
    os.chdir(scratch)  # the reference creates data/backups at import
    state = _install_stubs()
    import importlib

    fs_mod = importlib.import_module("app.services.face_service")
    fs = fs_mod.FaceService()
    ENC = state.ENCODINGS

    arrays = {}
    meta = {"cases": []}

    # --- confidence mapping known answers
    ds = [0.0, 0.1, 0.2, 0.3, 0.3999, 0.4, 0.45, 0.5, 0.5999, 0.6, 0.61, 0.8, 1.0, 1.2, 1.4142135, 2.0]
    meta["confidence"] = [
        {"d": d, "level": fs._get_confidence_level(d), "score": fs._calibrate_confidence(d)} for d in ds
    ]

    def strip(results):
        out = []
        for r in results:
            r = dict(r)
            out.append(r)
        return out

    case_id = 0
    for (N, D, seed, tol) in [(5, 128, 1, 0.6), (16, 512, 2, 0.6), (64, 512, 3, 1.1), (33, 128, 4, 0.45)]:
        rng = np.random.default_rng(seed)
        G = rng.standard_normal((N, D))
        G /= np.linalg.norm(G, axis=1, keepdims=True)
        names = [f"person_{seed}_{i:03d}" for i in range(N)]
        # queries: noisy copies of a few rows + one unrelated + an exact copy
        Q = []
        for j, noise in enumerate([0.0, 0.02, 0.05, 0.2]):
            q = G[(j * 3) % N] + noise * rng.standard_normal(D) / np.sqrt(D) * np.sqrt(D) * 0.05 if noise else G[(j * 3) % N].copy()
            q = q / np.linalg.norm(q)
            Q.append(q)
        q = rng.standard_normal(D)
        Q.append(q / np.linalg.norm(q))
        Q = np.stack(Q)
        ENC.clear()
        for n, g in zip(names, G):
            ENC[n] = g.tolist()
        fs.tolerance = tol
        key = f"case{case_id}"
        arrays[key + "_G"] = G
        arrays[key + "_Q"] = Q
        case = {"id": case_id, "N": N, "D": D, "tolerance": tol, "names": names,
                "compare": [], "knn": [], "batch": None, "subset": None}
        for q in Q:
            case["compare"].append(strip(fs.compare_faces(q)))
            case["knn"].append({str(k): strip(fs.find_k_nearest(q, k=k)) for k in (1, 5, N + 7)})
        case["batch"] = [strip(r) for r in fs.batch_compare_faces(list(Q))]
        sub = names[::2] + ["not_enrolled"]
        case["subset"] = {"target_names": sub, "result": strip(fs.compare_faces(Q[1], target_names=sub))}
        case["no_dist"] = strip(fs.compare_faces(Q[1], return_distances=False))
        case["clusters"] = {str(t): fs.cluster_faces(distance_threshold=t) for t in (0.6, 1.3, 1.45)}
        # store_face duplicate scan (first hit in dict order, d < 0.3)
        dup_probe = G[2] + 0.005 * rng.standard_normal(D)
        arrays[key + "_dup"] = dup_probe
        before = list(ENC.keys())
        r = fs.store_face("new_person", dup_probe)
        case["store_dup"] = r
        r2 = fs.store_face(names[0], G[0])
        case["store_update"] = r2
        case["targets_after"] = fs.get_all_targets()
        meta["cases"].append(case)
        case_id += 1

    # --- tolerance edge: distance exactly == tolerance must match (<=)
    ENC.clear()
    ENC["a"] = [0.0, 0.0, 0.0, 0.0]
    ENC["b"] = [0.6, 0.0, 0.0, 0.0]
    ENC["c"] = [0.0, 0.8, 0.0, 0.0]
    fs.tolerance = 0.6
    meta["edge_tolerance"] = strip(fs.compare_faces(np.zeros(4)))
    # --- empty gallery
    ENC.clear()
    meta["empty_compare"] = fs.compare_faces(np.zeros(4))
    meta["empty_knn"] = fs.find_k_nearest(np.zeros(4), k=3)
    meta["empty_batch"] = fs.batch_compare_faces([np.zeros(4), np.ones(4)])
    meta["empty_clusters"] = fs.cluster_faces()
    ENC["solo"] = [1.0, 0.0]
    meta["one_clusters"] = fs.cluster_faces()

    # --- assess_face_quality: geometry terms are reference arithmetic; the cv2
    # stub raises, so blur/lighting take the reference's own fallback value 50.0
    # (face_service.py:282-284, :295-297).
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    meta["quality"] = []
    for loc in [(100, 400, 300, 200), (0, 640, 480, 0), (10, 60, 40, 20), (200, 330, 280, 310), (50, 600, 120, 100)]:
        fsq = fs_mod.FaceService()
        meta["quality"].append({"shape": [480, 640, 3], "loc": list(loc), "result": fsq.assess_face_quality(img, loc)})

    # --- metrics bookkeeping after a known sequence
    fsm = fs_mod.FaceService()
    ENC.clear()
    for i in range(3):
        ENC[f"m{i}"] = [float(i), 0.0]
    fsm.compare_faces(np.zeros(2))
    fsm.compare_faces(np.ones(2))
    m = fsm.get_performance_metrics()
    meta["metrics_keys"] = sorted(m.keys())
    meta["metrics_total_comparisons"] = m["total_comparisons"]
    meta["metrics_history"] = m["comparison_history_size"]
    meta["health_empty"] = None
    ENC.clear()
    h = fs_mod.FaceService().health_check()
    meta["health_empty"] = h

    np.savez_compressed(os.path.join(OUT_DIR, "plumbing_golden.npz"), **arrays)
    with open(os.path.join(OUT_DIR, "plumbing_golden.json"), "w") as f:
        json.dump(meta, f, indent=None, sort_keys=False)
    print("wrote", OUT_DIR, "cases:", len(meta["cases"]))


if __name__ == "__main__":
    main()
