"""The plumbing oracle must reproduce the golden vectors captured from the reference's own
FaceService (tests/golden/make_plumbing_golden.py) -- this is what pins it."""
import json
import os

import numpy as np
import pytest

from oracle import plumbing

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    meta = json.load(open(os.path.join(HERE, "golden", "plumbing_golden.json")))
    arrays = np.load(os.path.join(HERE, "golden", "plumbing_golden.npz"))
    return meta, arrays


def _same(a, b, tol=1e-12):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert list(x.keys()) == list(y.keys())          # same keys, same order
        for k in x:
            if isinstance(x[k], float):
                assert abs(x[k] - y[k]) <= tol, (k, x[k], y[k])
            else:
                assert x[k] == y[k], (k, x[k], y[k])


def test_confidence_known_answers(golden):
    meta, _ = golden
    for row in meta["confidence"]:
        assert plumbing.confidence_level(row["d"]) == row["level"]
        assert plumbing.calibrate_confidence(row["d"]) == row["score"]
    # SURVEY.md 8a row a10
    assert [plumbing.calibrate_confidence(d) for d in (0, 0.3, 0.5, 0.6, 1.0)] == [99.75, 91.68, 50.0, 23.15, 0.25]


def test_compare_knn_batch_cluster_store(golden):
    meta, arrays = golden
    for case in meta["cases"]:
        G, Q = arrays[f"case{case['id']}_G"], arrays[f"case{case['id']}_Q"]
        o = plumbing.PlumbingOracle(case["tolerance"])
        for n, g in zip(case["names"], G):
            o.ENCODINGS[n] = g.tolist()
        for q, exp, knn in zip(Q, case["compare"], case["knn"]):
            _same(o.compare_faces(q), exp)
            for k, e in knn.items():
                _same(o.find_k_nearest(q, int(k)), e)
        got = o.batch_compare_faces(list(Q))
        for g_, e in zip(got, case["batch"]):
            _same(g_, e)
        _same(o.compare_faces(Q[1], target_names=case["subset"]["target_names"]), case["subset"]["result"])
        _same(o.compare_faces(Q[1], return_distances=False), case["no_dist"])
        for t, exp in case["clusters"].items():
            assert o.cluster_faces(float(t)) == exp
        assert o.store_face("new_person", arrays[f"case{case['id']}_dup"]) == case["store_dup"]
        assert o.store_face(case["names"][0], G[0]) == case["store_update"]
        assert list(o.ENCODINGS.keys()) == case["targets_after"]


def test_edges(golden):
    meta, _ = golden
    o = plumbing.PlumbingOracle(0.6)
    o.ENCODINGS.update({"a": [0.0] * 4, "b": [0.6, 0, 0, 0], "c": [0, 0.8, 0, 0]})
    _same(o.compare_faces(np.zeros(4)), meta["edge_tolerance"])       # d == tolerance matches (<=)
    e = plumbing.PlumbingOracle()
    assert e.compare_faces(np.zeros(4)) == meta["empty_compare"] == []
    assert e.find_k_nearest(np.zeros(4), 3) == meta["empty_knn"] == []
    assert e.batch_compare_faces([np.zeros(4), np.ones(4)]) == meta["empty_batch"]
    assert e.cluster_faces() == meta["empty_clusters"]
    e.ENCODINGS["solo"] = [1.0, 0.0]
    assert e.cluster_faces() == meta["one_clusters"]


def test_quality_geometry(golden):
    meta, _ = golden
    for q in meta["quality"]:
        assert plumbing.quality_geometry(tuple(q["shape"]), tuple(q["loc"])) == q["result"]


def test_camera_filter_loop():
    rng = np.random.default_rng(0)
    G = rng.standard_normal((20, 16))
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    o = plumbing.PlumbingOracle(0.6)
    for i, g in enumerate(G):
        o.ENCODINGS[f"p{i}"] = g.tolist()
    hits = plumbing.camera_filter_loop(o, 3, [G[4], G[7] + 0.01, rng.standard_normal(16)], 0.5)
    assert [h["target"] for h in hits] == ["p4", "p7"] and all(h["camera_id"] == 3 for h in hits)
