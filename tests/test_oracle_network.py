"""Network oracle: regression pins (tests/golden/network_golden.npz) + published-definition checks."""
import importlib.util
import os

import numpy as np

from oracle import network as onet

HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location("mkgold", os.path.join(HERE, "golden", "make_network_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.build()[0]


def test_oracle_outputs_are_frozen():
    gold = np.load(os.path.join(HERE, "golden", "network_golden.npz"))
    now = _gen()
    for k in gold.files:
        a, b = gold[k], now[k]
        assert a.shape == b.shape, k
        if a.dtype.kind == "f":
            assert np.allclose(a, b, rtol=1e-5, atol=1e-5), k     # torch-CPU conv may reorder sums across builds
        else:
            assert np.array_equal(a, b), k


def test_umeyama_recovers_known_similarity():
    rng = np.random.default_rng(0)
    ang, s, t = 0.3, 1.8, np.array([40.0, -12.0])
    R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    src = rng.uniform(0, 100, (5, 2))
    dst = s * src @ R.T + t
    M = onet.umeyama_similarity(src, dst)
    assert np.allclose(M[:, :2], s * R, atol=1e-9) and np.allclose(M[:, 2], t, atol=1e-9)
    # identity on the template itself
    assert np.allclose(onet.umeyama_similarity(onet.ARCFACE_TEMPLATE, onet.ARCFACE_TEMPLATE), [[1, 0, 0], [0, 1, 0]], atol=1e-6)


def test_warp_identity_and_border():
    img = np.random.default_rng(1).integers(0, 256, (112, 112, 3), dtype=np.uint8)
    same = onet.warp_affine_bilinear(img, np.array([[1.0, 0, 0], [0, 1.0, 0]]))
    assert np.array_equal(same, img.astype(np.float32))
    shifted = onet.warp_affine_bilinear(img, np.array([[1.0, 0, 200.0], [0, 1.0, 0]]))   # everything maps outside
    assert np.all(shifted == 0)
    half = onet.warp_affine_bilinear(img, np.array([[1.0, 0, 0.5], [0, 1.0, 0]]))
    assert np.allclose(half[:, 1:], 0.5 * (img[:, :-1].astype(np.float32) + img[:, 1:]))
    assert np.allclose(half[:, 0], 0.5 * img[:, 0])                                       # outside tap contributes 0


def test_decode_semantics():
    heads = [np.zeros((64 // s, 64 // s, 32), np.float16) for s in (8, 16, 32)]
    for h in heads:
        h[..., [0, 15]] = -5
    heads[1][1, 2, 0] = 3.0            # stride 16, anchor 0 at (x=2,y=1): centre (32,16)
    heads[1][1, 2, 1:5] = [1, 0.5, 2, 1.5]
    heads[1][1, 2, 5:15] = np.arange(10) * 0.25
    b, k, s, a = onet.decode_nms(heads, 0.5, 0.4, 5)
    assert len(b) == 1 and a[0] == 64 * 2 + (1 * 4 + 2) * 2
    assert np.allclose(b[0], [32 - 16, 16 - 8, 32 + 32, 16 + 24])
    assert np.allclose(k[0, :, 0], 32 + np.arange(0, 10, 2) * 0.25 * 16) and np.allclose(k[0, :, 1], 16 + np.arange(1, 10, 2) * 0.25 * 16)
    assert abs(s[0] - 1 / (1 + np.exp(-3.0))) < 1e-6
    assert onet.logit_threshold(0.5) == 0.0 and np.isneginf(onet.logit_threshold(0.0))
    assert len(onet.decode_nms(heads, 0.99, 0.4, 5)[0]) == 0


def test_match_and_distance_identities():
    rng = np.random.default_rng(2)
    G = rng.standard_normal((50, 512))
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    idx, cos = onet.match_topk(G, G[[7, 3]], 2)
    assert list(idx[:, 0]) == [7, 3] and np.allclose(cos[:, 0], 1.0)
    d = onet.cos_to_distance(G @ G[7])
    assert np.allclose(d, np.linalg.norm(G - G[7], axis=1), atol=1e-7)    # unit rows: d^2 = 2 - 2cos
