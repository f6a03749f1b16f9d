"""Stream mixer (SURVEY.md 8f-4 / BASELINE config 5): per-stream frame_skip / fps_limit as routes/camera.py:204-221, streams
mixed round-robin into batches, encoded stills decoded into the batch buffers, results handed back per stream."""
import io

import numpy as np
import pytest

from frp_amd import mixer as mx


def _frames(sid, n, H=6, W=8):
    """frame i of stream sid: every pixel = (sid * 50 + i) % 256 in B, sid in G, i in R"""
    f = np.zeros((n, H, W, 3), np.uint8)
    for i in range(n):
        f[i, ..., 0] = (sid * 50 + i) % 256
        f[i, ..., 1] = sid
        f[i, ..., 2] = i
    return f


def test_mixer_interleaves_streams_applies_frame_skip_and_pads_ended_streams():
    streams = {7: mx.SyntheticStream(_frames(7, 10)), 3: mx.SyntheticStream(_frames(3, 4)), 9: None}
    bufs = [np.full((4, 6, 8, 3), 255, np.uint8) for _ in range(3)]
    m = mx.StreamMixer(streams, batch=4, buffers=bufs, frame_skip=2)
    got = [(buf.copy(), meta) for buf, meta in m]
    m.close()
    # kept frames: stream 7 -> raw frames 1, 3, 5, 7, 9 (the last of every two reads); stream 3 -> 1, 3; stream 9 never opens.
    # slots go 7, 3, 9, 7 / 7, 3, 9, 7 / ...: slot 2 is always empty
    assert [meta for _, meta in got] == [[(7, 0), (3, 0), None, (7, 1)], [(7, 2), (3, 1), None, (7, 3)], [(7, 4), None, None, None]]
    b0, b1, b2 = (b for b, _ in got)
    assert b0[0, 0, 0].tolist() == [(7 * 50 + 1) % 256, 7, 1] and b0[1, 0, 0].tolist() == [(3 * 50 + 1) % 256, 3, 1]
    assert b0[3, 0, 0, 2] == 3 and b1[0, 0, 0, 2] == 5 and b1[1, 0, 0, 2] == 3 and b1[3, 0, 0, 2] == 7 and b2[0, 0, 0, 2] == 9
    assert np.all(b0[2] == 0) and np.all(b2[1:] == 0)                      # empty slots are zeroed, not stale
    assert streams[7].reads == 11 and streams[3].reads == 5                 # 2 reads per kept frame + the failed one
    d = mx.demix(got[1][1], ["a", "b", "c", "d"])
    assert d == {7: [(2, "a"), (3, "d")], 3: [(1, "b")]}
    with pytest.raises(ValueError):
        mx.StreamMixer({}, 4, bufs)
    with pytest.raises(ValueError):
        mx.StreamMixer(streams, 4, [np.zeros((3, 6, 8, 3), np.uint8)])


def test_mixer_decodes_stills_reopens_once_and_limits_fps():
    from PIL import Image
    raw = _frames(1, 3, 16, 16)
    stills = []
    for f in raw:
        bio = io.BytesIO()
        Image.fromarray(f[..., ::-1].copy()).save(bio, format="PNG")       # a camera sends RGB stills; the batch holds BGR
        stills.append(bio.getvalue())
    now = [100.0]
    slept = []

    def sleep(dt):
        slept.append(dt)
        now[0] += dt
    dropped = mx.SyntheticStream(_frames(2, 4, 16, 16), fail_at=1)           # the connection drops at the second read
    m = mx.StreamMixer({"jpg": mx.SyntheticStream(stills), "live": dropped}, batch=2, buffers=[np.zeros((2, 16, 16, 3), np.uint8)] * 2,
                       fps_limit={"jpg": 4.0}, decode_workers=2, clock=lambda: now[0], sleep=sleep)
    out = [(b.copy(), meta) for b, meta in m]
    m.close()
    assert [meta for _, meta in out] == [[("jpg", 0), ("live", 0)], [("jpg", 1), None], [("jpg", 2), ("live", 1)], [None, ("live", 2)],
                                         [None, ("live", 3)]]
    for i in range(3):
        assert np.array_equal(out[i][0][0], raw[i])                        # PNG stills decoded straight into the batch slot, as BGR
    assert np.array_equal(out[2][0][1], _frames(2, 4, 16, 16)[1])           # after ONE reopen the stream continues where it was
    assert len(slept) == 3 and all(abs(s - 0.25) < 1e-9 for s in slept)     # 4 fps: 0.25 s before every read after the first (the last one finds the end)
    bad = mx.StreamMixer({0: mx.SyntheticStream(_frames(0, 1, 5, 5))}, 1, [np.zeros((1, 16, 16, 3), np.uint8)])
    with pytest.raises(ValueError):
        list(bad)
    bad.close()


def test_run_mixed_hands_results_back_per_stream():
    class Svc:
        def process_stream(self, batches, **kw):
            for b in batches:                                            # one "result" per slot: (G, R) of the slot's first pixel
                yield [(int(f[0, 0, 1]), int(f[0, 0, 2])) for f in b]

    streams = {s: mx.SyntheticStream(_frames(s, 6)) for s in (4, 5)}
    m = mx.StreamMixer(streams, batch=4, buffers=[np.zeros((4, 6, 8, 3), np.uint8) for _ in range(2)])
    seen = {4: [], 5: []}
    for per_stream in mx.run_mixed(Svc(), m, max_faces=3):
        for sid, items in per_stream.items():
            for idx, (g, r) in items:
                assert g == sid and r == idx
                seen[sid].append(idx)
    m.close()
    assert seen == {4: list(range(6)), 5: list(range(6))}
