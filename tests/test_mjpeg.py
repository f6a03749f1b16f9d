"""Motion-JPEG ingest, host side: container demuxers, the capture surface, encoded batches through the mixer and the
service (fake engine: the batches are then decoded by PIL; the device decoder's turn is tests/test_gpu_pipeline.py)."""
import io
import os
import sys

import numpy as np
import pytest
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from fake_engine import FakeEngine  # noqa: E402
from frp_amd import mjpeg, mixer as mx, native  # noqa: E402
from frp_amd.face_service import FaceService  # noqa: E402
from oracle import jpeg as oj  # noqa: E402


def _frames(n, hw=(96, 128), seed=3, **kw):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        img = np.clip(rng.normal(120, 50, (hw[0] // 8, hw[1] // 8, 3)).repeat(8, 0).repeat(8, 1) + rng.normal(0, 6, hw + (3,)), 0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=85, **kw)
        out.append(b.getvalue())
    return out


def _strip_dht(d):
    out, i = bytearray(d[:2]), 2
    while True:
        m, L = d[i + 1], (d[i + 2] << 8) | d[i + 3]
        if m == 0xDA:
            return bytes(out + d[i:])
        if m != 0xC4:
            out += d[i:i + 2 + L]
        i += 2 + L


class _NoSeek:
    def __init__(self, b):
        self._b = io.BytesIO(b)

    def read(self, n=-1):
        return self._b.read(n)


def test_containers_give_back_the_frames_that_went_in():
    frames = _frames(7)
    thumb = b"Exif\0\0" + b"\xff\xd8\xff\xd9" * 3                      # an APP1 payload with SOI / EOI pairs inside: not a frame end
    b = io.BytesIO()
    Image.open(io.BytesIO(frames[2])).save(b, "JPEG", quality=85, exif=thumb)
    frames[2] = b.getvalue()
    raw = b"\x00junk" + b"".join(frames)
    for step in (1 << 20, 37, 5):
        got = list(mjpeg.split_stream(raw[i:i + step] for i in range(0, len(raw), step)))
        assert [bytes(g) for g in got] == frames, step
    for body in (mjpeg.write_multipart(frames), mjpeg.write_multipart(frames, with_length=False), mjpeg.write_avi(frames, (96, 128))):
        for src in (io.BytesIO(body), _NoSeek(body)):
            assert [bytes(g) for g in mjpeg.open_frames(src)] == frames
    assert [bytes(g) for g in mjpeg.open_frames(io.BytesIO(b"".join(frames)))] == frames
    # a frame cut in the middle is dropped at the next start-of-image, the stream goes on
    broken = frames[0] + frames[1][: len(frames[1]) // 2] + frames[2] + frames[3]
    got = [bytes(g) for g in mjpeg.split_stream([broken])]
    assert got[0] == frames[0] and got[-1] == frames[3] and len(got) in (3, 4)
    with pytest.raises(ValueError):
        list(mjpeg.avi_frames(io.BytesIO(b"RIFF\x00\x00\x00\x00WAVE")))
    h264 = bytearray(mjpeg.write_avi(frames[:1], (96, 128)))
    i = h264.find(b"vidsMJPG")
    h264[i + 4:i + 8] = b"H264"
    with pytest.raises(ValueError, match="not Motion-JPEG"):
        list(mjpeg.avi_frames(io.BytesIO(bytes(h264))))


def test_frames_without_huffman_tables_decode_with_the_annex_k_defaults():
    """AVI MJPG chunks and many cameras leave the DHT segment out: PIL (libjpeg-turbo) then assumes the Annex K tables; so does
    the host decoder - same coefficients, same pixels"""
    for kw in (dict(), dict(subsampling=0), dict(subsampling=1)):
        full = _frames(1, (120, 168), seed=9, **kw)[0]
        bare = _strip_dht(full)
        assert b"\xff\xc4" not in bare[: bare.find(b"\xff\xda")] and len(bare) < len(full)
        ref = np.asarray(Image.open(io.BytesIO(bare)).convert("RGB"))
        assert np.array_equal(ref, np.asarray(Image.open(io.BytesIO(full)).convert("RGB")))
        info, coef, q = native.jpeg_coefficients(bare)
        info_f, coef_f, q_f = native.jpeg_coefficients(full)
        assert info == info_f and np.array_equal(coef, coef_f) and np.array_equal(q, q_f)
        assert np.array_equal(oj.decode_from_coefficients(info, coef, q), ref)


def test_capture_surface_reopen_and_end_of_stream():
    frames = _frames(5)
    avi = mjpeg.write_avi(frames, (96, 128))
    opened = []

    def opener():
        opened.append(1)
        return io.BytesIO(avi)
    cap = mjpeg.MjpegCapture(opener)
    assert cap.isOpened()
    got = []
    while True:
        ok, f = cap.read()
        if not ok:
            break
        assert isinstance(f, mjpeg.JpegFrame)
        got.append(bytes(f))
    assert got == frames and cap.isOpened() and cap.frames_read == 5          # end of stream: still "open", read() fails (cv2 semantics)
    cap.release()
    assert not cap.isOpened() and cap.read() == (False, None)
    assert cap.open() and cap.read()[0] and len(opened) == 2
    assert not mjpeg.MjpegCapture(lambda: "/nonexistent/stream.avi").isOpened()


class _PixelEngine(FakeEngine):
    """a fake whose "detections" are a function of the pixels it is handed: one face per frame, box and embedding derived
    from the frame's content, matched against the fake gallery - any difference in the decoded pixels shows in the results"""

    def process_frames(self, frames, max_faces=10, det_thresh=0.5, nms_iou=0.4, flags=0):
        frames = np.asarray(frames)
        assert frames.dtype == np.uint8 and frames.ndim == 4
        B, K = len(frames), max_faces
        out = {"boxes": np.zeros((B, K, 4), np.float32), "kps": np.zeros((B, K, 5, 2), np.float32), "scores": np.zeros((B, K), np.float32),
               "counts": np.ones(B, np.int32), "emb": np.zeros((B, K, 512), np.float32),
               "match_idx": np.full((B, K), -1, np.int32), "match_cos": np.full((B, K), -2.0, np.float32)}
        for b, f in enumerate(frames):
            m = f.reshape(-1, 3).astype(np.float64).mean(0)
            out["boxes"][b, 0] = [m[0], m[1], m[0] + 40, m[1] + 50]
            out["scores"][b, 0] = 0.9
            e = np.resize(f[::7, ::5].astype(np.float32).ravel() - 128.0, 512)
            out["emb"][b, 0] = e / np.linalg.norm(e)
            if len(self.G):
                i, c = self.match(out["emb"][b, 0][None])
                out["match_idx"][b, 0], out["match_cos"][b, 0] = i[0], c[0]
        return out


def test_encoded_batches_through_the_mixer_and_the_service_equal_the_decoded_run():
    """two MJPEG cameras (one AVI file, one multipart stream, the second shorter) mixed into encoded batches; the service's
    results for them equal its results for the same frames decoded up front by PIL"""
    cams = {"usb0": _frames(6, seed=1), "ip7": _frames(4, seed=2)}
    def caps():
        return {"usb0": mjpeg.MjpegCapture(lambda: io.BytesIO(mjpeg.write_avi(cams["usb0"], (96, 128)))),
                "ip7": mjpeg.MjpegCapture(lambda: _NoSeek(mjpeg.write_multipart(cams["ip7"])))}
    geometry = [np.zeros((4, 96, 128, 3), np.uint8)]
    m = mx.StreamMixer(caps(), batch=4, buffers=geometry, encoded=True)
    batches = list(m)
    m.close()
    assert all(isinstance(b, mjpeg.JpegBatch) and b.shape == (4, 96, 128, 3) for b, _ in batches)
    metas = [meta for _, meta in batches]
    assert metas[0] == [("usb0", 0), ("ip7", 0), ("usb0", 1), ("ip7", 1)]
    assert metas[2] == [("usb0", 4), None, ("usb0", 5), None]                   # ip7 has ended: its slots repeat a frame, meta None
    assert bytes(batches[2][0][1]) == cams["usb0"][4]

    eng = _PixelEngine()
    svc = FaceService(engine=eng)
    rng = np.random.default_rng(0)
    for i in range(5):
        svc.store_face(f"p{i}", rng.standard_normal(512))
    m = mx.StreamMixer(caps(), batch=4, buffers=geometry, encoded=True)
    per_stream = {}
    for out in mx.run_mixed(svc, m, max_faces=3):
        for sid, rows in out.items():
            per_stream.setdefault(sid, []).extend(rows)
    m.close()
    assert [i for i, _ in per_stream["usb0"]] == list(range(6)) and [i for i, _ in per_stream["ip7"]] == list(range(4))
    for sid, rows in per_stream.items():
        for idx, faces in rows:
            rgb = np.asarray(Image.open(io.BytesIO(cams[sid][idx])).convert("RGB"))
            want = svc.process_frames(np.ascontiguousarray(rgb[..., ::-1])[None], max_faces=3)[0]
            assert len(faces) == len(want)
            for a, b in zip(faces, want):
                assert a["bbox"] == b["bbox"] and a["target"] == b["target"] and a["distance"] == b["distance"]
    # process_frames takes an encoded batch as well
    direct = svc.process_frames(batches[0][0], max_faces=3)
    assert len(direct) == 4


def test_sizes_taken_from_a_stream_are_not_believed_beyond_a_frame(monkeypatch):
    """round 4's advice: Content-Length (negative, huge), AVI chunk sizes and frames that never end are bounded"""
    fr = _frames(3)
    # multipart: a negative and an absurd Content-Length are ignored - the part is cut by its markers - and the stream goes on
    body = b""
    for i, f in enumerate(fr):
        cl = [b"-5", b"99999999999", str(len(f)).encode()][i]
        body += b"--b\r\nContent-Type: image/jpeg\r\nContent-Length: " + cl + b"\r\n\r\n" + f + b"\r\n"
    assert [bytes(x) for x in mjpeg.multipart_frames(io.BytesIO(body))] == fr
    # a frame that never ends is dropped once it has outgrown the limit; the frames behind it arrive
    endless = fr[0][:-2] + b"\x11" * 5000                                  # no EOI, 5 kB of entropy-like bytes
    stream = endless + fr[1] + fr[2]
    got = list(mjpeg.split_stream([stream[i:i + 512] for i in range(0, len(stream), 512)], max_frame_bytes=len(fr[0]) + 1000))
    assert [bytes(x) for x in got][-2:] == [fr[1], fr[2]] and len(got) <= 3
    got = list(mjpeg.multipart_frames(io.BytesIO(b"--b\r\n\r\n" + stream), max_frame_bytes=len(fr[0]) + 1000))
    assert [bytes(x) for x in got][-2:] == [fr[1], fr[2]]
    # AVI: a video chunk larger than the limit is skipped by its length, not read
    avi = mjpeg.write_avi(fr, (96, 128), fps=5)
    if avi is not None:
        assert [bytes(x) for x in mjpeg.avi_frames(io.BytesIO(avi))] == fr
        small = min(len(f) for f in fr)
        kept = [bytes(x) for x in mjpeg.avi_frames(io.BytesIO(avi), max_frame_bytes=small)]
        assert kept == [f for f in fr if len(f) <= small]


def test_jpeg_end_resumes_where_it_stopped():
    """the walk of a frame that arrives in pieces continues from its last position (state list) and finds the same end as a
    walk of the whole frame - for every chunking, including cuts inside markers, segment lengths and stuffed bytes"""
    f = _frames(1, hw=(160, 224), restart_marker_rows=1)[0] + b"\xff\xd8\xff"      # (a next frame's start behind it)
    whole = mjpeg.jpeg_end(f, 0)
    assert whole == len(f) - 3
    rng = np.random.default_rng(5)
    for trial in range(60):
        cuts = sorted(set(int(c) for c in rng.integers(1, len(f), size=int(rng.integers(1, 30)))))
        buf, state, end = bytearray(), [], None
        for a, b in zip([0] + cuts, cuts + [len(f)]):
            buf += f[a:b]
            end = mjpeg.jpeg_end(buf, 0, state)
            if end is not None:
                break
        assert end == whole, (trial, cuts)
    # one byte at a time
    buf, state = bytearray(), []
    for i in range(len(f)):
        buf.append(f[i])
        if mjpeg.jpeg_end(buf, 0, state) is not None:
            break
    assert mjpeg.jpeg_end(buf, 0, state) == whole
