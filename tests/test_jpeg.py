"""JPEG ingest, CPU side: the oracle's pixel pipeline (oracle/jpeg.py) is PINNED on PIL's decode of the committed
stills - the decode the reference's upload routes perform (face_recognition.load_image_file, face_service.py:139) -, and the
library's host half (csrc/jpeg_host.cpp through the C ABI, no GPU needed) is checked against an independent pure-Python
restatement of the entropy decoder and, through the oracle's pixel pipeline, against PIL.  The device half:
tests/test_gpu_pipeline.py::test_jpeg_stills_decode_on_the_device."""
import glob
import hashlib
import io
import json
import os

import numpy as np
import pytest
from PIL import Image

import frp_amd_loader  # noqa: F401
from frp_amd import native
from oracle import jpeg as oj

HERE = os.path.dirname(os.path.abspath(__file__))
STILLS = sorted(glob.glob(os.path.join(HERE, "golden", "stills", "*.jpg")))
META = json.load(open(os.path.join(HERE, "golden", "stills", "stills.json")))


def _pil_rgb(data: bytes) -> np.ndarray:
    return np.array(Image.open(io.BytesIO(data)).convert("RGB"))


def test_the_still_set_is_complete_and_pil_still_decodes_it_the_same_way():
    assert len(STILLS) == len(META) == 8
    for path in STILLS:
        name = os.path.basename(path)[:-4]
        data = open(path, "rb").read()
        assert len(data) == META[name]["bytes"]
        assert hashlib.md5(_pil_rgb(data).tobytes()).hexdigest() == META[name]["decoded_rgb_md5"], name
    assert b"\xff\xdd" in open(os.path.join(HERE, "golden", "stills", "c420_rst_q90.jpg"), "rb").read()        # restart intervals present


@pytest.mark.parametrize("path", STILLS, ids=[os.path.basename(p) for p in STILLS])
def test_oracle_and_host_decoder_reproduce_pil_bit_for_bit(path):
    data = open(path, "rb").read()
    ref = _pil_rgb(data)
    # oracle alone: pure-Python entropy decode + numpy pixel pipeline
    info_o, coef_o, q_o = oj.huffman_decode(data)
    assert np.array_equal(oj.decode_from_coefficients(info_o, coef_o, q_o), ref)
    # the library's host half: same header, same coefficients, same tables
    info, coef, q = native.jpeg_coefficients(data)
    assert info == info_o
    assert np.array_equal(coef, coef_o) and np.array_equal(q[:info["components"]], q_o[:info["components"]])
    assert np.array_equal(oj.decode_from_coefficients(info, coef, q), ref)


@pytest.mark.parametrize("kw", [dict(quality=90), dict(quality=50, subsampling=0), dict(quality=97, subsampling=1), dict(quality=75, optimize=True),
                                dict(quality=85, restart_marker_rows=2)])
def test_generated_camera_size_stills(kw):
    """720p / odd-sized stills written on the spot (not committed): host decoder + oracle pixel pipeline == PIL"""
    rng = np.random.default_rng(len(repr(kw)))
    for (h, w) in ((720, 1280), (243, 517)):
        img = np.clip(rng.normal(120, 50, (h // 8 + 1, w // 8 + 1, 3)).repeat(8, 0).repeat(8, 1)[:h, :w] + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", **kw)
        data = b.getvalue()
        info, coef, q = native.jpeg_coefficients(data)
        assert (info["height"], info["width"]) == (h, w)
        assert np.array_equal(oj.decode_from_coefficients(info, coef, q), _pil_rgb(data))


def test_long_codes_large_magnitudes_and_dense_ff_bytes():
    """quality-100 noise: every block full of large coefficients (magnitudes beyond the direct table's 8 bits, codes beyond its
    look-ahead, a stuffed 0xFF every few dozen bytes) - the general path of the host decoder, against PIL"""
    rng = np.random.default_rng(5)
    img = (rng.integers(0, 2, (34, 50, 3)) * 255).repeat(4, 0).repeat(4, 1).astype(np.uint8) ^ rng.integers(0, 32, (136, 200, 3), dtype=np.uint8)
    for kw in (dict(quality=100, subsampling=0), dict(quality=100, subsampling=2, optimize=True), dict(quality=100, subsampling=1, restart_marker_blocks=3)):
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", **kw)
        data = b.getvalue()
        assert data.count(b"\xff\x00") > 50
        info, coef, q = native.jpeg_coefficients(data)
        assert np.abs(coef).max() > 300
        assert np.array_equal(oj.decode_from_coefficients(info, coef, q), _pil_rgb(data)), kw


def test_damaged_scans_never_crash_the_host_decoder():
    """bytes of the entropy-coded segment overwritten at random: the call returns coefficients or reports the stream as corrupt;
    it never reads past the buffer (the buffer handed over ends exactly at the file's last byte)"""
    good = open(os.path.join(HERE, "golden", "stills", "c420_q90.jpg"), "rb").read() if os.path.exists(os.path.join(HERE, "golden", "stills", "c420_q90.jpg")) else open(STILLS[0], "rb").read()
    sos = good.find(b"\xff\xda")
    rng = np.random.default_rng(11)
    outcomes = set()
    for trial in range(120):
        bad = bytearray(good)
        for _ in range(int(rng.integers(1, 6))):
            i = int(rng.integers(sos + 14, len(bad) - 2))
            bad[i] = int(rng.integers(0, 256))
        if trial % 3 == 0:
            bad = bad[: int(rng.integers(sos + 20, len(bad)))]
        try:
            info, coef, q = native.jpeg_coefficients(bytes(bad))
            outcomes.add("decoded")
            assert coef.shape[0] > 0
        except native.FrpError:
            outcomes.add("refused")
    assert "decoded" in outcomes


def test_files_outside_the_decoders_scope_are_refused_not_half_decoded():
    img = Image.fromarray(np.random.default_rng(1).integers(0, 256, (40, 56, 3), dtype=np.uint8))
    b = io.BytesIO()
    img.save(b, "JPEG", progressive=True)
    assert native.jpeg_info(b.getvalue()) is None                          # progressive
    b = io.BytesIO()
    img.convert("CMYK").save(b, "JPEG")
    assert native.jpeg_info(b.getvalue()) is None                          # four components
    b = io.BytesIO()
    img.save(b, "PNG")
    assert native.jpeg_info(b.getvalue()) is None                          # not a JPEG
    good = open(STILLS[0], "rb").read()
    assert native.jpeg_info(good[:60]) is None                             # truncated inside the headers
    with pytest.raises(native.FrpError):
        native.jpeg_coefficients(good[:20])
    # a scan cut short decodes to zeros behind the cut (no crash, no read past the buffer); a corrupt table is an error
    info, coef, q = native.jpeg_coefficients(good[: len(good) * 2 // 3] + b"\xff\xd9")
    assert info["width"] > 0 and coef.shape[0] > 0
    bad = bytearray(good)
    i = bad.find(b"\xff\xc4")
    bad[i + 5:i + 21] = b"\xff" * 16                                       # an over-subscribed Huffman table
    assert native.jpeg_info(bytes(bad)) is None
