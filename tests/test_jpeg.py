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
    # a scan cut short is an ERROR (PIL: "image file is truncated"), not a frame that turns grey behind the cut; a file that
    # only lacks its EOI marker has every MCU and decodes
    with pytest.raises(native.FrpError):
        native.jpeg_coefficients(good[: len(good) * 2 // 3] + b"\xff\xd9")
    with pytest.raises(native.FrpError):
        native.jpeg_coefficients(good[: len(good) * 2 // 3])
    assert good.endswith(b"\xff\xd9")
    info, coef, q = native.jpeg_coefficients(good[:-2])
    info_f, coef_f, _ = native.jpeg_coefficients(good)
    assert info == info_f and np.array_equal(coef, coef_f)
    bad = bytearray(good)
    i = bad.find(b"\xff\xc4")
    bad[i + 5:i + 21] = b"\xff" * 16                                       # an over-subscribed Huffman table
    assert native.jpeg_info(bytes(bad)) is None


def test_truncation_inside_a_restart_interval_is_an_error_too():
    good = open(os.path.join(HERE, "golden", "stills", "c420_rst_q90.jpg"), "rb").read()
    sos = good.find(b"\xff\xda")
    rst = [i for i in range(sos, len(good) - 1) if good[i] == 0xFF and 0xD0 <= good[i + 1] <= 0xD7]
    assert len(rst) >= 3
    # half of the second interval's bytes removed: the interval ends (at its RST marker) before its MCUs are complete
    a, b = rst[0] + 2, rst[1]
    cut = good[:a + (b - a) // 2] + good[b:]
    with pytest.raises(native.FrpError):
        native.jpeg_coefficients(cut)


def _sof_patched(data: bytes, h: int, w: int) -> bytes:
    i = data.find(b"\xff\xc0")
    return data[:i + 5] + bytes([h >> 8, h & 255, w >> 8, w & 255]) + data[i + 9:]


def test_dimensions_from_untrusted_bytes_are_capped_before_anything_is_sized(monkeypatch):
    """round 4's advice: a ~200-byte header claiming 65,535 x 65,535 sized a 12.9 GB page-locked buffer and 32 GB of device
    buffers.  The limit is PIL's Image.MAX_IMAGE_PIXELS (89,478,485; FRP_JPEG_MAX_PIXELS overrides it)."""
    good = open(STILLS[0], "rb").read()
    huge = _sof_patched(good, 65535, 65535)
    assert native.jpeg_info(huge) is None
    with pytest.raises(native.FrpError):
        native.jpeg_coefficients(huge)
    assert native.jpeg_info(_sof_patched(good, 9459, 9459)) is not None      # 89,472,681 pixels: just inside
    assert native.jpeg_info(_sof_patched(good, 9460, 9460)) is None          # 89,491,600: outside
    assert Image.MAX_IMAGE_PIXELS == 89478485


def test_sos_segment_at_the_very_end_of_the_buffer():
    """round 4's advice: `FF DA 00 02` as the file's last bytes made the parser read the component count one byte past the buffer"""
    good = open(STILLS[0], "rb").read()
    sos = good.find(b"\xff\xda")
    assert native.jpeg_info(good[:sos] + b"\xff\xda\x00\x02") is None
    assert native.jpeg_info(good[:sos] + b"\xff\xda\x00\x03\x03") is None


def _with_adobe_marker(data: bytes, transform: int, keep_jfif: bool) -> bytes:
    """the same entropy-coded data behind an APP14 'Adobe' segment (and, optionally, without its JFIF APP0)"""
    out, i = bytearray(data[:2]), 2
    adobe = b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00" + bytes([transform])
    out += adobe
    while True:
        m, L = data[i + 1], (data[i + 2] << 8) | data[i + 3]
        if m == 0xDA:
            return bytes(out + data[i:])
        if not (m == 0xE0 and not keep_jfif):
            out += data[i:i + 2 + L]
        i += 2 + L


def test_rgb_stored_files_take_the_host_decoder():
    """round 4's advice: libjpeg (PIL) treats a 3-component file without a JFIF marker as RGB when its Adobe marker says
    transform 0, or when its component ids are 'R','G','B'; the device path converts YCbCr only and must not claim such files"""
    p444 = [p for p in STILLS if "444" in os.path.basename(p)][0]
    ycc = open(p444, "rb").read()
    rgb_stored = _with_adobe_marker(ycc, 0, keep_jfif=False)
    assert not np.array_equal(_pil_rgb(rgb_stored), _pil_rgb(ycc))            # PIL does read it as RGB: other pixels from the same bits
    assert native.jpeg_info(rgb_stored) is None
    with pytest.raises(native.FrpError):
        native.jpeg_coefficients(rgb_stored)
    # transform 1 (YCbCr), and transform 0 next to a JFIF marker (JFIF wins in libjpeg): still ours, and still PIL's pixels
    for variant in (_with_adobe_marker(ycc, 1, keep_jfif=False), _with_adobe_marker(ycc, 0, keep_jfif=True)):
        assert np.array_equal(_pil_rgb(variant), _pil_rgb(ycc))
        info, coef, q = native.jpeg_coefficients(variant)
        assert np.array_equal(oj.decode_from_coefficients(info, coef, q), _pil_rgb(ycc))
    # component ids 'R','G','B' without any marker
    i = ycc.find(b"\xff\xc0")
    ids = bytearray(_with_adobe_marker(ycc, 1, keep_jfif=False).replace(b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x01", b""))
    i = ids.find(b"\xff\xc0")
    old = [ids[i + 10 + 3 * c] for c in range(3)]
    for c, ch in enumerate(b"RGB"):
        ids[i + 10 + 3 * c] = ch
    j = ids.find(b"\xff\xda")
    for c, ch in enumerate(b"RGB"):
        assert ids[j + 5 + 2 * c] == old[c]
        ids[j + 5 + 2 * c] = ch
    assert not np.array_equal(_pil_rgb(bytes(ids)), _pil_rgb(ycc))
    assert native.jpeg_info(bytes(ids)) is None


def test_host_decoder_under_address_and_ub_sanitizers(tmp_path):
    """round 4's advice: csrc/jpeg_host.cpp (untrusted upload bytes) built with g++ -fsanitize=address,undefined and run over a
    corpus of damaged files - every still of the golden set truncated at every byte of its headers and at random points of its scan,
    with random byte flips, with markers inserted - each handed over in a heap block of exactly the file's size.  A report of
    either sanitizer fails the run."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++ here")
    root = os.path.dirname(HERE)
    exe = tmp_path / "jpeg_harness"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", os.path.join(root, "include"),
           "-I", os.path.join(root, "face-recognition-platform_amd", "csrc"), os.path.join(HERE, "native", "jpeg_sanitizer_harness.cpp"),
           os.path.join(root, "face-recognition-platform_amd", "csrc", "jpeg_host.cpp"), "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True)
    rng = np.random.default_rng(3)
    files = []

    def put(data):
        f = tmp_path / f"c{len(files):05d}.jpg"
        f.write_bytes(bytes(data))
        files.append(str(f))
    for path in STILLS:
        good = open(path, "rb").read()
        sos = good.find(b"\xff\xda")
        put(good)
        for cut in list(range(0, min(sos + 16, len(good)))) + [int(c) for c in rng.integers(sos, len(good), 40)]:
            put(good[:cut])
        for _ in range(60):
            bad = bytearray(good)
            for _ in range(int(rng.integers(1, 8))):
                bad[int(rng.integers(2, len(bad)))] = int(rng.integers(0, 256))
            put(bad)
        for _ in range(20):                                   # stray markers / segment lengths inside the headers and the scan
            bad = bytearray(good)
            i = int(rng.integers(2, len(bad) - 4))
            bad[i:i + 4] = bytes([0xFF, int(rng.choice([0xC0, 0xC4, 0xDA, 0xDB, 0xDD, 0xD9, 0xEE, 0xE0])), int(rng.integers(0, 3)), int(rng.integers(0, 256))])
            put(bad)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = ""
    for i in range(0, len(files), 400):
        r = subprocess.run([str(exe)] + files[i:i + 400], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        out += r.stdout
    assert "decoded" in out and len(files) > 1500
