"""Encrypted-watchlist loader: AES / Fernet known answers + the db.py format."""
import base64
import json

import numpy as np
import pytest

from fake_engine import FakeEngine
from frp_amd import watchlist as wl
from frp_amd.face_service import FaceService


def test_aes128_fips197_known_answer():
    key = bytes(range(16))
    pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    ct = wl.aes128_encrypt_blocks(key, np.frombuffer(pt, np.uint8))[0].tobytes()
    assert ct.hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"          # FIPS-197 appendix C.1
    assert wl.aes128_decrypt_blocks(key, np.frombuffer(ct, np.uint8))[0].tobytes() == pt
    # second published vector (FIPS-197 appendix B)
    k2 = bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c")
    p2 = bytes.fromhex("3243f6a8885a308d313198a2e0370734")
    assert wl.aes128_encrypt_blocks(k2, np.frombuffer(p2, np.uint8))[0].tobytes().hex() == "3925841d02dc09fbdc118597196a0b32"


def test_fernet_spec_vector_and_roundtrip():
    # published Fernet spec vector (fernet/spec generate.json / verify.json)
    key = "cw_0x689RpI-jtRR7oE8h_eQsKImvJapLeSbXpwF4e4="
    token = "gAAAAAAdwJ6wAAECAwQFBgcICQoLDA0ODy021cpGVWKZ_eEwCGM4BLLF_5CV9dOPmrhuVUPgJobwOz7JcbmrR64jVmpU4IwqDA=="
    f = wl.Fernet(key)
    assert f.decrypt(token) == b"hello"                              # HMAC verifies and CBC/PKCS7 decrypt
    assert f.encrypt(b"hello", now=499162800, iv=bytes(range(16))) == token.encode()
    with pytest.raises(wl.InvalidToken):
        f.decrypt(token[:-6] + "AAAA==")
    with pytest.raises(wl.InvalidToken):
        f.decrypt(token, ttl=60, now=499162800 + 61)
    g = wl.Fernet(wl.Fernet.generate_key())
    msg = bytes(range(256)) * 37 + b"tail"
    assert g.decrypt(g.encrypt(msg)) == msg
    with pytest.raises(wl.InvalidToken):
        f.decrypt(g.encrypt(b"x"))                                   # wrong key


def test_db_format_and_startup_load(tmp_path):
    rng = np.random.default_rng(0)
    f = wl.Fernet(wl.Fernet.generate_key())
    E = rng.standard_normal((7, 512))
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    records = [{"target": f"id{i}", "embedding": wl.encrypt_embedding(E[i].tolist(), f)} for i in range(7)]
    # stored string is base64 of the (already base64) Fernet token, as db.py:248-249 writes it
    inner = base64.b64decode(records[0]["embedding"])
    assert inner[:5] == b"gAAAA" and f.decrypt(inner) == json.dumps(E[0].tolist()).encode()
    assert wl.decrypt_embedding(records[3]["embedding"], f) == E[3].tolist()
    assert wl.decrypt_embedding("garbage", f) == [] and wl.decrypt_embedding(records[0]["embedding"], wl.Fernet(wl.Fernet.generate_key())) == []
    assert wl.decrypt_embedding(wl.encrypt_embedding([1.0, 2.0], None), None) == [1.0, 2.0]     # encryption disabled
    records += [{"target": "broken", "embedding": "@@@"}, {"target": "id2", "embedding": records[2]["embedding"]},
                {"target": "short", "embedding": wl.encrypt_embedding([0.1] * 128, f)}]
    fs = FaceService(engine=FakeEngine())
    assert wl.install_watchlist(fs, records, f) == {"loaded": 7, "skipped": 3}
    assert fs.get_all_targets() == [f"id{i}" for i in range(7)]
    top = fs.find_k_nearest(E[4], 1)[0]
    assert top["target"] == "id4" and top["distance"] < 1e-6
    # shard loader = what each rank decrypts before the all-gather
    mk = wl.shard_loader(records[:7], f)
    assert np.allclose(mk(2, 3), E[2:5].astype(np.float32))
    with pytest.raises(ValueError):
        wl.shard_loader(records, f)(6, 3)
    # JSON backups written by store_face
    import frp_amd.face_service as fsmod
    old = fsmod.BACKUP_DIR
    fsmod.BACKUP_DIR = tmp_path
    try:
        fs2 = FaceService(engine=FakeEngine())
        for i in range(3):
            fs2.store_face(f"p{i}", E[i])
        names, mat = wl.load_backup_dir(tmp_path)
        assert names == ["p0", "p1", "p2"] and np.allclose(mat, E[:3], atol=1e-6)
        d = json.loads((tmp_path / "p1_backup.json").read_text())
        assert list(d.keys()) == ["target", "encoding", "timestamp", "version"] and d["version"] == 1
    finally:
        fsmod.BACKUP_DIR = old
