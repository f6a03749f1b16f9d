"""Encrypted-watchlist loader: the host-side source of the gallery matrix.

Restates the storage format of backend/app/utils/db.py without its dependencies:
  * `encrypt_embedding` (:238-251): Fernet(json.dumps(list).encode()) -> base64.b64encode(token)
    (the Fernet token is itself url-safe base64, so the stored string is base64 twice), or the
    plain JSON string when encryption is disabled;
  * `decrypt_embedding` (:254-267): the inverse, `[]` on any error;
  * `retrieve_all_embeddings` (:484-490): records {"target", "embedding"} -- which the reference
    never calls at start-up (main.py:186 only prints len(ENCODINGS)); `load_records` is that
    missing loader;
  * per-identity JSON backups written by FaceService._backup_encoding_atomic
    (backend/app/services/face_service.py:731-741): {"target","encoding","timestamp","version"}.
Fernet (spec: version 0x80 | 8-byte timestamp | 16-byte IV | AES-128-CBC(PKCS7) | HMAC-SHA256,
key = 16-byte signing key + 16-byte encryption key, url-safe base64) is implemented here on
hashlib/hmac + a table-driven AES vectorised over blocks with numpy (CBC decryption is
block-parallel), because the `cryptography` package of the reference is not available offline.
The decrypted rows feed `Gallery.set_bulk` on one GPU, or -- one shard per rank --
`dist.allgather_gallery_into_engine` (the RCCL all-gather of the watch-list matrix).
"""
from __future__ import annotations

import base64
import hashlib
import hmac
import json
import os
import struct
import time
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

# ------------------------------------------------------------------ AES-128 (FIPS-197), numpy over blocks


def _build_tables():
    # GF(2^8) log/antilog with generator 3 -> multiplicative inverse -> S-box
    exp = np.zeros(512, dtype=np.int64)
    log = np.zeros(256, dtype=np.int64)
    x = 1
    for i in range(255):
        exp[i] = x
        log[x] = i
        x ^= (x << 1) ^ (0x11B if x & 0x80 else 0)
        x &= 0xFF
    exp[255:510] = exp[0:255]
    sbox = np.zeros(256, dtype=np.uint8)
    for v in range(256):
        inv = 0 if v == 0 else int(exp[255 - log[v]])
        s = inv
        for sh in (1, 2, 3, 4):
            s ^= ((inv << sh) | (inv >> (8 - sh))) & 0xFF
        sbox[v] = s ^ 0x63
    inv_sbox = np.zeros(256, dtype=np.uint8)
    inv_sbox[sbox] = np.arange(256, dtype=np.uint8)

    def mul(c):
        t = np.zeros(256, dtype=np.uint8)
        for v in range(1, 256):
            t[v] = exp[log[v] + log[c]]
        return t

    return sbox, inv_sbox, {c: mul(c) for c in (2, 3, 9, 11, 13, 14)}


_SBOX, _INV_SBOX, _MUL = _build_tables()
# state byte i = column i//4, row i%4.  ShiftRows: row r rotates left by r columns.
_SHIFT = np.array([(4 * ((c + r) % 4) + r) for c in range(4) for r in range(4)])
_INV_SHIFT = np.array([(4 * ((c - r) % 4) + r) for c in range(4) for r in range(4)])


def _expand_key(key: bytes) -> np.ndarray:
    assert len(key) == 16
    w = [list(key[4 * i:4 * i + 4]) for i in range(4)]
    rcon = 1
    for i in range(4, 44):
        t = list(w[i - 1])
        if i % 4 == 0:
            t = t[1:] + t[:1]
            t = [int(_SBOX[b]) for b in t]
            t[0] ^= rcon
            rcon = ((rcon << 1) ^ (0x11B if rcon & 0x80 else 0)) & 0xFF
        w.append([a ^ b for a, b in zip(w[i - 4], t)])
    return np.array(w, dtype=np.uint8).reshape(11, 16)


def _mix(state: np.ndarray, coef: Tuple[int, int, int, int]) -> np.ndarray:
    s = state.reshape(-1, 4, 4)          # [block, column, row]
    out = np.zeros_like(s)
    for r in range(4):
        acc = np.zeros(s.shape[:2], dtype=np.uint8)
        for k in range(4):
            c = coef[(k - r) % 4]
            v = s[:, :, k]
            acc ^= v if c == 1 else _MUL[c][v]
        out[:, :, r] = acc
    return out.reshape(-1, 16)


def aes128_encrypt_blocks(key: bytes, blocks: np.ndarray) -> np.ndarray:
    rk = _expand_key(key)
    s = blocks.reshape(-1, 16).astype(np.uint8) ^ rk[0]
    for rnd in range(1, 10):
        s = _mix(_SBOX[s][:, _SHIFT], (2, 3, 1, 1)) ^ rk[rnd]
    return _SBOX[s][:, _SHIFT] ^ rk[10]


def aes128_decrypt_blocks(key: bytes, blocks: np.ndarray) -> np.ndarray:
    rk = _expand_key(key)
    s = blocks.reshape(-1, 16).astype(np.uint8) ^ rk[10]
    for rnd in range(9, 0, -1):
        s = _INV_SBOX[s[:, _INV_SHIFT]] ^ rk[rnd]
        s = _mix(s, (14, 11, 13, 9))
    return _INV_SBOX[s[:, _INV_SHIFT]] ^ rk[0]


# ------------------------------------------------------------------ Fernet
class InvalidToken(Exception):
    pass


class Fernet:
    def __init__(self, key):
        raw = base64.urlsafe_b64decode(key if isinstance(key, bytes) else key.encode())
        if len(raw) != 32:
            raise ValueError("Fernet key must be 32 url-safe base64-encoded bytes")
        self._sign, self._enc = raw[:16], raw[16:]

    @staticmethod
    def generate_key() -> bytes:
        return base64.urlsafe_b64encode(os.urandom(32))

    def encrypt(self, data: bytes, now: Optional[int] = None, iv: Optional[bytes] = None) -> bytes:
        iv = os.urandom(16) if iv is None else iv
        pad = 16 - len(data) % 16
        plain = np.frombuffer(data + bytes([pad]) * pad, dtype=np.uint8).reshape(-1, 16)
        prev = np.frombuffer(iv, dtype=np.uint8)
        out = np.empty_like(plain)
        for i in range(len(plain)):                     # CBC encryption is sequential
            prev = aes128_encrypt_blocks(self._enc, (plain[i] ^ prev)[None])[0]
            out[i] = prev
        body = b"\x80" + struct.pack(">Q", int(time.time()) if now is None else now) + iv + out.tobytes()
        return base64.urlsafe_b64encode(body + hmac.new(self._sign, body, hashlib.sha256).digest())

    def decrypt(self, token, ttl: Optional[int] = None, now: Optional[int] = None) -> bytes:
        try:
            data = base64.urlsafe_b64decode(token if isinstance(token, bytes) else token.encode())
        except Exception as e:
            raise InvalidToken("not base64") from e
        if len(data) < 1 + 8 + 16 + 16 + 32 or data[0] != 0x80 or (len(data) - 57) % 16:
            raise InvalidToken("malformed token")
        body, mac = data[:-32], data[-32:]
        if not hmac.compare_digest(hmac.new(self._sign, body, hashlib.sha256).digest(), mac):
            raise InvalidToken("bad signature")
        if ttl is not None:
            ts = struct.unpack(">Q", data[1:9])[0]
            if ts + ttl < (int(time.time()) if now is None else now):
                raise InvalidToken("expired")
        iv, ct = data[9:25], np.frombuffer(data[25:-32], dtype=np.uint8).reshape(-1, 16)
        prev = np.concatenate([np.frombuffer(iv, dtype=np.uint8)[None], ct[:-1]])
        plain = (aes128_decrypt_blocks(self._enc, ct) ^ prev).tobytes()    # block-parallel
        pad = plain[-1]
        if not 1 <= pad <= 16 or plain[-pad:] != bytes([pad]) * pad:
            raise InvalidToken("bad padding")
        return plain[:-pad]


# ------------------------------------------------------------------ db.py format
def encrypt_embedding(embedding: Sequence[float], fernet: Optional[Fernet]) -> str:
    """db.py:238-251."""
    if fernet is None:
        return json.dumps(list(embedding))
    return base64.b64encode(fernet.encrypt(json.dumps(list(embedding)).encode("utf-8"))).decode("utf-8")


def decrypt_embedding(stored: str, fernet: Optional[Fernet]) -> List[float]:
    """db.py:254-267: `[]` on any error."""
    try:
        if fernet is None:
            return json.loads(stored)
        return json.loads(fernet.decrypt(base64.b64decode(stored)).decode("utf-8"))
    except Exception:
        return []


def load_records(records: Iterable[Dict], fernet: Optional[Fernet], dim: int = 512) -> Tuple[List[str], np.ndarray, List[str]]:
    """records as `retrieve_all_embeddings` returns them ({"target", "embedding"}) ->
    (names, float32 [n, dim] matrix, skipped targets).  Undecryptable / wrong-width rows are skipped
    (the reference's decrypt returns [] for them)."""
    names, rows, skipped = [], [], []
    seen = set()
    for rec in records:
        t = rec.get("target")
        vec = decrypt_embedding(rec.get("embedding", ""), fernet)
        if t is None or t in seen or len(vec) != dim:
            skipped.append(t)
            continue
        seen.add(t)
        names.append(t)
        rows.append(np.asarray(vec, dtype=np.float32))
    mat = np.stack(rows) if rows else np.zeros((0, dim), np.float32)
    return names, mat, skipped


def load_backup_dir(path, dim: int = 512) -> Tuple[List[str], np.ndarray]:
    """per-identity JSON backups (face_service.py:731-741), sorted by file name."""
    names, rows = [], []
    for f in sorted(Path(path).glob("*_backup.json")):
        try:
            d = json.loads(f.read_text(encoding="utf-8"))
            if len(d["encoding"]) == dim:
                names.append(d["target"])
                rows.append(np.asarray(d["encoding"], dtype=np.float32))
        except Exception:
            continue
    return names, (np.stack(rows) if rows else np.zeros((0, dim), np.float32))


def install_watchlist(service, records: Iterable[Dict], fernet: Optional[Fernet]) -> Dict[str, int]:
    """The start-up load the reference lacks: decrypt every stored identity and install the
    matrix in the device gallery of `service` (a FaceService) in one upload."""
    names, mat, skipped = load_records(records, fernet)
    service.ENCODINGS.set_bulk(names, mat)
    return {"loaded": len(names), "skipped": len(skipped)}


def shard_loader(records: Sequence[Dict], fernet: Optional[Fernet], dim: int = 512):
    """`make_rows(first, count)` for dist.allgather_gallery_into_engine: rank r decrypts only its
    own rows of the (already ordered, validated) record list."""
    def make_rows(first: int, count: int) -> np.ndarray:
        _, mat, skipped = load_records(records[first:first + count], fernet, dim)
        if skipped:
            raise ValueError(f"undecryptable watch-list rows in shard [{first}, {first + count}): {skipped[:3]}")
        return mat
    return make_rows
