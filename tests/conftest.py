import os
import sys

import pytest

# FaceService writes per-identity JSON backups (face_service.py:731-741 of the reference): keep the test
# runs out of the working tree
import tempfile
os.environ.setdefault("FACE_BACKUP_DIR", tempfile.mkdtemp(prefix="frp_backups_"))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import frp_amd_loader  # noqa: E402,F401  (registers package `frp_amd`)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    """One handle for the whole GPU session.  Fails loudly when libfrp.so or the GPU is missing."""
    from frp_amd import native
    eng = native.Engine(0, profile=False)
    yield eng
    eng.close()


_BLOBS = {}


def get_raw_and_blob(det_blocks, emb_blocks, seed=7):
    from frp_amd import weights
    key = (tuple(det_blocks), tuple(emb_blocks), seed)
    if key not in _BLOBS:
        raw = weights.make_synthetic_raw(seed, det_blocks, emb_blocks)
        _BLOBS[key] = (raw, weights.pack_blob(raw, det_blocks, emb_blocks))
    return _BLOBS[key]
