#!/usr/bin/env python3
"""Regression pins for the NETWORK oracle (oracle/network.py).  The reference holds no golden
vector for these stages (parity unpinned, see the oracle header); this script freezes the
oracle's own outputs on seeded inputs so that later edits to the oracle cannot drift silently.
Run from the repo root: python tests/golden/make_network_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import frp_amd_loader  # noqa: E402,F401
from frp_amd import weights  # noqa: E402
from oracle import network as onet  # noqa: E402


def build():
    out = {}
    rng = np.random.default_rng(2024)
    # decode + NMS on a seeded 64x96 head set
    heads = []
    for s in (8, 16, 32):
        h = rng.standard_normal((64 // s, 96 // s, 32)).astype(np.float32)
        h[..., [0, 15]] = h[..., [0, 15]] * 3 - 2
        h[..., 1:5] = np.abs(h[..., 1:5]) * 2 + 0.5
        h[..., 16:20] = np.abs(h[..., 16:20]) * 2 + 0.5
        heads.append(h.astype(np.float16))
        out[f"head{s}"] = heads[-1]
    b, k, sc, a = onet.decode_nms(heads, 0.5, 0.4, 6)
    out.update(dec_boxes=b, dec_kps=k, dec_scores=sc, dec_anchor=a)
    # similarity + warp
    src = onet.ARCFACE_TEMPLATE * 1.7 + [30, 12] + rng.standard_normal((5, 2)).astype(np.float32)
    M = onet.umeyama_similarity(src, onet.ARCFACE_TEMPLATE)
    img = rng.integers(0, 256, size=(160, 200, 3), dtype=np.uint8)
    chip = onet.warp_affine_bilinear(img, M)
    out.update(align_src=src.astype(np.float32), align_M=M, align_img=img, align_chip_sub=chip[::8, ::8])
    # tiny embedder / detector on seeded weights (weights regenerated from the seed, not stored)
    raw = weights.make_synthetic_raw(7, (1, 1, 1, 1), (1, 1, 1, 1))
    chips = rng.integers(0, 256, size=(2, 112, 112, 3), dtype=np.uint8)
    out.update(emb_chips_seed=np.array([2024]), emb_out=onet.emb_forward(raw, onet.emb_blob(chips)))
    out["emb_chips"] = chips[:, ::4, ::4]          # subsampled copy only as a checksum of the generator
    frames = rng.integers(0, 256, size=(1, 64, 96, 3), dtype=np.uint8)
    maps = onet.det_forward(raw, onet.det_blob(frames, (64, 96)))
    out.update(det_frames=frames, det_map8=maps[0], det_map32=maps[2])
    return out, chips


if __name__ == "__main__":
    out, _ = build()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "network_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
