#!/usr/bin/env python3
"""Build an frp weight blob whose embedder comes from a user-supplied ArcFace IResNet `.onnx` pack.

    python tools/onnx_to_blob.py arcface_r100.onnx frp_r100.blob [--det-onnx frpdet.onnx | --det-npz det.npz | --det-seed 7]

The detector of this repo is its own architecture (netspec.detector_layers; a public SCRFD pack is not loadable): its weights
come from an FRPDet `.onnx` file (`--det-onnx`, the interchange format onnx_pack.detector_to_onnx writes), from an .npz with
`det.*` arrays in `weights.make_synthetic_raw` naming (`--det-npz`), or are the seeded synthetic ones.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frp_amd_loader  # noqa: E402,F401
from frp_amd import onnx_pack, weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("onnx")
    ap.add_argument("out")
    ap.add_argument("--det-seed", type=int, default=7)
    ap.add_argument("--det-npz")
    ap.add_argument("--det-onnx")
    a = ap.parse_args()
    emb = onnx_pack.raw_from_onnx(a.onnx)
    blocks = weights.emb_blocks_of(emb)
    if a.det_onnx:
        raw = onnx_pack.det_raw_from_onnx(a.det_onnx)
    else:
        raw = weights.make_synthetic_raw(a.det_seed, want_emb=False)
        if a.det_npz:
            raw.update({k: v for k, v in np.load(a.det_npz).items() if k.startswith("det.")})
    det_blocks = onnx_pack.det_blocks_of(raw)
    raw.update(emb)
    blob = weights.pack_blob(raw, det_blocks=det_blocks, emb_blocks=blocks)
    with open(a.out, "wb") as f:
        f.write(blob)
    print(f"embedder IResNet stages {blocks}: {sum(v.size for v in emb.values()) / 1e6:.1f} M parameters -> {a.out} ({len(blob) / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
