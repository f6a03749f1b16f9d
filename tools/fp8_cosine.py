#!/usr/bin/env python3
"""Embedding cosine between the fp8-weight blob (BASELINE config 5) and the fp16 blob of the same seeded weights."""
import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frp_amd_loader
from frp_amd import native, weights as wts
eng = native.Engine(0)
rng = np.random.default_rng(55)
chips = rng.integers(0, 256, size=(32, 112, 112, 3), dtype=np.uint8)
for emb_blocks in [(1,1,1,1),(3,4,14,3),(3,13,30,3)]:
    raw = wts.make_synthetic_raw(17, (1,1,1,1), emb_blocks)
    e={}
    for f in ("fp16","fp8"):
        eng.load_weights(wts.pack_blob(raw,(1,1,1,1),emb_blocks,weight_format=f)); e[f]=eng.embed_aligned(chips)
    cos=(e["fp8"]*e["fp16"]).sum(1)
    print(emb_blocks, "cos min %.5f mean %.5f"%(cos.min(), cos.mean()))
