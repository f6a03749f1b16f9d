#!/usr/bin/env python3
"""64-cout 3x3 layers: 512 x 64 tile (two-slot patch ring) vs the 256 x 64 tile (dbg bit 128), per-launch time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (selects libfrp_lab.so)
import frp_amd_loader  # noqa
from frp_amd import native
eng = native.Engine(0)
for name, (N, H, W, Ci, Co, res) in {"det.layer1 64->64 272x480": (32, 272, 480, 64, 64, False), "det.layer1 +res": (32, 272, 480, 64, 64, True),
                                      "emb.stage1 64->64 56x56": (320, 56, 56, 64, 64, False), "emb.stage1 +res": (320, 56, 56, 64, 64, True),
                                      "emb.l1.0 conv1 112x112": (320, 112, 112, 64, 64, False),
                                      "det.head3 out 128->32": (32, 136, 240, 128, 32, False), "det.head4 out": (32, 68, 120, 128, 32, False)}.items():
    a = eng.conv_bench(N, H, W, Ci, Co, 3, 1, 1, 0, res, 20) * 1e3
    b = eng.conv_bench(N, H, W, Ci, Co, 3, 1, 1, 0 | (128 << 8), res, 20) * 1e3
    fl = 2.0 * N * H * W * 9 * Ci * Co
    print(f"{name:28s} 512x64 {a:7.1f} us {fl / a / 1e6:7.1f} TF | 256x64 {b:7.1f} us {fl / b / 1e6:7.1f} TF | x{b / a:.3f}")
