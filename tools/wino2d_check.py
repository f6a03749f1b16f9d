#!/usr/bin/env python3
"""The Winograd kernel's 2-D tiles (8 x 30 output pixels; lab build, flags bit 19) against the direct kernel - and, where the map is
narrow enough for both, bit for bit against the flattened tiles of the shipped form (same products, same accumulation order).
    python tools/wino2d_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (selects libfrp_lab.so)
import numpy as np  # noqa: E402
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

TILES_DEFAULT, WINO, T2D = 0x40000, 0x10000, 0x80000
CASES = [  # N, H, W, Cin, Cout, act, res, border
    (1, 8, 30, 64, 128, 0, False, 0),        # exactly one tile
    (1, 14, 14, 256, 256, 2, False, 1),      # narrow map: also through the flattened form
    (5, 14, 14, 256, 256, 0, True, 0),
    (2, 28, 28, 128, 128, 2, True, 1),
    (2, 20, 36, 64, 64, 1, False, 0),        # two column tiles, the second 6 wide
    (3, 17, 62, 128, 136, 2, True, 1),       # ragged rows, columns and couts
    (1, 9, 240, 64, 64, 0, True, 0),
    (2, 136, 240, 128, 128, 1, True, 1),     # det.layer2 shape
    (2, 68, 120, 256, 256, 1, False, 0),     # det.layer3 shape
    (1, 34, 60, 256, 256, 1, True, 0),
]


def main():
    eng = native.Engine(0)
    bad = 0
    for case in CASES:
        N, H, W, Cin, Cout, act, has_res, border = case
        rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
        x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
        w = (rng.standard_normal((Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float16)
        bias = rng.standard_normal((9, Cout) if border else (Cout,)).astype(np.float32) * 0.3
        slope = rng.uniform(0.1, 0.4, Cout).astype(np.float32) if act == 2 else None
        res = rng.standard_normal((N, H, W, Cout)).astype(np.float16) if has_res else None
        direct = eng.conv2d(x, w, bias, act=act, slope=slope, res=res, flags=border | TILES_DEFAULT).astype(np.float32)
        t2d = eng.conv2d(x, w, bias, act=act, slope=slope, res=res, flags=border | WINO | T2D)
        scale = max(1.0, float(np.abs(direct).max()))
        d = np.abs(t2d.astype(np.float32) - direct)
        line = f"{case}: max |2-D - direct| = {d.max() / (2.0 ** -10 * scale):.2f} fp16 ulps of the scale"
        ok = d.max() <= 3 * 2.0 ** -10 * scale
        if W <= 30:
            flat = eng.conv2d(x, w, bias, act=act, slope=slope, res=res, flags=border | WINO)
            same = np.array_equal(flat.view(np.uint16), t2d.view(np.uint16))
            line += f"; bit-identical to the flattened tiles: {same}"
            ok = ok and same
        if not ok:
            bad += 1
            w_ = np.argwhere(d > 3 * 2.0 ** -10 * scale)
            line += f"  <-- FAIL ({len(w_)} elements; first {w_[:4].tolist()})"
        print(line, flush=True)
    print("all good" if not bad else f"{bad} case(s) differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
