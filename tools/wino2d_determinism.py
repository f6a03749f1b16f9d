#!/usr/bin/env python3
"""Run-to-run determinism of ONE Winograd 2-D-tile launch (conv3x3_wino2_kernel<32>) on a detector shape: the same operands REPS
times through `Engine.conv2d(..., flags=WINO)`, outputs compared bit for bit with the first launch and with the direct kernel.
Mismatches are localised to (image, 8 x 30 tile, rows of the tile = the wave pair that owns them, cout half).
    python tools/wino2d_determinism.py [REPS=200] [noise=1] [N H W Cin Cout]"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

TILES_DEFAULT, WINO = 0x40000, 0x10000


def localise(first, got):
    idx = np.argwhere(first.view(np.uint16) != got.view(np.uint16))
    n, y, x, c = idx.T
    tiles = sorted(set(zip(n.tolist(), (y // 8).tolist(), (x // 30).tolist())))
    d = np.abs(first.astype(np.float32) - got.astype(np.float32)).max()
    out = [f"{len(idx)} elements in {len(tiles)} tile(s), max |diff| {d:.4f}"]
    for (tn, ty, tx) in tiles[:6]:
        m = (n == tn) & (y // 8 == ty) & (x // 30 == tx)
        yy, xx, cc = y[m] - 8 * ty, x[m] - 30 * tx, c[m]
        out.append(f"    tile (image {tn}, ty {ty}, tx {tx}): {m.sum()} elements, tile rows {sorted(set(yy.tolist()))}, cols {xx.min()}..{xx.max()}, "
                   f"couts {cc.min()}..{cc.max()} ({len(set(cc.tolist()))} distinct)")
    return "\n".join(out)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    noise_on = (int(sys.argv[2]) if len(sys.argv) > 2 else 1) != 0
    N, H, W, Cin, Cout = [int(a) for a in sys.argv[3:8]] if len(sys.argv) > 7 else (4, 136, 240, 128, 128)
    rng = np.random.default_rng(5)
    x = np.maximum(rng.standard_normal((N, H, W, Cin)) * 2.5, 0).astype(np.float16)          # ReLU-sparse, scale ~ 7 as in the pipeline
    w = (rng.standard_normal((Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float16)
    bias = (rng.standard_normal(Cout) * 0.3).astype(np.float32)
    eng = native.Engine(0)
    other = native.Engine(0)
    stop = threading.Event()
    xo = rng.standard_normal((9, 56, 56, 64)).astype(np.float16)
    wo = (rng.standard_normal((64, 3, 3, 64)) / 24).astype(np.float16)
    bo = np.zeros(64, np.float32)

    def noise():
        while not stop.is_set():
            other.conv2d(xo, wo, bo, flags=TILES_DEFAULT)
    th = threading.Thread(target=noise, daemon=True)
    if noise_on:
        th.start()
    bad = 0
    try:
        direct = eng.conv2d(x, w, bias, act=1, flags=TILES_DEFAULT)
        first = eng.conv2d(x, w, bias, act=1, flags=WINO)
        scale = max(1.0, float(np.abs(direct.astype(np.float32)).max()))
        print(f"shape {(N, H, W, Cin, Cout)}: first launch vs direct: {np.abs(first.astype(np.float32) - direct.astype(np.float32)).max() / (2.0 ** -10 * scale):.2f} fp16 ulps of the scale {scale:.2f}", flush=True)
        for r in range(reps):
            got = eng.conv2d(x, w, bias, act=1, flags=WINO)
            if not np.array_equal(first.view(np.uint16), got.view(np.uint16)):
                bad += 1
                print(f"launch {r}: " + localise(first, got), flush=True)
        print(f"{reps} launches, {bad} differed from the first", flush=True)
    finally:
        stop.set()
        if noise_on:
            th.join()
        other.close()
        eng.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
