#!/usr/bin/env python3
"""Per-shape timing of the MFMA conv kernel on the layer shapes of the headline workload
(32 x 1080p frames, 320 faces).  Usage on the GPU box: python tools/conv_bench.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (selects libfrp_lab.so)
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

SHAPES = [
    # name, N, H, W, Cin, Cout, k, stride, act, flags, res, launches per step
    ("emb.stage3 256->256 14x14", 320, 14, 14, 256, 256, 3, 1, 2, 1, False, 30),
    ("emb.stage3 conv2 +res",     320, 14, 14, 256, 256, 3, 1, 0, 0, True, 29),
    ("emb.stage2 128->128 28x28", 320, 28, 28, 128, 128, 3, 1, 2, 1, False, 25),
    ("emb.stage1 64->64 56x56",   320, 56, 56, 64, 64, 3, 1, 2, 1, False, 5),
    ("emb.stage4 512->512 7x7",   320, 7, 7, 512, 512, 3, 1, 2, 1, False, 5),
    ("emb.l1.0 conv1 112x112",    320, 112, 112, 64, 64, 3, 1, 2, 1, False, 1),
    ("emb.stem 8->64 112x112",    320, 112, 112, 8, 64, 3, 1, 2, 0, False, 1),
    ("emb.fc 25088->512",         320, 1, 1, 25088, 512, 1, 1, 0, 2, False, 1),
    ("det.layer1 64->64 272x480", 32, 272, 480, 64, 64, 3, 1, 1, 0, False, 2),
    ("det 128->128 136x240",      32, 136, 240, 128, 128, 3, 1, 1, 0, False, 9),
    ("det 256->256 68x120",       32, 68, 120, 256, 256, 3, 1, 1, 0, False, 4),
    ("det.stem1 8->32 s2 1088x1920", 32, 1088, 1920, 8, 32, 3, 2, 1, 0, False, 1),
    ("det.stem2 32->64 s2 544x960", 32, 544, 960, 32, 64, 3, 2, 1, 0, False, 1),
    ("det.head out 128->32",      32, 136, 240, 128, 32, 3, 1, 0, 0, False, 1),
    ("det 128->128 +res 136x240",  32, 136, 240, 128, 128, 3, 1, 1, 0, True, 1),
    ("det 64->64 +res 272x480",    32, 272, 480, 64, 64, 3, 1, 1, 0, True, 1),
    ("det 256->256 +res 68x120",   32, 68, 120, 256, 256, 3, 1, 1, 0, True, 1),
    # generic-kernel shapes (stride 2, 1x1)
    ("det 64->128 s2 272x480",    32, 272, 480, 64, 128, 3, 2, 1, 0, False, 1),
    ("det 128->256 s2 136x240",   32, 136, 240, 128, 256, 3, 2, 1, 0, False, 1),
    ("emb 64->64 s2 112x112",     320, 112, 112, 64, 64, 3, 2, 0, 0, True, 1),
    ("emb 128->128 s2 56x56",     320, 56, 56, 128, 128, 3, 2, 0, 0, True, 1),
    ("emb 256->256 s2 28x28",     320, 28, 28, 256, 256, 3, 2, 0, 0, True, 1),
    ("det lat3 1x1 128->128",     32, 136, 240, 128, 128, 1, 1, 0, 0, False, 1),
    ("det down 1x1 s2 64->128",   32, 272, 480, 64, 128, 1, 2, 0, 0, False, 1),
]


def main():
    """conv_bench.py [iters] [dbgA] [dbgB]: with dbgB given the two variants alternate in this one
    process (3 rounds, best of each) so that box-to-box and run-to-run drift cancels."""
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dbgs = [int(a) for a in sys.argv[2:4]] or [0]
    eng = native.Engine(0)
    tot = [0.0] * len(dbgs)
    for (name, N, H, W, Cin, Cout, k, s, act, flags, res, cnt) in SHAPES:
        best = [1e30] * len(dbgs)
        for _ in range(3 if len(dbgs) > 1 else 1):
            for v, dbg in enumerate(dbgs):
                best[v] = min(best[v], eng.conv_bench(N, H, W, Cin, Cout, k, s, act, flags | (dbg << 8), res, iters))
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        cin_r = 3 if Cin == 8 else Cin
        fl = 2.0 * N * Ho * Wo * k * k * cin_r * Cout
        cols = "  ".join(f"{ms*1e3:9.1f} us {fl/ms/1e9:7.1f} TF" for ms in best)
        ratio = f"  B/A time {best[1]/best[0]:.3f}" if len(dbgs) > 1 else ""
        print(f"{name:34s} {cols}   x{cnt:2d} = {best[0]*cnt:7.3f} ms{ratio}")
        for v in range(len(dbgs)):
            tot[v] += best[v] * cnt
    print("weighted total " + "  ".join(f"{x:.3f} ms" for x in tot))


if __name__ == "__main__":
    main()
