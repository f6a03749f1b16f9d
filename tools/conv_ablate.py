#!/usr/bin/env python3
"""Timing-only ablations of the row-patch conv k-step (tools/conv_bench.py shapes; wrong results by design).
dbg = ablation << 2: 1 no MFMA, 2 no fragment reads, 4 no DMA, 8 no barrier (bits combine)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (selects libfrp_lab.so)
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

SHAPES = [
    ("emb.stage3 256->256 14x14", 320, 14, 14, 256, 256),
    ("emb.stage2 128->128 28x28", 320, 28, 28, 128, 128),
]
VARIANTS = [("old k-step", 2), ("prefetch", 0), ("no MFMA", 1 << 2), ("no reads", 2 << 2), ("no DMA", 4 << 2), ("no barrier", 8 << 2),
            ("no reads+DMA", 6 << 2), ("MFMA only", 14 << 2), ("no MFMA+DMA", 5 << 2), ("reads only", 13 << 2), ("no DMA+barrier", 12 << 2), ("no reads+barrier", 10 << 2)]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    eng = native.Engine(0)
    for name, N, H, W, Cin, Cout in SHAPES:
        best = {v: 1e30 for v, _ in VARIANTS}
        for _ in range(3):
            for v, dbg in VARIANTS:
                best[v] = min(best[v], eng.conv_bench(N, H, W, Cin, Cout, 3, 1, 2, 1 | (dbg << 8), False, iters))
        fl = 2.0 * N * H * W * 9 * Cin * Cout
        print(name)
        for v, _ in VARIANTS:
            print(f"   {v:14s} {best[v]*1e3:8.1f} us  {fl/best[v]/1e9:7.1f} TF-equivalent")


if __name__ == "__main__":
    main()
