#!/usr/bin/env python3
"""Timing-only ablations of the Winograd conv k-loop (lab build; wrong results by design) on the stage-3 shape:
dbg code c -> flags bits (c << 1) << 8.  1 no MFMA, 2 no fragment reads, 4 no LDS-DMA, 8 no barrier, 9 no B^T d arithmetic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

NAMES = {0: "full kernel", 1: "no MFMA", 2: "no fragment reads", 4: "no LDS-DMA", 8: "no barrier", 3: "no MFMA, no reads", 6: "no reads, no DMA",
         5: "no MFMA, no DMA", 7: "only barriers + bookkeeping + transform", 12: "no DMA, no barrier", 9: "no B^T d arithmetic", 15: "bookkeeping only"}


def main():
    eng = native.Engine(0)
    shapes = [("stage 3 (320 x 14x14, 256->256)", 320, 14, 14, 256, 256), ("stage 2 (320 x 28x28, 128->128)", 320, 28, 28, 128, 128)]
    for name, N, H, W, Ci, Co in shapes:
        best = {c: 1e30 for c in NAMES}
        for _ in range(3):
            for c in NAMES:
                best[c] = min(best[c], eng.conv_bench(N, H, W, Ci, Co, 3, 1, 2, 1 | 0x10000 | ((c << 1) << 8), False, 20) * 1e3)
        print(name)
        for c, n in NAMES.items():
            print(f"   {n:44s} {best[c]:7.1f} us")


if __name__ == "__main__":
    main()
