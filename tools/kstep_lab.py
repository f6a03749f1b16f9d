#!/usr/bin/env python3
"""k-step schedule lab (csrc/kstep_lab.hip): TFLOP/s per variant, 3 interleaved rounds, best of each.
bits: 1 barrier per k-step, 2 pinned read-before-MFMA order, 4 cross-barrier prefetch, 8 stagger waves 4..7,
16 setprio for waves 4..7, 32 three fragment sets."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (selects libfrp_lab.so)
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

VARIANTS = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 17, 19, 21, 23, 25, 27, 32, 33, 34, 35, 49, 51]
NAMES = {1: "bar", 2: "pin", 4: "pf", 8: "stag", 16: "prio", 32: "tri"}


def main():
    eng = native.Engine(0)
    best = {v: 0.0 for v in VARIANTS}
    for _ in range(3):
        for v in VARIANTS:
            best[v] = max(best[v], eng.kstep_lab(v, 3000 // (3 if (v & 32 and not v & 512) else 1)))
    for v in VARIANTS:
        label = "+".join(n for b, n in NAMES.items() if v & b & 63) or "free"
        if (v >> 6) & 7:
            label += f"+dma{(v >> 6) & 7}"
        if v & 512:
            label += " [16x16x32]"
        if v & 1024:
            label += " [fp8 32x32x64 scaled]"
        print(f"variant {v:3d} {label:24s} {best[v]:8.1f} TF")


if __name__ == "__main__":
    main()
