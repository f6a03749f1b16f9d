#!/usr/bin/env python3
"""Mix lab (csrc/kstep_lab.hip: mix_lab_kernel): MFMA rate of candidate k-loop instruction mixes at one wave per SIMD
(256 accumulator registers per lane), next to the shipped k-step's skeleton (8 waves, 64 x 64 wave tiles).
TF = rate of the v_mfma_f32_32x32x16_f16 stream itself; "eff" = that rate x the useful-work factor of the scheme
(Winograd F(2,3) along W: 1.5, F(2x2,3x3): 2.25) = what the direct kernel would have to sustain to match it.
    python tools/mix_lab.py            (needs the FRP_LAB build of the library)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (selects libfrp_lab.so)
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

MODES = {0: ("direct 128x128 wave tile, 0.5 reads/MFMA", 1.0), 1: ("Winograd F(2,3) along W, 1 read/MFMA + 2 pk_add/MFMA", 1.5),
         2: ("Winograd F(2x2,3x3), 2 reads/MFMA + 8 pk_add/MFMA", 2.25), 3: ("as 2 without the transform arithmetic", 2.25)}
MIX = [(0, 0, 0), (0, 2, 0), (0, 2, 8), (0, 2, 12), (1, 0, 0), (1, 2, 0), (1, 2, 8), (1, 2, 12),
       (2, 0, 0), (2, 1, 0), (2, 1, 4), (2, 1, 9), (3, 1, 0), (3, 1, 4), (3, 1, 9)]
REF = [0, 1, 257, 263]


def main():
    eng = native.Engine(0)
    rows = []
    for _ in range(3):
        k = 0
        for v in REF:
            t = eng.kstep_lab(v, 3000)
            if len(rows) <= k:
                rows.append([f"shipped skeleton variant {v}", 1.0, 0.0])
            rows[k][2] = max(rows[k][2], t)
            k += 1
        for m, b, p in MIX:
            v = 2048 | m | (b << 4) | (p << 6)
            t = eng.kstep_lab(v, 2000)
            if len(rows) <= k:
                bar = {0: "no barrier", 1: "barrier per sub-step", 2: "barrier per 4 sub-steps"}[b]
                rows.append([f"mode {m} ({MODES[m][0]}), {bar}, {p} DMA pieces", MODES[m][1], 0.0])
            rows[k][2] = max(rows[k][2], t)
            k += 1
    for name, f, t in rows:
        print(f"{t:8.1f} TF  eff {t * f:8.1f}   {name}")


if __name__ == "__main__":
    main()
