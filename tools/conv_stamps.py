#!/usr/bin/env python3
"""Phase timeline of one conv launch from in-kernel 100 MHz stamps (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (selects libfrp_lab.so)
import frp_amd_loader  # noqa
from frp_amd import native
a = [int(x) for x in sys.argv[1:8]] if len(sys.argv) > 7 else [320, 14, 14, 256, 256, 3, 1]
eng = native.Engine(0)
res = len(sys.argv) > 8 and "res" in sys.argv[8:]
extra = (0x10000 if "wino" in sys.argv[8:] else 0) | ((32 << 8) if "wino4" in sys.argv[8:] else 0)     # Winograd kernel (8 / 4 waves)
if "wino" in sys.argv[8:] and "wino4" not in sys.argv[8:]:
    extra |= (8 << 1) << 8            # the hand-ordered kernel carries the phase stamps in its lab variant only (dbg code 8)
ms, st = eng.conv_bench(a[0], a[1], a[2], a[3], a[4], a[5], a[6], 0 if res else 2, 1 | extra, res, 20, stamps=True)
st = st[st[:, 0] > 0].astype(np.int64)
t0 = st[:, 0].min()
us = (st - t0) / 100.0
names = ["start", "prologue issued", "acc zeroed", "1st k-step done", "1st tile loop done", "1st tile epilogue", "end"]
print(f"kernel avg {ms*1e3:.1f} us; {len(st)} workgroups; stamps relative to the earliest workgroup start (us)")
for i, n in enumerate(names):
    c = us[:, i]
    print(f"  {n:22s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f}")
d = np.diff(us[:, :7], axis=1)
for i in range(6):
    print(f"  phase {names[i]:>20s} -> {names[i+1]:<20s} median {np.median(d[:, i]):7.2f} us")
