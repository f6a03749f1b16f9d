#!/usr/bin/env python3
"""Where one sub-step of the Winograd k-loop spends its cycles: shader-clock stamps (s_memtime) of every wave of the first 32
workgroups inside sub-steps 6 and 7 of the second channel block of their first tile (lab build, kernel variant 2).
   tools/wino_substep.py [N H W Cin Cout] [res]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import _lab  # noqa: E402,F401
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

a = [int(x) for x in sys.argv[1:6]] if len(sys.argv) > 5 else [320, 14, 14, 256, 256]
res = "res" in sys.argv
eng = native.Engine(0)
for _ in range(2):
    ms, st = eng.conv_bench(a[0], a[1], a[2], a[3], a[4], 3, 1, 0 if res else 2, 1 | 0x10000 | ((13 << 1) << 8), res, 20, stamps=True)
st = st.reshape(32, 8, 8).astype(np.int64)          # [workgroup][wave][point]
ok = st[:, :, 0] > 0
print(f"kernel avg {ms * 1e3:.1f} us (stamped build); {ok.sum()} waves sampled")
names = ["sub-step 6: top -> stage landed (vmcnt + lgkmcnt wait)", "-> barrier passed", "-> its 8 MFMAs issued"]
d = np.diff(st[:, :, :4], axis=2)
for i, n in enumerate(names):
    x = d[:, :, i][ok]
    lo = d[:, :4, i][ok[:, :4]]
    hi = d[:, 4:, i][ok[:, 4:]]
    print(f"  {n:58s} median {np.median(x):7.0f}  p10 {np.percentile(x, 10):6.0f}  p90 {np.percentile(x, 90):6.0f}   waves 0-3 {np.median(lo):6.0f}  waves 4-7 {np.median(hi):6.0f}")
full = ok.all(1)
print(f"  arrival skew at the barrier of sub-step 6 (max - min of 'stage landed' over a workgroup's waves): median "
      f"{np.median((st[:, :, 1].max(1) - st[:, :, 1].min(1))[full]):.0f} cycles")
cyc = (st[:, :, 6] - st[:, :, 4])[ok].astype(np.float64)
rt = (st[:, :, 7] - st[:, :, 5])[ok].astype(np.float64)
print(f"  kernel start -> end: {np.median(cyc):.0f} shader cycles in {np.median(rt) / 100:.2f} us (100 MHz clock): in-kernel clock {np.median(cyc / rt) * 0.1:.3f} GHz")
if "raw" in sys.argv:
    for w in (0, 4):
        print("workgroup 0 wave", w, [hex(int(v)) for v in st[0, w]])
