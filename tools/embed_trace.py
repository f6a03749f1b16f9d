#!/usr/bin/env python3
"""One embedder call of M aligned chips (full IResNet-100, synthetic weights), repeated: run under
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o kt -- python3 tools/embed_trace.py M [reps]
and list the last pass with tools/embed_trace.py --list <dir>/..._kernel_trace.csv M"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native, netspec as ns, weights  # noqa: E402


def main():
    if sys.argv[1] == "--list":
        rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
        M = int(sys.argv[3])
        starts = [i for i, r in enumerate(rows) if "emb_stem" in r["Kernel_Name"] or "chips_to_blob" in r["Kernel_Name"]]
        seq = rows[starts[-1]:]
        t0 = int(seq[0]["Start_Timestamp"])
        tot = 0.0
        for r in seq:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            tot += d
            print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {d:7.1f} us  grid {r['Grid_Size_X']:>8s}  {r['Kernel_Name'][:90]}")
        print(f"{len(seq)} kernels, sum {tot:.1f} us, span {(int(seq[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, M = {M}: "
              f"{M * 24.18e9 / ((int(seq[-1]['End_Timestamp']) - t0) * 1e-9) / 1e12:.1f} TFLOP/s")
        return
    M = int(sys.argv[1])
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    raw = weights.make_synthetic_raw(7)
    eng = native.Engine(0)
    eng.load_weights(weights.pack_blob(raw))
    chips = np.random.default_rng(3).integers(0, 256, size=(M, 112, 112, 3), dtype=np.uint8)
    for _ in range(reps):
        eng.embed_aligned(chips)
    eng.close()


if __name__ == "__main__":
    main()
