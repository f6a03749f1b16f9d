#!/usr/bin/env python3
"""The 64 -> 64 kernel (conv3x3_c64.hip) against the row-patch kernel's 512 x 64 tiles on the headline workload's shapes (32 x 1080p
frames, 320 faces): same process, alternating, best of 3; microseconds per launch, TFLOP/s, GB/s of algorithmic traffic.
    python tools/c64_probe.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (selects libfrp_lab.so)
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

NO_C64 = 0x100000
NO_PINGPONG = 0x400000       # all eight waves run an iteration in the same order (the kernel before the ping-pong)
SHAPES = [  # name, N, H, W, act, flags, res, launches per step
    ("det.layer1.0.conv1 272x480 ReLU", 32, 272, 480, 1, 0, False, 1),
    ("det.layer1.0.conv2 272x480 +res ReLU", 32, 272, 480, 1, 0, True, 1),
    ("emb.layer1.0.conv1 112x112 PReLU border", 320, 112, 112, 2, 1, False, 1),
    ("emb.layer1.x.conv1 56x56 PReLU border", 320, 56, 56, 2, 1, False, 2),
    ("emb.layer1.x.conv2 56x56 +res", 320, 56, 56, 0, 0, True, 2),
]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    eng = native.Engine(0)
    tot = [0.0, 0.0]
    for name, N, H, W, act, fl, res, cnt in SHAPES:
        best = [1e30, 1e30, 1e30]
        for _ in range(3):
            for v, extra in enumerate((NO_C64, 0, NO_PINGPONG)):
                best[v] = min(best[v], eng.conv_bench(N, H, W, 64, 64, 3, 1, act, fl | extra, res, iters) * 1e3)
        flops = 2.0 * N * H * W * 9 * 64 * 64
        bytes_ = N * H * W * 64 * 2 * (3 if res else 2)
        for v in range(2):
            tot[v] += best[v] * cnt
        print(f"{name:44s} row-patch {best[0]:7.1f} us {flops / best[0] / 1e6:7.1f} TF | c64 {best[1]:7.1f} us {flops / best[1] / 1e6:7.1f} TF "
              f"{bytes_ / best[1] / 1e3:7.1f} GB/s  x{best[0] / best[1]:.3f} | same order {best[2]:7.1f} us  x{best[2] / best[1]:.3f}", flush=True)
    print(f"per step (launch counts applied): row-patch {tot[0]:.1f} us, c64 {tot[1]:.1f} us")


if __name__ == "__main__":
    main()
