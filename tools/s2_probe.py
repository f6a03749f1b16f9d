#!/usr/bin/env python3
"""The stride-2 row-patch kernel (conv3x3_s2.hip) against the generic kernel's per-tap images on the headline workload's shapes (32 x 1080p
frames, 320 faces; the embedder's shapes without their shortcut segment): same process, alternating, best of 3; microseconds per
launch, TFLOP/s, GB/s of algorithmic traffic.
    python tools/s2_probe.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (selects libfrp_lab.so)
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

GENERIC = 1 << 8
SHAPES = [  # name, N, H, W, Cin, Cout, act
    ("det.layer2.0.conv1 272x480  64->128 ReLU", 32, 272, 480, 64, 128, 1),
    ("det.layer3.0.conv1 136x240 128->256 ReLU", 32, 136, 240, 128, 256, 1),
    ("det.layer4.0.conv1  68x120 256->256 ReLU", 32, 68, 120, 256, 256, 1),
    ("emb.layer2.0.conv2  56x56  128->128", 320, 56, 56, 128, 128, 0),
    ("emb.layer3.0.conv2  28x28  256->256", 320, 28, 28, 256, 256, 0),
    ("emb.layer4.0.conv2  14x14  512->512", 320, 14, 14, 512, 512, 0),
]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    eng = native.Engine(0)
    tot = [0.0, 0.0]
    for name, N, H, W, Cin, Cout, act in SHAPES:
        best = [1e30, 1e30]
        for _ in range(3):
            for v, extra in enumerate((GENERIC, 0x200000)):
                best[v] = min(best[v], eng.conv_bench(N, H, W, Cin, Cout, 3, 2, act, extra, False, iters) * 1e3)
        flops = 2.0 * N * (H // 2) * (W // 2) * 9 * Cin * Cout
        bytes_ = N * H * W * Cin * 2 + N * (H // 2) * (W // 2) * Cout * 2
        for v in range(2):
            tot[v] += best[v]
        print(f"{name:44s} generic {best[0]:7.1f} us {flops / best[0] / 1e6:7.1f} TF | s2 {best[1]:7.1f} us {flops / best[1] / 1e6:7.1f} TF "
              f"{bytes_ / best[1] / 1e3:7.1f} GB/s  x{best[0] / best[1]:.3f}", flush=True)
    print(f"sum: generic {tot[0]:.1f} us, s2 {tot[1]:.1f} us")


if __name__ == "__main__":
    main()
