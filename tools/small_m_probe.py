#!/usr/bin/env python3
"""Quarter tiles (128 pixels x 64 couts, two workgroups per CU; conv_common.h: conv_small_m) against the default tiles
(256 x 128, Winograd where eligible) on shapes with few tiles: the embedder at 1 / 8 / 36 / 64 / 100 / 128 faces and the
detector at the small pyramid scales of BASELINE config 4.  Same process, alternating, best of 3; microseconds per launch.
    python tools/small_m_probe.py [iters]          (lab build of the library)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

DEFAULT, QUARTER, WINO = 0x40000, 0x20000, 0x10000

EMB = [  # name, H, W, Cin, Cout, k, stride, act, flags, res, winograd-eligible
    ("stage 1 conv 56x56 64->64", 56, 56, 64, 64, 3, 1, 2, 1, False, False),
    ("stage 2 conv1 28x28 128->128", 28, 28, 128, 128, 3, 1, 2, 1, False, True),
    ("stage 2 conv2 28x28 +res", 28, 28, 128, 128, 3, 1, 0, 0, True, True),
    ("stage 3 conv1 14x14 256->256", 14, 14, 256, 256, 3, 1, 2, 1, False, True),
    ("stage 3 conv2 14x14 +res", 14, 14, 256, 256, 3, 1, 0, 0, True, True),
    ("stage 4 conv 7x7 512->512", 7, 7, 512, 512, 3, 1, 2, 1, False, False),
    ("layer3.0.conv2 28x28 256->256 s2", 28, 28, 256, 256, 3, 2, 0, 0, True, False),
    ("layer4.0.conv2 14x14 512->512 s2", 14, 14, 512, 512, 3, 2, 0, 0, True, False),
    ("layer3.0 shortcut 1x1 s2 128->256", 28, 28, 128, 256, 1, 2, 0, 0, False, False),
]
DET = [  # name, N, H, W, Cin, Cout, k, stride, act, flags, res
    ("det s=.25 stride 8: 4 x 68x120 128->128", 4, 68, 120, 128, 128, 3, 1, 1, 0, False),
    ("det s=.25 stride 16: 4 x 34x60 256->256", 4, 34, 60, 256, 256, 3, 1, 1, 0, True),
    ("det s=.25 stride 32: 4 x 17x30 256->256", 4, 17, 30, 256, 256, 3, 1, 1, 0, False),
    ("det s=.25 head out 4 x 68x120 128->32", 4, 68, 120, 128, 32, 3, 1, 0, 0, False),
    ("det s=.5 stride 32: 4 x 34x60 256->256", 4, 34, 60, 256, 256, 3, 1, 1, 0, False),
    ("det s=.5 stride 16: 4 x 68x120 256->256", 4, 68, 120, 256, 256, 3, 1, 1, 0, True),
    ("det 1080p x1 stride 32: 34x60 256->256", 1, 34, 60, 256, 256, 3, 1, 1, 0, False),
    ("det 1080p x1 stride 16: 68x120 256->256", 1, 68, 120, 256, 256, 3, 1, 1, 0, False),
    ("det 1080p x1 stride 8: 136x240 128->128", 1, 136, 240, 128, 128, 3, 1, 1, 0, False),
]


def run(eng, N, H, W, Ci, Co, k, s, act, fl, res, iters, wino_ok):
    variants = [fl | DEFAULT, fl | QUARTER] + ([fl | WINO] if wino_ok else [])
    best = [1e30] * 3
    for _ in range(3):
        for v, f in enumerate(variants):
            best[v] = min(best[v], eng.conv_bench(N, H, W, Ci, Co, k, s, act, f, res, iters) * 1e3)
    return best


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    eng = native.Engine(0)
    ncu = 256
    for faces in (1, 8, 36, 64, 100, 128, 186):
        print(f"== embedder, {faces} faces")
        for name, H, W, Ci, Co, k, s, act, fl, res, wok in EMB:
            Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
            M = faces * Ho * Wo
            dt = -(-M // 256) * -(-Co // 128) if Co > 64 or not (k == 3 and s == 1) else -(-M // 512)
            b = run(eng, faces, H, W, Ci, Co, k, s, act, fl, res, iters, wok)
            w = f"  winograd {b[2]:7.1f}" if wok else ""
            auto = "quarter" if dt * 2 <= ncu else "default"
            print(f"  {name:36s} default tiles {dt:4d}  direct {b[0]:7.1f} us  quarter {b[1]:7.1f} us  x{b[0] / b[1]:.2f}{w}   auto -> {auto}")
    print("== detector at small scales")
    for name, N, H, W, Ci, Co, k, s, act, fl, res in DET:
        M = N * H * W
        dt = -(-M // 256) * -(-Co // 128) if Co > 64 else -(-M // 512)
        b = run(eng, N, H, W, Ci, Co, k, s, act, fl, res, iters, False)
        auto = "quarter" if dt * 2 <= ncu else "default"
        print(f"  {name:44s} default tiles {dt:4d}  default {b[0]:7.1f} us  quarter {b[1]:7.1f} us  x{b[0] / b[1]:.2f}   auto -> {auto}")


if __name__ == "__main__":
    main()
