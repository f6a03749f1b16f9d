#!/usr/bin/env python3
"""Winograd F(2,3) conv kernel vs the direct (lean) kernel on the embedder's eligible shapes (320 faces), same process,
alternating, best of 3: microseconds per launch and effective TFLOP/s (algorithmic FLOPs of the direct convolution)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

SHAPES = [  # name, N, H, W, Cin, Cout, act, flags, res
    ("emb stage 3 conv1 (14x14 256->256, PReLU, border bias)", 320, 14, 14, 256, 256, 2, 1, False),
    ("emb stage 3 conv2 (14x14 256->256, residual)", 320, 14, 14, 256, 256, 0, 0, True),
    ("emb stage 2 conv1 (28x28 128->128, PReLU, border bias)", 320, 28, 28, 128, 128, 2, 1, False),
    ("emb stage 2 conv2 (28x28 128->128, residual)", 320, 28, 28, 128, 128, 0, 0, True),
    ("emb layer3.0.conv1 (28x28 128->256)", 320, 28, 28, 128, 256, 2, 1, False),
    ("emb layer4.0.conv1 (14x14 256->512)", 320, 14, 14, 256, 512, 2, 1, False),
    # row-patch form (W > 30): the detector's maps at 32 x 1080p, the embedder's 56 x 56
    ("det 136x240 128->128 (ReLU)", 32, 136, 240, 128, 128, 1, 0, False),
    ("det 136x240 128->128 (residual + ReLU)", 32, 136, 240, 128, 128, 1, 0, True),
    ("det 68x120 256->256 (residual + ReLU)", 32, 68, 120, 256, 256, 1, 0, True),
    ("det 68x120 128->128 (ReLU)", 32, 68, 120, 128, 128, 1, 0, False),
    ("det 34x60 256->256 (ReLU)", 32, 34, 60, 256, 256, 1, 0, False),
    ("emb layer2.0.conv1 (56x56 64->128)", 320, 56, 56, 64, 128, 2, 1, False),
]


def main():
    eng = native.Engine(0)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    tot = [0.0, 0.0, 0.0]
    for name, N, H, W, Ci, Co, act, fl, res in SHAPES:
        variants = [0, 0x10000, 0x10000 | (128 << 8)] + [0x10000 | ((int(x) << 1) << 8) for x in os.environ.get("WINO_VARIANTS", "").split(",") if x]
        t2d = len(variants) if os.environ.get("WINO_T2D") else -1          # + the 2-D tiles (8 x 30; flags bit 19), every shape
        if t2d >= 0:
            variants.append(0x10000 | 0x80000)
        best = [1e30] * len(variants)
        for _ in range(3):
            # direct, Winograd (hand-ordered k-loop, round 4), Winograd first generation (compiler-scheduled k-loop: dbg bit 128),
            # + lab variants of the hand-ordered loop (WINO_VARIANTS=14,12: dbg codes)
            for v, extra in enumerate(variants):
                if v >= 2 and W > 30 and v != t2d:                             # (wide maps: the row-patch form is first generation anyway)
                    best[v] = float("nan")
                    continue
                if v == 1 and W > 30:
                    extra |= 64 << 8                                           # wide maps: the lab's row-patch form (dbg bit 64)
                best[v] = min(best[v], eng.conv_bench(N, H, W, Ci, Co, 3, 1, act, fl | extra, res, iters) * 1e3)
        fl_ = 2.0 * N * H * W * 9 * Ci * Co
        tot = tot + [0.0] * (len(best) - len(tot))
        for v in range(len(best)):
            tot[v] += best[v]
        print(f"{name:58s} direct {best[0]:7.1f} us {fl_ / best[0] / 1e6:7.1f} TF | winograd {best[1]:7.1f} us {fl_ / best[1] / 1e6:7.1f} TF eff "
              f"x{best[0] / best[1]:.3f} | first generation {best[2]:7.1f} us x{best[0] / best[2]:.3f}" + "".join(f" | var {best[v]:7.1f}" for v in range(3, len(best))), flush=True)
    print(f"sum direct {tot[0]:.1f} us, winograd {tot[1]:.1f} us x{tot[0] / tot[1]:.3f}, first generation {tot[2]:.1f} us x{tot[0] / tot[2]:.3f}" + "".join(f", var {t:.1f}" for t in tot[3:]))


if __name__ == "__main__":
    main()
