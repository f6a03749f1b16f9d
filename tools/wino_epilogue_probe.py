#!/usr/bin/env python3
"""What the Winograd kernel's epilogue variants cost: the same shapes with (none), (ReLU), (PReLU + 9 border-bias classes), (residual):
microseconds per launch, best of 3, same process.   python tools/wino_epilogue_probe.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

WINO = 0x10000
SHAPES = [("emb stage 3 14x14 256->256", 320, 14, 14, 256, 256), ("emb stage 2 28x28 128->128", 320, 28, 28, 128, 128),
          ("det 136x240 128->128 (2-D tiles)", 32, 136, 240, 128, 128)]
VARIANTS = [("plain", 0, 0, False), ("ReLU", 1, 0, False), ("PReLU", 2, 0, False), ("PReLU + border bias", 2, 1, False), ("residual", 0, 0, True),
            ("residual + ReLU", 1, 0, True)]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    eng = native.Engine(0)
    for name, N, H, W, Ci, Co in SHAPES:
        best = [1e30] * len(VARIANTS)
        for _ in range(3):
            for v, (_, act, fl, res) in enumerate(VARIANTS):
                best[v] = min(best[v], eng.conv_bench(N, H, W, Ci, Co, 3, 1, act, fl | WINO, res, iters) * 1e3)
        print(f"{name:36s} " + " | ".join(f"{VARIANTS[v][0]} {best[v]:6.1f} us" for v in range(len(VARIANTS))), flush=True)


if __name__ == "__main__":
    main()
