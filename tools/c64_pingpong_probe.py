#!/usr/bin/env python3
"""The 64 -> 64 kernel with and without its ping-pong (flags bit 22 = all eight waves in the same order), six alternating rounds per shape
in one process, microseconds per launch (lab build).   python tools/c64_pingpong_probe.py"""
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import _lab, frp_amd_loader
from frp_amd import native
eng = native.Engine(0)
NO_C64, NO_PP = 0x100000, 0x400000
for name, N, H, W, act, fl, res in (("det1.conv1", 32, 272, 480, 1, 0, False), ("det1.conv2+res", 32, 272, 480, 1, 0, True), ("emb1.0.conv1", 320, 112, 112, 2, 1, False)):
    rows = []
    for _ in range(6):
        rows.append([eng.conv_bench(N, H, W, 64, 64, 3, 1, act, fl | extra, res, 20) * 1e3 for extra in (0, NO_PP)])
    print(name, "ping-pong", " ".join(f"{r[0]:6.1f}" for r in rows), "| same order", " ".join(f"{r[1]:6.1f}" for r in rows), flush=True)
