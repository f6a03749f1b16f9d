#!/usr/bin/env python3
"""Run ONE conv shape a few times (for rocprofv3 --pmc runs). args: N H W Cin Cout k stride iters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402,F401  (selects libfrp_lab.so)
import frp_amd_loader  # noqa
from frp_amd import native
a = [int(x) for x in sys.argv[1:9]]
eng = native.Engine(0)
ms = eng.conv_bench(a[0], a[1], a[2], a[3], a[4], a[5], a[6], 2, 1 | (0x10000 if "wino" in sys.argv[9:] else 0), False, a[7])   # "wino": the Winograd kernel
print("ms", ms)
