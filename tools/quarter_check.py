#!/usr/bin/env python3
"""Quarter tiles against default tiles on every distinct conv shape of the embedder / detector at a small batch: bit
equality, and determinism over repeated launches.    python tools/quarter_check.py [faces] [frames]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native, netspec  # noqa: E402

DEFAULT, QUARTER = 0x40000, 0x20000


def shapes(layers, n, h, w, in_name):
    dims = {in_name: (h, w)}
    seen = []
    for l in layers:
        ih, iw = dims[l.src]
        oh, ow = netspec.out_hw(ih, iw, l.k, l.stride)
        dims[l.dst] = (oh, ow)
        if l.flags & (netspec.FLAG_OUT_F32 | netspec.FLAG_FLATTEN) or l.cin % 64:
            continue
        key = (n, ih, iw, l.cin, l.cout, l.k, l.stride, l.act, l.res is not None, l.flags)
        if key not in [s[1] for s in seen]:
            seen.append((l.name, key))
    return seen


def main():
    faces = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    eng = native.Engine(0)
    todo = shapes(netspec.iresnet_layers(), faces, 112, 112, "emb.in")
    if frames:
        todo += [t for t in shapes(netspec.detector_layers(), frames, 1088, 1920, "det.in") if not (t[1][9] & 4)]
    only = os.environ.get("ONLY")
    if only:
        todo = [t for t in todo if only in t[0]]
    bad = 0
    for name, (N, H, W, Ci, Co, k, s, act, has_res, fl) in todo:
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31))
        x = rng.standard_normal((N, H, W, Ci)).astype(np.float16)
        w = (rng.standard_normal((Co, k, k, Ci)) / np.sqrt(k * k * Ci)).astype(np.float16)
        bias = rng.standard_normal((9, Co) if fl & 1 else (Co,)).astype(np.float32) * 0.3
        slope = rng.uniform(0.1, 0.4, Co).astype(np.float32) if act == 2 else None
        Ho, Wo = netspec.out_hw(H, W, k, s)
        res = None
        if has_res:
            res = rng.standard_normal((N, Ho // 2, Wo // 2, Co) if fl & 4 else (N, Ho, Wo, Co)).astype(np.float16)
        kw = dict(stride=s, act=act, slope=slope, res=res)
        big = eng.conv2d(x, w, bias, flags=fl | DEFAULT, **kw)
        reps = int(os.environ.get("REPS", "3"))
        big2 = [eng.conv2d(x, w, bias, flags=fl | DEFAULT, **kw) for _ in range(reps)]
        nd = [int((big.view(np.uint16) != b.view(np.uint16)).sum()) for b in big2]
        if max(nd):
            print(f"{name}: DEFAULT tiles not deterministic {nd}")
        q = [eng.conv2d(x, w, bias, flags=fl | QUARTER, **kw) for _ in range(reps)]
        diff = [int((big.view(np.uint16) != qq.view(np.uint16)).sum()) for qq in q]
        flag = "" if max(diff) == 0 else "   <-- MISMATCH"
        bad += max(diff) != 0
        print(f"{name:28s} {N}x{H}x{W} {Ci}->{Co} k{k}s{s} act{act} res{int(has_res)} fl{fl}: differing halfwords {diff}{flag}")
        for r, qq in enumerate(q):
            if diff[r]:
                neq = big.reshape(-1, Co).view(np.uint16) != qq.reshape(-1, Co).view(np.uint16)
                rows = np.nonzero(neq.any(1))[0]
                cols = np.nonzero(neq.any(0))[0]
                print(f"    run {r}: pixels {rows[:12].tolist()}{'...' if len(rows) > 12 else ''} ({len(rows)} rows; tile {rows[0] // 128}, in-tile {rows[0] % 128}..{rows[-1] % 128}) "
                      f"couts {cols.min()}..{cols.max()} ({len(cols)})  max |diff| {float(np.abs(big.astype(np.float32) - qq.astype(np.float32)).max()):.4f}")
    print("mismatching shapes:", bad)


if __name__ == "__main__":
    main()
