#!/usr/bin/env python3
"""Per-layer algorithmic vs measured HBM-side bytes of the conv launches of the LAST pipeline pass in two
rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv or rocpd .db) of the default bench workload.

    python tools/traffic_per_layer.py <fetch csv|db> <write csv|db> [passes=3]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frp_amd_loader  # noqa: E402,F401
from frp_amd import netspec as ns  # noqa: E402
from pmc_summarise import rows  # noqa: E402


def walk(layers, h0, w0, name, n):
    dims = {name: (h0, w0)}
    out = []
    for l in layers:
        h, w = dims[l.src]
        flat = bool(l.flags & ns.FLAG_FLATTEN)
        oh, ow = (1, 1) if flat else ns.out_hw(h, w, l.k, l.stride)
        inb = n * (l.cin if flat else h * w * l.cin) * 2
        if l.stride == 2 and l.k == 1:
            inb //= 4                                   # a 1x1 stride-2 conv touches a quarter of its input
        dims[l.dst] = (oh, ow)
        outb = n * oh * ow * l.cout * (4 if l.flags & ns.FLAG_OUT_F32 else 2)
        resb = 0
        if l.res:
            rh, rw = dims[l.res]
            resb = n * rh * rw * l.cout * 2
        wb = l.cout * l.cin * (1 if flat else l.k * l.k) * 2
        fl = 2 * n * oh * ow * l.cout * l.cin * (1 if flat else l.k * l.k)
        out.append((l, h, w, inb + resb + wb, outb, fl))
    return out


def main():
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    is_conv = lambda k: "conv_mfma" in k or "conv3x3_rows" in k or "emb_stem" in k or "stem12" in k  # noqa: E731
    f = [v for k, v in rows(sys.argv[1], "FETCH_SIZE") if is_conv(k)]
    w = [v for k, v in rows(sys.argv[2], "WRITE_SIZE") if is_conv(k)]
    n = len(f) // passes
    f, w = f[-n:], w[-n:]
    det = walk(ns.detector_layers(), 1088, 1920, "det.in", 32)
    emb = walk(ns.iresnet_layers(), 112, 112, "emb.in", 320)
    layers = det[2:] + emb if n == len(det) - 1 + len(emb) else det[1:] + emb      # fused two-layer stem?
    if n == len(det) - 1 + len(emb):                    # merge the two stems into one entry
        a, b = det[0], det[1]
        layers = [(b[0], a[1], a[2], a[3] - 0 + b[3] - a[4] - a[4], b[4], a[5] + b[5])] + layers
    tot_a = tot_m = 0.0
    print(f"{'layer':30s} {'HxW':>10s}  cin cout k s | alg rd   alg wr | meas rd  meas wr (MB) | x rd  x wr")
    for (l, h, wd, ard, awr, fl), fv, wv in zip(layers, f, w):
        mrd, mwr = 2 * fv * 1024 / 1e6, wv * 1024 / 1e6
        ard, awr = ard / 1e6, awr / 1e6
        tot_a += ard + awr
        tot_m += mrd + mwr
        print(f"{l.name:30s} {h:4d}x{wd:<5d} {l.cin:5d} {l.cout:4d} {l.k} {l.stride} | {ard:7.1f} {awr:7.1f} | {mrd:7.1f} {mwr:7.1f} | "
              f"{mrd / max(ard, 1e-9):5.2f} {mwr / max(awr, 1e-9):5.2f}")
    print(f"total algorithmic {tot_a / 1e3:.2f} GB, measured {tot_m / 1e3:.2f} GB")


if __name__ == "__main__":
    main()
