#!/usr/bin/env python3
"""Per-layer kernel durations and inter-kernel gaps of the LAST pipeline pass in a
`rocprofv3 --kernel-trace --output-format csv` trace of bench.py (default workload).

    python tools/layer_times.py gpurun_out/kt/kt_kernel_trace.csv [frames faces]
"""
import csv
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frp_amd_loader  # noqa: E402,F401
from frp_amd import netspec as ns  # noqa: E402


def walk(layers, h0, w0, name, n):
    dims = {name: (h0, w0)}
    out = []
    for l in layers:
        h, w = dims[l.src]
        flat = bool(l.flags & ns.FLAG_FLATTEN)
        oh, ow = (1, 1) if flat else ns.out_hw(h, w, l.k, l.stride)
        dims[l.dst] = (oh, ow)
        fl = 2 * n * oh * ow * (l.cout_real or l.cout) * (l.cin_real or l.cin) * (1 if flat else l.k * l.k)
        # algorithmic HBM bytes: the input pixels a stride-s kernel touches (every pixel for 3x3, every s-th row and column
        # for 1x1), the output, the residual, the weights; fp16 unless the output is fp32
        touched = n * h * w if (l.k > 1 or flat) else n * oh * ow
        by = touched * l.cin * 2 + n * oh * ow * l.cout * (4 if (l.flags & ns.FLAG_OUT_F32) else 2) + l.cout * l.cin * (1 if flat else l.k * l.k) * 2
        if l.res:
            rh, rw = dims[l.res]
            by += n * rh * rw * l.cout * 2
        out.append((l, h, w, fl, by))
    return out


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    faces = int(sys.argv[3]) if len(sys.argv) > 3 else 10 * frames
    starts = [i for i, r in enumerate(rows) if "stem_u8" in r["Kernel_Name"] or "stem12_u8" in r["Kernel_Name"]]
    seq = rows[starts[-1]:]
    fused2 = "stem12_u8" in seq[0]["Kernel_Name"]
    det = walk(ns.detector_layers(), 1088, 1920, "det.in", frames)
    emb = walk(ns.iresnet_layers(), 112, 112, "emb.in", faces)
    if not os.environ.get("FRP_NO_KCONCAT"):
        # K-concat (csrc/frp_api.cpp: frp_load_weights): a strided block's 1x1 shortcut conv is not a launch of its own - it
        # rides in the k-loop of the block's 3x3 stride-2 conv, which then reads the block input at every second pixel
        # instead of the shortcut map: FLOPs and bytes of the pair are charged to that launch
        fused, pend = [], None
        for (l, h, w, fl, by) in emb:
            if l.k == 1 and l.stride == 2 and l.name.endswith("downsample.0"):
                pend = (l, h, w, fl, by)
                continue
            if pend is not None and l.res == pend[0].dst:
                sl, sh, sw, sfl, sby = pend
                oh, ow = ns.out_hw(sh, sw, 1, 2)
                by = by - faces * oh * ow * l.cout * 2 + faces * oh * ow * sl.cin * 2 + sl.cout * sl.cin * 2     # no shortcut map; x at the centre taps
                fl += sfl
                pend = None
            fused.append((l, h, w, fl, by))
        emb = fused
    emb_stem = any("emb_stem" in r["Kernel_Name"] for r in seq)      # the embedder's first conv runs in its own kernel
    convs = iter(det[2 if fused2 else 1:] + emb[1 if emb_stem else 0:])
    prev_end, t0, tot, tot_gap = None, int(seq[0]["Start_Timestamp"]), 0.0, 0.0
    bounds = []
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        prev_end = e
        d = (e - s) / 1e3
        tot += d
        tot_gap += gap
        name = r["Kernel_Name"]
        if "conv_mfma" in name or "conv3x3_rows" in name or "conv3x3_lean" in name or "conv3x3_wino" in name or "conv3x3_c64" in name:
            l, h, w, fl, by = next(convs)
            stem_fused = "conv3x3_c64" in name and name.rstrip().endswith(", true>(frp::ConvParams)")
            if stem_fused:
                # the embedder's stem runs inside this launch (conv3x3_c64.hip, STEM): FLOPs of both convs; bytes = the chips, a quarter
                # of the stem's map (its even pixels, for the block's shortcut), the conv's output and weights
                l2, h2, w2, fl2, by2 = next(convs)
                n_img = by2 // (h2 * w2 * 64 * 2 * 2)
                by = n_img * h * w * 8 * 2 + n_img * h * w * 64 * 2 // 4 + n_img * h2 * w2 * 64 * 2 + (64 * 64 * 9 + 64 * 8 * 9) * 2
                fl += fl2
                l, h, w = l2, h2, w2
            cfg = re.search(r"<(\d+), (\d+),", name)
            wino = "conv3x3_wino" in name
            c64 = "conv3x3_c64" in name
            # what this launch cannot beat on this device: its FLOPs at the k-loop's own rate (1.25 PFLOP/s)
            # or its algorithmic bytes at 4.5 TB/s (the best streaming rate measured here), whichever is longer
            bound = max(fl / 1.25e9, by / 4.5e6)
            bounds.append((d, bound))
            print(f"{l.name:28s} {h:4d}x{w:<4d} {l.cin:5d}->{l.cout:3d} k{l.k}s{l.stride} grid {r['Grid_Size_X']:>7s} "
                  f"{d:8.1f} us  gap {gap:6.1f}  {fl / d / 1e6:7.1f} TF  {by / d / 1e3:7.1f} GB/s  x{d / bound:4.2f} of "
                  f"{'mfma' if fl / 1.25e9 >= by / 4.5e6 else 'hbm '} bound  " + (("winograd F(2,3), 8x30-pixel tiles x 128" if "wino2_kernel<32>" in name else "winograd F(2,3) 256x128") if wino else ((f"64 -> 64 kernel, {'16x16' if ', 16, ' in name else '32x8'}-pixel tiles, weights in registers" + (", the stem conv in the same launch" if stem_fused else "")) if c64 else f"tile {cfg.group(1)}x{cfg.group(2)}")))
        else:
            print(f"{name[:57]:57s} grid {r['Grid_Size_X']:>9s} {d:8.1f} us  gap {gap:6.1f}")
    print(f"sum of kernels {tot:.1f} us, sum of gaps {tot_gap:.1f} us, span {(prev_end - t0) / 1e3:.1f} us")
    if bounds:
        td, tb = sum(d for d, _ in bounds), sum(b for _, b in bounds)
        print(f"conv launches: {td:.1f} us against {tb:.1f} us of per-launch bounds (max of FLOPs at 1.25 PFLOP/s, algorithmic bytes "
              f"at 4.5 TB/s): x{td / tb:.2f}")


if __name__ == "__main__":
    main()
