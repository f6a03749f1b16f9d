#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py per kernel family.

    python tools/pmc_summarise.py <fetch csv|results.db> <write csv|results.db> <passes> > out.json

`passes` = pipeline passes the profiled command made (warmup + steps).  Counter unit is KB; FETCH_SIZE
gets the gfx950 x2 correction of MI355X_MICROARCH.md (validated on match_kernel whose algorithmic
bytes are known).  The start-up kernels (gallery normalise, random fill) are dropped.
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402

FAMILIES = ["conv3x3_lean_kernel", "conv3x3_wino", "conv3x3_c64_kernel", "conv3x3_rows_kernel", "conv_mfma_kernel", "stem12_u8_kernel", "emb_stem_kernel", "stem_u8_kernel", "gather_logits_kernel", "topk_rows_kernel", "preprocess_kernel", "decode_nms_kernel", "align_kernel",
            "match_kernel", "match_top1_kernel", "l2norm", "compact_faces", "chips_to_blob"]
SKIP = ["normalize_rows_kernel", "fill_random", "mfma_peak"]


def family(name):
    for s in SKIP:
        if s in name:
            return None
    for f in FAMILIES:
        if f in name:
            return f
    return "other"


def rows(path, counter):
    """(kernel name, value) per dispatch from a rocprofv3 counter_collection.csv or a rocpd results.db"""
    if path.endswith(".db"):
        import sqlite3
        con = sqlite3.connect(path)
        yield from con.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,))
        return
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                yield row["Kernel_Name"], float(row["Counter_Value"])


def collect(path, counter):
    out = {}
    for name, value in rows(path, counter):
        fam = family(name)
        if fam is None:
            continue
        d = out.setdefault(fam, [0, 0.0])
        d[0] += 1
        d[1] += float(value)
    return out


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    passes = int(sys.argv[3])
    kernels = {}
    for fam in sorted(set(fetch) | set(write)):
        n, fkb = fetch.get(fam, [0, 0.0])
        n2, wkb = write.get(fam, [0, 0.0])
        n = max(n, n2)
        kernels[fam] = {"launches": n, "fetch_KB_raw_sum": round(fkb), "write_KB_sum": round(wkb),
                        "hbm_bytes_per_launch_corrected": round((2 * fkb + wkb) * 1024 / max(n, 1))}
    conv = {k: sum(kernels.get(f, {}).get(k, 0) for f in ("conv3x3_lean_kernel", "conv3x3_wino", "conv3x3_c64_kernel", "conv3x3_rows_kernel", "conv_mfma_kernel", "stem12_u8_kernel", "emb_stem_kernel"))
            for k in ("fetch_KB_raw_sum", "write_KB_sum")}
    doc = {
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of `bench.py --steps 2 "
                "--warmup 1 --cpu-frames 0`. Counter unit KB. gfx950 correction (MI355X_MICROARCH.md, HBM): "
                "FETCH_SIZE counts half of a 16-B/lane streaming read -> x2; validated on match_kernel whose "
                "algorithmic bytes are known (102.4 MB gallery + 0.33 MB queries).",
        "passes": passes,
        "kernel_source_sha256_16": native.kernel_source_hash(),   # bench.py blanks roofline.traffic when the sources differ
        "kernels": kernels,
        "conv_traffic_bytes_per_step": round((2 * conv["fetch_KB_raw_sum"] + conv["write_KB_sum"]) * 1024 / passes),
    }
    json.dump(doc, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
