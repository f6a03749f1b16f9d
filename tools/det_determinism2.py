#!/usr/bin/env python3
"""Follow-up of tools/det_determinism.py: does the run-to-run difference of the head maps need the per-call frame upload?
mode `upload`: Engine.detect (hipMemcpy2DAsync of the pageable frames, then the detector) - after a differing run the device
copy of the frames is read back and compared with the host frames; mode `resident`: one upload, then detect_resident only.
    python tools/det_determinism2.py [B=4] [REPS=1500] [upload|resident|pinned]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import frp_amd_loader  # noqa: E402,F401
from frp_amd import native  # noqa: E402
from conftest import get_raw_and_blob  # noqa: E402
from test_gpu_pipeline import _frames  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    mode = sys.argv[3] if len(sys.argv) > 3 else "upload"
    os.environ["FRP_NO_WINO"] = "1"
    rng = np.random.default_rng(77)
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    fr = _frames(rng, B, 1080, 1920)
    eng = native.Engine(0)
    eng.load_weights(blob)
    if mode == "pinned":
        host = eng.host_frames(B, 1080, 1920)
        host[...] = fr
        fr_src = host
    else:
        fr_src = fr

    def run():
        if mode in ("upload", "pinned"):
            eng.detect(fr_src, max_faces=16, det_thresh=0.5)
        else:
            eng.detect_resident((1080, 1920), max_faces=16, det_thresh=0.5)
        return eng.head_maps()
    if mode == "resident":
        eng.upload_frames(fr)
    first = [h.copy() for h in run()]
    bad = 0
    for r in range(reps):
        got = run()
        diff = [lv for lv in range(3) if not np.array_equal(first[lv].view(np.uint16), got[lv].view(np.uint16))]
        if diff:
            bad += 1
            dev = eng.det_source()
            same = np.array_equal(dev, fr)
            where = ""
            if not same:
                idx = np.argwhere(dev != fr)
                where = f"; device frames differ in {len(idx)} bytes: images {sorted(set(idx[:, 0].tolist()))}, rows {idx[:, 1].min()}..{idx[:, 1].max()}, cols {idx[:, 2].min()}..{idx[:, 2].max()}"
            a, b = first[0].view(np.uint16), got[0].view(np.uint16)
            i = np.argwhere(a != b)
            print(f"[{mode}] rep {r}: maps {diff} differ; map 0 rows {i[:, 1].min()}..{i[:, 1].max()} cols {i[:, 2].min()}..{i[:, 2].max()} images {sorted(set(i[:, 0].tolist()))}; "
                  f"device frames equal the host frames afterwards: {same}{where}", flush=True)
    print(f"[{mode}] B={B}: {reps} repetitions, {bad} differed from the first run", flush=True)
    eng.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
