#!/usr/bin/env python3
"""Writes the committed JPEG still set of tests/test_jpeg.py (tests/golden/stills/*.jpg): small synthetic images encoded by PIL
(libjpeg-turbo) in the layouts uploads arrive in - 4:2:0 (PIL's default), 4:2:2, 4:4:4, grayscale, odd sizes that leave
partial MCUs, optimised Huffman tables, restart intervals, a low and a high quality.  Data only: the expected pixels are what
PIL decodes from these files at test time; `stills.json` records the md5 of that decode so that a PIL whose decoder differs
from the one the files were checked with is noticed."""
import hashlib
import io
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "stills")


def picture(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(np.sin(xx / 9.0) + np.cos(yy / 7.0)) * 70 + 120, xx * 255.0 / max(1, w - 1), 255 - yy * 255.0 / max(1, h - 1)], -1)
    img += rng.normal(0, 14, (h, w, 3))
    img[h // 4:h // 2, w // 3:w // 2] = (250, 10, 10)          # saturated patches: the clamps of the colour conversion
    img[h // 2:h // 2 + 5, :w // 4] = (5, 5, 250)
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = [("c420_q85", 96, 128, dict(quality=85)), ("c420_odd_q60", 67, 53, dict(quality=60)), ("c444_q92", 40, 72, dict(quality=92, subsampling=0)),
         ("c422_q75", 50, 90, dict(quality=75, subsampling=1)), ("gray_q80", 45, 61, dict(quality=80)), ("c420_opt_q30", 64, 64, dict(quality=30, optimize=True)),
         ("c420_rst_q90", 80, 112, dict(quality=90, restart_marker_blocks=3)), ("c444_rst_rows", 33, 48, dict(quality=88, subsampling=0, restart_marker_rows=1))]


def main():
    os.makedirs(OUT, exist_ok=True)
    meta = {}
    for i, (name, h, w, kw) in enumerate(CASES):
        im = Image.fromarray(picture(h, w, 100 + i))
        if name.startswith("gray"):
            im = im.convert("L")
        b = io.BytesIO()
        im.save(b, "JPEG", **kw)
        data = b.getvalue()
        with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
            f.write(data)
        dec = np.array(Image.open(io.BytesIO(data)).convert("RGB"))
        meta[name] = {"height": h, "width": w, "bytes": len(data), "decoded_rgb_md5": hashlib.md5(dec.tobytes()).hexdigest()}
    with open(os.path.join(OUT, "stills.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", len(CASES), "stills,", sum(m["bytes"] for m in meta.values()), "bytes")


if __name__ == "__main__":
    main()
