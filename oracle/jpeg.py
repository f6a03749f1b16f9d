"""CPU restatement of the pixel half of baseline JPEG decoding (TEST INFRASTRUCTURE: imported by tests/ only; the product path is
csrc/jpeg_host.cpp + csrc/jpeg_kernels.hip).

The reference decodes uploaded stills with PIL (`face_recognition.load_image_file`, backend/app/services/face_service.py:139;
routes/face.py:177-185,216), i.e. with libjpeg(-turbo) at its defaults: the integer "slow" inverse DCT, "fancy" (triangle
filter) chroma upsampling and the fixed-point YCbCr -> RGB tables.  Those three algorithms are restated here from their
published definitions (IJG jidctint.c / jdsample.c / jdcolor.c: Loeffler-Ligtenberg-Moschytz IDCT with 13-bit constants and two
passes; h2v1 / h2v2 triangle filters with their rounding terms; 16-bit fixed-point colour constants) in vectorised numpy.
PINNED: tests/test_jpeg.py checks this file against PIL's decode of the committed stills (tests/golden/stills/*.jpg, written by
tests/golden/make_stills.py) and of stills generated on the spot - bit for bit.
Inputs are the quantised coefficients and tables as csrc/jpeg_host.cpp (or `huffman_decode` below, a slow pure-Python
restatement of ITU-T T.81 Annex F for small files) extracts them.
"""
from __future__ import annotations

import numpy as np

CONST_BITS, PASS1_BITS = 13, 2
F_0_298631336, F_0_390180644, F_0_541196100, F_0_765366865 = 2446, 3196, 4433, 6270
F_0_899976223, F_1_175875602, F_1_501321110, F_1_847759065 = 7373, 9633, 12299, 15137
F_1_961570560, F_2_053119869, F_2_562915447, F_3_072711026 = 16069, 16819, 20995, 25172


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(v, shift, dc_shift_left=None):
    """one LL&M pass over the LAST axis of v [..., 8] (int64); returns the 8 outputs descaled by `shift`"""
    i0, i1, i2, i3, i4, i5, i6, i7 = (v[..., k] for k in range(8))
    z2, z3 = i2, i6
    z1 = (z2 + z3) * F_0_541196100
    tmp2 = z1 + z3 * (-F_1_847759065)
    tmp3 = z1 + z2 * F_0_765366865
    tmp0 = (i0 + i4) << CONST_BITS
    tmp1 = (i0 - i4) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = i7, i5, i3, i1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F_1_175875602
    t0 = t0 * F_0_298631336
    t1 = t1 * F_2_053119869
    t2 = t2 * F_3_072711026
    t3 = t3 * F_1_501321110
    z1 = z1 * (-F_0_899976223)
    z2 = z2 * (-F_2_562915447)
    z3 = z3 * (-F_1_961570560) + z5
    z4 = z4 * (-F_0_390180644) + z5
    t0 = t0 + z1 + z3
    t1 = t1 + z2 + z4
    t2 = t2 + z2 + z3
    t3 = t3 + z1 + z4
    out = [tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3]
    return np.stack([_descale(o, shift) for o in out], axis=-1)


def idct_islow(coef: np.ndarray, q: np.ndarray) -> np.ndarray:
    """coef [..., 64] int16 quantised (natural order), q [64] -> samples [..., 8, 8] uint8 (row, column).
    jidctint.c: pass 1 over columns (results scaled up by 2**PASS1_BITS), pass 2 over rows, + 128, clamp.  (The all-AC-zero
    column shortcut of the C code - dcval << PASS1_BITS - gives the same value as the full pass: the even part reduces to
    (dc << 13), descaled by 11.)"""
    blk = (coef.astype(np.int64) * q.astype(np.int64)).reshape(coef.shape[:-1] + (8, 8))         # [.., row, col]
    ws = _idct_1d(np.swapaxes(blk, -1, -2), CONST_BITS - PASS1_BITS)                               # over rows index per column -> [.., col, row']
    ws = np.swapaxes(ws, -1, -2)                                                                   # [.., row', col]
    out = _idct_1d(ws, CONST_BITS + PASS1_BITS + 3)                                                # over columns of each row
    return np.clip(out + 128, 0, 255).astype(np.uint8)


def blocks_to_plane(samples: np.ndarray, by: int, bx: int) -> np.ndarray:
    """[by * bx, 8, 8] -> [by * 8, bx * 8]"""
    return samples.reshape(by, bx, 8, 8).transpose(0, 2, 1, 3).reshape(by * 8, bx * 8)


def upsample_h2v1_fancy(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v1_fancy_upsample: each input row -> twice the columns, 3/4 nearer + 1/4 farther, rounding terms 1 / 2"""
    p = p.astype(np.int32)
    h, w = p.shape
    out = np.empty((h, 2 * w), np.int32)
    left = np.concatenate([p[:, :1], p[:, :-1]], axis=1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], axis=1)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out.astype(np.uint8)


def upsample_h2v2_fancy(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v2_fancy_upsample: rows first (3 * nearer + farther, kept at 4x scale), then columns
    ((3 * this + neighbour + 8 or 7) >> 4; the first / last column: (4 * this + 8) >> 4 and (4 * this + 7) >> 4);
    the rows above the first and below the last real row are those rows themselves (the main controller's context rows)"""
    p = p.astype(np.int32)
    h, w = p.shape
    above = np.concatenate([p[:1], p[:-1]], axis=0)
    below = np.concatenate([p[1:], p[-1:]], axis=0)
    out = np.empty((2 * h, 2 * w), np.int32)
    for phase, far in ((0, above), (1, below)):
        s = 3 * p + far                                          # column sums at 4x
        last = np.concatenate([s[:, :1], s[:, :-1]], axis=1)
        nxt = np.concatenate([s[:, 1:], s[:, -1:]], axis=1)
        even = (3 * s + last + 8) >> 4
        odd = (3 * s + nxt + 7) >> 4
        even[:, 0] = (4 * s[:, 0] + 8) >> 4
        odd[:, -1] = (4 * s[:, -1] + 7) >> 4
        out[phase::2, 0::2] = even
        out[phase::2, 1::2] = odd
    return out.astype(np.uint8)


def _fix(x):
    return int(x * 65536 + 0.5)


def ycc_to_rgb(y: np.ndarray, cb: np.ndarray, cr: np.ndarray) -> np.ndarray:
    """jdcolor.c ycc_rgb_convert with its 16-bit tables: R = y + Cr_r[cr], G = y + ((Cb_g[cb] + Cr_g[cr]) >> 16), B = y + Cb_b[cb]"""
    x = np.arange(256, dtype=np.int64) - 128
    cr_r = (_fix(1.40200) * x + 32768) >> 16
    cb_b = (_fix(1.77200) * x + 32768) >> 16
    cr_g = -_fix(0.71414) * x
    cb_g = -_fix(0.34414) * x + 32768
    yy = y.astype(np.int64)
    r = yy + cr_r[cr]
    g = yy + ((cb_g[cb] + cr_g[cr]) >> 16)
    b = yy + cb_b[cb]
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def decode_from_coefficients(info: dict, coef: np.ndarray, qtab: np.ndarray) -> np.ndarray:
    """info: width, height, components, h_samp, v_samp, mcus_x, mcus_y; coef: int16, components back to back;
    -> [height, width, 3] uint8 RGB (grayscale replicated), as PIL's Image.open(...).convert("RGB") gives it"""
    W, H, nc = info["width"], info["height"], info["components"]
    planes, off = [], 0
    for c in range(nc):
        bx, by = info["mcus_x"] * info["h_samp"][c], info["mcus_y"] * info["v_samp"][c]
        n = bx * by * 64
        planes.append(blocks_to_plane(idct_islow(coef[off:off + n].reshape(bx * by, 64), qtab[c]), by, bx))
        off += n
    if nc == 1:
        g = planes[0][:H, :W]
        return np.stack([g, g, g], axis=-1)
    hs, vs = info["h_samp"][0], info["v_samp"][0]
    y = planes[0][:H, :W]
    ch, cw = -(-H // vs), -(-W // hs)                                  # the chroma planes' REAL extent (their padding is not filtered in)
    cb, cr = planes[1][:ch, :cw], planes[2][:ch, :cw]
    if (hs, vs) == (2, 2):
        cb, cr = upsample_h2v2_fancy(cb), upsample_h2v2_fancy(cr)
    elif (hs, vs) == (2, 1):
        cb, cr = upsample_h2v1_fancy(cb), upsample_h2v1_fancy(cr)
    elif (hs, vs) != (1, 1):
        raise ValueError("unsupported sampling")
    return ycc_to_rgb(y, cb[:H, :W], cr[:H, :W])


# ----------------------------------------------------------------------------- entropy decoding (small files only: pure Python)
_ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
           35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def huffman_decode(data: bytes):
    """ITU-T T.81 Annex F (sequential Huffman) restated for the checks of csrc/jpeg_host.cpp: -> (info dict, int16 coefficients
    in the library's layout, uint16 [3, 64] tables).  One interleaved scan, 8-bit samples, restart intervals."""
    pos, qt, dc, ac = 2, {}, {}, {}
    comps, ri, W, H = [], 0, 0, 0
    assert data[:2] == b"\xff\xd8"
    while True:
        assert data[pos] == 0xFF
        m = data[pos + 1]
        pos += 2
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        ln = (data[pos] << 8) | data[pos + 1]
        seg = data[pos + 2:pos + ln]
        pos += ln
        if m == 0xDB:
            o = 0
            while o < len(seg):
                pq, tq = seg[o] >> 4, seg[o] & 15
                o += 1
                tab = [0] * 64
                for i in range(64):
                    tab[_ZIGZAG[i]] = ((seg[o + 2 * i] << 8) | seg[o + 2 * i + 1]) if pq else seg[o + i]
                o += 128 if pq else 64
                qt[tq] = tab
        elif m == 0xC4:
            o = 0
            while o < len(seg):
                tc, th = seg[o] >> 4, seg[o] & 15
                counts = list(seg[o + 1:o + 17])
                vals = list(seg[o + 17:o + 17 + sum(counts)])
                o += 17 + sum(counts)
                codes, code, k = {}, 0, 0
                for ln_ in range(1, 17):
                    for _ in range(counts[ln_ - 1]):
                        codes[(ln_, code)] = vals[k]
                        code += 1
                        k += 1
                    code <<= 1
                (ac if tc else dc)[th] = codes
        elif m in (0xC0, 0xC1):
            H, W = (seg[1] << 8) | seg[2], (seg[3] << 8) | seg[4]
            comps = [[seg[6 + 3 * c], seg[7 + 3 * c] >> 4, seg[7 + 3 * c] & 15, seg[8 + 3 * c]] for c in range(seg[5])]
        elif m == 0xDD:
            ri = (seg[0] << 8) | seg[1]
        elif m == 0xDA:
            tabs = [(seg[2 + 2 * c] >> 4, seg[2 + 2 * c] & 15) for c in range(seg[0])]
            break
    if len(comps) == 1:
        comps[0][1] = comps[0][2] = 1
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mx, my = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    info = {"width": W, "height": H, "components": len(comps), "h_samp": [c[1] for c in comps] + [0] * (3 - len(comps)),
            "v_samp": [c[2] for c in comps] + [0] * (3 - len(comps)), "mcus_x": mx, "mcus_y": my, "restart_interval": ri}
    bx = [mx * c[1] for c in comps]
    by = [my * c[2] for c in comps]
    offs = np.concatenate([[0], np.cumsum([bx[c] * by[c] * 64 for c in range(len(comps))])]).astype(int)
    coef = np.zeros(int(offs[-1]), np.int16)
    # unstuffed bit string of each restart interval
    scan = data[pos:]
    bits, i, chunks = [], 0, []
    while i < len(scan):
        b = scan[i]
        if b == 0xFF:
            nb = scan[i + 1]
            if nb == 0:
                bits.append(0xFF)
                i += 2
                continue
            chunks.append(bytes(bits))
            bits = []
            if nb == 0xD9:
                break
            i += 2
            continue
        bits.append(b)
        i += 1
    state = {"chunk": 0, "bitpos": 0, "s": "".join(f"{x:08b}" for x in chunks[0])}

    def take(n):
        v = state["s"][state["bitpos"]:state["bitpos"] + n]
        state["bitpos"] += n
        return int(v.ljust(n, "0"), 2) if n else 0

    def decode(tab):
        code = 0
        for ln_ in range(1, 17):
            code = (code << 1) | take(1)
            if (ln_, code) in tab:
                return tab[(ln_, code)]
        raise ValueError("bad code")

    def extend(v, s):
        return v - (1 << s) + 1 if s and v < (1 << (s - 1)) else v

    pred = [0] * len(comps)
    left = ri
    for y in range(my):
        for x in range(mx):
            if ri and left == 0:
                state["chunk"] += 1
                state["s"] = "".join(f"{v:08b}" for v in chunks[state["chunk"]])
                state["bitpos"] = 0
                pred = [0] * len(comps)
                left = ri
            for c, (cid, hs, vs, tq) in enumerate(comps):
                for v in range(vs):
                    for h in range(hs):
                        base = offs[c] + ((y * vs + v) * bx[c] + x * hs + h) * 64
                        s = decode(dc[tabs[c][0]])
                        pred[c] += extend(take(s), s)
                        coef[base] = pred[c]
                        k = 1
                        while k < 64:
                            rs = decode(ac[tabs[c][1]])
                            r, sz = rs >> 4, rs & 15
                            if sz == 0:
                                if r == 15:
                                    k += 16
                                    continue
                                break
                            k += r
                            coef[base + _ZIGZAG[k]] = extend(take(sz), sz)
                            k += 1
            left -= 1
    q = np.ones((3, 64), np.uint16)
    for c, comp in enumerate(comps):
        q[c] = qt[comp[3]]
    return info, coef, q
