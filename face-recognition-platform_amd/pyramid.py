"""Cross-scale merge of the detection pyramid (BASELINE config 4: "RetinaFace multi-scale pyramid").

Host logic over a handful of boxes per frame: every scale's detections (already NMS-ed within the scale
on the device, coordinates of the resized image) are mapped back to frame pixels, concatenated in scale
order, sorted by score and put through the same greedy +1-pixel-area NMS the device uses.  fp32
arithmetic, one rounding per operation, so the oracle's independent restatement reproduces it exactly.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

f32 = np.float32


def scaled_size(H: int, W: int, scale: float) -> Tuple[int, int]:
    return max(1, int(round(H * scale))), max(1, int(round(W * scale)))


def merge_scales(per_scale: Sequence[Tuple[Tuple[int, int], dict]], frame_hw: Tuple[int, int], max_faces: int,
                 nms_iou: float):
    """per_scale: [((Hs, Ws), {"boxes" [B,P,4], "kps" [B,P,5,2], "scores" [B,P], "counts" [B]})] in scale order.
    -> boxes [B,K,4], kps [B,K,5,2], scores [B,K], counts [B] in frame pixels."""
    H, W = frame_hw
    B = per_scale[0][1]["counts"].shape[0]
    K = max_faces
    out_b = np.zeros((B, K, 4), f32)
    out_k = np.zeros((B, K, 5, 2), f32)
    out_s = np.zeros((B, K), f32)
    out_c = np.zeros((B,), np.int32)
    for b in range(B):
        bs, ks, ss = [], [], []
        for (Hs, Ws), d in per_scale:
            n = int(d["counts"][b])
            ry, rx = f32(H) / f32(Hs), f32(W) / f32(Ws)
            sc = np.array([rx, ry, rx, ry], f32)
            bs.append((d["boxes"][b, :n].astype(f32) * sc).astype(f32))
            ks.append((d["kps"][b, :n].astype(f32) * np.array([rx, ry], f32)).astype(f32))
            ss.append(d["scores"][b, :n].astype(f32))
        bx, kp, scv = np.concatenate(bs), np.concatenate(ks), np.concatenate(ss)
        order = np.argsort(-scv, kind="stable")            # score desc; ties: scale order, then detector order
        bx, kp, scv = bx[order], kp[order], scv[order]
        x1, y1, x2, y2 = bx[:, 0], bx[:, 1], bx[:, 2], bx[:, 3]
        area = ((x2 - x1 + f32(1)) * (y2 - y1 + f32(1))).astype(f32)
        supp = np.zeros(len(bx), bool)
        keep: List[int] = []
        for i in range(len(bx)):
            if supp[i]:
                continue
            keep.append(i)
            if len(keep) >= K:
                break
            w = np.maximum(f32(0), (np.minimum(x2[i], x2[i + 1:]) - np.maximum(x1[i], x1[i + 1:]) + f32(1)).astype(f32))
            h = np.maximum(f32(0), (np.minimum(y2[i], y2[i + 1:]) - np.maximum(y1[i], y1[i + 1:]) + f32(1)).astype(f32))
            inter = (w * h).astype(f32)
            ovr = (inter / ((area[i] + area[i + 1:]).astype(f32) - inter).astype(f32)).astype(f32)
            supp[i + 1:] |= ovr > f32(nms_iou)
        n = len(keep)
        out_b[b, :n], out_k[b, :n], out_s[b, :n], out_c[b] = bx[keep], kp[keep], scv[keep], n
    return out_b, out_k, out_s, out_c
