"""ORACLE (test infrastructure, never shipped, never on the product path).

fp32 CPU restatement of the network half of the hot path, written from the
PUBLIC definitions of the model family the north star names -- NOT from the
product's layer tables (it walks the raw parameter dict by name), so it checks
the product's tables, BatchNorm folding, layouts and kernels independently.

PARITY UNPINNED: the reference holds no test, golden vector or fixture for any
stage below, and its implementations live in third-party packages that are
absent here and not vendored under /root/reference:
  * insightface==0.7.3 + onnxruntime==1.19.2 (backend/requirements.txt:41-43),
    only call site backend/app/utils/deepfake_utils.py:39-51,138
    (`FaceAnalysis(...).prepare(ctx_id=0, det_size=(640,640))`, `.get(frame_bgr)`);
  * face-recognition==1.3.0 / dlib==19.24.6 (requirements.txt:37-39), call sites
    backend/app/services/face_service.py:139,156,179 and routes/camera.py:232,237.
What is restated (published algorithms of those packages):
  * SCRFD/RetinaFace input blob: letterbox top-left into a zero canvas,
    BGR->RGB, (x - 127.5) / 128                      [insightface scrfd.py detect()/forward()]
  * anchor-free distance decode: centre = (x*stride, y*stride), 2 anchors per
    location, box = centre -/+ d*stride, kps = centre + off*stride, score
    threshold 0.5, greedy NMS IoU 0.4 with +1 pixel areas [scrfd.py distance2bbox/
    distance2kps/nms]
  * 5-point alignment: Umeyama similarity to the ArcFace 112x112 template,
    inverse bilinear warp, border 0          [insightface utils/face_align.py norm_crop]
  * ArcFace input blob (rgb - 127.5) / 127.5, IResNet, L2-normalised 512-d
    embedding, cosine similarity               [arcface_onnx.py, arcface_torch iresnet.py]
The oracle therefore *defines* these stages for this repo; its
own outputs on seeded inputs are committed under tests/golden/ as regression pins.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
ARCFACE_TEMPLATE = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366],
                             [41.5493, 92.3655], [70.7299, 92.2041]], dtype=np.float32)
STRIDES = (8, 16, 32)
NUM_ANCHORS = 2
PRE_NMS_CAP = 1024


def _t(raw, name) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(raw[name])).float()


def _bn(raw, name, x):
    return F.batch_norm(x, _t(raw, name + ".running_mean"), _t(raw, name + ".running_var"),
                        _t(raw, name + ".weight"), _t(raw, name + ".bias"), False, 0.0, BN_EPS)


def _conv(raw, name, x, stride=1, bias=False):
    w = _t(raw, name + ".weight")
    b = _t(raw, name + ".bias") if bias else None
    return F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2)


# ----------------------------------------------------------------------------- pre-process
def det_blob(frames_bgr: np.ndarray, canvas_hw: Tuple[int, int]) -> torch.Tensor:
    """[B,H,W,3] u8 BGR -> [B,3,Hc,Wc] f32: top-left letterbox into a zero u8 canvas,
    swap to RGB, (x-127.5)/128."""
    B, H, W, _ = frames_bgr.shape
    Hc, Wc = canvas_hw
    canvas = np.zeros((B, Hc, Wc, 3), dtype=np.uint8)
    canvas[:, :H, :W] = frames_bgr
    rgb = canvas[..., ::-1].astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray((rgb - 127.5) / 128.0)).permute(0, 3, 1, 2).contiguous()


# ----------------------------------------------------------------------------- detector
def _count_blocks(raw, prefix) -> int:
    n = 0
    while f"{prefix}.{n}.conv1.weight" in raw:
        n += 1
    return n


def det_forward(raw: Dict[str, np.ndarray], x: torch.Tensor) -> List[np.ndarray]:
    """-> per stride [B, H_l, W_l, 30] f32 head maps (channel = anchor*15 + value)."""
    with torch.no_grad():
        x = F.relu(_bn(raw, "det.stem1.bn", _conv(raw, "det.stem1.conv", x, 2)))
        x = F.relu(_bn(raw, "det.stem2.bn", _conv(raw, "det.stem2.conv", x, 2)))
        feats = {}
        for li in (1, 2, 3, 4):
            for bi in range(_count_blocks(raw, f"det.layer{li}")):
                p = f"det.layer{li}.{bi}"
                stride = 2 if (bi == 0 and li > 1) else 1
                ident = x
                if f"{p}.downsample.0.weight" in raw:
                    ident = _bn(raw, f"{p}.downsample.1", _conv(raw, f"{p}.downsample.0", x, stride))
                t = F.relu(_bn(raw, f"{p}.bn1", _conv(raw, f"{p}.conv1", x, stride)))
                t = _bn(raw, f"{p}.bn2", _conv(raw, f"{p}.conv2", t, 1))
                x = F.relu(t + ident)
            feats[li] = x
        p5 = _conv(raw, "det.fpn.lat5.conv", feats[4], 1, True)
        p4 = _conv(raw, "det.fpn.lat4.conv", feats[3], 1, True) + F.interpolate(p5, scale_factor=2, mode="nearest")
        p3 = _conv(raw, "det.fpn.lat3.conv", feats[2], 1, True) + F.interpolate(p4, scale_factor=2, mode="nearest")
        outs = []
        for lv, p in ((3, p3), (4, p4), (5, p5)):
            f = F.relu(_bn(raw, f"det.fpn.smooth{lv}.bn", _conv(raw, f"det.fpn.smooth{lv}.conv", p)))
            h = f"det.head{lv}"
            f = F.relu(_bn(raw, f"{h}.tower0.bn", _conv(raw, f"{h}.tower0.conv", f)))
            f = F.relu(_bn(raw, f"{h}.tower1.bn", _conv(raw, f"{h}.tower1.conv", f)))
            o = _conv(raw, f"{h}.out", f, 1, True)
            outs.append(o.permute(0, 2, 3, 1).contiguous().numpy())
        return outs


def logit_threshold(score_thresh: float) -> np.float32:
    """score >= t  <=>  logit >= log(t/(1-t)); evaluated in double, rounded to f32."""
    if score_thresh <= 0.0:
        return np.float32(-np.inf)
    if score_thresh >= 1.0:
        return np.float32(np.inf)
    return np.float32(math.log(score_thresh / (1.0 - score_thresh)))


def decode_nms(head_maps: Sequence[np.ndarray], score_thresh: float, nms_iou: float, max_faces: int,
               cap: int = PRE_NMS_CAP):
    """head_maps: per stride [H_l, W_l, >=30] (ONE frame; values as produced by the
    detector, f32 or f16).  Candidates = anchors with logit >= logit(score_thresh);
    if more than `cap`, the `cap` highest (logit desc, anchor index asc).  Greedy NMS
    in that order (suppress IoU > nms_iou, +1 areas), first `max_faces` kept.
    All arithmetic in float32, one rounding per operation (no fused multiply-add).
    -> boxes [K,4] (x1,y1,x2,y2), kps [K,5,2], scores [K], anchor_idx [K]"""
    f32 = np.float32
    logits, vals, cxs, cys, strs = [], [], [], [], []
    for hm, s in zip(head_maps, STRIDES):
        H, W = hm.shape[:2]
        v = hm[..., :NUM_ANCHORS * 15].astype(np.float32).reshape(H, W, NUM_ANCHORS, 15)
        ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        cx = np.repeat((xs * s).astype(np.float32)[..., None], NUM_ANCHORS, axis=2)
        cy = np.repeat((ys * s).astype(np.float32)[..., None], NUM_ANCHORS, axis=2)
        logits.append(v[..., 0].reshape(-1))
        vals.append(v.reshape(-1, 15))
        cxs.append(cx.reshape(-1))
        cys.append(cy.reshape(-1))
        strs.append(np.full(H * W * NUM_ANCHORS, s, dtype=np.float32))
    logit = np.concatenate(logits)
    vals = np.concatenate(vals)
    cx = np.concatenate(cxs)
    cy = np.concatenate(cys)
    st = np.concatenate(strs)
    lt = logit_threshold(score_thresh)
    # forced top-K mode (threshold <= 0, i.e. -inf): every anchor is a candidate, NaN logits included - they sort last
    cand = np.arange(len(logit)) if np.isneginf(lt) else np.nonzero(logit >= lt)[0]
    order = cand[np.argsort(-logit[cand], kind="stable")]  # desc logit (NaN last), ties by anchor index asc
    order = order[:cap]
    v = vals[order]
    s = st[order]
    x1 = (cx[order] - v[:, 1] * s).astype(f32)
    y1 = (cy[order] - v[:, 2] * s).astype(f32)
    x2 = (cx[order] + v[:, 3] * s).astype(f32)
    y2 = (cy[order] + v[:, 4] * s).astype(f32)
    kps = np.stack([(cx[order][:, None] + v[:, 5::2] * s[:, None]).astype(f32),
                    (cy[order][:, None] + v[:, 6::2] * s[:, None]).astype(f32)], axis=-1)
    score = (f32(1.0) / (f32(1.0) + np.exp(-logit[order].astype(f32)))).astype(f32)
    area = ((x2 - x1 + f32(1.0)) * (y2 - y1 + f32(1.0))).astype(f32)
    n = len(order)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if len(keep) >= max_faces:
            break
        xx1 = np.maximum(x1[i], x1[i + 1:])
        yy1 = np.maximum(y1[i], y1[i + 1:])
        xx2 = np.minimum(x2[i], x2[i + 1:])
        yy2 = np.minimum(y2[i], y2[i + 1:])
        w = np.maximum(f32(0.0), (xx2 - xx1 + f32(1.0)).astype(f32))
        h = np.maximum(f32(0.0), (yy2 - yy1 + f32(1.0)).astype(f32))
        inter = (w * h).astype(f32)
        denom = ((area[i] + area[i + 1:]).astype(f32) - inter).astype(f32)
        ovr = (inter / denom).astype(f32)
        suppressed[i + 1:] |= ovr > f32(nms_iou)
    keep = np.array(keep, dtype=np.int64)
    boxes = np.stack([x1, y1, x2, y2], axis=1)[keep] if len(keep) else np.zeros((0, 4), f32)
    return boxes.astype(f32), kps[keep].astype(f32), score[keep], order[keep].astype(np.int64)


# ----------------------------------------------------------------------------- alignment
def umeyama_similarity(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """Least-squares similarity (Umeyama 1991, with reflection guard) src->dst, float64.
    -> 2x3 matrix M with dst ~= M @ [src, 1]."""
    src = src.astype(np.float64)
    dst = dst.astype(np.float64)
    n = src.shape[0]
    mu_s, mu_d = src.mean(0), dst.mean(0)
    sc, dc = src - mu_s, dst - mu_d
    A = dc.T @ sc / n
    d = np.ones(2)
    if np.linalg.det(A) < 0:
        d[1] = -1
    U, S, Vt = np.linalg.svd(A)
    R = U @ np.diag(d) @ Vt
    var_s = sc.var(axis=0).sum()
    scale = (S * d).sum() / var_s
    M = np.zeros((2, 3))
    M[:, :2] = scale * R
    M[:, 2] = mu_d - scale * (R @ mu_s)
    return M


def warp_affine_bilinear(img: np.ndarray, M: np.ndarray, size: int = 112) -> np.ndarray:
    """img [H,W,C] u8; M 2x3 src->dst.  For each output pixel centre (u,v) the source
    point is M^-1 (u,v); float bilinear over the 4 neighbours, taps outside the image
    contribute 0 (constant border 0).  -> [size,size,C] f32 (no rounding)."""
    H, W, C = img.shape
    A = np.vstack([M, [0, 0, 1]])
    Ai = np.linalg.inv(A)[:2]
    v, u = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64), indexing="ij")
    sx = Ai[0, 0] * u + Ai[0, 1] * v + Ai[0, 2]
    sy = Ai[1, 0] * u + Ai[1, 1] * v + Ai[1, 2]
    x0 = np.floor(sx).astype(np.int64)
    y0 = np.floor(sy).astype(np.int64)
    fx = sx - x0
    fy = sy - y0
    out = np.zeros((size, size, C), dtype=np.float64)
    imgf = img.astype(np.float64)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xx, yy = x0 + dx, y0 + dy
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            xc, yc = np.clip(xx, 0, W - 1), np.clip(yy, 0, H - 1)
            out += (imgf[yc, xc] * ok[..., None]) * (wy * wx)[..., None]
    return out.astype(np.float32)


def align_faces(frame_bgr: np.ndarray, kps: np.ndarray) -> np.ndarray:
    """frame [H,W,3] u8 BGR, kps [K,5,2] -> aligned chips [K,112,112,3] f32 BGR (0..255)."""
    chips = [warp_affine_bilinear(frame_bgr, umeyama_similarity(k, ARCFACE_TEMPLATE)) for k in kps]
    return np.stack(chips) if chips else np.zeros((0, 112, 112, 3), np.float32)


# ----------------------------------------------------------------------------- embedder
def emb_blob(chips_bgr: np.ndarray) -> torch.Tensor:
    """[K,112,112,3] BGR (u8 or f32, 0..255) -> [K,3,112,112] f32 RGB (x-127.5)/127.5."""
    rgb = chips_bgr[..., ::-1].astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray((rgb - 127.5) / 127.5)).permute(0, 3, 1, 2).contiguous()


def emb_forward(raw: Dict[str, np.ndarray], x: torch.Tensor, normalize: bool = True) -> np.ndarray:
    """IResNet forward (arcface_torch iresnet.py IResNet/IBasicBlock), -> [K,512] f32."""
    with torch.no_grad():
        x = F.prelu(_bn(raw, "emb.bn1", _conv(raw, "emb.conv1", x, 1)), _t(raw, "emb.prelu.weight"))
        for li in (1, 2, 3, 4):
            for bi in range(_count_blocks(raw, f"emb.layer{li}")):
                p = f"emb.layer{li}.{bi}"
                stride = 2 if bi == 0 else 1
                ident = x
                if f"{p}.downsample.0.weight" in raw:
                    ident = _bn(raw, f"{p}.downsample.1", _conv(raw, f"{p}.downsample.0", x, stride))
                t = _bn(raw, f"{p}.bn1", x)
                t = _conv(raw, f"{p}.conv1", t, 1)
                t = F.prelu(_bn(raw, f"{p}.bn2", t), _t(raw, f"{p}.prelu.weight"))
                t = _bn(raw, f"{p}.bn3", _conv(raw, f"{p}.conv2", t, stride))
                x = t + ident
        x = _bn(raw, "emb.bn2", x)
        x = torch.flatten(x, 1)
        x = F.linear(x, _t(raw, "emb.fc.weight"), _t(raw, "emb.fc.bias"))
        x = F.batch_norm(x, _t(raw, "emb.features.running_mean"), _t(raw, "emb.features.running_var"),
                         _t(raw, "emb.features.weight"), _t(raw, "emb.features.bias"), False, 0.0, BN_EPS)
        e = x.numpy()
    if normalize:
        e = e / np.linalg.norm(e, axis=1, keepdims=True)
    return e.astype(np.float32)


# ----------------------------------------------------------------------------- match
def match_topk(gallery: np.ndarray, queries: np.ndarray, k: int = 1):
    """cosine scores of unit rows; -> (idx [M,k] by cosine desc, ties lower index; cos [M,k]) f64 math."""
    S = queries.astype(np.float64) @ gallery.astype(np.float64).T
    idx = np.argsort(-S, axis=1, kind="stable")[:, :k]
    return idx, np.take_along_axis(S, idx, axis=1)


def cos_to_distance(cos):
    """Euclidean distance between unit vectors: d^2 = 2 - 2cos (keeps the reference's
    `distance` field and its 0.4/0.6 buckets, face_service.py:410-411,486-492)."""
    return np.sqrt(np.maximum(0.0, 2.0 - 2.0 * np.asarray(cos, dtype=np.float64)))


# ----------------------------------------------------------------------------- multi-scale pyramid
def resize_bilinear_u8(img: np.ndarray, out_hw: Tuple[int, int]) -> np.ndarray:
    """[H,W,C] u8 -> [Hs,Ws,C] u8.  Pixel centres src = (dst+0.5)*(S/D) - 0.5 clamped to the image,
    fp32 lerp in x then y (one rounding per operation), floor(v+0.5).  (cv2.resize INTER_LINEAR
    geometry; cv2's own fixed-point rounding is not reproduced -- the oracle defines the rule.)"""
    f32 = np.float32
    H, W = img.shape[:2]
    Hs, Ws = out_hw
    ry, rx = f32(H) / f32(Hs), f32(W) / f32(Ws)
    sy = np.clip(((np.arange(Hs, dtype=f32) + f32(0.5)) * ry - f32(0.5)).astype(f32), f32(0), f32(H - 1))
    sx = np.clip(((np.arange(Ws, dtype=f32) + f32(0.5)) * rx - f32(0.5)).astype(f32), f32(0), f32(W - 1))
    y0, x0 = sy.astype(np.int64), sx.astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
    fy, fx = (sy - y0.astype(f32)).astype(f32)[:, None, None], (sx - x0.astype(f32)).astype(f32)[None, :, None]
    im = img.astype(f32)
    top = ((im[y0][:, x0] * (f32(1) - fx)).astype(f32) + (im[y0][:, x1] * fx).astype(f32)).astype(f32)
    bot = ((im[y1][:, x0] * (f32(1) - fx)).astype(f32) + (im[y1][:, x1] * fx).astype(f32)).astype(f32)
    v = ((top * (f32(1) - fy)).astype(f32) + (bot * fy).astype(f32)).astype(f32)
    return np.clip(np.floor(v + f32(0.5)), 0, 255).astype(np.uint8)


def detect_pyramid(raw, frame_bgr: np.ndarray, scales=(1.0, 0.5, 0.25), score_thresh=0.5, nms_iou=0.4, max_faces=10,
                   per_scale=64, head_maps_per_scale=None):
    """One frame.  Per scale: resize, detector, decode + NMS keeping `per_scale` boxes; map back to frame
    pixels (x * W/Ws, y * H/Hs); concatenate in scale order; stable sort by score desc; greedy +1-area NMS;
    first max_faces.  `head_maps_per_scale` (optional) substitutes the detector outputs (the GPU's own head
    maps) so that the merge logic can be checked bit for bit.  -> boxes, kps, scores"""
    f32 = np.float32
    H, W = frame_bgr.shape[:2]
    allb, allk, alls = [], [], []
    for si, s in enumerate(scales):
        Hs, Ws = max(1, int(round(H * s))), max(1, int(round(W * s)))
        if head_maps_per_scale is not None:
            maps = head_maps_per_scale[si]
        else:
            img = frame_bgr if (Hs, Ws) == (H, W) else resize_bilinear_u8(frame_bgr, (Hs, Ws))
            canvas = ((Hs + 31) // 32 * 32, (Ws + 31) // 32 * 32)
            maps = [m[0] for m in det_forward(raw, det_blob(img[None], canvas))]
        b, k, sc, _ = decode_nms(maps, score_thresh, nms_iou, per_scale)
        ry, rx = f32(H) / f32(Hs), f32(W) / f32(Ws)
        allb.append((b * np.array([rx, ry, rx, ry], f32)).astype(f32))
        allk.append((k * np.array([rx, ry], f32)).astype(f32))
        alls.append(sc.astype(f32))
    b, k, sc = np.concatenate(allb), np.concatenate(allk), np.concatenate(alls)
    order = np.argsort(-sc, kind="stable")
    b, k, sc = b[order], k[order], sc[order]
    keep, supp = [], np.zeros(len(b), bool)
    area = ((b[:, 2] - b[:, 0] + f32(1)) * (b[:, 3] - b[:, 1] + f32(1))).astype(f32)
    for i in range(len(b)):
        if supp[i]:
            continue
        keep.append(i)
        if len(keep) >= max_faces:
            break
        for j in range(i + 1, len(b)):
            w = max(f32(0), f32(min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0]) + f32(1)))
            h = max(f32(0), f32(min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1]) + f32(1)))
            inter = f32(w * h)
            if f32(inter / f32(f32(area[i] + area[j]) - inter)) > f32(nms_iou):
                supp[j] = True
    return b[keep], k[keep], sc[keep]


# ----------------------------------------------------------------------------- end to end
def process_frames(raw, frames_bgr: np.ndarray, gallery: Optional[np.ndarray], canvas_hw=None,
                   score_thresh=0.5, nms_iou=0.4, max_faces=10):
    B, H, W, _ = frames_bgr.shape
    if canvas_hw is None:
        canvas_hw = ((H + 31) // 32 * 32, (W + 31) // 32 * 32)
    maps = det_forward(raw, det_blob(frames_bgr, canvas_hw))
    out = []
    for b in range(B):
        boxes, kps, scores, aidx = decode_nms([m[b] for m in maps], score_thresh, nms_iou, max_faces)
        chips = align_faces(frames_bgr[b], kps)
        emb = emb_forward(raw, emb_blob(chips)) if len(chips) else np.zeros((0, 512), np.float32)
        r = {"boxes": boxes, "kps": kps, "scores": scores, "anchor_idx": aidx, "chips": chips, "emb": emb}
        if gallery is not None and len(emb):
            idx, cos = match_topk(gallery, emb, 1)
            r["match_idx"], r["match_cos"] = idx[:, 0], cos[:, 0]
        out.append(r)
    return out
