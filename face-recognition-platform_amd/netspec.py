"""Layer tables of the two networks on the hot path.

The reference delegates both to un-vendored packages (SURVEY.md §0.2, §8a rows
a2/a4: `insightface.app.FaceAnalysis` at backend/app/utils/deepfake_utils.py:39-51
is the only place the RetinaFace/ArcFace family appears); the tables below are
this repo's committed definition of that family:

* detector  "FRPDet": RetinaFace/SCRFD-style, strides {8,16,32}, 2 anchors per
  location, per anchor 1 score logit + 4 distances + 10 landmark offsets
  (15 values, 30 of the 32 head channels used).
* embedder  IResNet (ArcFace): stem conv3x3(3->64)+BN+PReLU, stages of
  IBasicBlock (BN -> conv3x3 -> BN -> PReLU -> conv3x3(stride) -> BN, 1x1-conv+BN
  shortcut on the strided block), BN2d -> flatten -> FC 512 -> BN1d.  R100 =
  blocks (3,13,30,3) = 12,089.6 MMAC per 112x112 face.

Each entry is a conv op in execution order; the packer (weights.py) folds the
BatchNorms into it and the native runtime (csrc/) executes the list.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

DET_STRIDES = (8, 16, 32)
DET_NUM_ANCHORS = 2
DET_VALUES_PER_ANCHOR = 15          # score, l, t, r, b, 5 x (dx, dy)
DET_HEAD_CH = 32                    # 30 used, padded to an MFMA-friendly width
DET_FPN_CH = 128
DET_IN_CH = 8                       # RGB padded to 8 channels (16-byte pixels)
EMB_IN_CH = 8
EMB_SIZE = 112
EMB_DIM = 512

ACT_NONE, ACT_RELU, ACT_PRELU = 0, 1, 2
FLAG_BORDER_BIAS = 1   # bias has 9 position classes (pre-conv BN shift folded exactly)
FLAG_OUT_F32 = 2       # output fp32 instead of fp16
FLAG_RES_UP2 = 4       # residual operand is read at (y>>1, x>>1): FPN top-down add
FLAG_FLATTEN = 8       # view the input [H,W,C] as [1,1,H*W*C] (FC as a 1x1 conv)


@dataclass
class ConvLayer:
    name: str                    # raw-weight prefix of the conv ("...conv1")
    src: str                     # logical input tensor name
    dst: str                     # logical output tensor name
    cin: int
    cout: int
    k: int = 3
    stride: int = 1
    act: int = ACT_NONE
    res: Optional[str] = None    # logical residual tensor added before activation
    flags: int = 0
    pre_bn: Optional[str] = None   # BN applied to the conv INPUT (IResNet bn1 / bn2-before-fc)
    post_bn: Optional[str] = None  # BN applied to the conv OUTPUT
    conv_bias: bool = False        # conv has its own bias
    prelu: Optional[str] = None    # raw name of the PReLU slope
    cin_real: Optional[int] = None   # real input channels when cin is padded (stem: 3)
    cout_real: Optional[int] = None  # real output channels when cout is padded (head: 30)

    def macs_per_out_pixel(self) -> int:
        return (self.cin_real or self.cin) * (self.cout_real or self.cout) * self.k * self.k


def detector_layers(blocks: Tuple[int, int, int, int] = (1, 2, 2, 2)) -> List[ConvLayer]:
    L: List[ConvLayer] = []
    L.append(ConvLayer("det.stem1.conv", "det.in", "det.s1", DET_IN_CH, 32, 3, 2, ACT_RELU,
                       post_bn="det.stem1.bn", cin_real=3))
    L.append(ConvLayer("det.stem2.conv", "det.s1", "det.s2", 32, 64, 3, 2, ACT_RELU, post_bn="det.stem2.bn"))
    widths = (64, 128, 256, 256)
    x, cin = "det.s2", 64
    feats = {}
    for li, (nb, w) in enumerate(zip(blocks, widths), start=1):
        for bi in range(nb):
            p = f"det.layer{li}.{bi}"
            stride = 2 if (bi == 0 and li > 1) else 1
            ident = x
            if stride != 1 or cin != w:
                L.append(ConvLayer(f"{p}.downsample.0", x, f"{p}.ds", cin, w, 1, stride, ACT_NONE,
                                   post_bn=f"{p}.downsample.1"))
                ident = f"{p}.ds"
            L.append(ConvLayer(f"{p}.conv1", x, f"{p}.t", cin, w, 3, stride, ACT_RELU, post_bn=f"{p}.bn1"))
            L.append(ConvLayer(f"{p}.conv2", f"{p}.t", f"{p}.out", w, w, 3, 1, ACT_RELU, res=ident,
                               post_bn=f"{p}.bn2"))
            x, cin = f"{p}.out", w
        feats[li] = (x, w)
    F = DET_FPN_CH
    # FPN: lateral 1x1 (+ nearest-2x top-down add fused as the residual operand), 3x3 smooth
    L.append(ConvLayer("det.fpn.lat5.conv", feats[4][0], "det.p5", feats[4][1], F, 1, 1, ACT_NONE, conv_bias=True))
    L.append(ConvLayer("det.fpn.lat4.conv", feats[3][0], "det.p4", feats[3][1], F, 1, 1, ACT_NONE, conv_bias=True,
                       res="det.p5", flags=FLAG_RES_UP2))
    L.append(ConvLayer("det.fpn.lat3.conv", feats[2][0], "det.p3", feats[2][1], F, 1, 1, ACT_NONE, conv_bias=True,
                       res="det.p4", flags=FLAG_RES_UP2))
    for lv in (3, 4, 5):
        L.append(ConvLayer(f"det.fpn.smooth{lv}.conv", f"det.p{lv}", f"det.f{lv}", F, F, 3, 1, ACT_RELU,
                           post_bn=f"det.fpn.smooth{lv}.bn"))
    for lv in (3, 4, 5):
        h = f"det.head{lv}"
        L.append(ConvLayer(f"{h}.tower0.conv", f"det.f{lv}", f"{h}.t0", F, F, 3, 1, ACT_RELU, post_bn=f"{h}.tower0.bn"))
        L.append(ConvLayer(f"{h}.tower1.conv", f"{h}.t0", f"{h}.t1", F, F, 3, 1, ACT_RELU, post_bn=f"{h}.tower1.bn"))
        L.append(ConvLayer(f"{h}.out", f"{h}.t1", f"det.out{lv}", F, DET_HEAD_CH, 3, 1, ACT_NONE, conv_bias=True,
                           cout_real=DET_NUM_ANCHORS * DET_VALUES_PER_ANCHOR))
    return L


def iresnet_layers(blocks: Tuple[int, int, int, int] = (3, 13, 30, 3)) -> List[ConvLayer]:
    L: List[ConvLayer] = []
    L.append(ConvLayer("emb.conv1", "emb.in", "emb.x0", EMB_IN_CH, 64, 3, 1, ACT_PRELU,
                       post_bn="emb.bn1", prelu="emb.prelu", cin_real=3))
    widths = (64, 128, 256, 512)
    x, cin = "emb.x0", 64
    for li, (nb, w) in enumerate(zip(blocks, widths), start=1):
        for bi in range(nb):
            p = f"emb.layer{li}.{bi}"
            stride = 2 if bi == 0 else 1
            ident = x
            if bi == 0:
                L.append(ConvLayer(f"{p}.downsample.0", x, f"{p}.ds", cin, w, 1, stride, ACT_NONE,
                                   post_bn=f"{p}.downsample.1"))
                ident = f"{p}.ds"
            L.append(ConvLayer(f"{p}.conv1", x, f"{p}.t", cin, w, 3, 1, ACT_PRELU, flags=FLAG_BORDER_BIAS,
                               pre_bn=f"{p}.bn1", post_bn=f"{p}.bn2", prelu=f"{p}.prelu"))
            L.append(ConvLayer(f"{p}.conv2", f"{p}.t", f"{p}.out", w, w, 3, stride, ACT_NONE, res=ident,
                               post_bn=f"{p}.bn3"))
            x, cin = f"{p}.out", w
    L.append(ConvLayer("emb.fc", x, "emb.out", 512 * 7 * 7, EMB_DIM, 1, 1, ACT_NONE,
                       flags=FLAG_OUT_F32 | FLAG_FLATTEN, pre_bn="emb.bn2", post_bn="emb.features", conv_bias=True))
    return L


def out_hw(h: int, w: int, k: int, stride: int) -> Tuple[int, int]:
    pad = k // 2
    return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1


def layer_macs(layers: List[ConvLayer], in_h: int, in_w: int, in_name: str) -> Tuple[int, List[Tuple[str, int]]]:
    """Algorithmic multiply-accumulates of a layer list at one input resolution
    (2*Cin*Cout*kh*kw*Hout*Wout summed = 2 x this; BN/act/pool excluded, padded
    channels not counted)."""
    dims = {in_name: (in_h, in_w)}
    total, per = 0, []
    for l in layers:
        h, w = dims[l.src]
        if l.flags & FLAG_FLATTEN:
            oh, ow = 1, 1
            m = (l.cin_real or l.cin) * (l.cout_real or l.cout)
        else:
            oh, ow = out_hw(h, w, l.k, l.stride)
            m = l.macs_per_out_pixel() * oh * ow
        dims[l.dst] = (oh, ow)
        total += m
        per.append((l.name, m))
    return total, per


def num_anchors(h: int, w: int) -> int:
    return sum((h // s) * (w // s) * DET_NUM_ANCHORS for s in DET_STRIDES)
