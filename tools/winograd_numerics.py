"""CPU experiment (no GPU): what would Winograd F(2x2,3x3) with fp16 transformed operands cost in accuracy?

Emulates the embedder PROGRAM the device runs (BatchNorms folded by weights.fold_layer, fp16 weights, fp16 storage
between layers, fp32 accumulate) twice: every 3x3 stride-1 conv as the direct sum (what the HIP kernels compute up to
summation order) and as Winograd F(2x2,3x3) with V = B^T d B evaluated in packed-fp16 steps (one rounding per add),
U = G g G^T evaluated in double from the folded weights and rounded ONCE to fp16, products accumulated in fp32 and the
output transform A^T M A in fp32.  Prints the cosine of both against the fp32 oracle network.

    python tools/winograd_numerics.py [--blocks 3,13,30,3] [--chips 4] [--min-cin 128] [--v32]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import frp_amd_loader  # noqa: E402,F401
from frp_amd import netspec as ns, weights as wts  # noqa: E402
from oracle import network as onet  # noqa: E402

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def _bt_half(d, dim):
    """B^T along `dim` (size 4) with one fp16 rounding per add (v_pk_add_f16)."""
    d0, d1, d2, d3 = d.unbind(dim)
    return torch.stack([(d0 - d2), (d1 + d2), (d2 - d1), (d1 - d3)], dim)


def winograd_conv(x16: torch.Tensor, w64: np.ndarray, v32: bool) -> torch.Tensor:
    """x16 [N,C,H,W] float32 tensor holding fp16 values; w64 [Cout,3,3,Cin] float64 folded weights -> [N,Cout,H,W] f32"""
    N, C, H, W = x16.shape
    He, We = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    xp = F.pad(x16, (1, 1 + We - W, 1, 1 + He - H))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # [N,C,th,tw,4,4]
    if v32:
        Bt = torch.from_numpy(BT).float()
        V = torch.einsum("ij,nctujk,lk->nctuil", Bt, d, Bt).half().float()
    else:
        V = _bt_half(_bt_half(d.half(), 4), 5).float()
    U = np.einsum("ij,ojkc,lk->oilc", G, w64, G).astype(np.float16).astype(np.float32)   # [Cout,4,4,Cin]
    M = torch.einsum("oilc,nctuil->notuil", torch.from_numpy(U), V)
    At = torch.from_numpy(AT).float()
    Y = torch.einsum("ai,notuil,bl->notaub", At, M, At)           # [N,O,th,2,tw,2]
    return Y.reshape(N, -1, He, We)[:, :, :H, :W]


def run_program(raw, layers, x16, wino_pred, v32):
    tens = {layers[0].src: x16}
    for l in layers:
        w16, bias, slope, w64 = wts.fold_layer(raw, l, return_w64=True)
        x = tens[l.src]
        if l.flags & ns.FLAG_FLATTEN:
            x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1, 1, 1)
        if wino_pred(l):
            y = winograd_conv(x, w64, v32)
        else:
            y = F.conv2d(x, torch.from_numpy(w16.astype(np.float32)).permute(0, 3, 1, 2), None, stride=l.stride, padding=l.k // 2)
        _, _, Ho, Wo = y.shape
        if l.flags & ns.FLAG_BORDER_BIAS:
            cy = np.where(np.arange(Ho) == 0, 0, np.where(np.arange(Ho) == Ho - 1, 2, 1))
            cx = np.where(np.arange(Wo) == 0, 0, np.where(np.arange(Wo) == Wo - 1, 2, 1))
            b = torch.from_numpy(bias[cy[:, None] * 3 + cx[None, :]]).permute(2, 0, 1)[None]
        else:
            b = torch.from_numpy(bias)[None, :, None, None]
        y = y + b
        if l.res:
            y = y + tens[l.res]
        if l.act == ns.ACT_RELU:
            y = torch.relu(y)
        elif l.act == ns.ACT_PRELU:
            y = torch.where(y > 0, y, y * torch.from_numpy(slope)[None, :, None, None])
        tens[l.dst] = y if (l.flags & ns.FLAG_OUT_F32) else y.half().float()
    return tens[layers[-1].dst]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", default="3,13,30,3")
    ap.add_argument("--chips", type=int, default=4)
    ap.add_argument("--min-cin", type=int, default=128)
    ap.add_argument("--v32", action="store_true", help="input transform in fp32, one rounding to fp16")
    ap.add_argument("--seed", type=int, default=7)
    a = ap.parse_args()
    blocks = tuple(int(v) for v in a.blocks.split(","))
    raw = wts.make_synthetic_raw(a.seed, (1, 1, 1, 1), blocks, want_det=False)
    layers = ns.iresnet_layers(blocks)
    rng = np.random.default_rng(9)
    chips = rng.integers(0, 256, size=(a.chips, 112, 112, 3), dtype=np.uint8)
    blob = onet.emb_blob(chips)
    ref = onet.emb_forward(raw, blob)
    x16 = torch.zeros(a.chips, ns.EMB_IN_CH, 112, 112)
    x16[:, :3] = blob.half().float()
    elig = lambda l: l.k == 3 and l.stride == 1 and l.cin >= a.min_cin and l.cin_real is None  # noqa: E731

    def norm(e):
        e = e.reshape(a.chips, -1).numpy()
        return e / np.linalg.norm(e, axis=1, keepdims=True)

    e_dir = norm(run_program(raw, layers, x16, lambda l: False, a.v32))
    e_win = norm(run_program(raw, layers, x16, elig, a.v32))
    n_w = sum(1 for l in layers if elig(l))
    print(f"blocks {blocks}: {n_w} of {len(layers)} convs as Winograd (cin >= {a.min_cin}), V in {'fp32' if a.v32 else 'packed fp16'}")
    print("1 - cos(direct fp16 program, fp32 oracle):", (1 - (e_dir * ref).sum(1)))
    print("1 - cos(winograd fp16 program, fp32 oracle):", (1 - (e_win * ref).sum(1)))
    print("1 - cos(winograd, direct):", (1 - (e_win * e_dir).sum(1)))


if __name__ == "__main__":
    main()
