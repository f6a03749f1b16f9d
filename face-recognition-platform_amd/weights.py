"""Raw weights -> folded fp16 program blob for the native runtime.

* `make_synthetic_raw(...)`: seeded synthetic raw parameters (no pretrained
  packs exist offline).  Names follow the public PyTorch
  conventions (`conv.weight` OIHW, `bn.{weight,bias,running_mean,running_var}`,
  IResNet names as in the public arcface_torch `iresnet` state_dict), so a
  loader for real checkpoints only has to produce the same dict.
* `pack_blob(raw, ...)`: folds every BatchNorm into its conv in fp32 (post-conv
  BN: scale into the weights + bias; pre-conv BN (IResNet bn1): scale into the
  weights per input channel, shift folded *exactly* as 9 border-class bias
  vectors because zero padding removes taps at the image border), casts to
  fp16 [Cout][kh][kw][Cin], assigns activation buffers by liveness and writes
  the blob `frp_load_weights` consumes (layout: include/frp_blob.h).
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, List, Tuple

import numpy as np

from . import netspec as ns

BN_EPS = 1e-5
BLOB_MAGIC = b"FRPBLOB1"
BLOB_VERSION = 2
HEADER_FMT = "<8sII" + "IIII" + "III" + "I" + "IIII" + "IIII" + "QQQQ" + "QQ"
HEADER_BYTES = struct.calcsize(HEADER_FMT)
OP_FMT = "<iiiiiiiiiiqqqiffi"
OP_FIELDS = ["in_buf", "out_buf", "res_buf", "cin", "cout", "ksize", "stride", "act", "flags", "real_ch", "w_off", "bias_off",
             "slope_off", "out2_buf", "in_scale", "out_scale", "reserved"]
OP_BYTES = struct.calcsize(OP_FMT)
assert HEADER_BYTES == 128 and OP_BYTES == 80


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def _bn(raw: Dict[str, np.ndarray], seed: int, name: str, c: int, gamma_scale: float = 1.0):
    r = _rng(seed, name)
    raw[name + ".weight"] = (r.uniform(0.9, 1.1, c) * gamma_scale).astype(np.float32)
    raw[name + ".bias"] = (r.standard_normal(c) * 0.05).astype(np.float32)
    raw[name + ".running_mean"] = (r.standard_normal(c) * 0.05).astype(np.float32)
    raw[name + ".running_var"] = r.uniform(0.9, 1.1, c).astype(np.float32)


def make_synthetic_raw(seed: int = 7, det_blocks=(1, 2, 2, 2), emb_blocks=(3, 13, 30, 3),
                       want_det: bool = True, want_emb: bool = True) -> Dict[str, np.ndarray]:
    """Variance-preserving seeded init; residual-branch-last BN gammas are scaled
    down so the fp16 residual streams stay O(1)-O(10)."""
    raw: Dict[str, np.ndarray] = {}
    layers: List[ns.ConvLayer] = []
    if want_det:
        layers += ns.detector_layers(det_blocks)
    if want_emb:
        layers += ns.iresnet_layers(emb_blocks)
    for l in layers:
        cin = l.cin_real or l.cin
        cout = l.cout_real or l.cout
        r = _rng(seed, l.name)
        is_det = l.name.startswith("det.")
        gain = np.sqrt(2.0) if (is_det and l.act == ns.ACT_RELU) else 1.0
        if l.name == "emb.fc":
            w = r.standard_normal((cout, cin), dtype=np.float32) * np.float32(1.0 / np.sqrt(cin))
            raw["emb.fc.weight"] = w
            raw["emb.fc.bias"] = (r.standard_normal(cout) * 0.05).astype(np.float32)
        else:
            fan_in = cin * l.k * l.k
            w = r.standard_normal((cout, cin, l.k, l.k), dtype=np.float32) * np.float32(gain / np.sqrt(fan_in))
            if l.name.endswith(".out"):
                # detector head: widen the output spread and bias the two score logits negative so
                # that the default 0.5 score threshold keeps a handful of anchors per frame
                w = w * np.float32(6.0)
            raw[l.name + ".weight"] = w
            if l.conv_bias:
                b = (r.standard_normal(cout) * 0.05).astype(np.float32)
                if l.name.endswith(".out"):
                    b[0::ns.DET_VALUES_PER_ANCHOR] -= np.float32(2.0)
                raw[l.name + ".bias"] = b
        if l.pre_bn:
            _bn(raw, seed, l.pre_bn, cin if l.name != "emb.fc" else 512)
        if l.post_bn:
            branch_last = l.res is not None and not (l.flags & ns.FLAG_RES_UP2)
            gs = (0.5 if is_det else 0.25) if branch_last else 1.0
            _bn(raw, seed, l.post_bn, cout, gs)
        if l.prelu:
            raw[l.prelu + ".weight"] = _rng(seed, l.prelu).uniform(0.15, 0.35, cout).astype(np.float32)
    return raw


def _bn_affine(raw, name) -> Tuple[np.ndarray, np.ndarray]:
    g = raw[name + ".weight"].astype(np.float64)
    b = raw[name + ".bias"].astype(np.float64)
    m = raw[name + ".running_mean"].astype(np.float64)
    v = raw[name + ".running_var"].astype(np.float64)
    s = g / np.sqrt(v + BN_EPS)
    return s, b - m * s


def fold_layer(raw: Dict[str, np.ndarray], l: ns.ConvLayer, return_w64: bool = False):
    """-> (w fp16 [cout][k][k][cin], bias fp32 [cout] or [9][cout], slope fp32 [cout] or None)
    `return_w64` appends the folded weights before the fp16 cast (float64, same layout, padded channels zero)."""
    cin_r = l.cin_real or l.cin
    cout_r = l.cout_real or l.cout
    if l.name == "emb.fc":
        # FC weight [512, C*7*7] with PyTorch flatten order (c, y, x) -> OIHW view [512, C, 7, 7]
        C = 512
        W = raw["emb.fc.weight"].astype(np.float64).reshape(cout_r, C, 7, 7)
        k_eff = 7
    else:
        W = raw[l.name + ".weight"].astype(np.float64)
        k_eff = l.k
    b = raw[l.name + ".bias"].astype(np.float64) if l.conv_bias else np.zeros(cout_r)
    T = np.zeros((cout_r, k_eff, k_eff))
    if l.pre_bn:
        s_in, t_in = _bn_affine(raw, l.pre_bn)
        T = np.einsum("ochw,c->ohw", W, t_in)
        W = W * s_in[None, :, None, None]
    if l.post_bn:
        s_out, t_out = _bn_affine(raw, l.post_bn)
    else:
        s_out, t_out = np.ones(cout_r), np.zeros(cout_r)
    W = W * s_out[:, None, None, None]
    if l.flags & ns.FLAG_BORDER_BIAS:
        assert l.k == 3 and l.stride == 1
        bias = np.zeros((9, l.cout))
        for cy in range(3):
            for cx in range(3):
                valid = np.ones((3, 3), dtype=bool)
                if cy == 0:
                    valid[0, :] = False
                if cy == 2:
                    valid[2, :] = False
                if cx == 0:
                    valid[:, 0] = False
                if cx == 2:
                    valid[:, 2] = False
                bias[cy * 3 + cx, :cout_r] = s_out * (b + (T * valid[None]).sum(axis=(1, 2))) + t_out
    else:
        bias = np.zeros((l.cout,))
        bias[:cout_r] = s_out * (b + T.sum(axis=(1, 2))) + t_out
    if l.name == "emb.fc":
        # -> [cout][y][x][c] flattened: matches the NHWC [7,7,512] activation viewed as 1x1x25088
        Wp = np.transpose(W, (0, 2, 3, 1)).reshape(cout_r, 1, 1, 7 * 7 * 512)
        w64 = np.zeros((l.cout, 1, 1, l.cin), dtype=np.float64)
        w64[:cout_r] = Wp
    else:
        w64 = np.zeros((l.cout, l.k, l.k, l.cin), dtype=np.float64)
        w64[:cout_r, :, :, :cin_r] = np.transpose(W, (0, 2, 3, 1))
    w16 = w64.astype(np.float16)
    slope = None
    if l.act == ns.ACT_PRELU:
        slope = np.zeros((l.cout,), dtype=np.float32)
        slope[:cout_r] = raw[l.prelu + ".weight"]
    if return_w64:
        return w16, bias.astype(np.float32), slope, w64
    return w16, bias.astype(np.float32), slope


KCONCAT_LIVENESS = True     # tests flip it to pack a blob with the buffer plan of older packers (the runtime must then leave the blocks unfused)


def assign_buffers(layers: List[ns.ConvLayer], pinned: List[str]) -> Tuple[Dict[str, int], int]:
    """Greedy liveness-based mapping logical tensor -> physical buffer id.
    `pinned` tensors (network inputs/outputs) get private buffers."""
    last_use: Dict[str, int] = {}
    producer = {l.dst: l for l in layers}
    for i, l in enumerate(layers):
        last_use[l.src] = i
        if l.res:
            last_use[l.res] = i
            # K-concat (csrc/frp_api.cpp: frp_load_weights): the runtime folds a block's 1x1 strided shortcut conv into the
            # 3x3 conv that adds it, which then reads the shortcut's INPUT - that tensor stays alive (and out of this
            # conv's output buffer) until here
            sc = producer.get(l.res)
            if KCONCAT_LIVENESS and sc is not None and sc.k == 1 and l.k == 3 and sc.stride == l.stride and not (l.flags & ns.FLAG_RES_UP2):
                last_use[sc.src] = max(last_use.get(sc.src, i), i)
        if getattr(l, "dst2", None):
            last_use.setdefault(l.dst2, i)          # a copy nobody reads still needs a buffer while it is written
    phys: Dict[str, int] = {}
    free: List[int] = []
    n = 0
    for p in pinned:
        phys[p] = n
        n += 1
    for i, l in enumerate(layers):
        for d in (l.dst, getattr(l, "dst2", None)):
            if d and d not in phys:
                if free:
                    phys[d] = free.pop(0)
                else:
                    phys[d] = n
                    n += 1
        # release tensors whose last use is this op (after allocating dst: in/out never alias)
        for t in {l.src, l.res, getattr(l, "dst2", None)}:
            if t and t not in pinned and last_use.get(t) == i and t in phys:
                free.append(phys[t])
    return phys, n


# ---------------------------------------------------------------------------- fp8 weight storage (BASELINE config 5)
OPFLAG_W_FP8 = 16          # include/frp_blob.h: FRP_OPFLAG_W_FP8
OPFLAG_FP8_MFMA = 32       # FRP_OPFLAG_FP8_MFMA
OPFLAG_OUT_FP8 = 64        # FRP_OPFLAG_OUT_FP8


def fp8_mfma_eligible(l: ns.ConvLayer) -> bool:
    """convs the fp8 matrix path covers: 3x3 stride 1 over whole 128-channel rows, fp16/fp8 outputs (the bulk of
    IResNet stages 2-4: every conv1 / conv2 except the strided conv2 and the 64 -> 128 conv1 of stage 2)"""
    return (l.k == 3 and l.stride == 1 and l.cin % 128 == 0 and l.cout % 8 == 0 and l.cin_real is None and
            not (l.flags & (ns.FLAG_OUT_F32 | ns.FLAG_FLATTEN | ns.FLAG_RES_UP2)))


def plan_fp8(layers: List[ns.ConvLayer]):
    """Which ops run on fp8 operands and which tensors exist in which precision.  -> per layer dict(f8, out8, dst2):
    f8: the op reads the fp8 copy of its input (tensor name + "@8"); out8: its primary output is fp8 (every consumer is an
    fp8 op and nobody needs it as a residual); dst2: name of the fp8 copy it writes next to its fp16 output."""
    elig = [fp8_mfma_eligible(l) for l in layers]
    need8, need16 = set(), set()
    for l, e in zip(layers, elig):
        (need8 if e else need16).add(l.src)
        if l.res:
            need16.add(l.res)
    need16.add(layers[-1].dst)
    plan = []
    for l, e in zip(layers, elig):
        out8 = e and l.dst in need8 and l.dst not in need16
        dst2 = (l.dst + "@8") if (l.dst in need8 and not out8) else None
        plan.append({"f8": e, "out8": out8, "dst2": dst2})
    return plan


def _fp8_e4m3_table() -> np.ndarray:
    """value of every OCP FP8 E4M3FN code (bias 7, no infinities, S.1111.111 = NaN)"""
    t = np.zeros(256, np.float32)
    for code in range(256):
        sgn = -1.0 if code & 0x80 else 1.0
        e, m = (code >> 3) & 0xF, code & 7
        if e == 15 and m == 7:
            v = np.nan
        elif e == 0:
            v = sgn * 2.0 ** -6 * (m / 8.0)
        else:
            v = sgn * 2.0 ** (e - 7) * (1.0 + m / 8.0)
        t[code] = v
    return t


FP8_E4M3 = _fp8_e4m3_table()


def fp8_e4m3_encode(x: np.ndarray) -> np.ndarray:
    """nearest E4M3FN code of each element (|x| <= 448), ties to the even code"""
    x = np.asarray(x, np.float32)
    pos = FP8_E4M3[:127]                                # codes 0..126: ascending magnitudes 0 .. 448
    a = np.minimum(np.abs(x), np.float32(448.0))
    hi = np.clip(np.searchsorted(pos, a, side="left"), 1, 126)
    lo = hi - 1
    dlo, dhi = a - pos[lo], pos[hi] - a
    code = np.where((dlo < dhi) | ((dlo == dhi) & ((lo & 1) == 0)), lo, hi).astype(np.uint8)
    return np.where(np.signbit(x) & (code != 0), code | 0x80, code).astype(np.uint8)


def fp8_quantize_rows(w16: np.ndarray):
    """per-output-channel E4M3 quantisation of a folded fp16 weight tensor [cout, ...]
    -> (codes u8 same shape, scale fp32 [cout]); dequantised value = fp16(fp32(table[code]) * scale)"""
    w = w16.astype(np.float32).reshape(w16.shape[0], -1)
    amax = np.abs(w).max(axis=1)
    scale = np.where(amax > 0, amax / np.float32(448.0), np.float32(1.0)).astype(np.float32)
    codes = fp8_e4m3_encode(w / scale[:, None]).reshape(w16.shape)
    return codes, scale


def fp8_dequantize_rows(codes: np.ndarray, scale: np.ndarray) -> np.ndarray:
    v = FP8_E4M3[codes.reshape(codes.shape[0], -1)] * scale[:, None].astype(np.float32)
    return v.astype(np.float32).astype(np.float16).reshape(codes.shape)


def run_program_fp32(raw: Dict[str, np.ndarray], layers: List[ns.ConvLayer], x: "np.ndarray", fp16_storage: bool = True,
                     weights_of=None, want_amax: bool = False):
    """The folded conv program on the CPU (torch fp32 convolutions): the arithmetic the device runs up to summation
    order when `fp16_storage` (fp16-rounded weights, every activation rounded to fp16 between layers), or the same
    program in plain fp32.  x: NHWC float array holding the network input.  `weights_of(layer, w16) -> array` may
    substitute a layer's folded weights (tests: the fp8-dequantised ones).  -> output of the last layer (NHWC / [N, D]),
    and with `want_amax` the dict tensor name -> max |value| over the batch (the fp8 calibration statistic)."""
    import torch
    import torch.nn.functional as F
    tens = {layers[0].src: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).permute(0, 3, 1, 2)}
    amax = {}
    with torch.no_grad():
        for l in layers:
            w16, bias, slope = fold_layer(raw, l)
            w = w16 if weights_of is None else weights_of(l, w16)
            t = tens[l.src]
            if l.flags & ns.FLAG_FLATTEN:
                t = t.permute(0, 2, 3, 1).reshape(t.shape[0], -1, 1, 1)
            y = F.conv2d(t, torch.from_numpy(np.asarray(w, dtype=np.float32)).permute(0, 3, 1, 2), None, stride=l.stride, padding=l.k // 2)
            Ho, Wo = y.shape[2], y.shape[3]
            if l.flags & ns.FLAG_BORDER_BIAS:
                cy = np.where(np.arange(Ho) == 0, 0, np.where(np.arange(Ho) == Ho - 1, 2, 1))
                cx = np.where(np.arange(Wo) == 0, 0, np.where(np.arange(Wo) == Wo - 1, 2, 1))
                y = y + torch.from_numpy(bias[cy[:, None] * 3 + cx[None, :]]).permute(2, 0, 1)[None]
            else:
                y = y + torch.from_numpy(bias)[None, :, None, None]
            if l.res:
                r = tens[l.res]
                if l.flags & ns.FLAG_RES_UP2:
                    r = r.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
                y = y + r
            if l.act == ns.ACT_RELU:
                y = torch.relu(y)
            elif l.act == ns.ACT_PRELU:
                y = torch.where(y > 0, y, y * torch.from_numpy(slope)[None, :, None, None])
            if fp16_storage and not (l.flags & ns.FLAG_OUT_F32):
                y = y.half().float()
            tens[l.dst] = y
            if want_amax:
                amax[l.dst] = float(y.abs().max())
    out = tens[layers[-1].dst]
    out = out.reshape(out.shape[0], -1).numpy() if (layers[-1].flags & ns.FLAG_FLATTEN) else out.permute(0, 2, 3, 1).numpy()
    return (out, amax) if want_amax else out


FP8_HEADROOM = 2.0      # calibration maps the largest observed |activation| to 448 / FP8_HEADROOM (E4M3 is floating point:
                        # headroom costs no relative precision until values reach the subnormal range, 2^-6 of the scale)


def default_calibration_chips(n: int = 4, seed: int = 1234) -> np.ndarray:
    """seeded aligned-chip stand-ins [n,112,112,3] u8 BGR (no face crops exist offline): smooth low-frequency content +
    noise, so that the activation statistics are not those of white noise alone.  Real packs: pass real aligned chips."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, ns.EMB_SIZE), np.linspace(-1, 1, ns.EMB_SIZE), indexing="ij")
    chips = np.empty((n, ns.EMB_SIZE, ns.EMB_SIZE, 3), np.float64)
    for i in range(n):
        c = rng.uniform(-0.3, 0.3, 2)
        blob = np.exp(-(((xx - c[0]) / 0.55) ** 2 + ((yy - c[1]) / 0.7) ** 2))
        base = rng.uniform(60, 120, 3)[None, None, :] + blob[..., None] * rng.uniform(40, 110, 3)[None, None, :]
        chips[i] = base + rng.standard_normal((ns.EMB_SIZE, ns.EMB_SIZE, 3)) * rng.uniform(6, 40)
    return np.clip(np.rint(chips), 0, 255).astype(np.uint8)


def emb_input_blob(chips_bgr: np.ndarray) -> np.ndarray:
    """aligned chips [n,112,112,3] u8 BGR -> the embedder's NHWC8 input as the device builds it: RGB (x - 127.5) / 127.5
    rounded to fp16, channels 3..7 zero"""
    x = np.zeros((len(chips_bgr), ns.EMB_SIZE, ns.EMB_SIZE, ns.EMB_IN_CH), np.float32)
    x[..., :3] = ((chips_bgr[..., ::-1].astype(np.float32) - 127.5) / 127.5).astype(np.float16)
    return x


def calibrate_fp8(raw: Dict[str, np.ndarray], layers: List[ns.ConvLayer], plan, chips_bgr: np.ndarray = None) -> Dict[str, float]:
    """Per-tensor scales of the fp8 activations of an embedder program: the fp16 program runs on the calibration chips
    (CPU emulation, run_program_fp32), every tensor that exists in E4M3 (the primary output of an `out8` op, the `dst2`
    copy of an fp16 output) gets scale = 2^ceil(log2(FP8_HEADROOM * amax / 448)).  Powers of two: scaling a tensor by a
    power of two then moves its scale and leaves its codes - and the network's result - bit for bit unchanged.
    -> tensor name -> scale (the stored byte is value / scale)."""
    if chips_bgr is None:
        chips_bgr = default_calibration_chips()
    _, amax = run_program_fp32(raw, layers, emb_input_blob(chips_bgr), True, want_amax=True)
    scales = {}
    for l, pl in zip(layers, plan):
        if pl["out8"] or pl["dst2"]:
            a = amax[l.dst]
            scales[l.dst] = float(2.0 ** np.ceil(np.log2(FP8_HEADROOM * a / 448.0))) if a > 0 and np.isfinite(a) else 1.0
    return scales


def fp8_dequantized_weights(layers: List[ns.ConvLayer], plan):
    """`weights_of` hook for run_program_fp32: the folded weights of the fp8 ops replaced by what their E4M3 codes
    decode to (value = table[code] * per-cout scale, NOT re-rounded to fp16: that is what the matrix unit multiplies)"""
    f8 = {l.name for l, pl in zip(layers, plan) if pl["f8"]}

    def hook(l, w16):
        if l.name not in f8:
            return w16.astype(np.float32)
        codes, scale = fp8_quantize_rows(w16)
        return (FP8_E4M3[codes.reshape(codes.shape[0], -1)] * scale[:, None]).reshape(codes.shape).astype(np.float32)
    return hook


def pack_blob(raw: Dict[str, np.ndarray], det_blocks=(1, 2, 2, 2), emb_blocks=(3, 13, 30, 3),
              weight_format: str = "fp16", w16_hook=None, calib_chips: np.ndarray = None, calibrate: bool = True) -> bytes:
    """weight_format "fp8": conv/FC weights are stored as E4M3 bytes + one fp32 scale per output channel
    (half the blob, half the upload); the library expands them to fp16 at load, the kernels are the fp16
    ones.  "fp8-mfma": the eligible embedder convs also run on E4M3 ACTIVATIONS; their per-tensor scales come from a
    calibration pass of the fp16 program over `calib_chips` (aligned chips [n,112,112,3] u8 BGR; default: seeded
    stand-ins) - `calibrate=False` writes unit scales (tests: what an uncalibrated pack does to scaled activations).
    `w16_hook(layer, w16) -> w16` lets tests substitute the folded fp16 weights of a layer."""
    if weight_format not in ("fp16", "fp8", "fp8-mfma"):
        raise ValueError("weight_format must be 'fp16', 'fp8' (storage only) or 'fp8-mfma' (fp8 storage + fp8 matrix path for the embedder)")
    det = ns.detector_layers(det_blocks)
    emb = ns.iresnet_layers(emb_blocks)
    data = bytearray()

    def put(arr: np.ndarray) -> int:
        pad = (-len(data)) % 256
        data.extend(b"\0" * pad)
        off = len(data)
        data.extend(np.ascontiguousarray(arr).tobytes())
        return off

    def pack_ops(layers, pinned, fp8_mfma=False):
        import copy
        plan = plan_fp8(layers) if fp8_mfma else [{"f8": False, "out8": False, "dst2": None}] * len(layers)
        scales = calibrate_fp8(raw, layers, plan, calib_chips) if (fp8_mfma and calibrate) else {}
        view = []                                   # the layer list as the buffer planner sees it (fp8 tensors by name)
        for l, pl in zip(layers, plan):
            v = copy.copy(l)
            if pl["f8"]:
                v.src = l.src + "@8" if not any(q["out8"] and ll.dst == l.src for ll, q in zip(layers, plan)) else l.src
            v.dst2 = pl["dst2"]
            view.append(v)
        phys, nb = assign_buffers(view, pinned)
        ops = bytearray()
        for l, v, pl in zip(layers, view, plan):
            w16, bias, slope = fold_layer(raw, l)
            if w16_hook is not None:
                w16 = w16_hook(l, w16)
            flags = l.flags
            if pl["f8"]:
                flags |= OPFLAG_FP8_MFMA
            if pl["out8"]:
                flags |= OPFLAG_OUT_FP8
            if weight_format == "fp8" or (weight_format == "fp8-mfma" and fp8_mfma):      # fp8-mfma: "fp8 ArcFace weights" - the
                codes, scale = fp8_quantize_rows(w16)                                     # detector keeps its fp16 weights
                w_off = put(codes)
                data.extend(b"\0" * ((-len(data)) % 16))
                data.extend(scale.astype("<f4").tobytes())
                flags |= OPFLAG_W_FP8
            else:
                w_off = put(w16)
            b_off = put(bias)
            s_off = put(slope) if slope is not None else -1
            ops += struct.pack(OP_FMT, phys[v.src], phys[l.dst], phys[l.res] if l.res else -1,
                               l.cin, l.cout, l.k, l.stride, l.act, flags,
                               (l.cin_real or l.cin) | ((l.cout_real or l.cout) << 16), w_off, b_off, s_off,
                               phys[v.dst2] if v.dst2 else -1,
                               scales.get(l.src, 1.0) if pl["f8"] else 1.0,                         # scale of the fp8 tensor it reads
                               scales.get(l.dst, 1.0) if (pl["out8"] or pl["dst2"]) else 1.0, 0)    # ... and of the one it writes
        return bytes(ops), phys, nb

    det_ops, det_phys, det_nb = pack_ops(det, ["det.in", "det.out3", "det.out4", "det.out5"])
    emb_ops, emb_phys, emb_nb = pack_ops(emb, ["emb.in", "emb.out"], fp8_mfma=(weight_format == "fp8-mfma"))
    det_macs, _ = ns.layer_macs(det, 1088, 1920, "det.in")
    emb_macs, _ = ns.layer_macs(emb, ns.EMB_SIZE, ns.EMB_SIZE, "emb.in")
    det_ops_off = HEADER_BYTES
    emb_ops_off = det_ops_off + len(det_ops)
    data_off = emb_ops_off + len(emb_ops)
    data_off += (-data_off) % 256
    hdr = struct.pack(
        HEADER_FMT, BLOB_MAGIC, BLOB_VERSION, HEADER_BYTES,
        len(det), det_nb, det_phys["det.in"], ns.DET_IN_CH,
        det_phys["det.out3"], det_phys["det.out4"], det_phys["det.out5"],
        ns.DET_NUM_ANCHORS,
        len(emb), emb_nb, emb_phys["emb.in"], ns.EMB_IN_CH,
        emb_phys["emb.out"], ns.EMB_SIZE, ns.EMB_DIM, 0,
        det_ops_off, emb_ops_off, data_off, len(data),
        det_macs, emb_macs)
    out = bytearray(hdr) + det_ops + emb_ops
    out.extend(b"\0" * (data_off - len(out)))
    out += data
    return bytes(out)


def load_arcface_state_dict(path: str, prefix: str = "emb.") -> Dict[str, np.ndarray]:
    """Embedder weights from a PyTorch checkpoint in the public arcface_torch `iresnet` naming
    (conv1.weight, bn1.*, prelu.weight, layer{1..4}.{i}.{bn1,conv1,bn2,prelu,conv2,bn3,downsample.0/1}.*,
    bn2.*, fc.{weight,bias}, features.*) -> the raw dict `pack_blob` consumes.  No such file exists
    offline; the layout is exercised on a checkpoint written by the tests."""
    import torch
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    raw = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        k = k[7:] if k.startswith("module.") else k
        raw[prefix + k] = v.detach().cpu().float().numpy()
    return raw


def emb_blocks_of(raw: Dict[str, np.ndarray]) -> Tuple[int, int, int, int]:
    """number of IBasicBlocks per stage present in a raw dict (R50 = (3,4,14,3), R100 = (3,13,30,3))"""
    out = []
    for li in (1, 2, 3, 4):
        n = 0
        while f"emb.layer{li}.{n}.conv1.weight" in raw:
            n += 1
        out.append(n)
    return tuple(out)


def synthetic_blob(seed: int = 7, det_blocks=(1, 2, 2, 2), emb_blocks=(3, 13, 30, 3)) -> bytes:
    return pack_blob(make_synthetic_raw(seed, det_blocks, emb_blocks), det_blocks, emb_blocks)
