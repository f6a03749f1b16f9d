"""GPU parity tests of the individual HIP kernels, through the C ABI, against the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import get_raw_and_blob
from oracle import network as onet

pytestmark = pytest.mark.gpu


def _conv_ref(x, w, bias, stride, act, slope, res, flags):
    """fp32 reference on the fp16-rounded operands.  x NHWC, w [Cout,k,k,Cin]."""
    xt = torch.from_numpy(x.astype(np.float32)).permute(0, 3, 1, 2)
    wt = torch.from_numpy(w.astype(np.float32)).permute(0, 3, 1, 2)
    k = w.shape[1]
    y = F.conv2d(xt, wt, None, stride=stride, padding=k // 2).permute(0, 2, 3, 1).numpy()
    N, Ho, Wo, Co = y.shape
    if flags & 1:
        cy = np.where(np.arange(Ho) == 0, 0, np.where(np.arange(Ho) == Ho - 1, 2, 1))
        cx = np.where(np.arange(Wo) == 0, 0, np.where(np.arange(Wo) == Wo - 1, 2, 1))
        cls = cy[:, None] * 3 + cx[None, :]
        y = y + bias[cls][None]
    else:
        y = y + bias[None, None, None, :]
    if res is not None:
        r = res.astype(np.float32)
        if flags & 4:
            r = np.repeat(np.repeat(r, 2, axis=1), 2, axis=2)
        y = y + r
    if act == 1:
        y = np.maximum(y, 0)
    elif act == 2:
        y = np.where(y > 0, y, y * slope[None, None, None, :])
    return y


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, act, res, flags
    (2, 16, 16, 64, 64, 3, 1, 1, False, 0),
    (1, 17, 13, 64, 128, 3, 2, 0, True, 0),
    (2, 9, 9, 128, 256, 1, 1, 2, False, 0),
    (1, 20, 20, 8, 32, 3, 2, 1, False, 0),      # stem: small-Cin path, Ktot=72
    (1, 12, 12, 32, 64, 3, 2, 1, False, 0),     # small-Cin path, Ktot=288
    (3, 7, 7, 256, 512, 3, 1, 2, False, 1),     # border-class bias
    (1, 8, 8, 128, 128, 1, 1, 0, True, 4),      # FPN lateral + 2x upsampled residual
    (4, 1, 1, 1024, 512, 1, 1, 0, False, 2),    # FC-shaped, fp32 out
    (1, 8, 8, 128, 32, 3, 1, 0, False, 0),      # head: Cout=32
    (2, 14, 14, 64, 128, 1, 2, 0, False, 0),    # 1x1 stride-2 shortcut
    (1, 5, 7, 64, 64, 3, 1, 1, True, 0),        # ragged: M=35 << tile
    (5, 28, 28, 128, 128, 3, 1, 2, True, 1),    # multi-tile, all epilogue features
    (1, 40, 24, 64, 64, 3, 2, 1, False, 0),
    # row-patch kernel (3x3 s1, Cin % 64 == 0): image width around / beyond the 256-pixel tile, many images
    # per tile, ragged last tile, two cout tiles, Cout not a multiple of the tile
    (1, 3, 300, 64, 64, 3, 1, 1, False, 0),     # W > tile: the three kh patches do not overlap
    (40, 7, 7, 128, 128, 3, 1, 2, True, 1),     # 5 images per tile, border bias + residual + PReLU
    (3, 30, 33, 192, 256, 3, 1, 1, True, 0),    # 3 channel blocks, 2 cout tiles, ragged M
    (2, 19, 21, 64, 96, 3, 1, 0, False, 0),     # Cout = 96: ragged cout tile
    (1, 1, 1, 64, 64, 3, 1, 0, False, 0),       # single pixel: every tap but the centre is padding
    (2, 2, 2, 128, 32, 3, 1, 1, False, 0),
    # 512 x 64 tile of the lean kernel (Cout <= 64: two-slot patch ring): ragged third tile, image rows longer than the
    # tile, two channel blocks with Cout = 32, ten images per tile with border bias + residual + PReLU, exactly one tile
    (3, 20, 20, 64, 64, 3, 1, 1, True, 0),
    (1, 2, 600, 64, 64, 3, 1, 0, False, 0),
    (2, 30, 33, 128, 32, 3, 1, 1, False, 0),
    (70, 7, 7, 64, 64, 3, 1, 2, True, 1),
    (2, 16, 16, 192, 64, 3, 1, 2, False, 1),    # M = 512: one full tile, three channel blocks
]


# Tile class of a conv launch (conv_common.h: conv_small_m): by default the launcher takes quarter tiles (128 pixels x 64 couts,
# two workgroups per CU) when the default tiling would leave half of the CUs idle - which is every case of this list.  The
# tests therefore name the class: bit 18 = default tiles (what the 32 x 1080p / 320-face workloads run), bit 17 = quarter tiles.
TILES_DEFAULT, TILES_QUARTER = 0x40000, 0x20000


@pytest.mark.parametrize("tiles", [TILES_DEFAULT, TILES_QUARTER], ids=["default-tiles", "quarter-tiles"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_parity(engine, case, tiles):
    N, H, W, Cin, Cout, k, stride, act, has_res, flags = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
    w = (rng.standard_normal((Cout, k, k, Cin)) / np.sqrt(k * k * Cin)).astype(np.float16)
    bias = rng.standard_normal((9, Cout) if flags & 1 else (Cout,)).astype(np.float32) * 0.3
    slope = rng.uniform(0.1, 0.4, Cout).astype(np.float32) if act == 2 else None
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = None
    if has_res:
        res = rng.standard_normal((N, Ho // 2, Wo // 2, Cout) if flags & 4 else (N, Ho, Wo, Cout)).astype(np.float16)
    out = engine.conv2d(x, w, bias, stride=stride, act=act, slope=slope, res=res, flags=flags | tiles)
    ref = _conv_ref(x, w, bias, stride, act, slope, res, flags)
    assert out.shape == ref.shape
    # fp32 accumulate: error is the fp16 output rounding (rel 2^-11) + summation order
    tol = 2e-3 * max(1.0, float(np.abs(ref).max())) if not (flags & 2) else 1e-4 * max(1.0, float(np.abs(ref).max()))
    err = np.abs(out.astype(np.float32) - ref).max()
    assert err <= tol, f"max err {err} > {tol}"


@pytest.mark.selfcheck
@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[5] == 3 and c[6] == 1 and c[3] % 64 == 0])
def test_conv_row_patch_kernel_equals_generic_kernel(engine, case):
    """conv3x3_rows.hip and conv_mfma.hip accumulate in the same k order -> identical bits."""
    N, H, W, Cin, Cout, k, stride, act, has_res, flags = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 1)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
    w = (rng.standard_normal((Cout, k, k, Cin)) / np.sqrt(k * k * Cin)).astype(np.float16)
    bias = rng.standard_normal((9, Cout) if flags & 1 else (Cout,)).astype(np.float32) * 0.3
    slope = rng.uniform(0.1, 0.4, Cout).astype(np.float32) if act == 2 else None
    res = rng.standard_normal((N, H, W, Cout)).astype(np.float16) if has_res else None
    a = engine.conv2d(x, w, bias, stride=1, act=act, slope=slope, res=res, flags=flags | TILES_DEFAULT)
    b = engine.conv2d(x, w, bias, stride=1, act=act, slope=slope, res=res, flags=flags | TILES_DEFAULT | (1 << 8))
    assert np.array_equal(a.view(np.uint16), b.view(np.uint16))


# 64 -> 64 layers on large maps (conv3x3_c64.hip: weights in registers, 8 x 32-pixel tiles, one barrier per tile; taken from two
# rounds of tiles - 512 - on): the four (activation, residual, border-bias) combinations of the two networks, maps whose last
# tile column / row is ragged (56 = 32 + 24 columns; 100 = 12.5 tile rows), images that end inside a workgroup's walk
C64_CASES = [
    # N, H, W, act, res, flags
    (4, 136, 240, 1, False, 0),      # detector layer1 conv1 shape class: ReLU
    (6, 100, 210, 1, True, 0),       # ... conv2: residual + ReLU, ragged rows and columns
    (40, 56, 56, 2, False, 1),       # IResNet conv1: PReLU + 9 border-bias classes
    (11, 112, 112, 0, True, 0),      # IResNet conv2: residual, no activation
    (600, 7, 33, 1, False, 0),       # many small images: one tile row, two tile columns (the second 1 pixel wide)
]
NO_C64 = 0x100000


@pytest.mark.parametrize("case", C64_CASES)
def test_conv_c64_vs_fp32_reference_and_generic_kernel(engine, case, monkeypatch):
    """the dedicated 64 -> 64 kernel against the fp32 reference (the direct kernels' bar) and - same k order, same epilogue
    arithmetic - BIT FOR BIT against the row-patch kernel (flags bit 20 keeps it off) and the generic kernel (dbg 1); 12 more
    launches return the bits of the first (its LDS ring runs two tiles ahead of the MFMAs)."""
    N, H, W, act, has_res, flags = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((N, H, W, 64)).astype(np.float16)
    w = (rng.standard_normal((64, 3, 3, 64)) / 24).astype(np.float16)
    bias = rng.standard_normal((9, 64) if flags & 1 else (64,)).astype(np.float32) * 0.3
    slope = rng.uniform(0.1, 0.4, 64).astype(np.float32) if act == 2 else None
    res = rng.standard_normal((N, H, W, 64)).astype(np.float16) if has_res else None
    kw = dict(stride=1, act=act, slope=slope, res=res)
    # (the launcher routes only maps whose tiles are at least 95 % full to this kernel; FRP_C64_ALL - read once per process - takes
    # every eligible shape: set in tests/conftest.py for the GPU session, so the ragged cases below do run it)
    assert os.environ.get("FRP_C64_ALL") == "1"
    out = engine.conv2d(x, w, bias, flags=flags | TILES_DEFAULT, **kw)
    lean = engine.conv2d(x, w, bias, flags=flags | TILES_DEFAULT | NO_C64, **kw)
    generic = engine.conv2d(x, w, bias, flags=flags | TILES_DEFAULT | (1 << 8), **kw)
    assert np.array_equal(lean.view(np.uint16), generic.view(np.uint16))
    bad = np.argwhere(out.view(np.uint16) != generic.view(np.uint16))
    assert len(bad) == 0, f"{len(bad)} elements differ from the generic kernel; first {bad[:4].tolist()}"
    sub = slice(0, min(N, 3))                                   # (the fp32 reference on the first images: CPU time)
    ref = _conv_ref(x[sub], w, bias, 1, act, slope, None if res is None else res[sub], flags)
    assert np.abs(out[sub].astype(np.float32) - ref).max() <= 2e-3 * max(1.0, float(np.abs(ref).max()))
    for _ in range(12):
        again = engine.conv2d(x, w, bias, flags=flags | TILES_DEFAULT, **kw)
        assert np.array_equal(out.view(np.uint16), again.view(np.uint16))


S2_CASES = [
    # N, H, W, Cin, Cout, act
    (3, 28, 28, 64, 128, 1),        # 14-wide output rows: 18 image rows begin inside a tile
    (2, 136, 240, 64, 128, 1),      # detector layer2.0.conv1 shape class: ReLU, 120-wide rows
    (5, 14, 14, 256, 256, 0),       # 7-wide rows (the narrowest the kernel takes), two cout tiles, less than one pixel tile
    (9, 56, 56, 128, 128, 0),       # IResNet layer2.0.conv2 shape class (without its shortcut segment)
    (40, 28, 28, 256, 256, 0),      # ... layer3.0.conv2
    (3, 30, 22, 128, 256, 1),       # odd output sizes (15 x 11), ragged last tile
    (2, 68, 120, 256, 256, 1),      # detector layer4.0.conv1
    (6, 14, 14, 512, 512, 0),       # 16 channel blocks, four cout tiles
]


@pytest.mark.parametrize("case", S2_CASES)
def test_conv_stride2_row_patch_kernel_vs_fp32_reference_and_generic_kernel(engine, case):
    """the stride-2 kernel (row patches with shared left / right neighbour entries, conv3x3_s2.hip) against the fp32 reference (the
    direct kernels' bar) and against the generic kernel (dbg 1), from which it differs in the k order (32- instead of 64-channel
    blocks) and therefore in the last bits of the fp32 sums: within 2 fp16 ulps of the output scale, most elements identical;
    12 more launches return the bits of the first (its rings run two steps ahead across tile boundaries)."""
    N, H, W, Cin, Cout, act = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
    w = (rng.standard_normal((Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float16)
    bias = rng.standard_normal((Cout,)).astype(np.float32) * 0.3
    kw = dict(stride=2, act=act)
    S2 = TILES_DEFAULT | 0x200000          # (the kernel is opt-in: flags bit 21 / FRP_S2=1 - it measures slower than the per-tap images)
    out = engine.conv2d(x, w, bias, flags=S2, **kw)
    generic = engine.conv2d(x, w, bias, flags=TILES_DEFAULT | (1 << 8), **kw)
    assert out.shape == generic.shape == (N, H // 2, W // 2, Cout)
    scale = max(1.0, float(np.abs(generic.astype(np.float32)).max()))
    ulp = float(np.spacing(np.float16(scale)))
    d = np.abs(out.astype(np.float32) - generic.astype(np.float32))
    worst = np.unravel_index(int(d.argmax()), d.shape)
    assert d.max() <= 2 * ulp, f"max |diff| {d.max():.4g} = {d.max() / ulp:.1f} ulps of {scale:.3g} at (n, y, x, c) = {worst}"
    assert (out.view(np.uint16) == generic.view(np.uint16)).mean() >= 0.9
    sub = slice(0, min(N, 3))
    ref = _conv_ref(x[sub], w, bias, 2, act, None, None, 0)
    assert np.abs(out[sub].astype(np.float32) - ref).max() <= 2e-3 * max(1.0, float(np.abs(ref).max()))
    for _ in range(12):
        again = engine.conv2d(x, w, bias, flags=S2, **kw)
        assert np.array_equal(out.view(np.uint16), again.view(np.uint16))
    # the two kernels differ in the k order: somewhere in a tensor this size a last bit does (i.e. the opt-in bit did select the kernel)
    if out.size >= 1 << 18:
        assert not np.array_equal(out.view(np.uint16), generic.view(np.uint16))


# cases the quarter-tile configurations cover with the k order of the default tiles (fp16 output, no split-K)
QUARTER_CASES = [c for c in CONV_CASES if c[3] % 64 == 0 and not (c[9] & 2)] + [
    (36, 14, 14, 256, 256, 3, 1, 2, True, 1),    # 36 faces at IResNet stage 3 (config 4's operating point): 56 default tiles
    (36, 28, 28, 128, 256, 3, 2, 0, True, 0),    # ... its stride-2 conv with the shortcut as residual
    (4, 34, 60, 128, 32, 3, 1, 0, False, 0),     # detector head at pyramid scale 0.25
    (1, 14, 14, 256, 512, 1, 2, 0, False, 0),    # one face: 1x1 stride-2 shortcut
    (9, 7, 7, 512, 512, 3, 1, 2, True, 1),       # stage 4, 4 channel blocks... 8 of them, 8 cout tiles
]


def test_conv_quarter_tiles_race_screen(engine):
    """Two workgroups per CU load the LDS enough to expose a write-after-read hole in the k-step protocol (a fragment read
    issued before a barrier, still in flight when another wave - released by that barrier - restages the ring slot; the
    compiler had sunk the MFMAs that retire such reads below the barrier): before `retire_lds_reads()` one launch in
    four of this layer came back with one stale 8-row weight piece in one wave.  490 quarter tiles on 512
    slots, 30 launches, every one bit-identical to the default tiles."""
    rng = np.random.default_rng(2024)
    N, H, W, C = 5, 112, 112, 64
    x = rng.standard_normal((N, H, W, C)).astype(np.float16)
    w = (rng.standard_normal((C, 3, 3, C)) / np.sqrt(9 * C)).astype(np.float16)
    bias = rng.standard_normal((9, C)).astype(np.float32) * 0.3
    slope = rng.uniform(0.1, 0.4, C).astype(np.float32)
    ref = engine.conv2d(x, w, bias, act=2, slope=slope, flags=1 | TILES_DEFAULT)
    for _ in range(30):
        got = engine.conv2d(x, w, bias, act=2, slope=slope, flags=1 | TILES_QUARTER)
        assert np.array_equal(ref.view(np.uint16), got.view(np.uint16))
    # ... and the generic kernel's quarter tiles (stride 2: 1,225 tiles, several per workgroup)
    ref = engine.conv2d(x, w, bias[0], stride=2, act=0, flags=TILES_DEFAULT)
    for _ in range(10):
        got = engine.conv2d(x, w, bias[0], stride=2, act=0, flags=TILES_QUARTER)
        assert np.array_equal(ref.view(np.uint16), got.view(np.uint16))


@pytest.mark.parametrize("case", QUARTER_CASES)
def test_conv_quarter_tiles_equal_default_tiles(engine, case):
    """Quarter tiles (small maps / few faces) keep the k order of the default tiles -> identical bits, in the row-patch
    kernel and in the generic one; and the automatic choice is one of the two."""
    N, H, W, Cin, Cout, k, stride, act, has_res, flags = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 2)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
    w = (rng.standard_normal((Cout, k, k, Cin)) / np.sqrt(k * k * Cin)).astype(np.float16)
    bias = rng.standard_normal((9, Cout) if flags & 1 else (Cout,)).astype(np.float32) * 0.3
    slope = rng.uniform(0.1, 0.4, Cout).astype(np.float32) if act == 2 else None
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = None
    if has_res:
        res = rng.standard_normal((N, Ho // 2, Wo // 2, Cout) if flags & 4 else (N, Ho, Wo, Cout)).astype(np.float16)
    kw = dict(stride=stride, act=act, slope=slope, res=res)
    big = engine.conv2d(x, w, bias, flags=flags | TILES_DEFAULT, **kw)
    quarter = engine.conv2d(x, w, bias, flags=flags | TILES_QUARTER, **kw)
    auto = engine.conv2d(x, w, bias, flags=flags, **kw)
    assert np.array_equal(big.view(np.uint16), quarter.view(np.uint16))
    assert np.array_equal(big.view(np.uint16), auto.view(np.uint16))
    if k == 3 and stride == 1:      # ... and the generic kernel's quarter tiles
        gq = engine.conv2d(x, w, bias, flags=flags | TILES_QUARTER | (1 << 8), **kw)
        assert np.array_equal(big.view(np.uint16), gq.view(np.uint16))


def test_conv_rejects_unsupported_shape(engine):
    from frp_amd.native import FrpError
    x = np.zeros((1, 8, 8, 24), np.float16)     # Cin=24: not a power of two below 64
    w = np.zeros((32, 3, 3, 24), np.float16)
    with pytest.raises(FrpError):
        engine.conv2d(x, w, np.zeros(32, np.float32))


@pytest.mark.parametrize("N,M", [(1000, 5), (128, 1), (4097, 37), (10000, 320), (1, 3)])
def test_match_parity(engine, N, M):
    rng = np.random.default_rng(N * 7 + M)
    G = rng.standard_normal((N, 512)).astype(np.float32)
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    planted = rng.integers(0, N, size=M)
    Q = G[planted] + 0.05 * rng.standard_normal((M, 512)).astype(np.float32) / np.sqrt(512) * 4
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    engine.gallery_set(G)
    assert engine.gallery_size() == N
    idx, cos = engine.match(Q)
    g16 = engine.gallery_get().astype(np.float32)
    assert np.abs(g16 - G).max() < 1e-3
    oidx, ocos = onet.match_topk(G, Q, 1)
    assert np.array_equal(idx, oidx[:, 0])          # identical top-1 identity
    assert np.abs(cos - ocos[:, 0]).max() < 1e-3    # north_star: within 1e-3 cosine
    if N <= 4097:
        S = engine.match_scores(Q)
        assert np.abs(S - Q.astype(np.float64) @ G.T.astype(np.float64)).max() < 1e-3


def test_match_ties_prefer_lower_index(engine):
    rng = np.random.default_rng(3)
    g = rng.standard_normal((1, 512)).astype(np.float32)
    G = np.repeat(g, 300, axis=0)                   # all rows identical -> every score ties
    engine.gallery_set(G)
    idx, cos = engine.match(g)
    assert idx[0] == 0 and abs(cos[0] - 1.0) < 1e-3


def test_gallery_update_remove(engine):
    rng = np.random.default_rng(11)
    G = rng.standard_normal((10, 512)).astype(np.float32)
    engine.gallery_set(G)
    new = rng.standard_normal(512).astype(np.float32)
    engine.gallery_update_row(10, new)              # append
    assert engine.gallery_size() == 11
    idx, cos = engine.match(new[None])
    assert idx[0] == 10 and cos[0] > 0.999
    engine.gallery_remove_row(3)                    # last row moves into slot 3
    assert engine.gallery_size() == 10
    idx, _ = engine.match(new[None])
    assert idx[0] == 3
    engine.gallery_update_row(0, -new)              # overwrite
    idx, cos = engine.match(-new[None])
    assert idx[0] == 0 and cos[0] > 0.999
    engine.gallery_set(np.zeros((0, 512), np.float32))
    assert engine.gallery_size() == 0
    from frp_amd.native import FrpError
    with pytest.raises(FrpError):
        engine.match(new[None])


def _random_heads(rng, B, Hc, Wc, logit_scale=3.0, logit_shift=-4.0):
    heads = []
    for s in (8, 16, 32):
        h = rng.standard_normal((B, Hc // s, Wc // s, 32)).astype(np.float32)
        h[..., 0] = h[..., 0] * logit_scale + logit_shift
        h[..., 15] = h[..., 15] * logit_scale + logit_shift
        h[..., 1:5] = np.abs(h[..., 1:5]) * 2 + 0.5       # positive distances: real boxes
        h[..., 16:20] = np.abs(h[..., 16:20]) * 2 + 0.5
        heads.append(h.astype(np.float16))
    return heads


@pytest.mark.parametrize("Hc,Wc,thresh,forced,shift", [
    (64, 96, 0.5, False, -4.0),       # few candidates
    (320, 320, 0.5, False, -1.0),     # thousands of candidates > CAP: radix-select path, heavy NMS
    (320, 320, 0.5, True, -4.0),      # forced top-K
    (1088, 1920, 0.5, False, -8.0),   # full 1080p canvas: 85,680 anchors
    (1088, 1920, 0.3, True, 0.0),
    (64, 64, 0.999, False, -9.0),     # nothing passes
])
def test_decode_nms_parity(engine, Hc, Wc, thresh, forced, shift):
    rng = np.random.default_rng(Hc * 31 + Wc + int(forced))
    B = 2
    heads = _random_heads(rng, B, Hc, Wc, logit_shift=shift)
    K = 10
    o = engine.decode_heads(heads, (Hc, Wc), max_faces=K, det_thresh=thresh, nms_iou=0.4, flags=1 if forced else 0)
    for b in range(B):
        ob, ok, osc, oa = onet.decode_nms([h[b] for h in heads], 0.0 if forced else thresh, 2.0 if forced else 0.4, K)
        n = len(oa)
        assert o["counts"][b] == n
        assert np.array_equal(o["anchor_idx"][b, :n], oa)             # same faces, same order
        assert np.array_equal(o["boxes"][b, :n], ob)                  # bit-exact fp32 box arithmetic
        assert np.array_equal(o["kps"][b, :n], ok)
        assert np.abs(o["scores"][b, :n] - osc).max() < 1e-6 if n else True
        assert np.all(o["boxes"][b, n:] == 0) and np.all(o["anchor_idx"][b, n:] == -1)


def test_decode_forced_k_with_nan_and_inf_logits(engine):
    """forced top-K keeps exactly K anchors whatever the head produced: +inf first, NaN logits last (a defined lowest
    key), so the compact face list the embedder walks never runs short (real fp16 weights can overflow a head)"""
    rng = np.random.default_rng(3)
    heads = _random_heads(rng, 1, 64, 64)
    h0 = heads[0]
    h0[0, :, :, 0] = np.nan                    # all first-anchor logits of the finest level: NaN
    h0[0, 2, 3, 15] = np.inf
    heads[1][0, :, :, 0] = -np.inf
    K = 20
    o = engine.decode_heads(heads, (64, 64), max_faces=K, det_thresh=0.5, nms_iou=0.4, flags=1)
    ob, ok, osc, oa = onet.decode_nms([h[0] for h in heads], 0.0, 2.0, K)
    assert o["counts"][0] == K == len(oa)
    assert np.array_equal(o["anchor_idx"][0], oa)
    assert oa[0] == 2 * (2 * 8 + 3) + 1        # the +inf anchor leads
    # every anchor NaN: still K faces, in anchor order
    for h in heads:
        h[..., 0] = np.nan
        h[..., 15] = np.nan
    o = engine.decode_heads(heads, (64, 64), max_faces=K, det_thresh=0.5, nms_iou=0.4, flags=1)
    assert o["counts"][0] == K and np.array_equal(o["anchor_idx"][0], np.arange(K))


def test_decode_ties_by_anchor_index(engine):
    heads = [np.zeros((1, 64 // s, 64 // s, 32), np.float16) for s in (8, 16, 32)]
    for h in heads:
        h[..., 0] = 2.0
        h[..., 15] = 2.0          # every anchor has the same logit
        h[..., 1:5] = 0.1
        h[..., 16:20] = 0.1       # tiny boxes: no overlap between locations
    o = engine.decode_heads(heads, (64, 64), max_faces=6, det_thresh=0.5, nms_iou=0.4)
    ob, _, _, oa = onet.decode_nms([h[0] for h in heads], 0.5, 0.4, 6)
    assert np.array_equal(o["anchor_idx"][0, :6], oa)
    assert list(oa[:2]) == [0, 2]   # anchor 1 sits on anchor 0's box (IoU 1) and is suppressed


def test_align_parity(engine):
    rng = np.random.default_rng(5)
    H, W = 240, 320
    frame = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    frame = (0.5 * frame + 0.5 * np.roll(frame, 1, axis=1)).astype(np.uint8)
    kps = []
    for (cx, cy, sc, ang) in [(160, 120, 1.5, 0.1), (60, 50, 0.6, -0.4), (300, 220, 2.0, 0.8), (10, 10, 1.0, 0.0)]:
        t = onet.ARCFACE_TEMPLATE - 56.0
        R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        kps.append((t @ R.T) * sc + [cx, cy] + rng.standard_normal((5, 2)) * 1.5)
    kps = np.array(kps, dtype=np.float32)
    chips = engine.align(frame, kps)                       # [M,112,112,8] fp16, RGB (x-127.5)/127.5
    ref = onet.align_faces(frame, kps)                     # BGR float 0..255
    ref_blob = (ref[..., ::-1] - 127.5) / 127.5
    assert np.all(chips[..., 3:] == 0)
    err = np.abs(chips[..., :3].astype(np.float32) - ref_blob)
    assert err.max() < 6e-3, err.max()                     # fp16 rounding + fp32 source coordinates
    assert err.mean() < 5e-4


@pytest.mark.selfcheck
def test_embedder_stem_kernel_agrees_with_generic_conv(engine, monkeypatch):
    """emb_stem_kernel vs the generic small-Cin conv path on the same chips: same fp16 inputs, fp32
    accumulation in a different order -> embeddings equal to ~1e-4"""
    rng = np.random.default_rng(19)
    chips = rng.integers(0, 256, size=(5, 112, 112, 3), dtype=np.uint8)
    raw, blob = get_raw_and_blob((1, 1, 1, 1), (1, 1, 1, 1))
    engine.load_weights(blob)
    a = engine.embed_aligned(chips)
    monkeypatch.setenv("FRP_NO_EMB_STEM", "1")
    b = engine.embed_aligned(chips)
    assert np.abs(a - b).max() < 2e-3 and (a * b).sum(1).min() > 1 - 1e-5
    ref = onet.emb_forward(raw, onet.emb_blob(chips))
    assert (a * ref).sum(1).min() > 1 - 1e-3 and (b * ref).sum(1).min() > 1 - 1e-3


def test_embed_parity_small_and_r100(engine):
    rng = np.random.default_rng(9)
    chips = rng.integers(0, 256, size=(3, 112, 112, 3), dtype=np.uint8)
    for blocks in [(1, 1, 1, 1), (3, 13, 30, 3)]:
        raw, blob = get_raw_and_blob((1, 1, 1, 1), blocks)
        engine.load_weights(blob)
        e = engine.embed_aligned(chips)
        ref = onet.emb_forward(raw, onet.emb_blob(chips))
        assert np.abs(np.linalg.norm(e, axis=1) - 1).max() < 1e-4
        cos = (e * ref).sum(1)
        assert cos.min() > 1 - 1e-3, (blocks, cos)          # north_star tolerance


@pytest.mark.parametrize("stem_env", [None, "FRP_NO_FUSED_STEM12", "FRP_NO_FUSED_STEM"])
@pytest.mark.parametrize("shape", [(2, 150, 200), (1, 97, 131), (3, 64, 64)])
def test_detector_head_maps_parity(engine, stem_env, shape, monkeypatch):
    """all three stem paths (u8 -> stem1 -> stem2 in one kernel; fused stem1 + generic stem2; preprocess +
    generic convs) against the fp32 oracle, on letterboxed canvases whose stem tiles are ragged"""
    if stem_env:
        monkeypatch.setenv(stem_env, "1")
    B, H, W = shape
    canvas = ((H + 31) // 32 * 32, (W + 31) // 32 * 32)
    rng = np.random.default_rng(21 + H)
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    engine.load_weights(blob)
    frames = rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)
    engine.detect(frames, max_faces=4, det_thresh=0.5)
    heads = engine.head_maps()
    ref = onet.det_forward(raw, onet.det_blob(frames, canvas))
    for g, r in zip(heads, ref):
        assert g.shape[:3] == r.shape[:3]
        scale = max(1.0, float(np.abs(r).max()))
        err = np.abs(g[..., :30].astype(np.float32) - r).max()
        assert err < 2e-2 * scale, (err, scale)
        assert np.all(g[..., 30:] == 0)


def _emulate_program_fp16(raw, layers, x16, want):
    """The detector / embedder PROGRAM as the device runs it, emulated in fp32 torch-CPU: BatchNorms folded by
    weights.fold_layer, fp16 weights, every activation rounded to fp16 between layers.  Against this the kernels may
    differ by fp32 summation order only, so the bound is a few fp16 ulps instead of the fp16-vs-fp32-network budget."""
    from frp_amd import weights as wts
    from frp_amd import netspec as ns
    tens = {layers[0].src: x16}
    for l in layers:
        w16, bias, slope = wts.fold_layer(raw, l)
        x = tens[l.src]
        if l.flags & ns.FLAG_FLATTEN:
            x = x.reshape(x.shape[0], 1, 1, -1)
        res = tens[l.res] if l.res else None
        y = _conv_ref(x, w16, bias, l.stride, l.act, slope, res, l.flags & 5)
        tens[l.dst] = y.astype(np.float32 if (l.flags & ns.FLAG_OUT_F32) else np.float16)
    return [tens[n] for n in want]


@pytest.mark.parametrize("shape", [(2, 150, 200), (1, 97, 131)])
def test_detector_program_matches_fp16_emulation(engine, shape):
    """head maps of the HIP detector vs the same folded fp16 program emulated layer by layer (fp16 storage between
    layers, fp32 accumulate): isolates kernel arithmetic from fp16 quantisation.  Bound: 4 fp16 ulps of the map's
    scale (an ulp flip of one intermediate activation perturbs downstream sums by ~2^-11 of one term)."""
    from frp_amd import netspec as ns
    B, H, W = shape
    Hc, Wc = (H + 31) // 32 * 32, (W + 31) // 32 * 32
    rng = np.random.default_rng(77 + H)
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    engine.load_weights(blob)
    frames = rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)
    engine.detect(frames, max_faces=4, det_thresh=0.5)
    heads = engine.head_maps()
    canvas = np.zeros((B, Hc, Wc, 3), np.uint8)
    canvas[:, :H, :W] = frames
    x = np.zeros((B, Hc, Wc, 8), np.float16)
    x[..., :3] = ((canvas[..., ::-1].astype(np.float32) - 127.5) / 128.0).astype(np.float16)     # exact in fp16
    emu = _emulate_program_fp16(raw, ns.detector_layers((1, 2, 2, 2)), x, ["det.out3", "det.out4", "det.out5"])
    for g, r in zip(heads, emu):
        g32, r32 = g.astype(np.float32), r.astype(np.float32)
        scale = max(1.0, float(np.abs(r32).max()))
        assert np.abs(g32 - r32).max() <= 4 * 2.0 ** -10 * scale, float(np.abs(g32 - r32).max() / scale)
        assert np.abs(g32 - r32).mean() <= 2.0 ** -12 * scale         # typical deviation: well below one ulp of the scale


def test_decode_threshold_one_ulp_and_exact_iou_tie(engine):
    """(1) the score threshold placed BETWEEN two adjacent fp16 logit values: the upper one is a candidate, the lower
    one is not - on the device exactly as in the oracle (no fp16/fp32 conversion slack);  (2) two boxes whose IoU
    is exactly the NMS threshold (80 / 200 with +1 areas, nms_iou 0.4): 'suppress if IoU > threshold' keeps both."""
    def head_set():
        return [np.zeros((1, 64 // s, 64 // s, 32), np.float16) for s in (8, 16, 32)]
    # (1) two far-apart anchors with logits L and prev(L)
    L = np.float16(1.25)
    Lm = np.nextafter(L, np.float16(-10), dtype=np.float16)
    h = head_set()
    for hm in h:
        hm[..., 0] = -9.0
        hm[..., 15] = -9.0
        hm[..., 1:5] = 0.1
        hm[..., 16:20] = 0.1
    h[0][0, 1, 1, 0] = L
    h[0][0, 5, 5, 0] = Lm
    mid = (float(L) + float(Lm)) / 2
    thr = 1.0 / (1.0 + np.exp(-mid))
    o = engine.decode_heads(h, (64, 64), max_faces=4, det_thresh=thr, nms_iou=0.4)
    ob, _, _, oa = onet.decode_nms([x[0] for x in h], np.float32(thr), 0.4, 4)
    assert list(oa) == [2 * (1 * 8 + 1)] and o["counts"][0] == 1 and o["anchor_idx"][0, 0] == oa[0]
    # threshold exactly AT the lower value: '>=' admits it
    thr2 = float(np.float32(1.0 / (1.0 + np.exp(-float(Lm)))))
    o2 = engine.decode_heads(h, (64, 64), max_faces=4, det_thresh=thr2, nms_iou=0.4)
    ob2, _, _, oa2 = onet.decode_nms([x[0] for x in h], np.float32(thr2), 0.4, 4)
    assert o2["counts"][0] == len(oa2) and np.array_equal(o2["anchor_idx"][0, :len(oa2)], oa2)
    # (2) boxes [0,0,9,13] and [0,6,9,19]: inter 10 x 8 = 80, union 140 + 140 - 80 = 200 -> IoU = 0.4f exactly
    h = head_set()
    for hm in h:
        hm[..., 0] = -9.0
        hm[..., 15] = -9.0
    h[0][0, 1, 1, 0:5] = [3.0, 1.0, 1.0, 0.125, 0.625]        # centre (8, 8): l, t, r, b in strides
    h[0][0, 2, 1, 0:5] = [2.0, 1.0, 1.25, 0.125, 0.375]       # centre (8, 16)
    for iou, want in ((0.4, 2), (0.399, 1)):
        o = engine.decode_heads(h, (64, 64), max_faces=4, det_thresh=0.5, nms_iou=iou)
        ob, _, _, oa = onet.decode_nms([x[0] for x in h], 0.5, iou, 4)
        assert len(oa) == want and o["counts"][0] == want
        assert np.array_equal(o["anchor_idx"][0, :want], oa) and np.array_equal(o["boxes"][0, :want], ob)
    assert np.array_equal(ob[0], np.array([0, 0, 9, 13], np.float32))


@pytest.mark.selfcheck
def test_fused_stems_agree_with_the_two_kernel_path(engine, monkeypatch):
    """stem12_u8_kernel vs stem_u8_kernel + generic conv: same fp16 stem1 values, fp32 accumulation in a
    different order -> head maps equal up to fp16 rounding noise; BGR and RGB inputs give the same result"""
    rng = np.random.default_rng(5)
    raw, blob = get_raw_and_blob((1, 2, 2, 2), (1, 1, 1, 1))
    engine.load_weights(blob)
    frames = rng.integers(0, 256, size=(2, 200, 328, 3), dtype=np.uint8)
    engine.detect(frames, max_faces=4, det_thresh=0.5)
    a = [h.astype(np.float32) for h in engine.head_maps()]
    engine.detect(frames[..., ::-1].copy(), max_faces=4, det_thresh=0.5, flags=2)       # FLAG_RGB
    a_rgb = [h.astype(np.float32) for h in engine.head_maps()]
    monkeypatch.setenv("FRP_NO_FUSED_STEM12", "1")
    engine.detect(frames, max_faces=4, det_thresh=0.5)
    b = [h.astype(np.float32) for h in engine.head_maps()]
    for x, xr, y in zip(a, a_rgb, b):
        assert np.array_equal(x, xr)
        scale = max(1.0, float(np.abs(y).max()))
        assert np.abs(x - y).max() < 4e-3 * scale


@pytest.mark.parametrize("N,M,k", [(1000, 5, 7), (4097, 3, 64), (5, 2, 8), (100000, 4, 10)])
def test_match_topk_parity(engine, N, M, k):
    """device top-k (a12, find_k_nearest): same rows in the same order as the oracle's stable argsort
    computed on the fp16-rounded operands the device holds; duplicate rows exercise the tie rule."""
    rng = np.random.default_rng(N + 31 * k)
    G = rng.standard_normal((N, 512)).astype(np.float32)
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    if N >= 1000:
        G[700] = G[3]                       # exact duplicates: the lower row must come first
        G[701] = G[3]
    Q = G[rng.integers(0, N, size=M)] + 0.3 * rng.standard_normal((M, 512)).astype(np.float32) / np.sqrt(512)
    if N >= 1000:
        Q[0] = G[3]
    engine.gallery_set(G)
    idx, cos = engine.match(Q, topk=k)
    assert idx.shape == (M, k) and cos.shape == (M, k)
    S = engine.match_scores(Q)              # the device's own score matrix: selection must be exact on it
    kk = min(k, N)
    oidx = np.argsort(-S.astype(np.float64), axis=1, kind="stable")[:, :kk]
    assert np.array_equal(idx[:, :kk], oidx)
    assert np.array_equal(cos[:, :kk], np.take_along_axis(S, oidx, axis=1))
    assert np.all(idx[:, kk:] == -1) and np.all(cos[:, kk:] == -2.0)
    # and against the fp64 oracle on the original operands: same identities where the margin exceeds fp16 noise
    o2, c2 = onet.match_topk(G, Q / np.linalg.norm(Q, axis=1, keepdims=True), kk)
    assert np.abs(cos[:, :kk] - c2).max() < 3e-3
    i1, c1 = engine.match(Q)                # top-1 path agrees with column 0
    assert np.array_equal(i1, idx[:, 0])
    if N >= 1000:
        assert list(idx[0, :3]) == [3, 700, 701]


@pytest.mark.selfcheck
@pytest.mark.parametrize("N,M", [(1, 1), (31, 5), (1000, 33), (4097, 320), (70001, 512), (300, 513)])
def test_match_running_best_kernel_equals_per_tile_kernel(engine, monkeypatch, N, M):
    """the persistent top-1 kernel (running winners in registers, M <= 512) and the per-tile kernel (FRP_MATCH_V1=1; also
    what M = 513 falls back to) return the same bits: same winner, same cosine, ties to the lower row - including
    galleries smaller than one block, ragged last blocks and duplicate rows spread over blocks, waves and workgroups"""
    rng = np.random.default_rng(N * 1000 + M)
    G = rng.standard_normal((N, 512)).astype(np.float32)
    if N > 64:
        G[N - 1] = G[7]                      # duplicates far apart: the lower row must win
        G[N // 2] = G[7]
    engine.gallery_set(G)
    Q = rng.standard_normal((M, 512)).astype(np.float32)
    if N > 64:
        Q[0] = G[7]
    a_idx, a_cos = engine.match(Q)
    monkeypatch.setenv("FRP_MATCH_V1", "1")
    b_idx, b_cos = engine.match(Q)
    assert np.array_equal(a_idx, b_idx) and np.array_equal(a_cos.view(np.uint32), b_cos.view(np.uint32))
    if N > 64:
        assert a_idx[0] == 7
    g16 = engine.gallery_get().astype(np.float64)
    qn = Q.astype(np.float64) / np.linalg.norm(Q.astype(np.float64), axis=1, keepdims=True)
    S = qn.astype(np.float16).astype(np.float64) @ g16.T
    assert np.abs(a_cos - S.max(1)).max() < 1e-3


def test_match_topk_rejects_bad_k(engine):
    from frp_amd.native import FrpError
    engine.gallery_set(np.eye(4, 512, dtype=np.float32))
    with pytest.raises(FrpError):
        engine.match(np.ones((1, 512), np.float32), topk=0)
    with pytest.raises(FrpError):
        engine.match(np.ones((1, 512), np.float32), topk=65)


def test_embedder_large_batch_equals_small_batch_and_oracle(engine, monkeypatch):
    """IResNet-100 on 330 faces (every stage spans several tiles per workgroup: XCD-interleaved walk, ragged last
    tiles) gives, face by face, the result of a 3-face call on the same tile class up to the fp32 summation order of the
    FC (its split-K factor depends on the batch), which the fp32 oracle confirms.  Left to itself a call of fewer than 128
    faces takes the DIRECT kernels also where the big batch runs the Winograd kernel (28 x 28, 14 x 14): the two
    families agree to 1 - cos <= 2e-5 (measured 1.6e-6), three orders of magnitude inside the 1e-3 bar."""
    rng = np.random.default_rng(404)
    chips = rng.integers(0, 256, size=(330, 112, 112, 3), dtype=np.uint8)
    raw, blob = get_raw_and_blob((1, 1, 1, 1), (3, 13, 30, 3))
    engine.load_weights(blob)
    big = engine.embed_aligned(chips)
    pick = [0, 151, 329]
    monkeypatch.setenv("FRP_WINO_MIN_FACES", "1")                 # the big batch's kernel family for the three faces
    small = engine.embed_aligned(chips[pick])
    monkeypatch.delenv("FRP_WINO_MIN_FACES")
    assert np.abs(big[pick] - small).max() < 1e-6
    few = engine.embed_aligned(chips[pick])                       # the default for three faces: direct kernels, quarter tiles
    sixty = engine.embed_aligned(chips[:64])                      # ... as for any call below FRP_WINO_MIN_FACES = 128 slots,
    assert np.abs(sixty[pick[0]] - few[0]).max() < 1e-6           # whatever tile sizes its layers take
    assert not np.array_equal(few, small)
    assert 1 - (few * small).sum(1).min() <= 2e-5
    ref = onet.emb_forward(raw, onet.emb_blob(chips[pick]))
    assert (small * ref).sum(1).min() > 1 - 1e-3 and (few * ref).sum(1).min() > 1 - 1e-3
    assert np.abs(np.linalg.norm(big, axis=1) - 1).max() < 1e-4


F8_CASES = [
    # N, H, W, Cin, Cout, act, res, flags(border), out_fp8, copy_fp8
    (2, 14, 14, 128, 128, 2, False, 1, True, False),     # conv1 of an IResNet block: PReLU + border bias, fp8 out
    (3, 14, 14, 256, 256, 0, True, 0, False, True),      # conv2: residual, fp16 out + fp8 copy, two channel blocks, two cout tiles
    (1, 7, 7, 512, 512, 2, False, 1, True, False),       # 4 channel blocks, 4 cout tiles, ragged pixel tile
    (5, 28, 28, 128, 128, 0, True, 0, False, True),      # several pixel tiles
    (1, 5, 9, 128, 96, 1, False, 0, False, False),       # ragged cout tile, fp16 out only
]


@pytest.mark.parametrize("case", F8_CASES)
def test_conv_fp8_mfma_parity(engine, case):
    """BASELINE config 5: the block-scaled fp8 MFMA conv (v_mfma_scale_f32_32x32x64_f8f6f4, E4M3 operands, unit block
    scales) against an fp32 reference on the DEQUANTISED operands.  Products of two E4M3 values are exact in fp32, so
    the two differ by fp32 summation order only: fp16 outputs to output rounding (2e-3 of the scale), fp8 outputs to
    the same E4M3 code except where the value sits on a rounding boundary (<= 1 code step, rarely)."""
    from frp_amd import weights as wts
    N, H, W, Cin, Cout, act, has_res, flags, out8, copy8 = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    xq = wts.fp8_e4m3_encode(rng.standard_normal((N, H, W, Cin)).astype(np.float32))
    x = wts.FP8_E4M3[xq]
    w16 = (rng.standard_normal((Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float16)
    wq, wscale = wts.fp8_quantize_rows(w16)
    wq = wq.reshape(Cout, 3, 3, Cin)
    wdq = wts.FP8_E4M3[wq] * wscale[:, None, None, None]
    bias = rng.standard_normal((9, Cout) if flags & 1 else (Cout,)).astype(np.float32) * 0.3
    slope = rng.uniform(0.1, 0.4, Cout).astype(np.float32) if act == 2 else None
    res = rng.standard_normal((N, H, W, Cout)).astype(np.float16) if has_res else None
    in_scale, out_scale = 0.5, 0.25
    got = engine.conv2d_f8(xq, wq, wscale, bias, act=act, slope=slope, res=res, flags=flags, in_scale=in_scale,
                           out_scale=out_scale, out_fp8=out8, copy_fp8=copy8)
    ref = _conv_ref((x * in_scale).astype(np.float32), wdq.astype(np.float32), bias, 1, act, slope, res, flags)
    scale = max(1.0, float(np.abs(ref).max()))

    def check_fp8(codes):
        want = wts.fp8_e4m3_encode(np.clip(ref / out_scale, -448, 448))
        gv, wv = wts.FP8_E4M3[codes], wts.FP8_E4M3[want]
        assert np.mean(codes == want) > 0.995
        step = np.maximum(np.abs(wv), 2.0 ** -6) * 2.0 ** -3        # one E4M3 code step at the value's binade
        assert np.all(np.abs(gv - wv) <= step)

    if out8:
        check_fp8(got)
    else:
        out16, out2 = got if copy8 else (got, None)
        assert np.abs(out16.astype(np.float32) - ref).max() <= 2e-3 * scale
        if copy8:
            check_fp8(out2)


WINO_CASES = [
    # N, H, W, Cin, Cout, act, res, flags(border)
    (3, 14, 14, 256, 256, 2, False, 1),      # IResNet stage 3 conv1: border-class bias + PReLU, 4 channel blocks, 2 cout tiles
    (5, 14, 14, 256, 256, 0, True, 0),       # conv2: residual; 980 pixels = 3 full tiles + a ragged one
    (2, 28, 28, 128, 128, 2, True, 1),       # stage 2, everything on
    (1, 28, 28, 128, 256, 1, False, 0),      # layer3.0.conv1 shape (128 -> 256), ReLU
    (40, 14, 14, 64, 64, 0, False, 0),       # one channel block, Cout = 64: half of the cout tile is padding
    (1, 2, 2, 64, 128, 0, False, 1),         # 2 x 2 map: every pair touches both borders
    (7, 6, 30, 192, 160, 2, True, 1),        # widest map the LDS holds (W = 30), ragged cout tile, 3 channel blocks
    (33, 4, 4, 128, 128, 1, True, 0),        # many tiny images per tile
    (1, 16, 16, 64, 64, 0, False, 0),        # exactly one tile
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv_winograd_parity(engine, case):
    """conv3x3_wino.hip (Winograd F(2,3) along the rows: transformed fp16 operands, fp32 accumulation) against the fp32
    reference on the fp16-rounded operands, and against the direct kernel: same bar as the direct kernels (2e-3 of the
    output scale: fp16 output rounding dominates), and within 3 fp16 ulps of the scale of the direct kernel's output."""
    N, H, W, Cin, Cout, act, has_res, flags = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 7)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
    w = (rng.standard_normal((Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float16)
    bias = rng.standard_normal((9, Cout) if flags & 1 else (Cout,)).astype(np.float32) * 0.3
    slope = rng.uniform(0.1, 0.4, Cout).astype(np.float32) if act == 2 else None
    res = rng.standard_normal((N, H, W, Cout)).astype(np.float16) if has_res else None
    direct = engine.conv2d(x, w, bias, stride=1, act=act, slope=slope, res=res, flags=flags | TILES_DEFAULT)
    wino = engine.conv2d(x, w, bias, stride=1, act=act, slope=slope, res=res, flags=flags | 0x10000)
    ref = _conv_ref(x, w, bias, 1, act, slope, res, flags)
    scale = max(1.0, float(np.abs(ref).max()))
    err = np.abs(wino.astype(np.float32) - ref).max()
    assert err <= 2e-3 * scale, (err, scale)
    assert np.abs(wino.astype(np.float32) - direct.astype(np.float32)).max() <= 3 * 2.0 ** -10 * scale
    # deterministic: a second launch gives the same bits
    again = engine.conv2d(x, w, bias, stride=1, act=act, slope=slope, res=res, flags=flags | 0x10000)
    assert np.array_equal(wino.view(np.uint16), again.view(np.uint16))


WINO_WIDE_CASES = [
    # N, H, W, Cin, Cout, act, res, flags(border): maps wider than the flattened tiles cover -> the kernel's 2-D tiles (8 x 30)
    (4, 136, 240, 128, 128, 1, False, 0),    # det.layer2 at 1080p, four frames: the smallest call that takes them (two rounds of tiles)
    (8, 68, 120, 256, 256, 1, True, 1),      # det.layer3 shape + residual: ragged tile rows (68 = 8 x 8 + 4), two cout tiles
    (11, 46, 118, 256, 160, 2, True, 1),     # ragged rows (46 = 5 x 8 + 6), columns (118 = 3 x 30 + 28) and couts, PReLU + residual + border classes
]


@pytest.mark.parametrize("case", WINO_WIDE_CASES)
def test_conv_winograd_2d_tiles_parity(engine, case):
    """The Winograd kernel on maps wider than 30 pixels (round 4: tiles of 8 rows x 30 columns of one image - the flattened kernel on
    a 32-wide virtual strip; conv3x3_wino.hip: VAR & 32): same bars as the flattened form - 2e-3 of the output scale against
    the fp32 reference, 3 fp16 ulps of the scale against the direct kernel, the same bits on a second launch."""
    N, H, W, Cin, Cout, act, has_res, flags = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 11)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
    w = (rng.standard_normal((Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float16)
    bias = rng.standard_normal((9, Cout) if flags & 1 else (Cout,)).astype(np.float32) * 0.3
    slope = rng.uniform(0.1, 0.4, Cout).astype(np.float32) if act == 2 else None
    res = rng.standard_normal((N, H, W, Cout)).astype(np.float16) if has_res else None
    direct = engine.conv2d(x, w, bias, stride=1, act=act, slope=slope, res=res, flags=flags | TILES_DEFAULT)
    wino = engine.conv2d(x, w, bias, stride=1, act=act, slope=slope, res=res, flags=flags | 0x10000)
    ref = _conv_ref(x, w, bias, 1, act, slope, res, flags)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(wino.astype(np.float32) - ref).max() <= 2e-3 * scale
    d = np.abs(wino.astype(np.float32) - direct.astype(np.float32))
    assert d.max() <= 3 * 2.0 ** -10 * scale
    assert d.max() > 0                                                   # (it IS the other kernel family)
    again = engine.conv2d(x, w, bias, stride=1, act=act, slope=slope, res=res, flags=flags | 0x10000)
    assert np.array_equal(wino.view(np.uint16), again.view(np.uint16))
    # a call of fewer than two rounds of tiles is refused (the engine takes the direct family there), not silently rerouted
    with pytest.raises(Exception):
        engine.conv2d(x[:1], w, bias, stride=1, act=act, slope=slope, res=None if res is None else res[:1], flags=flags | 0x10000)


@pytest.mark.selfcheck
def test_conv_winograd_ring_stress(engine):
    """The Winograd kernel's rings under timing pressure (round 4: the weight ring runs three sub-steps ahead and a slot is
    restaged by the sub-step that computes on the fragments read from it; round 3's review: no stress configuration of its
    own).  Several tiles per workgroup with 1, 3 and 4 channel blocks, with and without residual, ragged last tiles; 25
    launches each while a second handle keeps the chip busy with other kernels on another stream (the hardware scheduler
    then interleaves the workgroups of both and shifts every DMA's landing time): every launch returns the bits of the
    first one, which agree with the direct kernel to 3 fp16 ulps of the scale."""
    import threading
    from frp_amd import native
    other = native.Engine(0)
    stop = threading.Event()
    rng = np.random.default_rng(77)
    xo = rng.standard_normal((9, 56, 56, 64)).astype(np.float16)
    wo = (rng.standard_normal((64, 3, 3, 64)) / 24).astype(np.float16)
    bo = np.zeros(64, np.float32)

    def noise():
        while not stop.is_set():
            other.conv2d(xo, wo, bo, flags=TILES_DEFAULT)
    th = threading.Thread(target=noise, daemon=True)
    th.start()
    try:
        for (N, H, W, Cin, Cout, act, has_res, flags) in ((600, 14, 14, 256, 256, 2, False, 1), (333, 14, 14, 192, 128, 0, True, 0),
                                                           (170, 28, 28, 64, 256, 1, True, 0), (41, 30, 30, 128, 136, 2, False, 1),
                                                           (8, 68, 120, 256, 256, 1, True, 0)):       # (the last: the 2-D tiles)
            x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
            w = (rng.standard_normal((Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float16)
            bias = rng.standard_normal((9, Cout) if flags & 1 else (Cout,)).astype(np.float32) * 0.3
            slope = rng.uniform(0.1, 0.4, Cout).astype(np.float32) if act == 2 else None
            res = rng.standard_normal((N, H, W, Cout)).astype(np.float16) if has_res else None
            first = engine.conv2d(x, w, bias, act=act, slope=slope, res=res, flags=flags | 0x10000)
            direct = engine.conv2d(x, w, bias, act=act, slope=slope, res=res, flags=flags | TILES_DEFAULT)
            scale = max(1.0, float(np.abs(direct.astype(np.float32)).max()))
            assert np.abs(first.astype(np.float32) - direct.astype(np.float32)).max() <= 3 * 2.0 ** -10 * scale
            for _ in range(25):
                got = engine.conv2d(x, w, bias, act=act, slope=slope, res=res, flags=flags | 0x10000)
                assert np.array_equal(first.view(np.uint16), got.view(np.uint16))
    finally:
        stop.set()
        th.join()
        other.close()


def test_conv_winograd_rejects_what_it_does_not_cover(engine):
    from frp_amd.native import FrpError
    for shape, k, stride in (((1, 7, 7, 64), 3, 1), ((1, 8, 8, 32), 3, 1), ((1, 8, 8, 64), 1, 1), ((1, 8, 8, 64), 3, 2), ((1, 8, 63, 64), 3, 1), ((1, 8, 64, 64), 3, 1)):
        x = np.zeros(shape, np.float16)
        w = np.zeros((64, k, k, shape[3]), np.float16)
        with pytest.raises(FrpError):
            engine.conv2d(x, w, np.zeros(64, np.float32), stride=stride, flags=0x10000)


def test_embedder_winograd_layers_vs_direct_and_oracle(engine, monkeypatch):
    """IResNet-100 with its 28 x 28 and 14 x 14 stride-1 3x3 convs on the Winograd kernel (the default) against the same
    program on the direct kernels only (FRP_NO_WINO=1) and against the fp32 oracle: north-star bar 1 - 1e-3 against the
    oracle, and 1 - cos <= 2e-5 between the two kernel families (CPU emulation of both: tools/winograd_numerics.py)."""
    rng = np.random.default_rng(12)
    chips = rng.integers(0, 256, size=(5, 112, 112, 3), dtype=np.uint8)
    raw, blob = get_raw_and_blob((1, 1, 1, 1), (3, 13, 30, 3))
    engine.load_weights(blob)
    engine.reset_counters()
    # the kernel family and tile class of big batches (five chips alone take the direct kernels in quarter tiles)
    monkeypatch.setenv("FRP_WINO_MIN_FACES", "1")
    monkeypatch.setenv("FRP_SMALL_M", "0")
    wino = engine.embed_aligned(chips)
    monkeypatch.setenv("FRP_NO_WINO", "1")
    engine.load_weights(blob)
    direct = engine.embed_aligned(chips)
    monkeypatch.delenv("FRP_NO_WINO")
    engine.load_weights(blob)
    assert not np.array_equal(wino, direct)                       # the two paths really are different kernels
    # few faces: quarter tiles of the DIRECT kernels (same k order as their default tiles) - bit for bit the direct result
    monkeypatch.delenv("FRP_SMALL_M")
    monkeypatch.delenv("FRP_WINO_MIN_FACES")
    few = engine.embed_aligned(chips)
    assert np.array_equal(few, direct)
    monkeypatch.setenv("FRP_SMALL_M", "1")
    assert np.array_equal(engine.embed_aligned(chips), direct)
    monkeypatch.setenv("FRP_SMALL_M", "0")                        # direct family, default tiles
    assert np.array_equal(engine.embed_aligned(chips), direct)
    monkeypatch.delenv("FRP_SMALL_M")
    ref = onet.emb_forward(raw, onet.emb_blob(chips))
    assert (wino * ref).sum(1).min() > 1 - 1e-3 and (direct * ref).sum(1).min() > 1 - 1e-3
    assert 1 - (wino * direct).sum(1).min() <= 2e-5
    print("1 - cos: winograd vs oracle", 1 - (wino * ref).sum(1).min(), "direct vs oracle", 1 - (direct * ref).sum(1).min(),
          "winograd vs direct", 1 - (wino * direct).sum(1).min())


def test_shortcut_k_concat_equals_separate_launches(engine, monkeypatch):
    """K-concat (conv_mfma.hip): the 1x1 stride-2 shortcut conv of every strided IResNet block rides in the k-loop of the
    block's 3x3 stride-2 conv (its input at the centre tap as a second K segment, weights concatenated and biases summed at
    load time) - four launches fewer, the same algorithmic FLOPs on the counters, the shortcut sum kept in fp32 instead of
    being rounded to fp16 on its way through HBM: equal to the separate launches (FRP_NO_KCONCAT=1) to 1 - cos <= 1e-5 and
    as close to the fp32 oracle; 4 faces (quarter tiles) and 70 faces (default tiles where a launch has enough of them); Cin = 2 Cin2 and Cin = Cin2."""
    rng = np.random.default_rng(21)
    chips = rng.integers(0, 256, size=(70, 112, 112, 3), dtype=np.uint8)
    raw, blob = get_raw_and_blob((1, 1, 1, 1), (2, 2, 2, 2))
    ref = onet.emb_forward(raw, onet.emb_blob(chips[:4]))
    monkeypatch.setenv("FRP_NO_STEM_FUSE", "1")        # (the stem fusion of round 5 saves a launch of its own, only next to K-concat: its own test)
    out = {}
    for mode in ("fused", "separate"):
        if mode == "separate":
            monkeypatch.setenv("FRP_NO_KCONCAT", "1")
        engine.load_weights(blob)
        for n in (4, 70):
            engine.reset_counters()
            e = engine.embed_aligned(chips[:n])
            c = engine.counters()
            out[mode, n] = (e, c["emb_conv_launches"], c["emb_conv_flops"])
    monkeypatch.delenv("FRP_NO_KCONCAT")
    engine.load_weights(blob)
    for n in (4, 70):
        (a, la, fa), (b, lb, fb) = out["fused", n], out["separate", n]
        assert la == lb - 4 and abs(fa - fb) <= 1e-6 * fb
        assert not np.array_equal(a, b)
        assert 1 - (a * b).sum(1).min() <= 1e-5
    for mode in ("fused", "separate"):
        assert (out[mode, 4][0] * ref).sum(1).min() > 1 - 1e-3
    # a blob with the buffer plan of older packers (a strided block's output may reuse the block input's buffer): the
    # runtime's precondition check leaves those blocks unfused - the separate launches' bits, none fewer
    from frp_amd import weights
    monkeypatch.setattr(weights, "KCONCAT_LIVENESS", False)
    old_blob = weights.pack_blob(raw, (1, 1, 1, 1), (2, 2, 2, 2))
    monkeypatch.setattr(weights, "KCONCAT_LIVENESS", True)
    engine.load_weights(old_blob)
    engine.reset_counters()
    e = engine.embed_aligned(chips[:4])
    assert np.array_equal(e, out["separate", 4][0]) and engine.counters()["emb_conv_launches"] == out["separate", 4][1]
    engine.load_weights(blob)
    assert np.abs(out["fused", 70][0][:4] - out["fused", 4][0]).max() < 1e-6     # (same family, other tile sizes; the FC's split-K factor differs)


def test_embedder_stem_fused_into_the_first_64_channel_conv_is_bit_identical(engine, monkeypatch):
    """the embedder's stem conv (3 -> 64 + PReLU) computed INSIDE the launch of the 64 -> 64 conv behind it (conv3x3_c64.hip, STEM: the
    chips under every patch by LDS-DMA, a short MFMA phase writes the 64-channel patch into LDS and the tile's own pixels of the map
    to HBM for the block's shortcut): the same instructions on the same operands as emb_stem_kernel followed by the unfused conv -
    embeddings bit for bit those of the two launches (FRP_NO_STEM_FUSE=1), one launch fewer, the same FLOPs on the counters; below the
    kernel's two rounds of tiles (here: 8 faces) nothing is fused.  Repeated calls return the same bits."""
    rng = np.random.default_rng(29)
    chips = rng.integers(0, 256, size=(37, 112, 112, 3), dtype=np.uint8)
    raw, blob = get_raw_and_blob((1, 1, 1, 1), (2, 1, 1, 1))
    engine.load_weights(blob)
    out = {}
    for mode in ("fused", "separate"):
        if mode == "separate":
            monkeypatch.setenv("FRP_NO_STEM_FUSE", "1")
        for n in (37, 8):
            engine.reset_counters()
            e = engine.embed_aligned(chips[:n])
            c = engine.counters()
            out[mode, n] = (e, c["emb_conv_launches"], c["emb_conv_flops"])
    monkeypatch.delenv("FRP_NO_STEM_FUSE")
    (a, la, fa), (b, lb, fb) = out["fused", 37], out["separate", 37]
    assert la == lb - 1 and abs(fa - fb) <= 1e-6 * fb
    assert np.array_equal(a, b)
    assert out["fused", 8][1] == out["separate", 8][1] and np.array_equal(out["fused", 8][0], out["separate", 8][0])
    for _ in range(5):
        assert np.array_equal(engine.embed_aligned(chips), a)
    ref = onet.emb_forward(raw, onet.emb_blob(chips[:4]))
    assert (a[:4] * ref).sum(1).min() > 1 - 1e-3


def test_stride2_row_patch_kernel_in_the_embedder_with_the_shortcut_segments(engine, monkeypatch):
    """the opt-in stride-2 kernel (FRP_S2=1) inside IResNet: its three eligible strided convs (128, 256, 512 channels) carry the block's
    1x1 shortcut as a second K segment at the centre tap (ConvParams::x2 - the path the single-conv hook cannot reach).  Default
    tiles for both runs (FRP_SMALL_M=0: 70 faces would otherwise put stage 3 / 4 on quarter tiles, which the kernel does not
    have).  Equal to the per-tap kernel to 1 - cos <= 1e-5, not the same bits (another k order: the kernel did run), as close to
    the fp32 oracle."""
    rng = np.random.default_rng(23)
    chips = rng.integers(0, 256, size=(70, 112, 112, 3), dtype=np.uint8)
    raw, blob = get_raw_and_blob((1, 1, 1, 1), (2, 2, 2, 2))
    monkeypatch.setenv("FRP_SMALL_M", "0")
    engine.load_weights(blob)
    base = engine.embed_aligned(chips)
    monkeypatch.setenv("FRP_S2", "1")
    s2 = engine.embed_aligned(chips)
    again = engine.embed_aligned(chips)
    monkeypatch.delenv("FRP_S2")
    assert np.array_equal(engine.embed_aligned(chips), base)
    assert np.array_equal(s2, again)
    assert not np.array_equal(s2, base)
    assert 1 - (s2 * base).sum(1).min() <= 1e-5
    ref = onet.emb_forward(raw, onet.emb_blob(chips[:4]))
    assert (s2[:4] * ref).sum(1).min() > 1 - 1e-3
    print("1 - cos: stride-2 row-patch kernel vs per-tap kernel", 1 - (s2 * base).sum(1).min(), "vs oracle", 1 - (s2[:4] * ref).sum(1).min())


def test_gallery_reserve_commit_zero_copy_import(engine):
    """frp_gallery_reserve / _commit / _device_ptr (the RCCL all-gather lands in the snapshot the engine reserved,
    dist.allgather_gallery_into_engine): rows written into the reserved buffer by another producer on the GPU (torch here)
    become the gallery at commit, a second handle copies its snapshot from the first one's device pointer; while a
    reservation is pending the other gallery updates are refused and change nothing; cancel discards it"""
    from frp_amd import dist as fdist, native
    from frp_amd.native import FrpError
    rng = np.random.default_rng(15)
    rows = fdist.normalize_rows_f16(rng.standard_normal((777, 512)).astype(np.float32))
    engine.gallery_set(rng.standard_normal((5, 512)).astype(np.float32))
    ptr = engine.gallery_reserve(1000)                       # capacity beyond the rows that will be committed
    assert engine.gallery_size() == 5                        # not visible yet
    t = torch.as_tensor(fdist._DevicePtr(ptr, 1000, 512), device=torch.device("cuda", 0))
    t[:777].copy_(torch.from_numpy(rows))
    torch.cuda.synchronize()
    engine.gallery_commit(777)
    assert engine.gallery_size() == 777 and np.array_equal(engine.gallery_get(0, 777), rows)
    q = rows[[3, 500, 776]].astype(np.float32)
    idx, cos = engine.match(q)
    assert idx.tolist() == [3, 500, 776] and np.abs(cos - 1).max() < 2e-3
    e2 = native.Engine(0)
    e2.gallery_set_device(engine.gallery_device_ptr(), 777)
    assert np.array_equal(e2.gallery_get(0, 777), rows)
    e2.close()
    ptr2 = engine.gallery_reserve(64)
    for bad in (lambda: engine.gallery_update_row(0, q[0]), lambda: engine.gallery_remove_row(0),
                lambda: engine.gallery_set(q), lambda: engine.gallery_set_device(engine.gallery_device_ptr(), 5)):
        with pytest.raises(FrpError):                        # someone (RCCL) may be writing into the reservation: updates wait
            bad()
    assert engine.gallery_size() == 777 and np.array_equal(engine.gallery_get(0, 777), rows)      # nothing changed
    t2 = torch.as_tensor(fdist._DevicePtr(ptr2, 64, 512), device=torch.device("cuda", 0))
    t2[:10].copy_(torch.from_numpy(rows[100:110]))
    torch.cuda.synchronize()
    engine.gallery_commit(10)                                # ... and the reservation survived them
    assert engine.gallery_size() == 10 and np.array_equal(engine.gallery_get(0, 10), rows[100:110])
    engine.gallery_reserve(64)
    engine.gallery_cancel()
    engine.gallery_cancel()                                  # (no-op without a reservation)
    with pytest.raises(FrpError):
        engine.gallery_commit(5)                             # nothing to commit
    engine.gallery_update_row(0, q[0])                       # updates work again
    engine.gallery_set(np.zeros((0, 512), np.float32))
