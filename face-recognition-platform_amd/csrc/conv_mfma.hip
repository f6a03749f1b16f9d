// K2: implicit-GEMM convolution (3x3 / 1x1, stride 1 / 2) on CDNA4 matrix cores.
//
// Replaces the arithmetic the reference delegates to dlib / onnxruntime behind
// face_recognition.face_locations / face_encodings (backend/app/routes/camera.py:232,237,
// backend/app/services/face_service.py:156,179) for the RetinaFace-style detector and
// the ArcFace IResNet embedder (layer tables: ../netspec.py).
//
// Data layout (HBM): activations NHWC fp16, weights [Cout][kh][kw][Cin] fp16 (K-major
// for both MFMA operands), bias / PReLU slope fp32, accumulation fp32.
//
// GEMM view: D[cout][pixel] = sum_k Wt[cout][k] * X[pixel][k], k = (kh, kw, cin).  The
// weights are the MFMA A operand and the pixels the B operand, so an accumulator lane
// owns ONE output pixel and 4 consecutive output channels per register group: the
// epilogue (bias / 9-class border bias, residual, ReLU / PReLU) stores 8-byte fp16x4
// runs into the NHWC row of that pixel.
//
// Structure (persistent: one workgroup per CU walks a contiguous range of output tiles;
// 8 waves = 2 per SIMD, v_mfma_f32_32x32x16_f16):
//   * tile TP pixels x TC couts x 64 k per step; each wave owns a 128x64, 64x64 or 32x64 sub-tile;
//   * operands go HBM/L2 -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per
//     wave-instruction, no staging VGPRs) into a 2- or 3-slot ring that runs continuously
//     ACROSS tiles (the first stages of the next tile load during the tail and the epilogue
//     of the current one); the pieces of a stage are fired between the MFMA groups of the
//     running step, waits are counted (`s_waitcnt vmcnt(N)`), one raw `s_barrier` per k-step;
//   * zero padding, ragged pixel tiles and ragged cout tiles cost nothing: the per-lane
//     buffer offset is pushed out of range and the hardware range check writes zeros;
//   * the LDS image is rows of 128 B with chunk ^= (row>>1)&7 (conflict-free for the
//     ds_read_b128 fragment reads); the DMA destination is lane-linear, so the swizzle is
//     applied to the per-lane SOURCE chunk;
//   * persistent workgroups walk the tile list XCD-interleaved (neighbouring tiles, incl. the cout tiles of
//     one pixel tile, are in flight on the same L2 together), and k runs (64-channel block, kh, kw): the nine taps of a channel block
//     re-read nearly the same input lines within nine consecutive k-steps, so they hit in L2
//     (tap-major order thrashed the 4 MiB L2 on Cin >= 128: 2-10x the algorithmic reads);
//   * 3x3 stride-1 layers with Cin % 64 == 0 go to conv3x3_rows.hip (row patches) instead.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "frp_internal.h"
#include "conv_common.h"

namespace frp {

template <int TP, int TC, int WP, int WC, int NS, int NW, bool SMALL>
__global__ __launch_bounds__(NW * 64, TP == 128 ? 4 : 2) void conv_mfma_kernel(ConvParams p_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ConvParams p = p_in;
    if (p.n_dev) {                             // image count known on the device only (threshold mode)
        int n = *p.n_dev;
        n = n < 0 ? 0 : (n > p.N ? p.N : n);
        p.M = n * p.Ho * p.Wo;
        p.n_ptiles = (p.M + TP - 1) / TP;
        if (p.ksplit < 0) p.ksplit = conv_pick_ksplit(p.M, p.Cout, p.Ktot, p.flags, p.res != nullptr, p.n_cu);
    }
    constexpr int XB = TP * 128;              // bytes of one X stage
    constexpr int WB = TC * 128;
    constexpr int STAGE = XB + WB;
    constexpr int XI = TP / 8 / NW;      // X DMA instructions per wave per stage
    constexpr int WI = TC / 8 / NW;      // W DMA instructions per wave per stage
    constexpr int LPS = XI + WI;              // DMA instructions per wave per stage
    constexpr int MP = TP / WP / 32;          // MFMA tiles per wave (pixels)
    constexpr int MC = TC / WC / 32;          // MFMA tiles per wave (couts)
    constexpr int PRE = NS - 1;               // stages the DMA runs ahead of the MFMAs
    static_assert(WP * WC == NW && XI >= 1 && WI >= 1 && LPS <= 12 && (NS == 2 || NS == 3), "tile/wave layout");

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    // Persistent workgroup: a contiguous range of tiles (tile = ptile * n_ctiles + ctile, so the
    // cout tiles of one pixel tile follow each other on the same CU and L2).
    // With split-K (p.ksplit > 1, used for the FC whose M is tiny and K huge) a "tile" below is a
    // (tile, k-slice) pair: slice s covers k-steps [s*nk, (s+1)*nk) and writes raw fp32 partial
    // sums to a workspace slab that a finalize kernel reduces.
    const int n_tiles = p.n_ptiles * p.n_ctiles * p.ksplit;
    // XCD-interleaved walk (see conv3x3_rows.hip): the workgroups of one L2 (b, b+8, ...) share a contiguous
    // chunk of the tile list and walk it interleaved, so neighbouring tiles - same input rows, both cout tiles of
    // a pixel tile, the k-slices of one split-K tile - are in flight on the same L2 at the same time.
    const int G = (TP == 128 && p.n_workers) ? p.n_workers : (int)gridDim.x;      // workgroups that walk tiles (the others: conv_prefetch_weights)
    if constexpr (TP == 128) {
        if ((int)blockIdx.x >= G) { conv_prefetch_weights(p, (int)blockIdx.x - G, (int)gridDim.x - G, NW * 64); return; }
    }
    int t0, t1, tstep;
    if ((G & 7) == 0) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, per = G >> 3;
        const int cs = (int)((long)x * n_tiles / 8), ce = (int)((long)(x + 1) * n_tiles / 8);
        t0 = cs + j;
        t1 = ce;
        tstep = per;
    } else {
        t0 = (int)((long)blockIdx.x * n_tiles / G);
        t1 = (int)((long)(blockIdx.x + 1) * n_tiles / G);
        tstep = 1;
    }
    if (t0 >= t1) return;
    const int nk = p.nk / p.ksplit;            // k-steps per (tile, slice); launch_conv makes it exact

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    // K-concat: after the 9 * cpt (tap, channel block) stages of a tile come cpt2 stages of tensor x2 at the centre tap
    const __amdgpu_buffer_rsrc_t x2rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x), 0, p.x2 ? p.x2_bytes : 0u, 0x00020000);

    // ---------------- DMA lane state of the ISSUE cursor (it runs PRE stages ahead of the MFMAs
    // and crosses into the next tile while the current one is still being multiplied).
    // Instruction i of this wave fills LDS row group g = i*NW + wave (8 rows x 128 B); lane ->
    // row g*8 + lane/8, chunk position lane%8, which must hold logical chunk pos ^ ((row>>1)&7).
    // NW is even, so (row>>1)&7 is the same for every i: ((wave&1)<<2) | (lane>>4).
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));   // logical 16-B chunk of this lane
    const int HoWo = p.Ho * p.Wo;
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)p.Wo;
    int xoff[XI];          // byte offset of (n, iy0, ix0, c=0) + this lane's chunk; wraps at the border
    unsigned tapmask[XI];  // bit kh*3+kw set <=> that tap of this pixel row lies inside the image
    unsigned woff[WI];     // byte offset of (cout row, k=chunk); rows >= Cout are out of range -> zeros
    int kh = 0, kw = 0, cb = 0;               // aligned path: uniform (tap, 64-channel block) walk
    const int cpt = p.Cin >> 6;
    int it = t0, iks = 0, ibuf = 0;           // issue cursor: tile, k-step, ring slot
    int iks_base = 0;                         // first k-step of the issue cursor's slice
    auto setup_issue_tile = [&](int vtile) {
        const int tile = vtile / p.ksplit;
        iks_base = (vtile - tile * p.ksplit) * nk;
        const int ptile = tile / p.n_ctiles;
        const int m0i = ptile * TP;
        const int c0i = (tile - ptile * p.n_ctiles) * TC;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int m = m0i + (i * NW + wave) * 8 + lrow;
            unsigned mask = 0;
            int off = 0;
            if (m < p.M) {
                int n, rem, oy, ox;
                fast_divmod(m, HoWo, inv_howo, n, rem);
                fast_divmod(rem, p.Wo, inv_wo, oy, ox);
                const int y0 = oy * p.stride - p.pad, x0 = ox * p.stride - p.pad;
                off = (((n * p.H + y0) * p.W + x0) * p.Cin) * 2;
                unsigned ym = 0, xm = 0;      // 3-bit validity of y0+{0,1,2}, x0+{0,1,2}
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    ym |= ((unsigned)(y0 + d) < (unsigned)p.H ? 1u : 0u) << d;
                    xm |= ((unsigned)(x0 + d) < (unsigned)p.W ? 1u : 0u) << d;
                }
                if (p.KS == 1) { ym &= 1u; xm &= 1u; }
#pragma unroll
                for (int d = 0; d < 3; ++d) mask |= ((ym >> d) & 1u) ? (xm << (3 * d)) : 0u;
            }
            xoff[i] = off;
            tapmask[i] = mask;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int co = c0i + (i * NW + wave) * 8 + lrow;
            woff[i] = co < p.Cout ? (unsigned)(co * p.Ktot + lchunk * 8) * 2u : CONV_OOB;
        }
        if constexpr (!SMALL) {                  // start of the slice in the (channel block, tap) walk
            const int taps = p.KS * p.KS;
            cb = iks_base / taps;
            const int tap = iks_base - cb * taps;
            kh = tap / p.KS;
            kw = tap - kh * p.KS;
        }
    };

    // One stage = LPS DMA pieces per wave.  `prep_stage` computes the per-lane source offsets of
    // the stage under the issue cursor (VALU only); the pieces are fired a few at a time BETWEEN
    // the MFMA groups of the current step, so their issue cost hides under matrix time.
    // per-stage uniform part of the source offset (computed once per stage, see `prep_stage`)
    int st_tapoff = 0;
    unsigned st_bit = 0, st_wadd = 0;
    bool st_kin = true;
    bool st_seg2 = false;                     // the prepared stage belongs to the second K segment (tensor x2)
    auto prep_stage = [&]() {
        const int ks = iks_base + iks;
        if constexpr (!SMALL) {
            if (cb < cpt) {
                st_seg2 = false;
                st_tapoff = ((kh * p.W + kw) * p.Cin + (cb << 6) + lchunk * 8) * 2;
                st_bit = 1u << (kh * 3 + kw);
                st_wadd = (unsigned)(((kh * p.KS + kw) * cpt + cb) << 7);
                if (++kw == p.KS) { kw = 0; if (++kh == p.KS) { kh = 0; ++cb; } }
            } else {                          // K-concat: channel block cb - cpt of x2, one pixel down and right of the window's corner
                st_seg2 = true;
                st_tapoff = ((p.W + 1) * p.Cin2 + ((cb - cpt) << 6) + lchunk * 8) * 2;
                st_bit = 1u << 4;
                st_wadd = (unsigned)((p.KS * p.KS * cpt + (cb - cpt)) << 7);
                ++cb;
            }
        } else {
            const int kg = (ks << 6) + lchunk * 8;
            const int tap = kg >> p.cin_shift;
            const int ci = kg & (p.Cin - 1);
            const int tkh = (p.KS == 3) ? tap / 3 : 0;
            const int tkw = tap - tkh * p.KS;
            st_tapoff = ((tkh * p.W + tkw) * p.Cin + ci) * 2;
            st_bit = kg < p.Ktot ? 1u << (tkh * 3 + tkw) : 0u;
            // weights: a k chunk beyond Ktot (small-Cin tail) would read the NEXT row, so guard it
            st_kin = kg < p.Ktot;
            st_wadd = (unsigned)(ks << 7);
        }
    };
    // (second segment: the window corner's byte offset in x2 = that in x scaled by Cin2 / Cin - exact, offsets are multiples of 2 Cin)
    auto x_off = [&](int i) -> unsigned {
        const int base = (!SMALL && st_seg2) ? (xoff[i] >> p.x2_shift) : xoff[i];
        return (tapmask[i] & st_bit) ? (unsigned)(base + st_tapoff) : CONV_OOB;
    };
    auto w_off = [&](int i) -> unsigned { return (st_kin && woff[i] != CONV_OOB) ? woff[i] + st_wadd : CONV_OOB; };
    // after all pieces of the prepared stage are fired: advance the cursor
    auto advance_issue = [&]() {
        ibuf = ibuf == NS - 1 ? 0 : ibuf + 1;
        if (++iks == nk) {
            iks = 0;
            if ((it += tstep) < t1) setup_issue_tile(it);
        }
    };
    // piece j of a stage: j < XI -> pixel rows, else weight rows (j is a literal at every use)
#define DMA_PIECE(bufv, j)                                                                              \
    do {                                                                                                \
        if constexpr ((j) < LPS) {                                                                      \
            constexpr int jo_ = (j) < WI ? (j) + XI : (j) - WI;   /* weight pieces first: +2 % */            \
            unsigned char* base_ = smem + (bufv) * STAGE;                                               \
            if constexpr (jo_ < XI) {                                                                   \
                if (!SMALL && st_seg2) dma16(x2rsrc, base_ + (jo_ * NW + wave) * 1024, x_off(jo_ < XI ? jo_ : 0)); \
                else dma16(xrsrc, base_ + (jo_ * NW + wave) * 1024, x_off(jo_ < XI ? jo_ : 0));         \
            }                                                                                           \
            else                                                                                        \
                dma16(wrsrc, base_ + XB + ((jo_ - XI) * NW + wave) * 1024, w_off(jo_ >= XI && jo_ < LPS ? jo_ - XI : 0)); \
        }                                                                                               \
    } while (0)
    // pieces [lo, hi) with lo, hi compile-time constants and hi - lo <= 6
#define DMA_RANGE(bufv, lo, hi)                                 \
    do {                                                        \
        if constexpr ((lo) + 0 < (hi)) DMA_PIECE(bufv, (lo) + 0); \
        if constexpr ((lo) + 1 < (hi)) DMA_PIECE(bufv, (lo) + 1); \
        if constexpr ((lo) + 2 < (hi)) DMA_PIECE(bufv, (lo) + 2); \
        if constexpr ((lo) + 3 < (hi)) DMA_PIECE(bufv, (lo) + 3); \
        if constexpr ((lo) + 4 < (hi)) DMA_PIECE(bufv, (lo) + 4); \
        if constexpr ((lo) + 5 < (hi)) DMA_PIECE(bufv, (lo) + 5); \
    } while (0)
#define DMA_ALL(bufv)                      \
    do {                                   \
        DMA_RANGE(bufv, 0, LPS < 6 ? LPS : 6); \
        DMA_RANGE(bufv, 6, LPS);           \
    } while (0)

    const int wave_p = wave / WC, wave_c = wave - wave_p * WC;
    const int prow0 = wave_p * (TP / WP), crow0 = wave_c * (TC / WC);
    const int fr = lane & 31, fh = lane >> 5;

    // fragment registers are double-buffered: the ds_reads of sub-step kk+1 are in flight
    // while the MFMAs of sub-step kk run
    constexpr int FB = (MP * MC <= 4) ? 2 : 1;   // big wave tiles: single fragment set (register budget)
    floatx16 acc[MP][MC];
    half8 bf[FB][MP], af[FB][MC];
    auto read_frags = [&](const unsigned char* xs, const unsigned char* ws, int kk, int S) {
#pragma unroll
        for (int i = 0; i < MP; ++i)
            bf[S][i] = *reinterpret_cast<const half8*>(xs + lds_off(prow0 + i * 32 + fr, 2 * kk + fh));
#pragma unroll
        for (int j = 0; j < MC; ++j)
            af[S][j] = *reinterpret_cast<const half8*>(ws + lds_off(crow0 + j * 32 + fr, 2 * kk + fh));
    };
    auto mfma_group = [&](int S) {
#pragma unroll
        for (int i = 0; i < MP; ++i)
#pragma unroll
            for (int j = 0; j < MC; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[S][j], bf[S][i], acc[i][j], 0, 0, 0);
    };
    // DMA pieces fired after MFMA group kk: an even split of LPS over the 4 sub-steps
    constexpr int Q1 = (LPS + 3) / 4, Q2 = (LPS + 1) / 2, Q3 = (3 * LPS + 3) / 4;

    // ---------------- per-tile epilogue parameters in LDS (behind the ring): bias (1 or 9 classes) and
    // PReLU slope of this tile's TC couts.  They are fetched at the start of the tile's first k-step and
    // written at its end, so the epilogue reads them with short LDS latencies instead of serialising
    // on global loads.
    float* lds_bias = reinterpret_cast<float*>(smem + NS * STAGE);       // [9][TC] (class-major)
    float* lds_slope = lds_bias + 9 * TC;                                  // [TC]
    constexpr int PPT = (9 * TC + NW * 64 - 1) / (NW * 64);               // bias values per thread
    float pb[PPT], ps = 0.f;
    const bool border_ = p.flags & FRP_FLAG_BORDER_BIAS;
    auto fetch_params = [&](int vtile) {
        const int tile = vtile / p.ksplit;
        const int c0p = (tile % p.n_ctiles) * TC;
        const int nb = (border_ ? 9 : 1) * TC;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int idx = t + q * NW * 64;
            const int cls = idx / TC, co = c0p + (idx - cls * TC);
            pb[q] = (idx < nb && co < p.Cout) ? p.bias[(long)cls * p.Cout + co] : 0.f;
        }
        if (p.act == FRP_ACT_PRELU && t < TC) ps = (c0p + t < p.Cout) ? p.slope[c0p + t] : 0.f;
    };
    auto store_params = [&]() {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int idx = t + q * NW * 64;
            if (idx < 9 * TC) lds_bias[idx] = pb[q];
        }
        if (t < TC) lds_slope[t] = ps;
    };

    // ---------------- the stage stream: NS-slot ring, one barrier per k-step, continuous over tiles
    const int total = ((t1 - t0 + tstep - 1) / tstep) * nk;   // stages this workgroup consumes
    stamp(p.stamps, 0);
    setup_issue_tile(t0);
    int issued = 0;
#pragma unroll
    for (int s0 = 0; s0 < PRE; ++s0) {
        if (issued < total) {
            prep_stage();
            if (s0 == 0) DMA_ALL(0); else DMA_ALL(1);
            advance_issue();
            ++issued;
        }
    }
    int buf = 0, consumed = 0;
    const bool has_res = p.res != nullptr;

    stamp(p.stamps, 1);
    for (int ct = t0; ct < t1; ct += tstep) {
#pragma unroll
        for (int i = 0; i < MP; ++i)
#pragma unroll
            for (int j = 0; j < MC; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        if (ct == t0) stamp(p.stamps, 2);

        for (int ks = 0; ks < nk; ++ks) {
            // The stage to consume has landed for THIS wave once only the newer stages may still be
            // pending (NS=3: one stage = LPS pieces; NS=2: none).  Right after an epilogue its
            // stores are pending too (vmcnt counts them, in issue order, behind the DMAs), so
            // everything is drained there.
            if (NS == 3 && issued - consumed >= 2 && !(ks == 0 && ct != t0)) wait_vmcnt<(NS == 3 ? LPS : 0)>(); else wait_vmcnt<0>();
            // every wave's DMA of this stage is in LDS, and every wave is done reading the previous
            // stage, whose ring slot the next issue goes into
            retire_lds_reads();
            __builtin_amdgcn_s_barrier();
            const unsigned char* xs = smem + buf * STAGE;
            const unsigned char* ws = xs + XB;
            const int nbuf = ibuf;
            if (ks == 0 && p.ksplit == 1) fetch_params(ct);   // every wave is past the previous epilogue here
            read_frags(xs, ws, 0, 0);
            if (issued < total) {                       // uniform; false only for the last PRE steps
                prep_stage();
                if constexpr (FB == 2) {
                    read_frags(xs, ws, 1, 1); mfma_group(0);
                    DMA_RANGE(nbuf, 0, Q1);
                    read_frags(xs, ws, 2, 0); mfma_group(1);
                    DMA_RANGE(nbuf, Q1, Q2);
                    read_frags(xs, ws, 3, 1); mfma_group(0);
                    DMA_RANGE(nbuf, Q2, Q3);
                    mfma_group(1);
                    DMA_RANGE(nbuf, Q3, LPS);
                } else {
                    mfma_group(0); DMA_RANGE(nbuf, 0, Q1);
                    read_frags(xs, ws, 1, 0); mfma_group(0); DMA_RANGE(nbuf, Q1, Q2);
                    read_frags(xs, ws, 2, 0); mfma_group(0); DMA_RANGE(nbuf, Q2, Q3);
                    read_frags(xs, ws, 3, 0); mfma_group(0); DMA_RANGE(nbuf, Q3, LPS);
                }
                advance_issue();
                ++issued;
            } else {
                if constexpr (FB == 2) {
                    read_frags(xs, ws, 1, 1); mfma_group(0);
                    read_frags(xs, ws, 2, 0); mfma_group(1);
                    read_frags(xs, ws, 3, 1); mfma_group(0);
                    mfma_group(1);
                } else {
                    mfma_group(0);
                    read_frags(xs, ws, 1, 0); mfma_group(0);
                    read_frags(xs, ws, 2, 0); mfma_group(0);
                    read_frags(xs, ws, 3, 0); mfma_group(0);
                }
            }
            if (ks == 0 && p.ksplit == 1) store_params();
            ++consumed;
            buf = buf == NS - 1 ? 0 : buf + 1;
            if (ct == t0 && ks == 0) stamp(p.stamps, 3);      // first k-step done (includes the first DMA latency)
        }
        if (ct == t0) stamp(p.stamps, 4);                      // first tile's k-loop done

        // ---------------- epilogue of tile ct (accumulator layout: lane = one pixel, 4 consecutive
        // couts per register group): bias / border-class bias, residual, activation in fp32, fp16
        // (or fp32) stores straight from registers while the next tile's first stages are already
        // in flight.  All bias / slope / residual loads of a pixel row are issued back to back
        // from clamped (always valid) addresses before any is consumed; stores are predicated.
        const int ctile_id = ct / p.ksplit;
        const int ptile = ctile_id / p.n_ctiles;
        const int m0 = ptile * TP;
        const int c0 = (ctile_id - ptile * p.n_ctiles) * TC;
        if (p.ksplit > 1) {                       // raw fp32 partial sums -> workspace slab of this slice
            float* slab = reinterpret_cast<float*>(p.out) + (long)(ct - ctile_id * p.ksplit) * p.M * p.Cout;
#pragma unroll
            for (int i = 0; i < MP; ++i) {
                const int m = m0 + prow0 + i * 32 + fr;
#pragma unroll
                for (int j = 0; j < MC; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = c0 + crow0 + j * 32 + 8 * g + 4 * fh;
                        if (m < p.M && co < p.Cout) {
                            floatx4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                            *reinterpret_cast<floatx4*>(slab + (long)m * p.Cout + co) = v;
                        }
                    }
            }
            continue;
        }
        // bias / slope come from the LDS parameter cache; only the residual needs global loads, all
        // issued up front (32 VGPRs).  Interior tiles (the common case) take a copy of the body with
        // unconditional stores; ragged tiles clamp the load addresses and predicate the stores.
        if (nk == 1) __syncthreads();             // params were written in this very k-step
        {
            uint4 rres[MP][MC][2];
            if (has_res) conv_residual_loads<MP, MC>(p, rres, m0, c0, prow0, crow0, fr, fh, HoWo, inv_howo, inv_wo);
            if (p.out2)       // fp16 output plus an fp8 copy for an fp8 consumer (BASELINE config 5)
                conv_epilogue8<MP, MC, TC, false>(p, acc, rres, lds_bias, lds_slope, lds_slope, m0, c0, TP, prow0, crow0, fr, fh, HoWo, inv_howo, inv_wo);
            else
                conv_epilogue<MP, MC, TC>(p, acc, rres, lds_bias, lds_slope, m0, c0, TP, prow0, crow0, fr, fh, HoWo, inv_howo, inv_wo);
        }
        if (ct == t0) stamp(p.stamps, 5);                      // first tile's epilogue issued
    }
    stamp(p.stamps, 6);                                        // all tiles done (after the last epilogue's issue)
#undef DMA_ALL
#undef DMA_RANGE
#undef DMA_PIECE
}

template <int TP, int TC, int WP, int WC, int NS, int NW, bool SMALL>
static hipError_t launch_cfg(const ConvParams& p0, hipStream_t stream) {
    ConvParams p = p0;
    p.n_ptiles = (p.M + TP - 1) / TP;
    p.n_ctiles = (p.Cout + TC - 1) / TC;
    const int lds = NS * (TP + TC) * 128 + 10 * TC * 4;   // operand ring + epilogue parameter cache
    static bool attr_set[64] = {};
    auto kern = conv_mfma_kernel<TP, TC, WP, WC, NS, NW, SMALL>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const int ncu = device_cu_count(dev);
    if (ncu <= 0) return hipErrorInvalidDevice;
    // (device-side split choice: the grid is sized for the largest split, that of a single image)
    const int ks_grid = p.ksplit < 0 ? conv_pick_ksplit(p.Ho * p.Wo, p.Cout, p.Ktot, p.flags, p.res != nullptr, ncu) : p.ksplit;
    const long ntiles = (long)p.n_ptiles * p.n_ctiles * ks_grid;
    if (ntiles <= 0 || ntiles > 0x7fffffffL) return hipErrorInvalidValue;
    p.n_cu = ncu;
    // persistent: as many workgroups per CU as the LDS ring allows (1 or 2), each walks a
    // contiguous range of tiles
    const long slots = (long)ncu * (lds <= 80 * 1024 ? 2 : 1);
    unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
    p.n_workers = 0;
    // a launch of at most 128 tiles (a call of up to ~20 faces: where the latency of ONE call is what counts) leaves three quarters
    // of the slots empty: 64 more workgroups warm the L2s for the next launch.  Larger quarter-tile launches (config 4's ~36 faces
    // on two lanes) keep their spare CUs for the other lane's kernels: there the prefetchers cost 4 % of the throughput.
    if (TP == 128 && p.pf_ptr && p.pf_bytes >= 4096 && grid <= 128 && (long)grid + CONV_PF_WGS <= slots && conv_prefetch_enabled()) {
        p.n_workers = (int)grid;
        grid += CONV_PF_WGS;
    }
    if (grid > 256) p.stamps = nullptr;                                // (the diagnostic stamp buffer holds 256 workgroups)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, stream, p);
    return hipGetLastError();
}

// Host-side shape checks + tile selection.  Returns hipErrorInvalidValue on a shape the
// kernel does not cover instead of launching (a faulting kernel can take the node down).
hipError_t launch_conv(const ConvParams& in, hipStream_t stream) {
    ConvParams p = in;
    if (!(p.KS == 1 || p.KS == 3) || !(p.stride == 1 || p.stride == 2)) return hipErrorInvalidValue;
    if (p.Cin < 8 || (p.Cin & 7) || (p.Cout & 7) || p.N <= 0 || p.H <= 0 || p.W <= 0) return hipErrorInvalidValue;
    p.pad = p.KS / 2;
    p.Ho = (p.H + 2 * p.pad - p.KS) / p.stride + 1;
    p.Wo = (p.W + 2 * p.pad - p.KS) / p.stride + 1;
    const long M = (long)p.N * p.Ho * p.Wo;
    if (M <= 0 || M > 0x7fffffffL) return hipErrorInvalidValue;
    p.M = (int)M;
    p.Ktot = p.KS * p.KS * p.Cin;
    p.x2_bytes = 0;
    p.x2_shift = 0;
    if (p.x2) {                                  // K-concat: generic kernel, whole channel blocks, plain fp16 epilogue input
        if (p.KS != 3 || (p.Cin & 63) || p.Cin2 <= 0 || (p.Cin2 & 63) || !(p.Cin == p.Cin2 || p.Cin == 2 * p.Cin2) || p.ksplit > 1 ||
            (p.flags & (FRP_FLAG_F8 | FRP_FLAG_BORDER_BIAS)) || p.wino_w)
            return hipErrorInvalidValue;
        const long x2b = (long)p.N * p.H * p.W * p.Cin2 * 2;
        if (x2b >= 0x7fffffffL) return hipErrorInvalidValue;
        p.x2_bytes = (unsigned)x2b;
        p.x2_shift = p.Cin == p.Cin2 ? 0 : 1;
        p.Ktot += p.Cin2;
    }
    p.nk = (p.Ktot + 63) / 64;
    if (p.ksplit < 0 && !(p.n_dev && (p.flags & FRP_FLAG_OUT_F32) && !p.res && !(p.Cin & 63))) return hipErrorInvalidValue;
    if (p.ksplit == 0) p.ksplit = 1;
    if (p.ksplit > 1 && (p.nk % p.ksplit != 0 || !(p.flags & FRP_FLAG_OUT_F32) || p.res || (p.Cin & 63)))
        return hipErrorInvalidValue;             // split-K: exact slices, fp32 slabs, aligned path only
    // buffer descriptors carry 32-bit sizes and the kernel does signed 32-bit offset math
    const bool f8 = (p.flags & FRP_FLAG_F8) != 0;
    const int es = f8 ? 1 : 2;
    const long xb = (long)p.N * p.H * p.W * p.Cin * es, wb = (long)p.Cout * p.Ktot * es;
    if (xb >= 0x7fffffffL || wb >= 0x7fffffffL) return hipErrorInvalidValue;
    if ((p.flags & FRP_FLAG_OUT_FP8) && !f8) return hipErrorInvalidValue;
    if ((p.out2 || f8) && ((p.flags & FRP_FLAG_OUT_F32) || p.ksplit != 1 || !(p.out_scale > 0.f))) return hipErrorInvalidValue;
    p.x_bytes = (unsigned)xb;
    p.w_bytes = (unsigned)wb;
    const bool small = (p.Cin & 63) != 0;
    p.cin_shift = 0;
    if (small) {
        if (p.Cin & (p.Cin - 1)) return hipErrorInvalidValue;   // small path needs power-of-two Cin
        while ((1 << p.cin_shift) < p.Cin) ++p.cin_shift;
    }
    if ((p.flags & FRP_FLAG_BORDER_BIAS) && !(p.KS == 3 && p.stride == 1 && p.Ho >= 2 && p.Wo >= 2))
        return hipErrorInvalidValue;
    if ((p.flags & FRP_FLAG_RES_UP2) && (!p.res || p.Hr * 2 != p.Ho || p.Wr * 2 != p.Wo)) return hipErrorInvalidValue;
    if (!p.x || !p.w || !p.bias || !p.out) return hipErrorInvalidValue;
    if (p.act == FRP_ACT_PRELU && !p.slope) return hipErrorInvalidValue;
    // 3x3 stride-1 layers with whole 64-channel blocks: row-patch kernel (a third of the LDS-DMA traffic)
    if (f8) {                                        // fp8 operands: only the static-loop row-patch kernel covers them
        if (!conv3x3_rows_eligible(p) || (p.Cin & 127) || !p.wscale || !(p.in_scale > 0.f)) return hipErrorInvalidValue;
        return launch_conv3x3_lean(p, stream);
    }
    int ncu_ = p.n_cu;                               // (set by the engine; standalone callers: ask the runtime)
    if (ncu_ <= 0) {
        int dev_ = 0;
        if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= 64) return hipErrorInvalidDevice;
        ncu_ = device_cu_count(dev_);
    }
    // few 256 x 128 tiles (small pyramid scales, a handful of faces): quarter tiles, two workgroups per CU (conv_small_m)
    const bool few = conv_small_m(p, (long)((p.M + 255) / 256) * ((p.Cout + 127) / 128), ncu_);
    // (a caller that hands over the Winograd image has chosen the kernel FAMILY - frp_api.cpp: run_embed, by the slots of the
    // call -; the tile count of a single launch does not overrule it: the families differ in the last bits)
    if (p.wino_w && !p.x2 && !(p.dbg & 1) && conv3x3_wino_eligible(p)) {      // Winograd F(2,3) along the rows: 1.5 x fewer MFMAs
        p.w = p.wino_w;
        return launch_conv3x3_wino(p, stream);
    }
    // 3x3 stride-2 layers: row patches with shared neighbour entries (conv3x3_s2.hip).  OPT-IN (dbg bit 2048: FRP_S2=1 in the engine,
    // flags bit 21 of frp_conv2d_nhwc): on the headline shapes it measures 9-13 % SLOWER than this kernel's per-tap images
    // (profiles/r5/s2_probe.txt)
    if ((p.dbg & 2048) && !(p.dbg & 1) && !few && conv3x3_s2_eligible(p)) return launch_conv3x3_s2(p, stream);
    // the embedder's stem fused into the conv behind it: only conv3x3_c64.hip does that (the caller asked conv3x3_c64_fuses_stem first)
    if (p.stem_x) return (p.small_m <= 0 && conv3x3_c64_eligible(p)) ? launch_conv3x3_c64(p, stream) : hipErrorInvalidValue;
    // 64 -> 64 layers on large maps: weights in registers, 2-D tiles (conv3x3_c64.hip; dbg bit 512 / FRP_NO_C64=1: the row-patch
    // kernel instead - A/B runs; bit-identical results either way)
    {
        static const bool no_c64 = getenv("FRP_NO_C64") != nullptr;
        if (!no_c64 && !(p.dbg & (1 | 512)) && p.small_m <= 0 && conv3x3_c64_eligible(p)) return launch_conv3x3_c64(p, stream);
    }
#ifdef FRP_LAB   // lab build: dbg bits select the first-generation kernel and its timing ablations (conv3x3_rows.hip)
    if (!(p.dbg & 1) && !p.x2 && conv3x3_rows_eligible(p)) return launch_conv3x3_rows(p, stream);
#else
    if (!(p.dbg & 1) && !p.x2 && conv3x3_rows_eligible(p)) return launch_conv3x3_lean(p, stream);
#endif
    // Tile selection (measured on MI355X, tools/conv_bench.py):
    //   Cout > 64 : 256 pixels x 128 couts, 8 waves (64x64 each), 3-slot ring (144 KiB, one
    //               workgroup per CU).  Two independent 4-wave 128x128 groups per CU (2-slot
    //               rings) were 20-25 % slower; a 128x64 per-wave tile does not fit 256 VGPRs.
    //   Cout <= 64: 256 x 64, 8 waves (32x64 each), 3-slot ring (a 512 x 64 tile with 64x64 per wave
    //               measured 10 % slower, two co-resident 4-wave 128 x 64 groups 0-4 % slower).
    if (!small && few) return launch_cfg<128, 64, 4, 2, 3, 8, false>(p, stream);
    if (p.Cout > 64)
        return small ? launch_cfg<256, 128, 4, 2, 3, 8, true>(p, stream) : launch_cfg<256, 128, 4, 2, 3, 8, false>(p, stream);
    return small ? launch_cfg<256, 64, 8, 1, 3, 8, true>(p, stream) : launch_cfg<256, 64, 8, 1, 3, 8, false>(p, stream);
}

}  // namespace frp
