// K2e: 3x3 stride-2 convolution (the first conv of the detector's strided blocks, the second conv of the embedder's - there with
// the block's 1x1 stride-2 shortcut riding in the k-loop as a second K segment: ConvParams::x2).
//
// Why its own kernel.  In the generic kernel (conv_mfma.hip) every tap of a stride-2 window is its own pixel image: nine images of
// 256 pixels per 64-channel block, 32 KiB of LDS-DMA per k-step next to 16 KiB of weights - these launches run AT the LDS-DMA
// rate, x1.4-1.6 of their bounds (profiles/r5/layer_times.txt).  The three taps of a kernel row read the input row 2 oy + kh - 1 at
// the columns 2 ox - 1, 2 ox, 2 ox + 1, and the right neighbour of one output pixel is the left neighbour of the next.  Here a ROW
// PATCH per (32-channel block, kh) holds, for the tile's 256 flattened output pixels,
//   E[i]      the pixel under the centre tap of output pixel i                                   (256 entries)
//   L[i + r]  its left neighbour, L[i + r + 1] its right neighbour; r = image rows begun since the tile's first pixel
//             (one extra entry per row: a row's last right neighbour is not the next row's first left one)   (<= 294 entries)
// - 2.1 instead of 3 pixel entries per output pixel and kernel row, de-interleaved by the DMA itself (the LDS destination of an
// LDS-DMA piece is lane-linear, the SOURCE pixel of a lane is free), so that the fragment reads of consecutive output pixels are
// consecutive 64-byte entries: conflict-free ds_read_b128 with the chunk swizzle c ^ ((entry >> 2) & 3).  Out-of-image entries
// (left padding, the row above the image, pixels past the last image) are zero-filled by the DMA (source offset out of range).
// One barrier per (block, kh): 24 MFMAs per wave (3 taps x 2 k-slices x 2 x 2 blocks of 32) on ONE weight stage of
// 3 taps x 128 couts x 32 channels; patches in a 3-slot ring two steps ahead, weight stages in a 2-slot ring one step ahead; the
// stream of LDS-DMA pieces never stops at a tile boundary and every wave issues the same number of pieces per step, so all waits
// are counted (`s_waitcnt vmcnt(N)`).  K order (32-channel block, kh, kw, 16-channel slice), then the blocks of x2 at the centre tap:
// not the generic kernel's order (64-channel blocks) - results agree to the last bits of the fp32 accumulation, not bit for bit.
//
// Replaces the same reference calls as conv_mfma.hip (face_recognition.face_locations / face_encodings,
// backend/app/routes/camera.py:232,237).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "frp_internal.h"
#include "conv_common.h"

namespace frp {

#define S2_TP 256                                // output pixels per tile (flattened n, oy, ox)
#define S2_CT 128                                // couts per tile
#define S2_PPW 5                                 // patch pieces per wave and patch: 40 piece slots, of which the launch's
#define S2_PATCH_PIECES 35                       //   ceil((LN + 256) / 16) <= 35 exist (the others: zero-fill into a dummy KiB)
#define S2_PATCH (S2_PATCH_PIECES * 1024)        // 560 entries of 64 B
#define S2_NP 3
#define S2_WST (3 * S2_CT * 64)                  // a weight stage: rows (kw, cout) of 32 channels
#define S2_NWS 2
#define S2_OFF_W (S2_NP * S2_PATCH)
#define S2_OFF_BIAS (S2_OFF_W + S2_NWS * S2_WST)
#define S2_MAX_COUT 512
#define S2_OFF_DUMMY (S2_OFF_BIAS + S2_MAX_COUT * 4)
#define S2_LDS (S2_OFF_DUMMY + 1024)             // 159,744 B

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

// lab builds only (tools/s2_ablate.sh): timing ablations - 1: no patch DMA, 2: no MFMAs, 4: no weight DMA, 8: no fragment reads,
// 16: no stores (1, 4, 16: every wait drains the counter).  Wrong results, of course.
#ifndef S2_ABL
#define S2_ABL 0
#endif

__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_base, unsigned voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, voffset, soffset, 0, 0);
}

template <int ACT>
__global__ __launch_bounds__(512, 2) void conv3x3_s2_kernel(ConvParams p_in) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const ConvParams& p = p_in;
    // (every field the loops use is copied into a local: a modified copy of the struct lives in scratch memory)
    int N_ = p.N, M_ = p.M, n_pt = p.n_ptiles;
    if (p.n_dev) {                             // image count known on the device only (threshold mode)
        int n = *p.n_dev;
        n = n < 0 ? 0 : (n > p.N ? p.N : n);
        N_ = n;
        M_ = n * p.Ho * p.Wo;
        n_pt = (M_ + S2_TP - 1) / S2_TP;
    }
    const int M = M_, n_ct = p.n_ctiles, pW = p.W, pH = p.H, pCin = p.Cin, pCout = p.Cout;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n_tiles = n_pt * n_ct;
    int t0, t1, tstep;
    if ((gridDim.x & 7) == 0) {                // XCD-interleaved tile walk (see conv3x3_lean.hip)
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, pr = gridDim.x >> 3;
        const int cs = (int)((long)x * n_tiles / 8), ce = (int)((long)(x + 1) * n_tiles / 8);
        t0 = cs + j;
        t1 = ce;
        tstep = pr;
    } else {
        t0 = (int)((long)blockIdx.x * n_tiles / gridDim.x);
        t1 = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
        tstep = 1;
    }
    if (t0 >= t1) return;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t x2rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x), 0, p.x2 ? p.x2_bytes : 0u, 0x00020000);
    const unsigned o_bytes = (unsigned)((long)M * pCout * 2);                         // (< 2 GiB: launcher)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, o_bytes, 0x00020000);

    float* lds_bias = reinterpret_cast<float*>(smem + S2_OFF_BIAS);
    for (int i = t; i < pCout; i += 512) lds_bias[i] = p.bias[i];

    // ---------------- geometry of the launch
    const int Wo = p.Wo, Ho = p.Ho, NR = N_ * p.Ho;
    const int LN = 256 + (Wo + 254) / Wo + 1;                 // L entries of a patch; E entries follow
    const int n_pieces = (LN + 256 + 15) >> 4;                // <= 35
    const float inv_wo = 1.0f / (float)Wo, inv_wo1 = 1.0f / (float)(Wo + 1), inv_ho = 1.0f / (float)Ho;
    const int cin_sh = p.cin_shift, cin2_sh = p.x2_shift;     // log2(2 Cin), log2(2 Cin2) (set by the launcher)
    const int nfull = 3 * (pCin >> 5);                       // steps (32-channel block, kh) of a tile ...
    const int U = nfull + (p.x2 ? (p.Cin2 >> 5) : 0);         // ... + one per 32-channel block of x2 (centre tap only)
    const int Ktot2 = p.Ktot * 2;

    // ---------------- lane constants
    const int fr = lane & 31, fh = lane >> 5;
    const int wave_p = wave >> 1, wave_c = wave & 1;
    const int chunk16 = (((lane & 3) ^ ((lane >> 4) & 3)) << 4);          // source chunk of a DMA lane (both operands: entry / row = 16 piece + lane / 4)
    // weights: piece wave + 8 i, row = 16 piece + lane / 4 = (kw, cout)
    unsigned wv[3], wvc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row = (wave + 8 * i) * 16 + (lane >> 2);
        const int kw = row >> 7, co = row & 127;
        wv[i] = (unsigned)(co * Ktot2 + ((kw * pCin) << 1) + chunk16);
        wvc[i] = kw == 1 ? (unsigned)(co * Ktot2 + chunk16) : CONV_OOB;
    }
    // A fragments: row (kw = 0) of this lane's cout in block j, k-slice kk
    int aad[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int row = wave_c * 64 + j * 32 + fr;
            aad[j][kk] = S2_OFF_W + row * 64 + (((2 * kk + fh) ^ ((row >> 2) & 3)) << 4);
        }

    // ---------------- cursors: compute (ct, u), weights one step ahead, patches two steps ahead
    struct Cur { int tile, u, kh, cb; };
    auto advance = [&](Cur& c) __attribute__((always_inline)) {
        ++c.u;
        if (c.u == U) { c.u = 0; c.kh = 0; c.cb = 0; c.tile += tstep; }
        else if (c.u < nfull) { if (++c.kh == 3) { c.kh = 0; ++c.cb; } }
        else { c.kh = 1; c.cb = c.u - nfull; }
    };
    // per-lane state of the patch cursor's tile: pixel index of the entry's column in the row 2 oy - 1, validity of kh = 0..2 (+ bit 3: an E entry)
    int ppix[S2_PPW];
    unsigned pmask[S2_PPW];
    auto setup_patch_tile = [&](int tile) __attribute__((always_inline)) {
        const bool live = tile < t1;
        const int ptile = live ? tile / n_ct : 0;
        const int m0 = ptile * S2_TP;
        const int R_s = m0 / Wo, ox_s = m0 - R_s * Wo;
#pragma unroll
        for (int i = 0; i < S2_PPW; ++i) {
            const int e_ = (wave + 8 * i) * 16 + (lane >> 2);
            int R, ix;
            bool ok, isE;
            if (e_ < LN) {
                int k, e;
                fast_divmod(e_ + ox_s, Wo + 1, inv_wo1, k, e);
                R = R_s + k;
                ix = 2 * e - 1;
                ok = R < NR;
                isE = false;
            } else {
                const int ii = e_ - LN, m = m0 + ii;
                int ox;
                fast_divmod(m, Wo, inv_wo, R, ox);
                ix = 2 * ox;
                ok = ii < S2_TP && m < M;
                isE = true;
            }
            int n, oy;
            fast_divmod(ok ? R : 0, Ho, inv_ho, n, oy);
            const int iy0 = 2 * oy - 1;
            ok = ok && live && (unsigned)ix < (unsigned)pW;
            ppix[i] = (n * pH + iy0) * pW + ix;
            pmask[i] = ok ? ((iy0 >= 0 ? 1u : 0u) | (iy0 + 1 < pH ? 2u : 0u) | (iy0 + 2 < pH ? 4u : 0u) | (isE ? 8u : 0u)) : 0u;
        }
    };
    auto patch_piece = [&](const Cur& c, int i, int pslot) __attribute__((always_inline)) {
        if (S2_ABL & 1) return;
        const int pc = wave + 8 * i;
        unsigned char* dst = pc < n_pieces ? smem + pslot * S2_PATCH + pc * 1024 : smem + S2_OFF_DUMMY;
        // (one DMA call with selected operands: two calls in the arms of a branch are merged by the optimiser into one call on a
        // pointer INTO the lambda's closure, which then has to live in scratch memory)
        const bool full = c.u < nfull;
        const __amdgpu_buffer_rsrc_t r = full ? xrsrc : x2rsrc;
        const int sh = full ? cin_sh : cin2_sh;
        const int row = full ? c.kh * pW : pW;
        const bool ok = full ? ((pmask[i] >> c.kh) & 1u) != 0u : (pmask[i] & 10u) == 10u;
        const unsigned v = ok ? (((unsigned)(ppix[i] + row)) << sh) + (unsigned)chunk16 : CONV_OOB;
        dma16s(r, dst, v, c.cb << 6);
    };
    auto weight_pieces = [&](const Cur& c, int wslot) __attribute__((always_inline)) {
        if (S2_ABL & 4) return;
        const bool live = c.tile < t1;
        const int c0 = live ? (c.tile % n_ct) * S2_CT : 0;
        const bool full = c.u < nfull;
        const int soff = c0 * Ktot2 + (full ? ((c.kh * 3 * pCin + (c.cb << 5)) << 1) : ((9 * pCin + (c.cb << 5)) << 1));
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned v = !live ? CONV_OOB : (full ? wv[i] : wvc[i]);
            dma16s(wrsrc, smem + S2_OFF_W + wslot * S2_WST + (wave + 8 * i) * 1024, v, soff);
        }
    };

    __syncthreads();                           // bias visible; no LDS-DMA in flight yet
    Cur cw{t0, 0, 0, 0}, cp{t0, 0, 0, 0};
    setup_patch_tile(t0);
#pragma unroll
    for (int i = 0; i < S2_PPW; ++i) patch_piece(cp, i, 0);           // P(0)
    weight_pieces(cw, 0);                                             // W(0)
    advance(cp);
    advance(cw);
#pragma unroll
    for (int i = 0; i < S2_PPW; ++i) patch_piece(cp, i, 1);           // P(1)
    advance(cp);                                                      // cp = step 2, cw = step 1
    int pslot = 0, wslot = 0;                                         // slots of the step being computed
    bool after_epilogue = false;

    for (int ct = t0; ct < t1; ct += tstep) {
        const int ptile = ct / n_ct;
        const int m0 = ptile * S2_TP, c0 = (ct - ptile * n_ct) * S2_CT;
        // B fragment addresses of this tile: block b = pixels 32 (2 wave_p + b) + fr; entries L[i + r], E[i], L[i + r + 1]
        int bad[2][3][2];
        {
            const int R_s = m0 / Wo, ox_s = m0 - R_s * Wo;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int i = (wave_p * 2 + b) * 32 + fr;
                int r, ox;
                fast_divmod(ox_s + i, Wo, inv_wo, r, ox);
                const int ent[3] = {i + r, LN + i, i + r + 1};
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) bad[b][k][kk] = ent[k] * 64 + (((2 * kk + fh) ^ ((ent[k] >> 2) & 3)) << 4);
            }
        }
        floatx16 acc[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][j][e] = 0.f;

        for (int u = 0; u < U; ++u) {
            // P(g) and W(g) have landed once only P(g + 1)'s five pieces - and, behind them, the eight stores of an epilogue that has just
            // been issued - may still be pending (in-order counter; every wave issues the same pieces)
            if (S2_ABL & (1 | 4 | 16)) wait_vmcnt<0>();               // (the counts below assume every piece and store)
            else if (after_epilogue) wait_vmcnt<S2_PPW + 8>(); else wait_vmcnt<S2_PPW>();
            after_epilogue = false;
            retire_lds_reads();
            __builtin_amdgcn_s_barrier();
            const int nws = wslot ^ 1;
            int nps = pslot + 2;
            nps = nps >= S2_NP ? nps - S2_NP : nps;
            const unsigned char* pb = smem + pslot * S2_PATCH;
            const unsigned char* wb = smem + wslot * S2_WST;
            const bool center = u >= nfull;
            half8 bf[2][2], af[2][2];                                  // [buffer][block]
            auto rd = [&](int kw, int kk, int S) __attribute__((always_inline)) {
                if (S2_ABL & 8) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) { asm volatile("" : "=v"(bf[S][b])); asm volatile("" : "=v"(af[S][b])); }
                    return;
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[S][b] = *reinterpret_cast<const half8*>(pb + bad[b][kw][kk]);
#pragma unroll
                for (int j = 0; j < 2; ++j) af[S][j] = *reinterpret_cast<const half8*>(wb + aad[j][kk] + kw * (S2_CT * 64));
            };
            auto mm = [&](int S) __attribute__((always_inline)) {
                if (S2_ABL & 2) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) { asm volatile("" ::"v"(bf[S][b]), "v"(af[S][b])); }
                    return;
                }
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[b][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[S][j], bf[S][b], acc[b][j], 0, 0, 0);
            };
            if (!center) {
                rd(0, 0, 0);
                weight_pieces(cw, nws);                                // W(g + 1): a whole step to land
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    if (s + 1 < 6) rd((s + 1) >> 1, (s + 1) & 1, (s + 1) & 1);
                    mm(s & 1);
                    // the patch cursor enters its next tile (or runs off the end): ~250 vector instructions, behind the step's first MFMAs
                    if (s == 0) { if (cp.u == 0) setup_patch_tile(cp.tile); }
                    else patch_piece(cp, s - 1, nps);                  // P(g + 2)
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                if (cp.u == 0) setup_patch_tile(cp.tile);
                rd(1, 0, 0);
                weight_pieces(cw, nws);
                __builtin_amdgcn_sched_barrier(0);
                rd(1, 1, 1);
                mm(0);
#pragma unroll
                for (int i = 0; i < S2_PPW; ++i) patch_piece(cp, i, nps);
                __builtin_amdgcn_sched_barrier(0);
                mm(1);
            }
            advance(cw);
            advance(cp);
            pslot = pslot + 1 == S2_NP ? 0 : pslot + 1;
            wslot = nws;
        }

        // ---------------- epilogue: lane = pixel, registers 4g .. 4g+3 of block j = couts c0 + 64 wave_c + 32 j + 8g + 4 fh .. + 3
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = m0 + (wave_p * 2 + b) * 32 + fr;
            const unsigned ooff = m < M ? (unsigned)(m * pCout * 2 + (c0 + wave_c * 64) * 2 + fh * 16) : CONV_OOB;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                union { half4 h; unsigned u[2]; } pk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = c0 + wave_c * 64 + j * 32 + 8 * g + 4 * fh;
                    const floatx4 b4 = *reinterpret_cast<const floatx4*>(lds_bias + cl);
                    floatx4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[b][j][4 * g + e] + b4[e];
                    if (ACT == FRP_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[g].h[e] = (_Float16)v[e];
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    swap_halves(pk[2 * q].u[0], pk[2 * q + 1].u[0]);
                    swap_halves(pk[2 * q].u[1], pk[2 * q + 1].u[1]);
                    const u32x4s o = {pk[2 * q].u[0], pk[2 * q].u[1], pk[2 * q + 1].u[0], pk[2 * q + 1].u[1]};
                    if (S2_ABL & 16) asm volatile("" ::"v"(o)); else
                    __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, ooff + j * 64 + q * 32, 0, 0);   // couts 32 j + 16 q + 8 fh .. + 7 of the wave's 64
                }
            }
        }
        after_epilogue = true;
    }
    // nothing of this workgroup's DMA stream may still be in flight when its LDS is handed to the next workgroup
    wait_vmcnt<0>();
}

static int log2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
}

// Shapes the kernel covers (`p` with launch_conv()'s derived fields: M, Ho, Wo, Ktot incl. the x2 segment, x_bytes, w_bytes, x2_bytes).
bool conv3x3_s2_eligible(const ConvParams& p) {
    if (p.KS != 3 || p.stride != 2 || p.ksplit != 1 || p.out2 || p.res) return false;
    if (p.flags & (FRP_FLAG_BORDER_BIAS | FRP_FLAG_OUT_F32 | FRP_FLAG_F8 | FRP_FLAG_OUT_FP8 | FRP_FLAG_RES_UP2 | FRP_FLAG_FLATTEN)) return false;
    if (!(p.act == FRP_ACT_NONE || p.act == FRP_ACT_RELU)) return false;
    if ((p.H & 1) || (p.W & 1) || p.Ho * 2 != p.H || p.Wo * 2 != p.W || p.Wo < 7) return false;
    if (p.Cin < 64 || p.Cin > 512 || log2_exact(p.Cin) < 0) return false;
    if ((p.Cout & (S2_CT - 1)) || p.Cout > S2_MAX_COUT) return false;
    if (p.x2 && (p.Cin2 < 32 || p.Cin2 > 512 || log2_exact(p.Cin2) < 0)) return false;
    if (p.M <= 0 || p.M >= (1 << 24) - 1024) return false;                        // (fast_divmod's exact range)
    if ((long)p.M * p.Cout * 2 >= 0x7f000000L) return false;                        // 32-bit byte offsets of the stores
    return true;
}

template <int ACT>
static hipError_t launch_s2_cfg(const ConvParams& p0, hipStream_t stream) {
    ConvParams p = p0;
    static bool attr_set[64] = {};
    auto kern = conv3x3_s2_kernel<ACT>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S2_LDS);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const int ncu = device_cu_count(dev);
    if (ncu <= 0) return hipErrorInvalidDevice;
    p.n_ptiles = (p.M + S2_TP - 1) / S2_TP;
    p.n_ctiles = p.Cout / S2_CT;
    p.cin_shift = log2_exact(p.Cin) + 1;                       // this kernel: log2 of a pixel's bytes in x ...
    p.x2_shift = p.x2 ? log2_exact(p.Cin2) + 1 : 0;            // ... and in x2
    const long tiles = (long)p.n_ptiles * p.n_ctiles;
    if (tiles <= 0 || tiles > 0x7fffffffL) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(tiles < ncu ? tiles : ncu);       // persistent: one workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), S2_LDS, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv3x3_s2(const ConvParams& p, hipStream_t stream) {
    if (!conv3x3_s2_eligible(p)) return hipErrorInvalidValue;
    return p.act == FRP_ACT_RELU ? launch_s2_cfg<FRP_ACT_RELU>(p, stream) : launch_s2_cfg<FRP_ACT_NONE>(p, stream);
}

}  // namespace frp
