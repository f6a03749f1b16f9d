// K2c: 3x3 stride-1 convolution as Winograd F(2,3) ALONG THE IMAGE ROWS - 1.5 x fewer matrix instructions for the same
// result - for the maps the embedder spends its time on (14 x 14 and 28 x 28, Cin % 64 == 0; any even width the LDS holds).
//
// Why: the direct kernels (conv3x3_lean.hip) run their k-loop at the rate its own instruction stream sustains; the chip
// is power-bound near 1.5 PFLOP/s of fp16 MFMA, so the next gain has to come from issuing fewer MFMAs per output.
// tools/mix_lab.py (profiles/r3/mix_lab.txt) measured the candidate instruction mixes at one wave per SIMD: the 2-D
// F(2x2,3x3) form needs 2 fragment reads + 8 packed adds per MFMA and twice the LDS-DMA bytes per MFMA; the 1-D form keeps
// the direct kernel's 1 read per MFMA, adds 2 packed fp16 adds per MFMA and reuses its whole data path.
//
// Arithmetic.  Output pixels are taken in horizontal PAIRS (x = 2j, 2j+1; W even).  Per kernel row kh and input channel:
//     d0..d3 = x[y+kh-1][2j-1 .. 2j+2]                       (zero outside the image)
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3            (B^T d, packed fp16, in registers)
//     U0 = g0        U1 = (g0+g1+g2)/2   U2 = (g0-g1+g2)/2   U3 = g2        (G g, once at load time: frp_api.cpp)
//     M_f += U_f * V_f  summed over (kh, cin) on the matrix cores, fp32      (4 frequencies instead of 6 products per pair)
//     y[2j] = M0 + M1 + M2        y[2j+1] = M1 - M2 - M3                     (A^T M, fp32, in the epilogue)
// fp16 products of exactly transformed fp16 operands, fp32 accumulation: on the seeded IResNet-100 the embedding moves by
// 1 - cos = 1.6e-6 against the direct kernels and stays 1.2e-6 from the fp32 oracle (tools/winograd_numerics.py; bar 1e-3).
//
// Structure: one 512-thread workgroup per CU = two waves per SIMD, each wave owning all four frequencies of a
// (32 pairs = 64 pixels) x 64 couts block in 128 accumulator registers: per 16-channel sub-step 8 MFMAs, 4 raw + 8 weight
// fragment reads and 16 packed adds.  (The lab's one-wave-per-SIMD form - 64 pairs x 64 couts in 256 AGPRs, 1 read per MFMA -
// is x0.7: a single issue stream per SIMD is longer than its matrix time; `conv3x3_wino_kernel<4>`, lab build only.)
// Tile = 256 pixels x 128 couts, persistent XCD-interleaved walk as the other conv kernels.  LDS: the pixel operand of a
// 64-channel block is ONE super-patch (all three kernel rows: 256 + 2W + 2 pixel rows of 128 B, even and odd pixels in
// separate halves so that a pair-strided fragment read hits consecutive rows), double-buffered across channel blocks; the
// weight operand is a 4-slot ring of 16 KiB stages, one per (channel block, kernel row, 16-channel slice), stored in global
// memory as the LDS image itself (128 rows = couts x 8 chunks: chunk 2f + h = frequency f, 8-channel half h; already
// xor-swizzled), so its LDS-DMA is a linear copy; the tile's epilogue parameters arrive by LDS-DMA too.  One barrier per
// sub-step; the DMA runs two sub-steps ahead of the fragment reads, three ahead of the MFMAs; the raw fragments and the
// first weight fragments of sub-step s+1 are read - and B^T d of s+1 is computed - during the MFMAs of s, so the matrix pipe
// starts right behind every barrier.
//
// Replaces the same reference calls as conv_mfma.hip (face_recognition.face_encodings, backend/app/routes/camera.py:237,
// backend/app/services/face_service.py:179).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "frp_internal.h"
#include "conv_common.h"

namespace frp {

// packed fp16 add / subtract of an 8-halfword fragment (a plain <8 x half> subtraction is scalarised by this compiler)
struct WH8 { unsigned p[4]; };
__device__ __forceinline__ half8 wadd(half8 a, half8 b) {
    WH8 x = __builtin_bit_cast(WH8, a), y = __builtin_bit_cast(WH8, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) asm("v_pk_add_f16 %0, %1, %2" : "=v"(x.p[i]) : "v"(x.p[i]), "v"(y.p[i]));
    return __builtin_bit_cast(half8, x);
}
__device__ __forceinline__ half8 wsub(half8 a, half8 b) {
    WH8 x = __builtin_bit_cast(WH8, a), y = __builtin_bit_cast(WH8, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x.p[i]) : "v"(x.p[i]), "v"(y.p[i]));
    return __builtin_bit_cast(half8, x);
}

#define WN_TC_ 128
// Tile epilogue of the Winograd kernel: A^T M in fp32 (y_even = M0 + M1 + M2, y_odd = M1 - M2 - M3) and the shared per-block
// epilogue body (bias / border-class bias, residual, activation, fp16 stores), ONE (32 pairs x 32 couts, parity) unit at
// a time: 16 output registers and 8 residual registers live next to the accumulators (both parities of both cout blocks at
// once - 64 + 64 registers - spilled, and a spilled epilogue took 19 us per tile).  The residual of unit u + 1 is requested
// before unit u is finished.  Specialised at compile time on (full tile, activation, residual) like the direct kernels.
// residual of one (cout block c, parity) unit in the store layout, from clamped - always valid - addresses
template <int PB, bool OVER = false>
__device__ __forceinline__ void wino_load_res(const ConvParams& p, uint4 (&r)[PB][1][2], int c, int par, int m0, int c0, int pair0,
                                              int crow0, int fr, int fh, int m_over = 0) {
#pragma unroll
    for (int b = 0; b < PB; ++b) {
        const int mraw = OVER ? (m_over >= 0 ? m_over + par : p.M) : m0 + (pair0 + b * 32 + fr) * 2 + par;
        const int m = mraw < p.M ? mraw : 0;
        // (round 4, lab experiment since removed: every tile's residual read from the same L2-resident 64 KiB - the best any prefetch
        // could do - took stage 3 conv2 from 59.5 to 58.4 us, stage 2 conv2 from 75.5 to 71.4: profiles/r4/wino_probe_residual_from_l2.txt)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int co = c0 + crow0 + c * 32 + 16 * q + 8 * fh;
            r[b][0][q] = *reinterpret_cast<const uint4*>(p.res + (long)m * p.Cout + (co < p.Cout ? co : 0));
        }
    }
}

// (Requesting the first unit's residual under the tile's last MFMAs - 8 more live registers in the k-loop's last sub-step -
// measured slower: 69.1 vs 66.2 us on the stage-3 shape.  Tried again in the second half of round 3 in two forms that should
// not have cost registers - the request written into the raw-fragment registers, dead in a tile's last sub-step; the last
// channel block peeled into its own copy of the body, `last_cb` a literal -: 25 / 26 spilled registers instead of 12, some of
// them in the per-tile head, embedder 8.02 vs 7.83 ms in the same-box A/B.  The ~2 us stay exposed: 0.26 ms of a step.)
template <int PB, bool FULL, int ACT, int RES, bool OVER = false>
__device__ __forceinline__ void wino_tile_epilogue(const ConvParams& p, floatx16 (&acc)[4][PB][2],
                                                   const float* lds_bias, const float* lds_slope, int m0, int c0, int pair0, int crow0,
                                                   int fr, int fh, int HoWo, float inv_howo, float inv_wo, int m_over = 0) {
    // OVER / m_over (2-D tiles): pixel index of the lane's pair (its even pixel), -1 for a pair that does not exist
    const bool has_res = RES < 0 ? p.res != nullptr : RES != 0;
#ifndef WN_RES_UPFRONT
#define WN_RES_UPFRONT 1
#endif
    // Round 5: the residual of ALL four units is requested before the first one is finished (32 registers: the k-loop's fragment
    // registers are dead here) - one L2 round trip per tile exposed instead of one per unit (the request for unit u + 1 went out
    // when unit u started, and a unit's arithmetic is shorter than the latency).  WN_RES_UPFRONT=0: the former order (A/B).
    constexpr int NRR = (WN_RES_UPFRONT && PB == 1) ? 4 : 2;
    uint4 rr[NRR][PB][1][2];
    auto load_res = [&](int c, int par, uint4 (&r)[PB][1][2]) { wino_load_res<PB, OVER>(p, r, c, par, m0, c0, pair0, crow0, fr, fh, m_over); };
    if (has_res) {
        if constexpr (NRR == 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) load_res(u >> 1, u & 1, rr[u]);
        } else load_res(0, 0, rr[0]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = u >> 1, par = u & 1;
        if constexpr (NRR == 2) { if (has_res && u < 3) load_res((u + 1) >> 1, (u + 1) & 1, rr[(u + 1) & 1]); }
        floatx16 y[PB][1];
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                y[b][0][e] = par ? (acc[1][b][c][e] - acc[2][b][c][e]) - acc[3][b][c][e] : (acc[0][b][c][e] + acc[1][b][c][e]) + acc[2][b][c][e];
        conv_epilogue_body<PB, 1, WN_TC_, FULL, ACT, RES, OVER>(p, y, rr[NRR == 4 ? u : (u & 1)], lds_bias, lds_slope, m0, c0, pair0, crow0 + c * 32, fr, fh, HoWo,
                                                                inv_howo, inv_wo, 2, par, m_over);
    }
}

#define WN_TP 256          // output pixels per tile (128 pairs)
#define WN_TC 128          // couts per tile
#define WN_NSW 4           // weight ring slots (one 16-channel sub-step each)
#define WN_WSLOT (WN_TC * 128)
#define WN_PIECES 48       // patch-piece issue slots per workgroup and patch
#define WN_ROWP_HALF 160   // row patches: (256 + 2 + 1) / 2 rows per half, rounded up to 32

// Weight stage image (16 KiB per (channel block, kernel row, 16-channel slice); global memory holds the LDS image itself, so its
// LDS-DMA is a linear copy): [frequency f: 4 KiB][32-cout block: 1 KiB][32 x 16-byte slots], slot (2 r + h) ^ ((r >> 3) & 1) of a
// block = channels 8h .. 8h+7 of cout r.  A lane (r = lane & 31, h = lane >> 5) reads EVERY fragment of a stage at one per-lane
// base + an immediate (f * 4096 + block * 1024): no address arithmetic in the k-loop.  The xor keeps a ds_read_b128's 16-lane
// groups ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32) on 16 different 16-byte slots modulo 256 B: conflict-free.
__host__ __device__ inline int wino_u_slot(int r, int h) { return (2 * r + h) ^ ((r >> 3) & 1); }
__host__ __device__ inline int wino_u_lane(int r, int h) { return wino_u_slot(r, h) * 16; }

// LDS rows of one half (even / odd pixels) of a super-patch: (256 + 2W + 2 + 1) / 2 rounded up to 32 (whole pieces per wave
// for 4 and for 8 waves)
__host__ __device__ inline int wino_half_rows(int W) { return ((WN_TP + 2 * W + 3) / 2 + 31) / 32 * 32; }
__host__ __device__ inline int wino_lds_bytes(int W) {
    return 2 * (2 * wino_half_rows(W) * 128) + WN_NSW * WN_WSLOT + 256 + 11 * WN_TC * 4 + 8 * 1024;
}
// the super-patch of a channel block fits (W <= 30); wider maps take the row-patch form, whose LDS does not depend on W
__host__ __device__ inline bool wino_super_patch(int W) { return wino_half_rows(W) / 4 <= WN_PIECES && wino_lds_bytes(W) <= 160 * 1024; }
__host__ __device__ inline int wino_rowp_lds_bytes() { return 2 * (2 * WN_ROWP_HALF * 128) + WN_NSW * WN_WSLOT + 256 + 11 * WN_TC * 4 + 8 * 1024; }

// NW = 4: one wave per SIMD, each owning 64 pairs x 64 couts x 4 frequencies in 256 accumulator registers (1 fragment read
// per MFMA; the wave's own issue stream - 16 reads, 16 packed adds, 5 DMA pieces per 16 MFMAs - is what bounds it).
// NW = 8: two waves per SIMD, 32 pairs x 64 couts x 4 frequencies in 128 accumulator registers each (1.5 reads per MFMA, but
// the two issue streams of a SIMD overlap, as in the direct kernels).
// ABL (lab build only; wrong results by design): timing ablations of the k-loop - 1 no MFMA, 2 no fragment reads, 4 no LDS-DMA,
// 8 no barrier, 16 no B^T d arithmetic
// ROWP (lab build only - measured, not shipped): one patch per (channel block, KERNEL ROW) - 258 pixel rows whatever the map
// width - in a two-slot ring fired one kernel row ahead, instead of the super-patch of a channel block: for maps too wide
// for it (W > 30: the detector's).  Three times the patch bytes per channel block (the three row patches of a block overlap
// in all but 2 W pixels), the same fragment reads - and with 3.5 instead of 2.4 LDS-DMA pieces per wave and sub-step it runs
// at x0.87...0.99 of the DIRECT kernel on the detector's shapes (profiles/r3/wino_probe.txt): the DMA issue, already the
// kernel's largest cost next to the matrix time, eats the 1.5 x fewer MFMAs.  Bit-identical to the super-patch form.
template <int NW, int ABL = 0, bool ROWP = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void conv3x3_wino_kernel(ConvParams p_in) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    ConvParams p = p_in;
    constexpr int TP = WN_TP, TC = WN_TC;
    constexpr int PB = 8 / NW;                 // 32-pair blocks per wave
    constexpr int WPC = 16 / NW;               // weight pieces per wave and stage
    constexpr int PPS = WN_PIECES / NW;        // patch-piece issue slots per wave and patch
    constexpr int PPT_STEP = ROWP ? PPS / 2 : PPS / 6;   // ... per sub-step (super-patch: sub-steps 0..5 of a channel block; row patches:
                                                         // sub-steps 0, 1 of every kernel row)
    if (p.n_dev) {                             // image count known on the device only (threshold mode)
        int n = *p.n_dev;
        n = n < 0 ? 0 : (n > p.N ? p.N : n);
        p.M = n * p.Ho * p.Wo;
        p.n_ptiles = (p.M + TP - 1) / TP;
    }
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n_tiles = p.n_ptiles * p.n_ctiles;
    int t0, t1, tstep;
    if ((gridDim.x & 7) == 0) {                // XCD-interleaved tile walk (see conv3x3_lean.hip)
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
        const int cs = (int)((long)x * n_tiles / 8), ce = (int)((long)(x + 1) * n_tiles / 8);
        t0 = cs + j;
        t1 = ce;
        tstep = per;
    } else {
        t0 = (int)((long)blockIdx.x * n_tiles / gridDim.x);
        t1 = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
        tstep = 1;
    }
    if (t0 >= t1) return;
    const int cpt = p.Cin >> 6;                // 64-channel blocks
    const int cin2 = p.Cin * 2;                // bytes per pixel
    const int HALF = ROWP ? WN_ROWP_HALF : wino_half_rows(p.W);
    const int XSLOT = 2 * HALF * 128;
    const int OFF_W = 2 * XSLOT;
    const int OFF_Z = OFF_W + WN_NSW * WN_WSLOT;
    const int OFF_PAR = OFF_Z + 256;
    const int OFF_DUMP = OFF_PAR + 11 * TC * 4;
    const int npw = (HALF >> 2) / NW;          // real patch pieces per wave and channel block (<= PPS)

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    // ---------------- DMA lane geometry.  A piece fills 8 LDS rows x 128 B: lane -> row lane/8, 16-byte position lane%8,
    // which must hold logical chunk pos ^ ((row>>1)&7) (source-side swizzle; pieces start at multiples of 8 rows).
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));   // row group wave + NW j: (row >> 1) & 7 = (wave & 1) * 4 + lane / 16
    // patch piece j of this wave = LDS row group g = wave + NW j (8 rows).  Rows below HALF hold the even pixel offsets
    // r = 2 * row, the others r = 2 * (row - HALF) + 1 of the super-patch, whose pixel offset 0 is pixel m0 - W - 1.
    // x_src(j): per-lane byte offset of that pixel row's chunk relative to pixel m0, channel block 0 (recomputed per piece:
    // a dozen scalar-ish operations against six registers held through the k-loop)
    // = a per-lane constant (its row inside the piece, its chunk) + a per-piece scalar (HALF is a multiple of 32: a piece lies
    // in one half): one vector add per piece
    const int xlane = (2 * lrow - (ROWP ? 0 : p.W) - 1) * cin2 + lchunk * 16;   // (row patches: the kernel row's offset comes per patch)
    auto x_src = [&](int j) -> int {
        const int row0 = (wave + NW * j) * 8;                                      // scalar
        const int r0 = row0 < HALF ? 2 * row0 : 2 * (row0 - HALF) + 1;
        return xlane + r0 * cin2;
    };
    constexpr int DEAD = (int)0x80000000;

    // ---------------- consumer geometry: wave (wp, wc) owns pairs [32 PB wp, +32 PB) x couts [64 wc, +64) of the tile
    const int wave_p = wave >> 1, wave_c = wave & 1;
    const int pair0 = wave_p * (32 * PB), crow0 = wave_c * 64;
    const int fr = lane & 31, fh = lane >> 5;
    int fr_e = fr, fh_e = fh;                  // copies made opaque per tile / channel block (address arithmetic stays where it is used)
    const int HoWo = p.Ho * p.Wo;
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)p.Wo;
    // raw fragment (pair block b, position i = 0..3, kernel row kh): pixel offset r = 2 pair + i + kh W of the super-patch
    // -> LDS row (r >> 1) + (r & 1) HALF = pair + (i >> 1) + kh W/2 + (i & 1) HALF; byte address of its kk = 0 fragment
    // (chunk fh) inside a patch slot.  kk flips address bits 5..6.
    auto pv_of = [&](int kh, int b, int i) -> int {
        const int row = pair0 + b * 32 + fr_e + (i >> 1) + (ROWP ? 0 : kh * (p.W >> 1)) + (i & 1) * HALF;
        return row * 128 + ((fh ^ ((row >> 1) & 7)) << 4);
    };
    // weight fragment (cout block c, frequency f) of a stage: wino_u_lane(fr, fh) + f * 4096 + (2 wave_c + c) * 1024 (see wino_u_lane)
    const int aoff = OFF_W + wave_c * 2048 + wino_u_lane(fr, fh);

    floatx16 acc[4][PB][2];                    // [frequency][pair block][cout block]
    half8 raw[PB][4];                          // [pair block][position]: raw fragments of the NEXT sub-step
    half8 vcur[4][PB];                         // [frequency][pair block]: B^T d of the running sub-step
    half8 uf[4][2];                            // [frequency][cout block]
    int radr[PB][4];                           // per (channel block, kernel row): address of this lane's raw fragments (patch or zero block)

    // ---------------- per-tile epilogue parameters in LDS: bias [9][TC] (class-major; one class without border bias) and
    // PReLU slope [TC], 5 KiB, brought in by LDS-DMA with per-lane global addresses at the first sub-step of a tile (one
    // piece per wave: every wave must issue the same number of counted operations) - no registers, and no drain of the
    // in-order counter as a register round trip (load, wait, ds_write) costs.  Couts beyond Cout and unused classes read
    // clamped addresses: their values are never stored.
    float* lds_bias = reinterpret_cast<float*>(smem + OFF_PAR);            // [9][TC]
    float* lds_slope = lds_bias + 9 * TC;                                   // [TC]
    constexpr int PARP = (8 + NW - 1) / NW;                                 // parameter pieces per wave (5 real ones in all)
    const bool border = p.flags & FRP_FLAG_BORDER_BIAS;
    auto dma_params = [&](int tile) {
        const int c0p = (tile % p.n_ctiles) * TC;
#pragma unroll
        for (int j = 0; j < PARP; ++j) {
            const int q = wave + NW * j;
            const int o = q * 1024 + lane * 16;
            const char* src = reinterpret_cast<const char*>(p.bias);
            if (o < 9 * TC * 4) {
                const int cls = border ? o / (TC * 4) : 0;
                int co = c0p + ((o % (TC * 4)) >> 2);
                co = co + 4 <= p.Cout ? co : 0;
                src = reinterpret_cast<const char*>(p.bias + (long)cls * p.Cout + co);
            } else if (o < 10 * TC * 4 && p.slope) {
                int co = c0p + ((o - 9 * TC * 4) >> 2);
                co = co + 4 <= p.Cout ? co : 0;
                src = reinterpret_cast<const char*>(p.slope + co);
            }
            unsigned char* dst = q < 5 ? smem + OFF_PAR + q * 1024 : smem + OFF_DUMP + wave * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    // ---------------- the DMA stream.  Weight stage = 16 KiB, contiguous in the image: ((ctile * cpt + cb) * 3 + kh) * 4 + kk.
    // A tile's stages follow each other in the image; the stream crosses into the next tile of this workgroup.
    struct Tile { int m0b; int wbase; unsigned vmask[PB]; };
    // vmask[b]: validity bits of pair block b's pair of this lane: 1 pair exists, 2 row above inside, 4 row below inside,
    // 8 left neighbour (d0) inside, 16 right neighbour (d3) inside
    auto make_tile = [&](int tile, Tile& d) {
        if (tile >= t1) {
            d.m0b = DEAD; d.wbase = DEAD;
#pragma unroll
            for (int b = 0; b < PB; ++b) d.vmask[b] = 0;
            return;
        }
        const int pt = tile / p.n_ctiles;
        const int ct_ = tile - pt * p.n_ctiles;
        d.m0b = pt * TP * cin2;
        d.wbase = ct_ * cpt * 12 * WN_WSLOT;
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int m = pt * TP + 2 * (pair0 + b * 32 + fr);
            unsigned mask = 0;
            if (m < p.M) {
                int n, rem, oy, ox;
                fast_divmod(m, HoWo, inv_howo, n, rem);
                fast_divmod(rem, p.Wo, inv_wo, oy, ox);
                mask = 1u | (oy > 0 ? 2u : 0u) | (oy < p.H - 1 ? 4u : 0u) | (ox > 0 ? 8u : 0u) | (ox + 2 < p.W ? 16u : 0u);
            }
            d.vmask[b] = mask;
        }
    };
    // weight stage `st` (0..12 cpt - 1 of the tile, or beyond: the next tile's) into ring slot `slot`: this wave's pieces
    // (the stage / piece part of the source address rides in the SGPR offset operand - inside the image by construction -, the
    // per-lane part is the constant lane * 16: no vector arithmetic per piece; past the last tile the lane offset is out of range)
    auto w_stage = [&](int wbase, int st, int slot) {
        const unsigned voff = wbase == DEAD ? CONV_OOB : (unsigned)(lane * 16);
#pragma unroll
        for (int j = 0; j < WPC; ++j) {
            const int q = wave + NW * j;
            if constexpr (!(ABL & 4))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(smem + OFF_W + slot * WN_WSLOT + q * 1024), 16,
                                                         voff, wbase == DEAD ? 0 : wbase + st * WN_WSLOT + q * 1024, 0, 0);
        }
    };
    // patch piece slot j (0..PPS-1) of channel block byte offset cbs of the tile at m0b into patch slot `xs`
    auto x_piece = [&](int m0b, int cbs, int j, int xs, int kh = 1) {
        const bool real = j < npw && m0b != DEAD;
        const unsigned off = real ? (unsigned)(m0b + x_src(j) + cbs + (ROWP ? (kh - 1) * p.W * cin2 : 0)) : CONV_OOB;
        unsigned char* dst = real ? smem + xs + (wave + NW * j) * 1024 : smem + OFF_DUMP + wave * 1024;
        if constexpr (!(ABL & 4)) dma16(xrsrc, dst, off);
    };

    // ---------------- prologue
    stamp(p.stamps, 0);
    if (t < 64) reinterpret_cast<unsigned*>(smem + OFF_Z)[t] = 0u;
    Tile cur, nt;
    make_tile(t0, cur);
#pragma unroll
    for (int j = 0; j < PPS; ++j) x_piece(cur.m0b, 0, j, 0, 0);
    w_stage(cur.wbase, 0, 0);
    w_stage(cur.wbase, 1, 1);
    w_stage(cur.wbase, 2, 2);
    int xs = 0;                                // byte offset of the patch slot the running channel block reads
    const bool has_res = p.res != nullptr;

    auto set_radr = [&](const Tile& tl, int kh, int xslot) {
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const unsigned m = tl.vmask[b];
            const bool rowok = (m & 1u) && (kh == 0 ? (m & 2u) : kh == 2 ? (m & 4u) : true);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = rowok && (i == 0 ? (m & 8u) : i == 3 ? (m & 16u) : true);
                const int a_ = pv_of(kh, b, i);
                radr[b][i] = ok ? xslot + a_ : OFF_Z + (a_ & 255);
            }
        }
    };
    auto read_raw = [&](int kk) {
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (!(ABL & 2)) raw[b][i] = *reinterpret_cast<const half8*>(smem + (radr[b][i] ^ (kk << 5)));
                else asm volatile("" : "+v"(raw[b][i]) : "v"(radr[b][i]));
            }
    };
    auto transform_raw = [&]() {               // B^T d: V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            if constexpr (!(ABL & 16)) {
                vcur[0][b] = wsub(raw[b][0], raw[b][2]);
                vcur[1][b] = wadd(raw[b][1], raw[b][2]);
                vcur[2][b] = wsub(raw[b][2], raw[b][1]);
                vcur[3][b] = wsub(raw[b][1], raw[b][3]);
            } else {
#pragma unroll
                for (int f = 0; f < 4; ++f) vcur[f][b] = raw[b][f];
            }
        }
    };
    auto read_u = [&](int slot, int f) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if constexpr (!(ABL & 2)) uf[f][c] = *reinterpret_cast<const half8*>(smem + aoff + f * 4096 + c * 1024 + slot * WN_WSLOT);
            else asm volatile("" : "+v"(uf[f][c]) : "v"(aoff));
        }
    };
    auto mfma_f = [&](int f, const half8 (&v)[PB]) {
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if constexpr (!(ABL & 1)) acc[f][b][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uf[f][c], v[b], acc[f][b][c], 0, 0, 0);
                else asm volatile("" : "+v"(acc[f][b][c]) : "v"(uf[f][c]), "v"(v[b]));
            }
    };

    // first sub-step's operands: wait for the patch and stages 0, 1, then read as every later sub-step does one step ahead
    wait_vmcnt<WPC>();
    __syncthreads();
    set_radr(cur, 0, 0);
    read_raw(0);
    read_u(0, 0);
    read_u(0, 1);
    transform_raw();
    stamp(p.stamps, 1);

    for (int ct = t0; ct < t1; ct += tstep) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int b = 0; b < PB; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[f][b][c][e] = 0.f;
        const int ptile = ct / p.n_ctiles;
        const int m0 = ptile * TP;
        const int c0 = (ct - ptile * p.n_ctiles) * TC;
        make_tile(ct + tstep, nt);
        asm volatile("" : "+v"(fr_e), "+v"(fh_e));
        if (ct == t0) stamp(p.stamps, 2);

        // One sub-step T = kh * 4 + kk of channel block cb (all static but cb): raw set T & 1 and uf[0], uf[1] hold its
        // operands already.  It waits until weight stage T + 1 has landed (issued two sub-steps ago; behind it in the
        // in-order counter: the pieces of sub-step T - 1 = 2 patch pieces if (T - 1) mod 12 <= 5, and 4 weight pieces),
        // fires the patch pieces of the next channel block (sub-steps 0..5) and weight stage T + 3, reads the raw fragments
        // and the first weight fragments of sub-step T + 1 under its MFMAs.
#define WN_STEP(KH, KK)                                                                                                \
    do {                                                                                                               \
        constexpr int T = (KH) * 4 + (KK);                                                                             \
        constexpr int TN = (T + 1) % 12, KHN = TN / 4, KKN = TN % 4;                                                   \
        if (T == 0 && cb == 0 && ct != t0) wait_vmcnt<0>();          /* the epilogue's stores sit in the counter too */ \
        else if (T == 1 && cb == 0) wait_vmcnt<WPC + PPT_STEP + PARP>();   /* sub-step 0 of a tile also issued the parameter pieces */ \
        else if (ROWP ? (((T + 11) % 12) % 4) <= 1 : ((T + 11) % 12) <= 5) wait_vmcnt<WPC + PPT_STEP>(); else wait_vmcnt<WPC>(); \
        if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();                                                        \
        /* V of this sub-step and uf[0], uf[1] are in registers: the matrix pipe starts right behind the barrier, everything  \
           else of the step - DMA issue, fragment reads, the next step's B^T d - is placed between its MFMAs */        \
        mfma_f(0, vcur[0]);                                                                                            \
        read_u(T & 3, 2);                                                                                              \
        read_u(T & 3, 3);                                                                                              \
        /* operands of the next sub-step (not across a tile's epilogue: the registers held there spill; the next tile  \
           reads them after it): its kernel row's addresses (next channel block at T = 11), its raw fragments */       \
        const bool pf = !(T == 11 && last_cb);                                                                         \
        if (KKN == 0 && pf) {                                                                                          \
            if (T == 11 || ROWP) set_radr(cur, KHN, xs ^ XSLOT_X); else set_radr(cur, KHN, xs);                        \
        }                                                                                                              \
        if (pf) read_raw(KKN);                                                                                         \
        /* (the DMA pieces sit here, right behind the first MFMAs: issued late in the sub-step - behind the fragment reads, where  \
           a piece is said to be cheapest - the six shapes ran at x1.10 instead of x1.13 of the direct kernel) */            \
        /* epilogue parameters of this tile: LDS-DMA, first in this sub-step's issue order (visible to all waves from the   \
           barrier of sub-step 2 on; the previous tile's epilogue, their last reader, lies before this barrier) */       \
        if (T == 0 && cb == 0) dma_params(ct);                                                                         \
        if constexpr (!ROWP && T <= 5) {                                                                               \
            _Pragma("unroll") for (int q_ = 0; q_ < PPT_STEP; ++q_) x_piece(nxm0b, ncbs, (PPT_STEP * T + q_) % PPS, xs ^ XSLOT_X); \
        }                                                                                                              \
        if constexpr (ROWP && (KK) <= 1) {     /* the next kernel row's patch: row KH + 1 of this block, or row 0 of what lies beyond it */ \
            _Pragma("unroll") for (int q_ = 0; q_ < PPT_STEP; ++q_) {                                                  \
                if ((KH) < 2) x_piece(cur.m0b, cb << 7, (PPT_STEP * (KK) + q_) % PPS, xs ^ XSLOT_X, (KH) + 1);          \
                else x_piece(nxm0b, ncbs, (PPT_STEP * (KK) + q_) % PPS, xs ^ XSLOT_X, 0);                              \
            }                                                                                                          \
        }                                                                                                              \
        if (T < 9) w_stage(cur.wbase, cb * 12 + T + 3, (T + 3) & 3); else w_stage(nxw, nxst + T - 9, (T + 3) & 3);     \
        mfma_f(1, vcur[1]);                                                                                            \
        mfma_f(2, vcur[2]);                                                                                            \
        /* write after read (conv_common.h: retire_lds_reads): every fragment read issued so far - the last ones of weight  \
           slot T & 3 among them, old by now - has returned before the reads of the NEXT sub-step's operands go out; those  \
           are the only ones in flight at the next barrier, and they and the rest of slot (T + 1) & 3's reads return at this  \
           point of sub-step T + 1, a barrier before sub-step T + 2 restages that slot.  (In front of the barrier the wait cost  \
           1.7 % of the embedder: it held the prefetched operands of the first MFMAs back.) */                          \
        retire_lds_reads();                                                                                            \
        if (pf) { read_u((T + 1) & 3, 0); read_u((T + 1) & 3, 1); }                                                    \
        mfma_f(3, vcur[3]);                                                                                            \
        if (pf) transform_raw();                                                                                       \
        if (ROWP && (KK) == 3) xs ^= XSLOT_X;  /* row patches: the slots alternate per kernel row */                    \
    } while (0)

        const int XSLOT_X = XSLOT;
        for (int cb = 0; cb < cpt; ++cb) {
            const bool last_cb = cb + 1 == cpt;
            // what lies beyond this channel block: the next block of this tile, or block 0 of the next tile
            const int nxm0b = last_cb ? nt.m0b : cur.m0b;
            const int ncbs = last_cb ? 0 : (cb + 1) << 7;
            const int nxw = last_cb ? nt.wbase : cur.wbase;
            const int nxst = last_cb ? 0 : (cb + 1) * 12;
            asm volatile("" : "+v"(fr_e));     // (opaque per channel block: the address arithmetic is not hoisted out of the loop)
            WN_STEP(0, 0);
            if (cb == 0 && ct == t0) stamp(p.stamps, 3);
            WN_STEP(0, 1); WN_STEP(0, 2); WN_STEP(0, 3);
            WN_STEP(1, 0); WN_STEP(1, 1); WN_STEP(1, 2); WN_STEP(1, 3);
            WN_STEP(2, 0); WN_STEP(2, 1); WN_STEP(2, 2); WN_STEP(2, 3);
            if (!ROWP) xs ^= XSLOT;
        }
#undef WN_STEP

        if (ct == t0) stamp(p.stamps, 4);
        // ---------------- epilogue (wino_tile_epilogue above)
        {
#define WN_EPI(FULL_, ACT_, RES_) \
    wino_tile_epilogue<PB, FULL_, ACT_, RES_>(p, acc, lds_bias, lds_slope, m0, c0, pair0, crow0, fr_e, fh_e, HoWo, inv_howo, inv_wo)
            const bool full = m0 + TP <= p.M && c0 + TC <= p.Cout;
            if (!full) WN_EPI(false, -1, -1);
            else if (p.act == FRP_ACT_PRELU) { if (has_res) WN_EPI(true, FRP_ACT_PRELU, 1); else WN_EPI(true, FRP_ACT_PRELU, 0); }
            else if (p.act == FRP_ACT_RELU) { if (has_res) WN_EPI(true, FRP_ACT_RELU, 1); else WN_EPI(true, FRP_ACT_RELU, 0); }
            else { if (has_res) WN_EPI(true, FRP_ACT_NONE, 1); else WN_EPI(true, FRP_ACT_NONE, 0); }
#undef WN_EPI
        }
        cur = nt;
        if (ct + tstep < t1) {                 // first sub-step's operands of the next tile (its patch and stage 0 landed before this
            set_radr(cur, 0, xs);              // tile's last barrier)
            read_raw(0);
            read_u(0, 0);
            read_u(0, 1);
            transform_raw();
        }
        if (ct == t0) stamp(p.stamps, 5);
    }
    stamp(p.stamps, 6);
}


// =====================================================================================================================
// Second generation of the k-loop (round 4): the same tile, rings, arithmetic and epilogue - bit-identical results - with
// the sub-step's instruction stream written out by hand.  What the compiler made of the first generation (ISA of round 3):
// every fragment register was reused as soon as it was free, so four `s_waitcnt lgkmcnt(0)` per sub-step sat directly in
// front of MFMAs whose operand had been requested one or two instructions earlier (the whole LDS latency exposed, in both
// waves of a SIMD at the same time); per-read address arithmetic (xor per weight fragment); 12 spilled vector registers
// and 75 scalar spills.  Here every instruction of the loop is a pinned `asm volatile` statement in the order below - the
// compiler only allocates registers - and
//   * ALL fragments of sub-step T + 1 (8 weight fragments, 4 raw pixel fragments) are requested during sub-step T, spread
//     between its MFMAs, into registers that the MFMAs of T have already released: nothing an MFMA needs is ever
//     younger than half a sub-step, the only LDS wait inside a sub-step is a counted one for the raw fragments before
//     B^T d, and the sub-step after a barrier starts on operands that sit in registers;
//   * weight fragments are read at ONE per-lane base + immediates (stage layout: wino_u_slot), raw fragments at
//     per-kernel-row addresses kept in registers for a whole channel block;
// Per sub-step and wave: 8 MFMAs, 12 ds_read_b128, 16 v_pk_add_f16, <= 3 xors, 2-3 LDS-DMA pieces, 3 waits, 1 barrier.
// The weight ring runs THREE sub-steps ahead (stage T + 4 is issued in sub-step T, into the slot whose fragments sub-step T
// itself computes on): with two, the late waves of a workgroup waited ~200 cycles for their own pieces in front of every
// barrier (tools/wino_substep.py).  Write-after-read on the rings: every wave waits for ALL its fragment requests in front of
// the barrier of sub-step T (lgkmcnt(0): they are the operands of T, requested during T - 1 from slot T & 3), so no read of
// that slot is in flight anywhere when the DMA behind the barrier restages it; a patch slot is restaged one to five
// sub-steps after the last read of it was consumed.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ZERO: the first MFMA into an accumulator of a tile - C is the inline constant 0, the accumulator is only written (no 128 v_mov
// per wave and tile to clear it first; early-clobber: the destination may not overlap the sources)
template <bool ZERO = false>
__device__ __forceinline__ void wn2_mfma(floatx16& acc, const half8& a, const half8& b) {
    if constexpr (ZERO) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int OFF>
__device__ __forceinline__ void wn2_read(half8& d, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset is a 16-bit field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// LDS-DMA piece (64 lanes x 16 B -> LDS m0v + lane * 16) from buffer `rsrc` at per-lane offset voff + scalar offset soff.
// M0 is written in the statement that uses it (the compiler does not preserve it around asm).
__device__ __forceinline__ void wn2_dma(const u32x4& rsrc, unsigned m0v, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void wn2_dma_global(unsigned m0v, const void* src) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src) : "memory");
}
__device__ __forceinline__ void wn2_barrier() { asm volatile("s_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wn2_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wn2_pk(unsigned& d, unsigned a, unsigned b) { asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); }
__device__ __forceinline__ void wn2_pk_sub(unsigned& d, unsigned a, unsigned b) {
    asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ half8 wn2_add(const half8& a, const half8& b) {
    WH8 x = __builtin_bit_cast(WH8, a), y = __builtin_bit_cast(WH8, b), r;
#pragma unroll
    for (int i = 0; i < 4; ++i) wn2_pk(r.p[i], x.p[i], y.p[i]);
    return __builtin_bit_cast(half8, r);
}
__device__ __forceinline__ half8 wn2_sub(const half8& a, const half8& b) {
    WH8 x = __builtin_bit_cast(WH8, a), y = __builtin_bit_cast(WH8, b), r;
#pragma unroll
    for (int i = 0; i < 4; ++i) wn2_pk_sub(r.p[i], x.p[i], y.p[i]);
    return __builtin_bit_cast(half8, r);
}
template <int T> struct wn2_step_tag { static constexpr int value = T; };
struct wn2_true { static constexpr bool value = true; };
struct wn2_false { static constexpr bool value = false; };
// lab build, VAR & 2: shader-clock stamps inside ONE sub-step (tools/wino_substep.py); s_memtime returns through lgkmcnt, out of
// order with the LDS reads: the values are only read behind an lgkmcnt(0) at the end of the sampled sub-steps
// (a result still in flight at the end of an asm statement may be copied - spilled to a vector lane - before it has arrived, and
// eight 64-bit values in scalar registers did not survive the loop's scalar pressure: the stamp is waited for inside its
// statement - so stamps are only placed where the wave's LDS queue is nearly empty anyway - and parked in two lanes of ONE
// vector register)
template <int SLOT>
__device__ __forceinline__ void wn2_clock(unsigned& park) {
    asm volatile("s_memtime vcc\n\ts_waitcnt lgkmcnt(0)\n\tv_writelane_b32 %0, vcc_lo, %1\n\tv_writelane_b32 %0, vcc_hi, %2" : "+v"(park) : "n"(2 * SLOT), "n"(2 * SLOT + 1) : "vcc");
}

template <int SLOT>
__device__ __forceinline__ void wn2_clock_real(unsigned& park) {         // the constant 100 MHz clock next to it (in-kernel clock = ratio of the deltas)
    asm volatile("s_memrealtime vcc\n\ts_waitcnt lgkmcnt(0)\n\tv_writelane_b32 %0, vcc_lo, %1\n\tv_writelane_b32 %0, vcc_hi, %2" : "+v"(park) : "n"(2 * SLOT), "n"(2 * SLOT + 1) : "vcc");
}

#define WN2_PPW 5            // patch pieces per wave and channel block (half_rows = 160 for every W <= 30: 40 row groups / 8 waves)
                             // = the sub-steps of a block that carry patch pieces

// VAR (lab build): 1 = waves 4-7 at s_setprio 1
template <int VAR = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_wino2_kernel(ConvParams p_in) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    ConvParams p = p_in;
    constexpr int TP = WN_TP, TC = WN_TC;
    constexpr int AHEAD = (VAR & 16) ? 2 : 3;  // sub-steps the weight ring runs ahead (lab: VAR & 16 = two)
    // VAR & 32: 2-D tiles for maps wider than the super-patch allows (the detector's): a tile = 8 rows x 30 columns of one image.
    // Its halo'd patch is 10 x 32 pixels = the 320 LDS rows of the flattened form, and with 16 pairs per tile row (15 real + one that
    // never exists) every consumer-side address is the flattened formula at W = 32: the kernel runs on a 32-wide virtual strip.
    // What differs: the source side of the patch DMA (a piece = half a patch row), the validity masks, the epilogue's pixel index.
    constexpr bool T2D = (VAR & 32) != 0;
    constexpr int T2H = 8, T2W = 30;
    const int t2x = T2D ? (p.W + T2W - 1) / T2W : 1, t2y = T2D ? (p.H + T2H - 1) / T2H : 1;      // tiles per image row / column
    if (!T2D && p.n_dev) {                     // image count known on the device only (threshold mode)
        int n = *p.n_dev;
        n = n < 0 ? 0 : (n > p.N ? p.N : n);
        p.M = n * p.Ho * p.Wo;
        p.n_ptiles = (p.M + TP - 1) / TP;
    }
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n_tiles = p.n_ptiles * p.n_ctiles;
    int t0, t1, tstep;
    if ((gridDim.x & 7) == 0) {                // XCD-interleaved tile walk (see conv3x3_lean.hip)
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
        const int cs = (int)((long)x * n_tiles / 8), ce = (int)((long)(x + 1) * n_tiles / 8);
        t0 = cs + j;
        t1 = ce;
        tstep = per;
    } else {
        t0 = (int)((long)blockIdx.x * n_tiles / gridDim.x);
        t1 = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
        tstep = 1;
    }
    if (t0 >= t1) return;
    const int cpt = p.Cin >> 6;                // 64-channel blocks
    const int cin2 = p.Cin * 2;                // bytes per pixel
    const int HALF = T2D ? 160 : wino_half_rows(p.W);      // 160 (launcher: 40 row groups of 8 rows per patch)
    const int XSLOT = 2 * HALF * 128;
    const int OFF_W = 2 * XSLOT;
    const int OFF_Z = OFF_W + WN_NSW * WN_WSLOT;
    const int OFF_PAR = OFF_Z + 256;
    const int OFF_DUMP = OFF_PAR + 11 * TC * 4;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;   // LDS byte address of smem

    // buffer descriptors as plain SGPR quads (asm operands): base, base_hi (stride 0), num_records, flags
    const unsigned long xa = (unsigned long)p.x, wa = (unsigned long)p.w;
    const u32x4 xrsrc = {(unsigned)xa, (unsigned)(xa >> 32) & 0xffffu, p.x_bytes, 0x00020000u};
    const u32x4 wrsrc = {(unsigned)wa, (unsigned)(wa >> 32) & 0xffffu, p.w_bytes, 0x00020000u};

    // The first tile's weight stages 0..AHEAD go out before anything else is set up: their operands are a handful of scalars, and the
    // ~1.3 us until a first stage has landed then run under the rest of the prologue (lane geometry, masks, the patch's addresses)
    // instead of behind it.  (The patch pieces follow below, so the in-order counter holds: stages, then patch - see the wait.)
    {
        constexpr int NDW0 = (VAR & 8) ? 4 : 8, WPW0 = 16 / NDW0;
        const int wbase0 = (t0 % p.n_ctiles) * cpt * 12 * WN_WSLOT;
        if (!(VAR & 8) || wave < 4) {
#pragma unroll
            for (int st = 0; st < (AHEAD + 1); ++st)
#pragma unroll
                for (int j = 0; j < WPW0; ++j)
                    wn2_dma(wrsrc, lds0 + OFF_W + st * WN_WSLOT + (wave + NDW0 * j) * 1024, (unsigned)(lane * 16),
                            (unsigned)(wbase0 + st * WN_WSLOT + (wave + NDW0 * j) * 1024));
        }
    }

    // ---------------- DMA lane geometry (as the first generation): a piece fills 8 LDS rows x 128 B, lane -> row lane / 8,
    // 16-byte position lane % 8 holding logical chunk pos ^ ((row >> 1) & 7) (source-side swizzle)
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
    const int xlane = (T2D ? 2 * lrow : 2 * lrow - p.W - 1) * cin2 + lchunk * 16;     // (2-D: the patch origin is part of the tile's base)
    // patch piece j of this wave = LDS row group wave + 8 j; rows below HALF hold the even pixel offsets r = 2 row, the
    // others r = 2 (row - HALF) + 1 of the super-patch; its scalar source offset (added to the per-lane part per piece)
    // VAR & 8 (lab): only waves 0-3 - the older wave of every SIMD, which reaches the barrier ~170 cycles before its partner -
    // issue the DMA stream (twice the pieces each): NDW issuing waves, XPW patch / WPW weight pieces per issuing wave
    constexpr bool HALF_DMA = (VAR & 8) != 0;
    constexpr int NDW = HALF_DMA ? 4 : 8, XPW = 40 / NDW, WPW = 16 / NDW, XPS = XPW / 5;
    const bool dma_wave = !HALF_DMA || wave < 4;
    int sxo[XPW];
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
        const int row0 = (wave + NDW * j) * 8;
        if constexpr (T2D) {                   // row group g: patch row py = g' / 2, its pair columns 8 (g' & 1) .. + 7, g' = g mod 20; parity g / 20
            const int g = wave + NDW * j, gg = g < 20 ? g : g - 20;
            sxo[j] = ((gg >> 1) * p.W + ((gg & 1) << 4) + (g < 20 ? 0 : 1)) * cin2;
        } else
        sxo[j] = (row0 < HALF ? 2 * row0 : 2 * (row0 - HALF) + 1) * cin2;
    }
    constexpr int DEAD = (int)0x80000000;

    // ---------------- consumer geometry: wave (wp, wc) owns pairs [32 wp, +32) x couts [64 wc, +64) of the tile
    const int wave_p = wave >> 1, wave_c = wave & 1;
    const int pair0 = wave_p * 32, crow0 = wave_c * 64;
    const int fr = lane & 31, fh = lane >> 5;
    int fr_e = fr, fh_e = fh, lane_e = lane;   // copies made opaque per tile (address arithmetic stays where it is used: hoisted out of
                                               // the tile loop it is spilled at kernel start and reloaded in every epilogue)
    const int HoWo = p.Ho * p.Wo;
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)p.Wo;
    const unsigned ubase = lds0 + OFF_W + wave_c * 2048 + wino_u_lane(fr, fh);
    // raw fragment (position i = 0..3, kernel row kh): LDS row pair + (i >> 1) + kh W/2 + (i & 1) HALF of a patch slot;
    // byte address of its kk = 0 fragment (kk flips address bits 5..6); lanes whose tap falls off the image (or whose pair
    // does not exist) read a 256-byte zero block at the bank offset their real address would have had
    auto raw_addr = [&](unsigned mask, int kh, int i, int xslot) -> unsigned {
        const int row = pair0 + fr_e + (i >> 1) + kh * (T2D ? 16 : (p.W >> 1)) + (i & 1) * HALF;
        const int a_ = row * 128 + ((fh ^ ((row >> 1) & 7)) << 4);
        const bool rowok = (mask & 1u) && (kh == 0 ? (mask & 2u) : kh == 2 ? (mask & 4u) : true);
        const bool ok = rowok && (i == 0 ? (mask & 8u) : i == 3 ? (mask & 16u) : true);
        return lds0 + (ok ? xslot + a_ : OFF_Z + (a_ & 255));
    };

    floatx16 acc[4][2];                        // [frequency][cout block]
    half8 U[2][8];                             // [sub-step parity][frequency * 2 + cout block]: weight fragments
    half8 V[2][4];                             // [sub-step parity][frequency]: B^T d
    half8 raw[4];                              // raw pixel fragments of the next sub-step (dead after B^T d)
    unsigned rad[4];                           // raw-fragment addresses (kk = 0) of the kernel row the next reads belong to

    float* lds_bias = reinterpret_cast<float*>(smem + OFF_PAR);            // [9][TC]
    float* lds_slope = lds_bias + 9 * TC;                                   // [TC]
    const bool border = p.flags & FRP_FLAG_BORDER_BIAS;
    // epilogue parameters of a tile (bias classes + slopes, 5 KiB): one LDS-DMA piece per wave with per-lane global addresses
    auto dma_params = [&](int tile) {
        const int c0p = (tile % p.n_ctiles) * TC;
        const int o = wave * 1024 + lane_e * 16;
        const char* src = reinterpret_cast<const char*>(p.bias);
        if (o < 9 * TC * 4) {
            const int cls = border ? o / (TC * 4) : 0;
            int co = c0p + ((o % (TC * 4)) >> 2);
            co = co + 4 <= p.Cout ? co : 0;
            src = reinterpret_cast<const char*>(p.bias + (long)cls * p.Cout + co);
        } else if (o < 10 * TC * 4 && p.slope) {
            int co = c0p + ((o - 9 * TC * 4) >> 2);
            co = co + 4 <= p.Cout ? co : 0;
            src = reinterpret_cast<const char*>(p.slope + co);
        }
        wn2_dma_global(lds0 + (wave < 5 ? OFF_PAR + wave * 1024 : OFF_DUMP + wave * 1024), src);
    };

    struct Tile { int m0b; int wbase; unsigned vmask; };
    int mt_cur = 0, mt_nxt = 0;                // (2-D tiles) pixel index of the tile's first pixel
    // vmask: validity bits of this lane's pair: 1 pair exists, 2 row above inside, 4 row below inside, 8 left neighbour (d0)
    // inside, 16 right neighbour (d3) inside
    auto make_tile = [&](int tile, Tile& d) {
        if (tile >= t1) { d.m0b = DEAD; d.wbase = DEAD; d.vmask = 0; return; }
        const int pt = tile / p.n_ctiles;
        const int ct_ = tile - pt * p.n_ctiles;
        d.wbase = ct_ * cpt * 12 * WN_WSLOT;
        if constexpr (T2D) {
            const int per = t2x * t2y;
            const int n = pt / per, r = pt - n * per;
            const int ty = r / t2x, tx = r - ty * t2x;
            const int y0 = ty * T2H, x0 = tx * T2W;
            mt_nxt = (n * p.H + y0) * p.W + x0;
            d.m0b = (mt_nxt - p.W - 1) * cin2;                        // the patch starts one row above, one column left of the tile
            const int pl = pair0 + fr_e, y = y0 + (pl >> 4), x = x0 + 2 * (pl & 15);
            unsigned mask = 0;
            if ((pl & 15) < 15 && y < p.H && x < p.W)
                mask = 1u | (y > 0 ? 2u : 0u) | (y + 1 < p.H ? 4u : 0u) | (x > 0 ? 8u : 0u) | (x + 2 < p.W ? 16u : 0u);
            d.vmask = mask;
            return;
        }
        d.m0b = pt * TP * cin2;
        const int m = pt * TP + 2 * (pair0 + fr_e);
        unsigned mask = 0;
        if (m < p.M) {
            int n, rem, oy, ox;
            fast_divmod(m, HoWo, inv_howo, n, rem);
            fast_divmod(rem, p.Wo, inv_wo, oy, ox);
            mask = 1u | (oy > 0 ? 2u : 0u) | (oy + 1 < p.H ? 4u : 0u) | (ox > 0 ? 8u : 0u) | (ox + 2 < p.W ? 16u : 0u);
        }
        d.vmask = mask;
    };

    // ---------------- prologue: patch of the first tile's first channel block, weight stages 0..3
    unsigned clkk = 0;                         // (lab, VAR & 2: shader clock and 100 MHz clock at kernel start / end in its lanes 8..15)
    if constexpr (VAR & 2) { if (p.stamps && blockIdx.x < 32) { wn2_clock<4>(clkk); wn2_clock_real<5>(clkk); } }
    else if constexpr (VAR & 64) stamp(p.stamps, 0);          // (100 MHz phase stamps: lab variant VAR & 64 only - the stamp address is a
                                                              //  per-lane value the shipped kernel would keep spilled from start to end)
    if (t < 64) reinterpret_cast<unsigned*>(smem + OFF_Z)[t] = 0u;
    Tile cur, nt;
    make_tile(t0, cur);
    if constexpr (T2D) mt_cur = mt_nxt;
    {
        const unsigned xb0 = (unsigned)(cur.m0b + xlane);
        if (dma_wave) {
#pragma unroll
            for (int j = 0; j < XPW; ++j) wn2_dma(xrsrc, lds0 + (wave + NDW * j) * 1024, xb0 + sxo[j], 0u);
        }
    }
    int xs = 0;                                // byte offset of the patch slot the running channel block reads
    const bool has_res = p.res != nullptr;
    if constexpr (VAR & 1) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }

    // first sub-step's operands of a tile (its patch in slot `xslot`, weight stage 0 in ring slot 0): plain loads, waited for by the
    // compiler; every later sub-step's operands are requested one sub-step ahead inside the loop
    auto first_operands = [&](const Tile& tl, int xslot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) raw[i] = *reinterpret_cast<const half8*>(smem + (raw_addr(tl.vmask, 0, i, xslot) - lds0));
#pragma unroll
        for (int e = 0; e < 8; ++e) U[0][e] = *reinterpret_cast<const half8*>(smem + (ubase - lds0) + (e >> 1) * 4096 + (e & 1) * 1024);
        V[0][0] = wn2_sub(raw[0], raw[2]);
        V[0][1] = wn2_add(raw[1], raw[2]);
        V[0][2] = wn2_sub(raw[2], raw[1]);
        V[0][3] = wn2_sub(raw[1], raw[3]);
    };
    wait_vmcnt<0>();                        // the patch - the youngest of the prologue's requests - is in, and with it stages 0..AHEAD
    __syncthreads();
    first_operands(cur, 0);
    if constexpr (VAR & 64) stamp(p.stamps, 1);

    for (int ct = t0; ct < t1; ct += tstep) {
        // (flattened tiles: the accumulators are not cleared - the tile's first sub-step writes them with C = 0, wn2_mfma<true>: 128 v_mov
        // per wave and tile less, and with them gone the compiler moves the tile's address set-up into the first sub-step's gaps;
        // in the 2-D form the same change made the allocator spill four registers and cost more than it saved: 136 x 240 x 128
        // 261-266 us with the clears, 275 with C = 0 (-DWN_T2D_C0=1))
#ifndef WN_T2D_C0
#define WN_T2D_C0 0
#endif
        if constexpr (T2D && !WN_T2D_C0) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[f][c][e] = 0.f;
        }
        const int ptile = ct / p.n_ctiles;
        const int m0 = ptile * TP;
        const int c0 = (ct - ptile * p.n_ctiles) * TC;
        make_tile(ct + tstep, nt);
        asm volatile("" : "+v"(fr_e), "+v"(fh_e), "+v"(lane_e));
        // Raw-fragment addresses of the tile, once: the patch-slot-independent part (or the zero block for taps off the image) and a bit
        // per address that says which kind it is; a kernel row's four addresses are then slot offset + base where the bit is set
        // (per kernel row and channel block the full address arithmetic - row, swizzle, two selects - was ~130 vector
        // instructions per channel block).
        unsigned abase[3][4], okm = 0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned mask = cur.vmask;
                const int row = pair0 + fr_e + (i >> 1) + kh * (T2D ? 16 : (p.W >> 1)) + (i & 1) * HALF;
                const int a_ = row * 128 + ((fh ^ ((row >> 1) & 7)) << 4);
                const bool rowok = (mask & 1u) && (kh == 0 ? (mask & 2u) : kh == 2 ? (mask & 4u) : true);
                const bool ok = rowok && (i == 0 ? (mask & 8u) : i == 3 ? (mask & 16u) : true);
                abase[kh][i] = lds0 + (ok ? a_ : OFF_Z + (a_ & 255));
                okm |= (ok ? 1u : 0u) << (kh * 4 + i);
            }
        auto rad_of = [&](int kh, int xslot) {          // -> rad[0..3]: this lane's raw-fragment addresses for kernel row kh in patch slot xslot
#pragma unroll
            for (int i = 0; i < 4; ++i) rad[i] = abase[kh][i] + ((okm >> (kh * 4 + i)) & 1u ? (unsigned)xslot : 0u);
        };
        rad_of(0, xs);
        if constexpr (VAR & 64) { if (ct == t0) stamp(p.stamps, 2); }

        auto block = [&](int cb, auto first_tag) {          // one 64-channel block = 12 sub-steps; first_tag: the tile's first block
            constexpr bool FIRST_CB = decltype(first_tag)::value;
            const bool last_cb = cb + 1 == cpt;
            // what lies beyond this channel block: the next block of this tile, or block 0 of the next tile
            const int nxm0b = last_cb ? nt.m0b : cur.m0b;
            const int nxw = last_cb ? nt.wbase : cur.wbase;
            // per-block operands of the pinned statements (fr_e opaque per block: hoisted out of the loop, both patch slots' variants
            // of the 16 addresses cost 28 registers and spilled)
            asm volatile("" : "+v"(fr_e));
            const unsigned xb = nxm0b == DEAD ? 0x80000000u : (unsigned)(nxm0b + (last_cb ? 0 : (cb + 1) << 7) + xlane);
            const unsigned lx = lds0 + (xs ^ XSLOT) + wave * 1024;          // next patch: row group wave (+ NDW j)
            const unsigned lw = lds0 + OFF_W + wave * 1024;                 // weight ring: piece wave (+ NDW j) of a slot
            const unsigned wsa = (unsigned)(cur.wbase + cb * 12 * WN_WSLOT + wave * 1024);
            const unsigned wsb = nxw == DEAD ? 0u : (unsigned)(nxw + (last_cb ? 0 : (cb + 1) * 12) * WN_WSLOT + wave * 1024);
            const unsigned wva = (unsigned)(lane * 16);
            const unsigned wvb = nxw == DEAD ? CONV_OOB : (unsigned)(lane * 16);

            // One sub-step T = kh * 4 + kk (everything static but the operands above).  U[T & 1], V[T & 1] hold its operands.
            // In front of its barrier: weight stage T + 1 has landed (issued two sub-steps ago; behind it in the in-order counter
            // only the pieces of sub-step T - 1: 2 weight pieces, + 1 patch piece if (T - 1) mod 12 <= 4).
            unsigned clk = 0;                   // (lab: 8 stamps x 2 dwords in its lanes 0..15)
            const bool sample = (VAR & 2) && p.stamps && blockIdx.x < 32 && ct == t0 && cb == (cpt > 1 ? 1 : 0);
            auto step = [&](auto tag) {
                constexpr int T = decltype(tag)::value;
                constexpr bool Z = FIRST_CB && T == 0 && (!T2D || WN_T2D_C0);                  // the first MFMA into every accumulator of the tile
                constexpr int TN = (T + 1) % 12, KHN = TN / 4, KKN = TN % 4;
                constexpr int P = T & 1, Q = P ^ 1;
                constexpr int SLN = (T + 1) & 3;                               // ring slot of stage T + 1
                constexpr int SLD = (T + AHEAD + 1) & 3;                       // ring slot stage T + AHEAD + 1 goes to (three ahead: the one this
                                                                               // sub-step's own operands came from, all in registers by now)
                constexpr int TS = T + AHEAD + 1;                              // ... of this block, or stage TS - 12 of what lies beyond it
                auto w_piece = [&](int j) {
                    if (!dma_wave) return;
                    if constexpr (TS < 12) wn2_dma(wrsrc, lw + SLD * WN_WSLOT + j * NDW * 1024, wva, wsa + TS * WN_WSLOT + j * NDW * 1024);
                    else wn2_dma(wrsrc, lw + SLD * WN_WSLOT + j * NDW * 1024, wvb, wsb + (TS - 12) * WN_WSLOT + j * NDW * 1024);
                };
                auto x_piece = [&](int q) {                                    // the next channel block's patch
                    if constexpr (T < 5) { if (dma_wave) wn2_dma(xrsrc, lx + (XPS * T + q) * NDW * 1024, xb + sxo[XPS * T + q], 0u); }
                };
                if constexpr ((VAR & 2) && T == 6) { if (sample) wn2_clock<0>(clk); }
                // stage T + 1 has landed: issued in sub-step T - 3; younger than its pieces are the patch piece of T - 3 and the
                // pieces of T - 2 and T - 1 (2 weight pieces each, + 1 patch piece in sub-steps 0..4 of a block)
                // (per issuing wave: WPW weight pieces per sub-step, XPS patch pieces in each of sub-steps 0..4 of a block; with two
                // sub-steps of lead the pieces of T - 1 only)
                constexpr int VMC = AHEAD == 3 ? 2 * WPW + XPS * (((T + 9) % 12 < 5) + ((T + 10) % 12 < 5) + ((T + 11) % 12 < 5))
                                               : WPW + XPS * ((T + 11) % 12 < 5);
                wait_vmcnt<VMC>();
                // every fragment requested so far is in (the operands of this sub-step, requested during T - 1): behind the barrier
                // another wave restages the slot they came from
                wn2_wait_lgkm<0>();
                if constexpr ((VAR & 2) && T == 6) { if (sample) wn2_clock<1>(clk); }
                wn2_barrier();
                if constexpr ((VAR & 2) && T == 6) { if (sample) wn2_clock<2>(clk); }
                if constexpr (VAR & 4) __builtin_amdgcn_s_setprio(3);         // (lab) the wave that is behind in its sub-step wins the issue arbitration
                wn2_mfma<Z>(acc[0][0], U[P][0], V[P][0]);
                if constexpr (T == 0) { if (cb == 0) dma_params(ct); }         // (visible from the barrier of sub-step 2 on; the previous
                                                                               //  tile's epilogue, the last reader, lies before this barrier)
                // (T = 11 of a tile's last channel block requests the next tile's first operands like any other sub-step, but nothing is
                // KEPT across the epilogue: fragments held there - 48 registers, or 24 for B^T d and two weight fragments - spilled
                // in it and cost more than requesting them again behind it; the dead requests cost nothing)
                // the next sub-step's kernel row changes behind T = 3, 7, 11 (11: row 0 of the next channel block, in the other patch slot;
                // behind a tile's last block those requests are dead, whatever they read)
                if constexpr (KKN == 0) rad_of(KHN, T == 11 ? xs ^ XSLOT : xs);
                {
                    const unsigned a0 = rad[0] ^ (KKN << 5), a1 = rad[1] ^ (KKN << 5);
                    wn2_read<0>(raw[0], a0);
                    wn2_read<0>(raw[1], a1);
                }
                wn2_mfma<Z>(acc[0][1], U[P][1], V[P][0]);
                {
                    const unsigned a2 = rad[2] ^ (KKN << 5), a3 = rad[3] ^ (KKN << 5);
                    wn2_read<0>(raw[2], a2);
                    wn2_read<0>(raw[3], a3);
                }
                w_piece(0);
                if constexpr (VAR & 4) __builtin_amdgcn_s_setprio(2);
                wn2_mfma<Z>(acc[1][0], U[P][2], V[P][1]);
                wn2_read<SLN * WN_WSLOT + 0 * 4096 + 0>(U[Q][0], ubase);
                wn2_read<SLN * WN_WSLOT + 0 * 4096 + 1024>(U[Q][1], ubase);
                w_piece(1);
                wn2_mfma<Z>(acc[1][1], U[P][3], V[P][1]);
                wn2_read<SLN * WN_WSLOT + 1 * 4096 + 0>(U[Q][2], ubase);
                wn2_read<SLN * WN_WSLOT + 1 * 4096 + 1024>(U[Q][3], ubase);
                if constexpr (HALF_DMA) w_piece(2);
                x_piece(0);
                if constexpr (VAR & 4) __builtin_amdgcn_s_setprio(1);
                wn2_mfma<Z>(acc[2][0], U[P][4], V[P][2]);
                wn2_read<SLN * WN_WSLOT + 2 * 4096 + 0>(U[Q][4], ubase);
                wn2_read<SLN * WN_WSLOT + 2 * 4096 + 1024>(U[Q][5], ubase);
                wn2_wait_lgkm<6>();                                            // the four raw fragments are in
                if constexpr (HALF_DMA) w_piece(3);
                V[Q][0] = wn2_sub(raw[0], raw[2]);                             // B^T d: V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3
                V[Q][1] = wn2_add(raw[1], raw[2]);
                wn2_mfma<Z>(acc[2][1], U[P][5], V[P][2]);
                if constexpr (HALF_DMA) x_piece(1);
                wn2_read<SLN * WN_WSLOT + 3 * 4096 + 0>(U[Q][6], ubase);
                wn2_read<SLN * WN_WSLOT + 3 * 4096 + 1024>(U[Q][7], ubase);
                V[Q][2] = wn2_sub(raw[2], raw[1]);
                if constexpr (VAR & 4) __builtin_amdgcn_s_setprio(0);
                wn2_mfma<Z>(acc[3][0], U[P][6], V[P][3]);
                V[Q][3] = wn2_sub(raw[1], raw[3]);
                wn2_mfma<Z>(acc[3][1], U[P][7], V[P][3]);
                if constexpr ((VAR & 2) && T == 6) { if (sample) wn2_clock<3>(clk); }          // end of sub-step 6
                if constexpr ((VAR & 2) && T == 11) {
                    if (sample && lane < 8) reinterpret_cast<unsigned*>(p.stamps)[((long)blockIdx.x * 8 + wave) * 16 + lane] = clk;
                }
            };
            step(wn2_step_tag<0>{}); step(wn2_step_tag<1>{}); step(wn2_step_tag<2>{}); step(wn2_step_tag<3>{});
            step(wn2_step_tag<4>{}); step(wn2_step_tag<5>{}); step(wn2_step_tag<6>{}); step(wn2_step_tag<7>{});
            step(wn2_step_tag<8>{}); step(wn2_step_tag<9>{}); step(wn2_step_tag<10>{}); step(wn2_step_tag<11>{});
            wn2_wait_lgkm<0>();                // every request of the block has landed: the compiler may move the registers now
            xs ^= XSLOT;
        };
        block(0, wn2_true{});
        for (int cb = 1; cb < cpt; ++cb) block(cb, wn2_false{});

        if constexpr (VAR & 64) { if (ct == t0) stamp(p.stamps, 4); }
        // the epilogue reads the accumulators with vector instructions: the last MFMAs must have left the matrix pipe (the
        // compiler pads what it schedules itself, not what sits in asm statements)
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]),
                     "+v"(acc[3][0]), "+v"(acc[3][1]));
        {
            floatx16 (&acc4)[4][1][2] = *reinterpret_cast<floatx16 (*)[4][1][2]>(&acc);
#define WN_EPI(FULL_, ACT_, RES_) \
    wino_tile_epilogue<1, FULL_, ACT_, RES_, T2D>(p, acc4, lds_bias, lds_slope, m0, c0, pair0, crow0, fr_e, fh_e, HoWo, inv_howo, inv_wo, m_over)
            int m_over = 0;
            if constexpr (T2D) {               // every 2-D tile has pairs that do not exist (the 16th of a tile row): masked stores throughout
                const int pl = pair0 + fr_e;
                m_over = (cur.vmask & 1u) ? mt_cur + (pl >> 4) * p.W + 2 * (pl & 15) : -1;
            }
            const bool full = !T2D && m0 + TP <= p.M && c0 + TC <= p.Cout;
            if ((T2D || m0 + TP > p.M) && c0 + TC <= p.Cout) {      // masked stores, still specialised (the run-time form below keeps both the
                                                                     // slope and the residual path alive: it is the one that spills)
                if (p.act == FRP_ACT_PRELU) { if (has_res) WN_EPI(false, FRP_ACT_PRELU, 1); else WN_EPI(false, FRP_ACT_PRELU, 0); }
                else if (p.act == FRP_ACT_RELU) { if (has_res) WN_EPI(false, FRP_ACT_RELU, 1); else WN_EPI(false, FRP_ACT_RELU, 0); }
                else { if (has_res) WN_EPI(false, FRP_ACT_NONE, 1); else WN_EPI(false, FRP_ACT_NONE, 0); }
            } else
            if (!full) WN_EPI(false, -1, -1);
            else if (p.act == FRP_ACT_PRELU) { if (has_res) WN_EPI(true, FRP_ACT_PRELU, 1); else WN_EPI(true, FRP_ACT_PRELU, 0); }
            else if (p.act == FRP_ACT_RELU) { if (has_res) WN_EPI(true, FRP_ACT_RELU, 1); else WN_EPI(true, FRP_ACT_RELU, 0); }
            else { if (has_res) WN_EPI(true, FRP_ACT_NONE, 1); else WN_EPI(true, FRP_ACT_NONE, 0); }
#undef WN_EPI
        }
        cur = nt;
        if constexpr (T2D) mt_cur = mt_nxt;
        if (ct + tstep < t1) first_operands(cur, xs);       // (its patch and stage 0 landed before this tile's last barrier)
        if constexpr (VAR & 64) { if (ct == t0) stamp(p.stamps, 5); }
    }
    // nothing of this workgroup's DMA stream may still be in flight when its LDS is handed to the next workgroup
    wait_vmcnt<0>();
    if constexpr (VAR & 2) {
        if (p.stamps && blockIdx.x < 32) {
            wn2_clock<6>(clkk);
            wn2_clock_real<7>(clkk);
            if (lane >= 8 && lane < 16) reinterpret_cast<unsigned*>(p.stamps)[((long)blockIdx.x * 8 + wave) * 16 + lane] = clkk;
        }
    } else if constexpr (VAR & 64) stamp(p.stamps, 6);
}

// 2-D tiles (8 rows x 30 columns of one image; conv3x3_wino2_kernel<32>) for maps wider than the flattened tiles cover: where they pay.
// A tile carries 240 real pixels in 256 pixel slots and the edge tiles of a map carry less; against that stand the kernel's x1.12
// (128 channels: 24 sub-steps per tile, a third of a tile is prologue + epilogue) and x1.2 (256 channels and more) over the direct
// kernel.  Measured (tools/wino_probe.py, 32 frames): 136 x 240 x 128: x1.08-1.11, 68 x 120 x 256: x1.09, 68 x 120 x 128: x1.00,
// 34 x 60 x 256: x0.98.  Fewer than two rounds of tiles: the direct family (its quarter tiles fill the chip better).
bool conv3x3_wino_wide_pays(int N, int H, int W, int Cin, int Cout, int n_cu, bool has_res) {
    if ((W & 1) || W <= 30 || Cin < 128 || (Cin & 63)) return false;
    // (round 4 kept the 136 x 240 x 128 RESIDUAL layers direct: 302 / 318 us against 310 / 305 in the pipeline.  With the residual of
    // all four epilogue units requested up front - round 5 - they take the 2-D tiles too: detector 5.065 -> 5.038 ms per 32 frames,
    // gpurun_out -> profiles/r5/ab_residual_epilogue.txt.)
    (void)has_res;
    const long ty = (H + 7) / 8, tx = (W + 29) / 30;
    const long tiles = (long)N * ty * tx * ((Cout + WN_TC - 1) / WN_TC);
    if (tiles < 2L * (n_cu > 0 ? n_cu : 256)) return false;
    const double slots = (double)(ty * 8) * (double)(tx * 32), real = (double)H * W;
    return (Cin >= 256 ? 1.2 : 1.12) * real / slots >= 1.03;
}
static bool wino_2d_pays(const ConvParams& p) {
    return !p.n_dev && (long)p.N * p.H * p.W == (long)p.M && conv3x3_wino_wide_pays(p.N, p.H, p.W, p.Cin, p.Cout, p.n_cu, p.res != nullptr);
}

// which form of the kernel a launch takes: 0 none, 1 flattened tiles (maps up to 30 wide), 2 the 2-D tiles
static int wino_form(const ConvParams& p) {
    if (p.KS != 3 || p.stride != 1 || (p.Cin & 63) || p.ksplit != 1 || (p.W & 1) || p.W < 2) return 0;
    if (p.Ho != p.H || p.Wo != p.W) return 0;
    if (p.flags & (FRP_FLAG_OUT_F32 | FRP_FLAG_F8 | FRP_FLAG_OUT_FP8 | FRP_FLAG_RES_UP2) || p.out2) return 0;
#ifdef FRP_LAB    // dbg bit 256 (conv2d / conv_bench flags bit 19): the 2-D tiles whatever the shape; dbg bit 64: the first generation's
    if ((p.dbg & 256) && !p.n_dev && conv3x3_wino_lab_shape_ok(p.W, p.Cin, p.KS, p.stride)) return 2;      // row-patch form for wide maps
    if ((p.dbg & 64) && conv3x3_wino_lab_shape_ok(p.W, p.Cin, p.KS, p.stride)) return 1;                   // (slower than the direct kernel)
#endif
    if (!p.wino_wide_only && conv3x3_wino_shape_ok(p.W, p.Cin, p.KS, p.stride)) return 1;
    return wino_2d_pays(p) ? 2 : 0;
}

// Shapes the Winograd kernel covers; `p` carries the derived fields of launch_conv().
bool conv3x3_wino_eligible(const ConvParams& p) {
    if (!wino_form(p)) return false;
    const long reach = ((long)p.M + 2L * p.W + 600) * p.Cin * 2;       // signed 32-bit patch offsets
    return reach < 0x7fffffffL;
}

// The same answer in the shipped and in the lab build: frp_load_weights builds Winograd images - and run_net routes layers to
// this kernel - for exactly the same layers in both, so engine-level runs on libfrp_lab.so time and compute what libfrp.so does.
bool conv3x3_wino_shape_ok(int W, int Cin, int ksize, int stride) {
    return ksize == 3 && stride == 1 && !(Cin & 63) && !(W & 1) && W >= 2 && wino_super_patch(W);
}
#ifdef FRP_LAB
bool conv3x3_wino_lab_shape_ok(int W, int Cin, int ksize, int stride) {      // + wide maps (row-patch form, dbg bit 64)
    return ksize == 3 && stride == 1 && !(Cin & 63) && !(W & 1) && W >= 2;
}
#endif

// bytes of the transformed weight image of a layer (the `w` operand of the Winograd kernel)
size_t conv3x3_wino_image_bytes(int Cin, int Cout) {
    return (size_t)((Cout + WN_TC - 1) / WN_TC) * (Cin / 64) * 12 * WN_WSLOT;
}

template <int NW, int ABL = 0, bool ROWP = false>
static hipError_t launch_wino_cfg(const ConvParams& p0, hipStream_t stream) {
    ConvParams p = p0;
    p.n_ptiles = (p.M + WN_TP - 1) / WN_TP;
    p.n_ctiles = (p.Cout + WN_TC - 1) / WN_TC;
    const size_t img = conv3x3_wino_image_bytes(p.Cin, p.Cout);
    if (img >= 0x7fffffffUL) return hipErrorInvalidValue;
    p.w_bytes = (unsigned)img;
    const int lds = ROWP ? wino_rowp_lds_bytes() : wino_lds_bytes(p.W);
    static int attr_lds[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (attr_lds[dev] < lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>((conv3x3_wino_kernel<NW, ABL, ROWP>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_lds[dev] = 160 * 1024;
    }
    const long ntiles = (long)p.n_ptiles * p.n_ctiles;
    if (ntiles <= 0 || ntiles > 0x7fffffffL) return hipErrorInvalidValue;
    const int ncu = device_cu_count(dev);
    if (ncu <= 0) return hipErrorInvalidDevice;
    const unsigned grid = (unsigned)(ntiles < ncu ? ntiles : ncu);     // persistent: one workgroup per CU
    hipLaunchKernelGGL((conv3x3_wino_kernel<NW, ABL, ROWP>), dim3(grid), dim3(NW * 64), lds, stream, p);
    return hipGetLastError();
}

template <int VAR>
static hipError_t launch_wino2_cfg(const ConvParams& p0, hipStream_t stream) {
    ConvParams p = p0;
    constexpr bool T2D = (VAR & 32) != 0;
    p.n_ptiles = T2D ? p.N * ((p.H + 7) / 8) * ((p.W + 29) / 30) : (p.M + WN_TP - 1) / WN_TP;
    p.n_ctiles = (p.Cout + WN_TC - 1) / WN_TC;
    const size_t img = conv3x3_wino_image_bytes(p.Cin, p.Cout);
    if (img >= 0x7fffffffUL) return hipErrorInvalidValue;
    p.w_bytes = (unsigned)img;
    if (T2D && (p.n_dev || (long)p.N * p.H * p.W != (long)p.M)) return hipErrorInvalidValue;
    if (!T2D && wino_half_rows(p.W) != WN2_PPW * 32) return hipErrorInvalidValue;          // 5 patch pieces per wave
    const int lds = wino_lds_bytes(T2D ? 14 : p.W);
    static int attr_lds[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (attr_lds[dev] < lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>((conv3x3_wino2_kernel<VAR>)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_lds[dev] = 160 * 1024;
    }
    const long ntiles = (long)p.n_ptiles * p.n_ctiles;
    if (ntiles <= 0 || ntiles > 0x7fffffffL) return hipErrorInvalidValue;
    const int ncu = device_cu_count(dev);
    if (ncu <= 0) return hipErrorInvalidDevice;
    const unsigned grid = (unsigned)(ntiles < ncu ? ntiles : ncu);     // persistent: one workgroup per CU
    hipLaunchKernelGGL((conv3x3_wino2_kernel<VAR>), dim3(grid), dim3(512), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv3x3_wino(const ConvParams& p, hipStream_t stream) {
    const int form = wino_form(p);
    if (!form || !conv3x3_wino_eligible(p)) return hipErrorInvalidValue;
    if (form == 2) return launch_wino2_cfg<32>(p, stream);            // 2-D tiles
#ifdef FRP_LAB   // dbg bit 128: the first generation of the k-loop (compiler-scheduled; A/B partner of the hand-ordered one)
    if (!(p.dbg & 128) && wino_super_patch(p.W) && !(p.dbg & (32 | 64)) && ((p.dbg >> 1) & 15) == 13) return launch_wino2_cfg<2>(p, stream);   // sub-step stamps
    if (!(p.dbg & 128) && wino_super_patch(p.W) && !(p.dbg & (32 | 64)) && ((p.dbg >> 1) & 15) == 8) return launch_wino2_cfg<64>(p, stream);   // 100 MHz phase stamps
    if (!(p.dbg & 128) && wino_super_patch(p.W) && !(p.dbg & (32 | 64)) && ((p.dbg >> 1) & 15) == 14) return launch_wino2_cfg<1>(p, stream);   // waves 4-7 at priority 1
    if (!(p.dbg & 128) && wino_super_patch(p.W) && !(p.dbg & (32 | 64)) && ((p.dbg >> 1) & 15) == 12) return launch_wino2_cfg<4>(p, stream);   // priority falls with progress
    if (!(p.dbg & 128) && wino_super_patch(p.W) && !(p.dbg & (32 | 64)) && ((p.dbg >> 1) & 15) == 11) return launch_wino2_cfg<8>(p, stream);   // DMA by waves 0-3 only
    if (!(p.dbg & 128) && wino_super_patch(p.W) && !(p.dbg & (32 | 64)) && ((p.dbg >> 1) & 15) == 10) return launch_wino2_cfg<16>(p, stream);  // ring two ahead
    if (!(p.dbg & 128) && wino_super_patch(p.W) && !(p.dbg & (32 | 64)) && ((p.dbg >> 1) & 15) == 9) return launch_wino2_cfg<9>(p, stream);    // DMA by waves 0-3, waves 4-7 at priority 1
    if (!(p.dbg & 128) && wino_super_patch(p.W) && !(p.dbg & (32 | 64)) && !((p.dbg >> 1) & 15)) return launch_wino2_cfg<0>(p, stream);
#else
    return wino_super_patch(p.W) ? launch_wino2_cfg<0>(p, stream) : hipErrorInvalidValue;
#endif
#ifdef FRP_LAB   // dbg bit 32: the one-wave-per-SIMD configuration (A/B runs in the lab build; 0.7 x the speed of the default)
    if ((p.dbg & 32) && wino_super_patch(p.W)) return launch_wino_cfg<4>(p, stream);
    if (wino_super_patch(p.W)) switch ((p.dbg >> 1) & 15) {               // dbg bits 1..4: timing ablations (tools/wino_ablate.py)
        case 1: return launch_wino_cfg<8, 1>(p, stream);
        case 2: return launch_wino_cfg<8, 2>(p, stream);
        case 4: return launch_wino_cfg<8, 4>(p, stream);
        case 8: return launch_wino_cfg<8, 8>(p, stream);
        case 3: return launch_wino_cfg<8, 3>(p, stream);
        case 6: return launch_wino_cfg<8, 6>(p, stream);
        case 5: return launch_wino_cfg<8, 5>(p, stream);
        case 7: return launch_wino_cfg<8, 7>(p, stream);
        case 12: return launch_wino_cfg<8, 12>(p, stream);
        case 9: return launch_wino_cfg<8, 16>(p, stream);      // (code 9: no B^T d arithmetic)
        case 15: return launch_wino_cfg<8, 31>(p, stream);     // everything off: the loop's own bookkeeping
        default: break;
    }
#endif
#ifdef FRP_LAB
    if (p.dbg & 64) return launch_wino_cfg<8, 0, true>(p, stream);     // row-patch form (first generation; wide maps, or forced for an A/B)
#endif
    if (!wino_super_patch(p.W)) return hipErrorInvalidValue;
    return launch_wino_cfg<8>(p, stream);
}

}  // namespace frp
