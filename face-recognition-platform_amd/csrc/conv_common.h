// Device helpers shared by the MFMA convolution kernels (conv_mfma.hip, conv3x3_rows.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "frp_internal.h"

namespace frp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define CONV_NS 3            // LDS ring stages
#define CONV_OOB 0x80000000u // buffer offset beyond any tensor (< 2 GiB each): reads as zero

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// one LDS-DMA piece: 64 lanes x 16 B from per-lane buffer offsets to lds_base + lane*16.
// (kept in a __device__ function: the builtin does not exist for the host pass)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_base, unsigned voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, voffset, 0, 0, 0);
}

// (device-only constructs must live in __device__ functions: written directly in the __global__
// template body they make the HOST pass drop the kernel stub without a diagnostic)
__device__ __forceinline__ void keep_alive(floatx16 v) { asm volatile("" ::"v"(v)); }
// Half-wave exchange: lane<32 ends up with this pixel's couts [lo | upper lane's lo] (16 contiguous
// bytes), lane>=32 with [lower lane's hi | hi]: two 8-byte stores per lane become one 16-byte store
// (the epilogue store tail is issue-bound, not bandwidth-bound).
__device__ __forceinline__ void swap_halves(unsigned& a, unsigned& b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
// diagnostic stamps (conv_bench only, p.stamps != null): 100 MHz wall clock per workgroup phase,
// written to a buffer nothing else reads.  Lab build only: the shipped kernels carry neither the tests of the stamp pointer nor
// the ~12 scalar registers they keep spilled (-0.3 % of a step; the lab library's kernels are otherwise the shipped ones).
__device__ __forceinline__ void stamp(unsigned long long* buf, int slot) {
#ifndef FRP_LAB
    (void)buf; (void)slot;
    return;
#endif
    if (buf && threadIdx.x == 0) buf[(long)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memrealtime();
}
// keeps hipcc from hoisting the loads of every epilogue slice above the first one (which
// would need several hundred live registers)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

static int device_cu_count(int dev) {
    static int n_cu[64] = {};
    if (!n_cu[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        n_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n_cu[dev];
}

// Small-M rule shared by the conv launchers: quarter tiles (128 pixels x 64 couts, two workgroups per CU) when the default
// tiling would leave at least half of the CUs without a tile - small pyramid scales, a handful of faces, single images.
// With the image count on the device (n_dev) the decision - like the grid - goes by the capacity.  Same k order: bit-identical
// to the default tiles of the same kernel family.
static inline bool conv_small_m(const ConvParams& p, long default_tiles, int ncu) {
    if (p.small_m < 0 || p.ksplit != 1 || p.out2 || (p.flags & (FRP_FLAG_F8 | FRP_FLAG_OUT_FP8 | FRP_FLAG_OUT_F32))) return false;
    if (p.small_m > 0) return true;
    return ncu > 0 && default_tiles * 2 <= ncu;
}

// Weight prefetch of a quarter-tile launch (ConvParams::pf_ptr): workgroup k of the n_pf extra ones (k = blockIdx - workers; round-robin
// dispatch puts workgroup b on XCD b % 8) pulls its share of the next launch's weights through the L2 of ITS XCD - every XCD reads the
// whole region, in n_pf / 8 slices.  Read-only, results unused: nothing to synchronise with.
#define CONV_PF_WGS 64
static inline bool conv_prefetch_enabled() {                         // FRP_NO_PREFETCH=1 (read once): A/B runs
    static const bool on = getenv("FRP_NO_PREFETCH") == nullptr;
    return on;
}
__device__ __forceinline__ void conv_prefetch_weights(const ConvParams& p, int k, int n_pf, int nthreads) {
    const int per_xcd = n_pf >> 3;
    if (per_xcd <= 0 || !p.pf_ptr) return;
    const int q = k >> 3;                                            // slice index within this XCD's workgroups
    // at most 3 MiB (an XCD's L2 holds 4): the head of a larger tensor (stage 4's 4.7 MB; the FC's 25.7 MB would only evict itself)
    const unsigned n16 = (p.pf_bytes < (3u << 20) ? p.pf_bytes : (3u << 20)) >> 4;
    const unsigned lo = (unsigned)((unsigned long long)n16 * q / per_xcd), hi = (unsigned)((unsigned long long)n16 * (q + 1) / per_xcd);
    const uint4* src = reinterpret_cast<const uint4*>(p.pf_ptr);
    unsigned acc = 0;
    for (unsigned i = lo + threadIdx.x; i < hi; i += nthreads) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u && p.stamps) p.stamps[0] = acc;          // (keeps the loads; never true in practice, harmless if it is)
}

// q = m / d, r = m % d for 0 <= m < 2^24 via a float reciprocal and one correction step
// (an integer division costs ~40 instructions; the prologue needs two per pixel row)
__device__ __forceinline__ void fast_divmod(int m, int d, float inv_d, int& q, int& r) {
    q = (int)((float)m * inv_d);
    r = m - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
}

// ---------------------------------------------------------------------------------------------------------------
// Tile epilogue shared by the conv kernels (accumulator layout of v_mfma_f32_32x32x16: lane = one pixel (fr) of a
// 32-pixel row block, registers 4g..4g+3 = couts 8g + 4*fh .. +3 of a 32-cout block): bias / 9-class border bias
// from the LDS parameter cache, residual, activation in fp32, then fp16 (16-byte stores after a half-wave
// exchange) or fp32 output.
//
// FULL tiles (the common case) run a copy specialised at compile time on (activation, residual): with the flags
// as run-time values the compiler kept ~70 uniform branches and ~90 s_nops in every epilogue (it does not unswitch
// a body this large), a quarter of its instructions.  Ragged tiles and fp32 output take the generic copy.
//   ACT: FRP_ACT_* or -1 = run-time p.act;  RES: 0 / 1 or -1 = run-time.
template <int MP, int MC, int TC, bool FULL, int ACT, int RES, bool OVER = false>
__device__ __forceinline__ void conv_epilogue_body(const ConvParams& p, floatx16 (&acc)[MP][MC], const uint4 (&rres)[MP][MC][2],
                                                   const float* lds_bias, const float* lds_slope, int m0, int c0, int prow0,
                                                   int crow0, int fr, int fh, int HoWo, float inv_howo, float inv_wo,
                                                   int ps = 1, int po = 0, int m_over = 0) {
    // (ps, po): lane row r of the tile holds output pixel m0 + r * ps + po: (1, 0) in the direct kernels, (2, parity) in
    // the Winograd kernel, whose lanes own pixel PAIRS (conv3x3_wino.hip).  OVER (MP = 1; the Winograd kernel's 2-D tiles):
    // the lane's pixel is m_over + po instead, and does not exist when m_over < 0.
    const bool border = p.flags & FRP_FLAG_BORDER_BIAS;
    const bool out32 = (ACT < 0) && (p.flags & FRP_FLAG_OUT_F32);
    const bool has_res = RES < 0 ? p.res != nullptr : RES != 0;
    const int act = ACT < 0 ? p.act : ACT;
    half4 r4[MP][MC][4];
    bool mok[MP];
    long obase[MP];
    int cls[MP];
#pragma unroll
    for (int i = 0; i < MP; ++i) {
        const int mraw = OVER ? (m_over >= 0 ? m_over + po : p.M) : m0 + (prow0 + i * 32 + fr) * ps + po;
        mok[i] = FULL || mraw < p.M;
        const int m = mok[i] ? mraw : 0;
        cls[i] = 0;
        if (border) {
            int n, rem, oy, ox;
            fast_divmod(m, HoWo, inv_howo, n, rem);
            fast_divmod(rem, p.Wo, inv_wo, oy, ox);
            cls[i] = ((oy == 0) ? 0 : (oy == p.Ho - 1) ? 2 : 1) * 3 + ((ox == 0) ? 0 : (ox == p.Wo - 1) ? 2 : 1);
        }
        obase[i] = (long)m * p.Cout;
        if (has_res) {
            // the residual arrived as 16-byte chunks in the STORE layout; the same half-wave exchange as for the
            // stores (it is its own inverse) turns them into the accumulator layout
#pragma unroll
            for (int j = 0; j < MC; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    uint4 rr = rres[i][j][q];
                    swap_halves(rr.x, rr.z);
                    swap_halves(rr.y, rr.w);
                    union { unsigned u[2]; half4 h; } lo, hi;
                    lo.u[0] = rr.x; lo.u[1] = rr.y; hi.u[0] = rr.z; hi.u[1] = rr.w;
                    r4[i][j][2 * q] = lo.h;
                    r4[i][j][2 * q + 1] = hi.h;
                }
        }
    }
#pragma unroll
    for (int i = 0; i < MP; ++i) {
#pragma unroll
        for (int j = 0; j < MC; ++j) {
            floatx4 v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = crow0 + j * 32 + 8 * g + 4 * fh;
                const floatx4 b4 = *reinterpret_cast<const floatx4*>(lds_bias + cls[i] * TC + cl);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[g][e] = acc[i][j][4 * g + e] + b4[e];
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] += (float)r4[i][j][g][e];
                }
                if (act == FRP_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] = fmaxf(v[g][e], 0.f);
                } else if (act == FRP_ACT_PRELU) {
                    const floatx4 s4 = *reinterpret_cast<const floatx4*>(lds_slope + cl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] = v[g][e] > 0.f ? v[g][e] : v[g][e] * s4[e];
                }
            }
            if (out32) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = c0 + crow0 + j * 32 + 8 * g + 4 * fh;
                    if (FULL || (mok[i] && co < p.Cout))
                        *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(p.out) + obase[i] + co) = v[g];
                }
            } else {
                union { half4 h; unsigned u[2]; } pk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[g].h[e] = (_Float16)v[g][e];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    swap_halves(pk[2 * q].u[0], pk[2 * q + 1].u[0]);
                    swap_halves(pk[2 * q].u[1], pk[2 * q + 1].u[1]);
                    const int co = c0 + crow0 + j * 32 + 16 * q + 8 * fh;   // 8 consecutive couts
                    if (FULL || (mok[i] && co < p.Cout))
                        *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(p.out) + obase[i] + co) =
                            make_uint4(pk[2 * q].u[0], pk[2 * q].u[1], pk[2 * q + 1].u[0], pk[2 * q + 1].u[1]);
                }
            }
        }
    }
}

template <int MP, int MC, int TC>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, floatx16 (&acc)[MP][MC], const uint4 (&rres)[MP][MC][2],
                                              const float* lds_bias, const float* lds_slope, int m0, int c0, int TP, int prow0,
                                              int crow0, int fr, int fh, int HoWo, float inv_howo, float inv_wo,
                                              int ps = 1, int po = 0) {
#define FRP_EPI(FULL_, ACT_, RES_) \
    conv_epilogue_body<MP, MC, TC, FULL_, ACT_, RES_>(p, acc, rres, lds_bias, lds_slope, m0, c0, prow0, crow0, fr, fh, HoWo, inv_howo, inv_wo, ps, po)
    const bool full = m0 + TP <= p.M && c0 + TC <= p.Cout;
    if (!full) { FRP_EPI(false, -1, -1); return; }
    if (p.flags & FRP_FLAG_OUT_F32) { FRP_EPI(true, -1, -1); return; }
    const bool has_res = p.res != nullptr;
    if (p.act == FRP_ACT_PRELU) { if (has_res) FRP_EPI(true, FRP_ACT_PRELU, 1); else FRP_EPI(true, FRP_ACT_PRELU, 0); }
    else if (p.act == FRP_ACT_RELU) { if (has_res) FRP_EPI(true, FRP_ACT_RELU, 1); else FRP_EPI(true, FRP_ACT_RELU, 0); }
    else { if (has_res) FRP_EPI(true, FRP_ACT_NONE, 1); else FRP_EPI(true, FRP_ACT_NONE, 0); }
#undef FRP_EPI
}

// ---------------------------------------------------------------------------------------------------------------
// Epilogue with fp8 outputs (BASELINE config 5).  WS: the accumulator is first multiplied by lds_wscale[cout] (fp8
// weights x fp8 input: per-cout weight scale x input-tensor scale).  Outputs: fp16 to p.out unless FRP_FLAG_OUT_FP8;
// OCP E4M3 bytes (value / out_scale, clamped to +-448) to p.out when FRP_FLAG_OUT_FP8, else to p.out2 when set.
// fp8 store layout: a lane packs its 4-cout runs into dwords, exchanges half-waves like the fp16 path and stores
// 8 consecutive couts (8 bytes) per (32-cout block, half).
__device__ __forceinline__ int pack_fp8x4(floatx4 v, float inv_scale) {
    float a[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = fminf(fmaxf(v[e] * inv_scale, -448.f), 448.f);
    int d = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], 0, false);
    return __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], d, true);
}

template <int MP, int MC, int TC, bool FULL, int ACT, int RES, bool WS>
__device__ __forceinline__ void conv_epilogue8_body(const ConvParams& p, floatx16 (&acc)[MP][MC], const uint4 (&rres)[MP][MC][2],
                                                    const float* lds_bias, const float* lds_slope, const float* lds_wscale, int m0,
                                                    int c0, int prow0, int crow0, int fr, int fh, int HoWo, float inv_howo, float inv_wo) {
    const bool border = p.flags & FRP_FLAG_BORDER_BIAS;
    const bool out8 = p.flags & FRP_FLAG_OUT_FP8;
    unsigned char* dst8 = reinterpret_cast<unsigned char*>(out8 ? p.out : p.out2);
    const float inv_os = 1.0f / p.out_scale;
    const bool has_res = RES < 0 ? p.res != nullptr : RES != 0;
    const int act = ACT < 0 ? p.act : ACT;
#pragma unroll
    for (int i = 0; i < MP; ++i) {
        const int mraw = m0 + prow0 + i * 32 + fr;
        const bool mok = FULL || mraw < p.M;
        const int m = mok ? mraw : 0;
        int cls = 0;
        if (border) {
            int n, rem, oy, ox;
            fast_divmod(m, HoWo, inv_howo, n, rem);
            fast_divmod(rem, p.Wo, inv_wo, oy, ox);
            cls = ((oy == 0) ? 0 : (oy == p.Ho - 1) ? 2 : 1) * 3 + ((ox == 0) ? 0 : (ox == p.Wo - 1) ? 2 : 1);
        }
        const long obase = (long)m * p.Cout;
#pragma unroll
        for (int j = 0; j < MC; ++j) {
            half4 r4[4];
            if (has_res) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    uint4 rr = rres[i][j][q];
                    swap_halves(rr.x, rr.z);
                    swap_halves(rr.y, rr.w);
                    union { unsigned u[2]; half4 h; } lo, hi;
                    lo.u[0] = rr.x; lo.u[1] = rr.y; hi.u[0] = rr.z; hi.u[1] = rr.w;
                    r4[2 * q] = lo.h;
                    r4[2 * q + 1] = hi.h;
                }
            }
            floatx4 v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = crow0 + j * 32 + 8 * g + 4 * fh;
                const floatx4 b4 = *reinterpret_cast<const floatx4*>(lds_bias + cls * TC + cl);
                if constexpr (WS) {
                    const floatx4 w4 = *reinterpret_cast<const floatx4*>(lds_wscale + cl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] = acc[i][j][4 * g + e] * w4[e] + b4[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] = acc[i][j][4 * g + e] + b4[e];
                }
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] += (float)r4[g][e];
                }
                if (act == FRP_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] = fmaxf(v[g][e], 0.f);
                } else if (act == FRP_ACT_PRELU) {
                    const floatx4 s4 = *reinterpret_cast<const floatx4*>(lds_slope + cl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] = v[g][e] > 0.f ? v[g][e] : v[g][e] * s4[e];
                }
            }
            if (!out8) {                      // fp16 primary output
                union { half4 h; unsigned u[2]; } pk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[g].h[e] = (_Float16)v[g][e];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    swap_halves(pk[2 * q].u[0], pk[2 * q + 1].u[0]);
                    swap_halves(pk[2 * q].u[1], pk[2 * q + 1].u[1]);
                    const int co = c0 + crow0 + j * 32 + 16 * q + 8 * fh;
                    if (FULL || (mok && co < p.Cout))
                        *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(p.out) + obase + co) =
                            make_uint4(pk[2 * q].u[0], pk[2 * q].u[1], pk[2 * q + 1].u[0], pk[2 * q + 1].u[1]);
                }
            }
            if (dst8) {                       // fp8 output (primary or copy)
                unsigned d[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) d[g] = (unsigned)pack_fp8x4(v[g], inv_os);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    swap_halves(d[2 * q], d[2 * q + 1]);        // -> couts 16q + 8fh .. +7 of this pixel
                    const int co = c0 + crow0 + j * 32 + 16 * q + 8 * fh;
                    if (FULL || (mok && co < p.Cout))
                        *reinterpret_cast<uint2*>(dst8 + obase + co) = make_uint2(d[2 * q], d[2 * q + 1]);
                }
            }
        }
    }
}

template <int MP, int MC, int TC, bool WS>
__device__ __forceinline__ void conv_epilogue8(const ConvParams& p, floatx16 (&acc)[MP][MC], const uint4 (&rres)[MP][MC][2],
                                               const float* lds_bias, const float* lds_slope, const float* lds_wscale, int m0, int c0,
                                               int TP, int prow0, int crow0, int fr, int fh, int HoWo, float inv_howo, float inv_wo) {
#define FRP_EPI8(FULL_, ACT_, RES_) \
    conv_epilogue8_body<MP, MC, TC, FULL_, ACT_, RES_, WS>(p, acc, rres, lds_bias, lds_slope, lds_wscale, m0, c0, prow0, crow0, fr, fh, HoWo, inv_howo, inv_wo)
    const bool full = m0 + TP <= p.M && c0 + TC <= p.Cout;
    if (!full || !WS) { FRP_EPI8(false, -1, -1); return; }        // ragged tiles, and fp16 kernels that add an fp8 copy
    const bool has_res = p.res != nullptr;
    if (p.act == FRP_ACT_PRELU) { if (has_res) FRP_EPI8(true, FRP_ACT_PRELU, 1); else FRP_EPI8(true, FRP_ACT_PRELU, 0); }
    else { if (has_res) FRP_EPI8(true, FRP_ACT_NONE, 1); else FRP_EPI8(true, -1, 0); }
#undef FRP_EPI8
}

// residual of a tile in the STORE layout (16-byte chunks: couts 16q + 8*fh .. +7 of this lane's pixel), from clamped -
// always valid - addresses; consumed by conv_epilogue
template <int MP, int MC>
__device__ __forceinline__ void conv_residual_loads(const ConvParams& p, uint4 (&rres)[MP][MC][2], int m0, int c0, int prow0, int crow0,
                                                    int fr, int fh, int HoWo, float inv_howo, float inv_wo, int ps = 1, int po = 0) {
    const bool up2 = p.flags & FRP_FLAG_RES_UP2;
#pragma unroll
    for (int i = 0; i < MP; ++i) {
        const int mraw = m0 + (prow0 + i * 32 + fr) * ps + po;
        const int m = mraw < p.M ? mraw : 0;
        long ridx = (long)m * p.Cout;
        if (up2) {
            int n, rem, oy, ox;
            fast_divmod(m, HoWo, inv_howo, n, rem);
            fast_divmod(rem, p.Wo, inv_wo, oy, ox);
            ridx = (((long)n * p.Hr + (oy >> 1)) * p.Wr + (ox >> 1)) * p.Cout;
        }
#pragma unroll
        for (int j = 0; j < MC; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int co = c0 + crow0 + j * 32 + 16 * q + 8 * fh;
                rres[i][j][q] = *reinterpret_cast<const uint4*>(p.res + ridx + (co < p.Cout ? co : 0));
            }
    }
}

__device__ __forceinline__ void wait_lgkmcnt0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// In front of a k-step's barrier: every fragment read of the step before must have RETURNED before another wave, released by
// this barrier, fires the LDS-DMA that restages the ring slot it read (write after read).  The source order alone does not
// give that: the compiler sinks the last MFMA group of a step - and with it the lgkmcnt wait that retires its reads - below
// the barrier, and under LDS load (two workgroups per CU, 2 reads per MFMA: the quarter-tile configurations) a read issued
// before the barrier lost the race against a DMA from L2 issued after it: one wave with one stale 8-row piece, one launch in
// four (tools/quarter_check.py).  FRP_WAR_RELAXED builds leave the wait out (A/B of its cost).
__device__ __forceinline__ void retire_lds_reads() {
#ifndef FRP_WAR_RELAXED
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace frp
