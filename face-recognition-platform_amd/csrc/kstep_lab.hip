// Tuning lab (not on the product path): schedules of the conv k-step's inner loop in isolation - 8 waves per CU,
// 64x64 wave tiles, fragments from a random-filled LDS image by ds_read_b128, 16 MFMAs per wave and k-step - to find
// out what a workgroup barrier per k-step costs and which instruction order wins it back, before the real kernels
// (conv3x3_rows.hip, conv_mfma.hip) are restructured.  frp_kstep_lab(variant) reports TFLOP/s; tools/kstep_lab.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frp_internal.h"
#include "conv_common.h"

namespace frp {

// V bit 0: s_barrier per k-step          bit 1: reads of group g+1 pinned BEFORE the MFMAs of group g (sched_group_barrier)
//   bit 2: kk = 0 fragments of the next k-step requested before the barrier (cross-barrier prefetch)
//   bit 3: waves 4..7 run half a k-step behind the barrier (stagger)      bit 4: s_setprio 1 for waves 4..7
//   bit 5: three fragment sets (reads two groups ahead)
//   bits 6..8: P = LDS-DMA pieces (1 KiB each) per wave and k-step into the ring slot two stages ahead, counted vmcnt
//              wait at the top of every step (P = 4: the row-patch kernel at 128 couts; 6: the generic kernel)
template <int V>
__global__ __launch_bounds__(512, 2) void kstep_lab_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * 384 * 128];
    for (int i = threadIdx.x; i < 3 * 384 * 128 / 16; i += 512)
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(src)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int prow0 = (wave >> 1) * 64, crow0 = 256 + (wave & 1) * 64;
    constexpr bool BAR = V & 1, PIN = V & 2, PF = V & 4, STAG = V & 8, PRIO = V & 16, TRI = V & 32;
    constexpr int NSET = TRI ? 3 : 2;
    floatx16 acc[2][2] = {};
    half8 f[NSET][4];
    int addr[4][4];                       // [fragment][kk] byte offset inside a stage
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int row = (q < 2 ? prow0 + q * 32 : crow0 + (q - 2) * 32) + fr;
            addr[q][kk] = row * 128 + (((2 * kk + fh) ^ ((row >> 1) & 7)) << 4);
        }
    auto rd = [&](int soff, int kk, int S) {
#pragma unroll
        for (int q = 0; q < 4; ++q) f[S][q] = *reinterpret_cast<const half8*>(lds + soff + addr[q][kk]);
    };
    auto mm = [&](int S) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[S][2], f[S][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[S][3], f[S][0], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[S][2], f[S][1], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[S][3], f[S][1], acc[1][1], 0, 0, 0);
    };
    // "4 DS reads, then 4 MFMAs": the order the source asks for, pinned
    auto pin = [&]() {
        if constexpr (PIN) {
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    };
    constexpr int P = (V >> 6) & 7;
    // DMA source: a 4 MiB window of `src` (L2 / Infinity-Cache resident), a different 1 KiB per (workgroup, wave, piece, step)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4u << 20, 0x00020000);
    unsigned goff = ((blockIdx.x * 8 + wave) * 8192u + lane * 16u) & ((4u << 20) - 1);
    auto dma = [&](int slot, int piece) {
        if constexpr (P > 0) {
            dma16(rsrc, lds + slot * (384 * 128) + ((piece * 8 + wave) % 48) * 1024, goff);
            goff = (goff + 1024u) & ((4u << 20) - 1);
        }
    };
    // pieces of one step, spread over the four groups like the kernels do
    auto dma_group = [&](int slot, int g) {
#pragma unroll
        for (int q = 0; q < P; ++q)
            if ((q & 3) == g) dma(slot, q);
    };
    auto dma_wait = [&]() {
        if constexpr (P > 0) wait_vmcnt<P>();
    };
    if constexpr (P > 0) {           // two stages in flight before the loop
#pragma unroll
        for (int q = 0; q < P; ++q) dma(1, q);
#pragma unroll
        for (int q = 0; q < P; ++q) dma(2, q);
    }
    if (PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
    const bool late = STAG && wave >= 4;
    int stage = 0;
    if constexpr (TRI) {
        // three sets: group g's fragments were requested two groups earlier
        rd(0, 0, 0);
        rd(0, 1, 1);
        for (int it = 0; it < iters; ++it) {
            const int soff = stage * (384 * 128);
            const int nstage = stage == 2 ? 0 : stage + 1;
            const int noff = nstage * (384 * 128);
            if constexpr (BAR) __builtin_amdgcn_s_barrier();
            rd(soff, 2, 2); mm(0); pin();
            rd(soff, 3, 0); mm(1); pin();
            rd(noff, 0, 1); mm(2); pin();
            rd(noff, 1, 2); mm(0); pin();
            // rotate the roles by 1 each k-step is not expressible with static indices: do a second k-step with the
            // sets shifted, and a third, so that one loop body = 3 k-steps
            if constexpr (BAR) __builtin_amdgcn_s_barrier();
            rd(noff, 2, 0); mm(1); pin();
            rd(noff, 3, 1); mm(2); pin();
            rd(soff, 0, 2); mm(0); pin();
            rd(soff, 1, 0); mm(1); pin();
            if constexpr (BAR) __builtin_amdgcn_s_barrier();
            rd(soff, 2, 1); mm(2); pin();
            rd(soff, 3, 2); mm(0); pin();
            rd(noff, 0, 0); mm(1); pin();
            rd(noff, 1, 1); mm(2); pin();
            stage = nstage;
        }
    } else if constexpr (PF) {
        rd(0, 0, 0);
        for (int it = 0; it < iters; ++it) {
            const int soff = stage * (384 * 128);
            const int nstage = stage == 2 ? 0 : stage + 1;
            const int wslot = stage == 0 ? 2 : stage - 1;      // the slot read in the previous step
            dma_wait();
            if constexpr (BAR) { wait_lgkmcnt0(); __builtin_amdgcn_s_barrier(); }
            rd(soff, 1, 1); mm(0); pin(); dma_group(wslot, 0);
            rd(soff, 2, 0); mm(1); pin(); dma_group(wslot, 1);
            rd(soff, 3, 1); mm(0); pin(); dma_group(wslot, 2);
            rd(nstage * (384 * 128), 0, 0); mm(1); pin(); dma_group(wslot, 3);
            stage = nstage;
        }
    } else if constexpr (STAG) {
        // waves 4..7: [groups 2,3 of step s-1 | barrier s | groups 0,1 of step s] ... i.e. their barrier sits in the
        // middle of their k-step; waves 0..3 as usual
        rd(0, 0, 0);
        for (int it = 0; it < iters; ++it) {
            const int soff = stage * (384 * 128);
            const int nstage = stage == 2 ? 0 : stage + 1;
            if (!late) {
                if constexpr (BAR) __builtin_amdgcn_s_barrier();
                rd(soff, 1, 1); mm(0); pin();
                rd(soff, 2, 0); mm(1); pin();
                rd(soff, 3, 1); mm(0); pin();
                rd(nstage * (384 * 128), 0, 0); mm(1); pin();
            } else {
                rd(soff, 1, 1); mm(0); pin();
                rd(soff, 2, 0); mm(1); pin();
                if constexpr (BAR) __builtin_amdgcn_s_barrier();
                rd(soff, 3, 1); mm(0); pin();
                rd(nstage * (384 * 128), 0, 0); mm(1); pin();
            }
            stage = nstage;
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            const int soff = stage * (384 * 128);
            const int wslot = stage == 0 ? 2 : stage - 1;
            dma_wait();
            if constexpr (BAR) __builtin_amdgcn_s_barrier();
            rd(soff, 0, 0);
            rd(soff, 1, 1); mm(0); pin(); dma_group(wslot, 0);
            rd(soff, 2, 0); mm(1); pin(); dma_group(wslot, 1);
            rd(soff, 3, 1); mm(0); pin(); dma_group(wslot, 2);
            mm(1); dma_group(wslot, 3);
            stage = stage == 2 ? 0 : stage + 1;
        }
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += acc[0][0][e] + acc[0][1][e] + acc[1][0][e] + acc[1][1][e];
    dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The same k-step on v_mfma_f32_16x16x32_f16: 64x64 wave tile = 4x4 blocks of 16x16, a 64-deep stage = two 32-deep
// slabs of 8 fragment reads (4 A + 4 B) and 16 MFMAs each.  V: bit 0 barrier, bit 2 cross-barrier prefetch, bits 6..8 DMA.
template <int V>
__global__ __launch_bounds__(512, 2) void kstep_lab16_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * 384 * 128];
    for (int i = threadIdx.x; i < 3 * 384 * 128 / 16; i += 512)
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(src)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, kg = lane >> 4;
    const int prow0 = (wave >> 1) * 64, crow0 = 256 + (wave & 1) * 64;
    constexpr bool BAR = V & 1, PF = V & 4;
    constexpr int P = (V >> 6) & 7;
    floatx4 acc[4][4] = {};
    half8 f[2][8];
    int addr[8][2];                       // [fragment][slab] byte offset inside a stage
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int row = (q < 4 ? prow0 + q * 16 : crow0 + (q - 4) * 16) + fr;
            addr[q][sl] = row * 128 + (((4 * sl + kg) ^ ((row >> 1) & 7)) << 4);
        }
    auto rd = [&](int soff, int sl, int S) {
#pragma unroll
        for (int q = 0; q < 8; ++q) f[S][q] = *reinterpret_cast<const half8*>(lds + soff + addr[q][sl]);
    };
    auto mm = [&](int S) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[S][4 + j], f[S][i], acc[i][j], 0, 0, 0);
    };
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4u << 20, 0x00020000);
    unsigned goff = ((blockIdx.x * 8 + wave) * 8192u + lane * 16u) & ((4u << 20) - 1);
    auto dma = [&](int slot, int piece) {
        if constexpr (P > 0) {
            dma16(rsrc, lds + slot * (384 * 128) + ((piece * 8 + wave) % 48) * 1024, goff);
            goff = (goff + 1024u) & ((4u << 20) - 1);
        }
    };
    auto dma_half = [&](int slot, int hlf) {
#pragma unroll
        for (int q = 0; q < P; ++q)
            if ((q & 1) == hlf) dma(slot, q);
    };
    if constexpr (P > 0) {
#pragma unroll
        for (int q = 0; q < P; ++q) dma(1, q);
#pragma unroll
        for (int q = 0; q < P; ++q) dma(2, q);
    }
    int stage = 0;
    if constexpr (PF) rd(0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        const int soff = stage * (384 * 128);
        const int nstage = stage == 2 ? 0 : stage + 1;
        const int wslot = stage == 0 ? 2 : stage - 1;
        if constexpr (P > 0) wait_vmcnt<P>();
        if constexpr (BAR) { if constexpr (PF) wait_lgkmcnt0(); __builtin_amdgcn_s_barrier(); }
        if constexpr (!PF) rd(soff, 0, 0);
        rd(soff, 1, 1); mm(0); dma_half(wslot, 0);
        if constexpr (PF) rd(nstage * (384 * 128), 0, 0);
        mm(1); dma_half(wslot, 1);
        stage = nstage;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The k-step on the block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, E4M3 operands, unit scales): the same LDS
// image read as bytes - a 128-byte row is 128 k - so a step is 2 kk x (4 fragments of 32 B = 2 ds_read_b128 each) and
// 8 MFMAs of 64 cycles: the LDS traffic and the matrix time of the fp16 step, twice its FLOPs.
typedef int intx8 __attribute__((ext_vector_type(8)));
template <int V>
__global__ __launch_bounds__(512, 2) void kstep_lab8_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * 384 * 128];
    // fill with small-magnitude fp8 bit patterns (exponent field <= 8): no NaN codes, bounded sums
    for (int i = threadIdx.x; i < 3 * 384 * 128 / 16; i += 512) {
        uint4 v = reinterpret_cast<const uint4*>(src)[i];
        v.x &= 0xc7c7c7c7u; v.y &= 0xc7c7c7c7u; v.z &= 0xc7c7c7c7u; v.w &= 0xc7c7c7c7u;
        reinterpret_cast<uint4*>(lds)[i] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int prow0 = (wave >> 1) * 64, crow0 = 256 + (wave & 1) * 64;
    constexpr bool BAR = V & 1;
    constexpr int P = (V >> 6) & 7;
    floatx16 acc[2][2] = {};
    intx8 f[2][4];
    int addr[4][2][2];                    // [fragment][kk][half of the 32-byte fragment]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int row = (q < 2 ? prow0 + q * 32 : crow0 + (q - 2) * 32) + fr;
                addr[q][kk][c] = row * 128 + (((4 * kk + 2 * fh + c) ^ ((row >> 1) & 7)) << 4);
            }
    auto rd = [&](int soff, int kk, int S) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 lo = *reinterpret_cast<const uint4*>(lds + soff + addr[q][kk][0]);
            const uint4 hi = *reinterpret_cast<const uint4*>(lds + soff + addr[q][kk][1]);
            f[S][q] = intx8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
    };
    auto mm = [&](int S) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f[S][2 + j], f[S][i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    };
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4u << 20, 0x00020000);
    unsigned goff = ((blockIdx.x * 8 + wave) * 8192u + lane * 16u) & ((4u << 20) - 1);
    auto dma = [&](int slot, int piece) {
        if constexpr (P > 0) {
            dma16(rsrc, lds + slot * (384 * 128) + ((piece * 8 + wave) % 48) * 1024, goff);
            goff = (goff + 1024u) & ((4u << 20) - 1);
        }
    };
    if constexpr (P > 0) {
#pragma unroll
        for (int q = 0; q < P; ++q) dma(1, q);
#pragma unroll
        for (int q = 0; q < P; ++q) dma(2, q);
    }
    int stage = 0;
    for (int it = 0; it < iters; ++it) {
        const int soff = stage * (384 * 128);
        const int wslot = stage == 0 ? 2 : stage - 1;
        if constexpr (P > 0) wait_vmcnt<P>();
        if constexpr (BAR) __builtin_amdgcn_s_barrier();
        rd(soff, 0, 0);
        rd(soff, 1, 1); mm(0);
#pragma unroll
        for (int q = 0; q < P; ++q) dma(wslot, q);
        mm(1);
        stage = stage == 2 ? 0 : stage + 1;
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += acc[0][0][e] + acc[0][1][e] + acc[1][0][e] + acc[1][1][e];
    dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
}



// packed-fp16 add / subtract on an 8-halfword fragment as four v_pk_add_f16 (a plain <8 x half> subtraction is
// scalarised into v_sub_f16 + sdwa + v_pack by this compiler)
struct H8 { unsigned p[4]; };
__device__ __forceinline__ half8 h8add(half8 a, half8 b) {
    H8 x = __builtin_bit_cast(H8, a), y = __builtin_bit_cast(H8, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) asm("v_pk_add_f16 %0, %1, %2" : "=v"(x.p[i]) : "v"(x.p[i]), "v"(y.p[i]));
    return __builtin_bit_cast(half8, x);
}
__device__ __forceinline__ half8 h8sub(half8 a, half8 b) {
    H8 x = __builtin_bit_cast(H8, a), y = __builtin_bit_cast(H8, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(x.p[i]) : "v"(x.p[i]), "v"(y.p[i]));
    return __builtin_bit_cast(half8, x);
}

// ---------------------------------------------------------------------------------------------------------------
// Mix lab (round 3): instruction mixes of candidate k-loops at ONE wave per SIMD (256-thread workgroups, 256 accumulator
// registers per lane), 16 MFMAs of 32x32x16 per wave and sub-step (16 channels deep):
//   MODE 0  direct conv, 128 x 128 wave tile: 4 A + 4 B fragment reads per sub-step (0.5 reads per MFMA)
//   MODE 1  Winograd F(2,3) along W only (4 frequencies x 64 x 64): 8 raw + 8 U reads, 32 v_pk_add_f16
//   MODE 2  Winograd F(2x2,3x3) (16 frequencies x 32 x 32): 16 raw + 16 U reads, 128 v_pk_add_f16 (B^T d B in registers)
//   MODE 3  as 2 without the transform arithmetic (what the 2 reads per MFMA alone cost)
// BARN: 0 no barrier, 1 one per sub-step, 2 one per four sub-steps.  P: LDS-DMA pieces (1 KiB) per wave and barrier period.
// Reads run one sub-step ahead of their MFMAs (two for the raw operands of MODE 2, whose transform runs between the MFMAs
// of the sub-step before).  Not a convolution: operands are random LDS contents, addresses conflict-free.
template <int MODE, int BARN, int P>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mix_lab_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * 384 * 128];
    for (int i = threadIdx.x; i < 3 * 384 * 128 / 16; i += 256)
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(src)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    floatx16 acc[16] = {};
    // 16 conflict-free fragment addresses per operand side: fragment q = row block q / 4 (32 rows), 16-byte chunk
    // 2 (q % 4) + fh of a 128-byte row; four base addresses per side, the row block rides in the immediate offset
    int baseA[4], baseB[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int rowa = 192 + fr, rowb = (wave & 1) * 16 + fr;
        baseA[c] = rowa * 128 + (((2 * c + fh) ^ ((rowa >> 1) & 7)) << 4);
        baseB[c] = rowb * 128 + (((2 * c + fh) ^ ((rowb >> 1) & 7)) << 4);
    }
#define addrA(q) (baseA[(q) & 3] + ((q) >> 2) * 4096)
#define addrB(q) (baseB[(q) & 3] + ((q) >> 2) * 4096)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4u << 20, 0x00020000);
    unsigned goff = ((blockIdx.x * 4 + wave) * 8192u + lane * 16u) & ((4u << 20) - 1);
    auto dma = [&](int slot, int piece) {
        dma16(rsrc, lds + slot * (384 * 128) + ((piece * 4 + wave) % 48) * 1024, goff);
        goff = (goff + 1024u) & ((4u << 20) - 1);
    };
    auto ld = [&](int off) -> half8 { return *reinterpret_cast<const half8*>(lds + off); };
    if constexpr (P > 0) {
#pragma unroll
        for (int q = 0; q < P; ++q) dma(1, q);
#pragma unroll
        for (int q = 0; q < P; ++q) dma(2, q);
    }
    int stage = 0, sub = 0;
    auto period = [&](int& soff, int& wslot) {      // start of a barrier period
        soff = stage * (384 * 128);
        wslot = stage == 0 ? 2 : stage - 1;
        if constexpr (P > 0) wait_vmcnt<P>();
        if constexpr (BARN > 0) __builtin_amdgcn_s_barrier();
    };
    int soff = 0, wslot = 2;
    if constexpr (MODE == 0) {
        half8 a[2][4], b[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[0][q] = ld(addrA(q)); b[0][q] = ld(addrB(q)); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (BARN == 1 || (BARN == 2 && (sub & 3) == 0) || (BARN == 0 && P > 0 && (sub & 3) == 0)) period(soff, wslot);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[s ^ 1][i] = ld(soff + addrA(i));
                    b[s ^ 1][i] = ld(soff + addrB(i));
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][j], b[s][i], acc[i * 4 + j], 0, 0, 0);
                    if constexpr (P > 0) { if (BARN == 1) { for (int q = i; q < P; q += 4) dma(wslot, q); } else if ((sub & 3) == i) { for (int q = 0; q < P; ++q) if ((q & 3) == i) dma(wslot, q); } }
                }
                ++sub;
                if ((BARN == 1) || (sub & 3) == 0) stage = stage == 2 ? 0 : stage + 1;
            }
        }
    } else if constexpr (MODE == 1) {
        // 4 frequencies x (2 tile blocks x 2 cout blocks); raw d[tile block][4 positions], V in place
        half8 d[2][2][4], u[2][8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { d[0][q >> 2][q & 3] = ld(addrB(q)); u[0][q] = ld(addrA(q)); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (BARN == 1 || (BARN == 2 && (sub & 3) == 0) || (BARN == 0 && P > 0 && (sub & 3) == 0)) period(soff, wslot);
                half8 v[2][4];
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) {
                    v[tb][0] = h8sub(d[s][tb][0], d[s][tb][2]);
                    v[tb][1] = h8add(d[s][tb][1], d[s][tb][2]);
                    v[tb][2] = h8sub(d[s][tb][2], d[s][tb][1]);
                    v[tb][3] = h8sub(d[s][tb][1], d[s][tb][3]);
                }
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    d[s ^ 1][f >> 1][(f & 1) * 2] = ld(soff + addrB(f * 2));
                    d[s ^ 1][f >> 1][(f & 1) * 2 + 1] = ld(soff + addrB(f * 2 + 1));
                    u[s ^ 1][f * 2] = ld(soff + addrA(f * 2));
                    u[s ^ 1][f * 2 + 1] = ld(soff + addrA(f * 2 + 1));
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                        for (int cbk = 0; cbk < 2; ++cbk)
                            acc[f * 4 + tb * 2 + cbk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u[s][f * 2 + cbk], v[tb][f], acc[f * 4 + tb * 2 + cbk], 0, 0, 0);
                    if constexpr (P > 0) { if (BARN == 1) { for (int q = f; q < P; q += 4) dma(wslot, q); } else if ((sub & 3) == f) { for (int q = 0; q < P; ++q) if ((q & 3) == f) dma(wslot, q); } }
                }
                ++sub;
                if ((BARN == 1) || (sub & 3) == 0) stage = stage == 2 ? 0 : stage + 1;
            }
        }
    } else {
        // 16 frequencies x one 32 x 32 block.  r[x]: raw 4x4 patch of sub-step x (3 in rotation: being multiplied as V,
        // being transformed, arriving); u: weight fragments in two groups of 8
        half8 r[3][16], u[2][8];
#pragma unroll
        for (int q = 0; q < 16; ++q) { r[0][q] = ld(addrB(q)); r[1][q] = ld(addrB(q)); }
#pragma unroll
        for (int q = 0; q < 8; ++q) u[0][q] = ld(addrA(q));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                period(soff, wslot);
                constexpr int dummy = 0; (void)dummy;
                half8 (&V)[16] = r[s];                  // already transformed
                half8 (&T)[16] = r[(s + 1) % 3];        // transform during this sub-step
                half8 (&N)[16] = r[(s + 2) % 3];        // arriving
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // MFMAs of frequencies 4g..4g+3; U fragments of group g were read one group earlier
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int f = 4 * g + e;
                        acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u[g & 1][(g & 1) * 0 + e + 4 * ((g >> 0) & 0)], V[f], acc[f], 0, 0, 0);
                        N[f] = ld(soff + addrB(f));
                    }
                    // next U group (frequencies 4(g+1).. of this sub-step, or 0..3 of the next one)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        u[(g + 1) & 1][e] = ld(soff + addrA((4 * (g + 1) + e) & 15));
                    if constexpr (MODE == 2) {
                        // a quarter of B^T d B on T: g = 0, 1: columns 2g, 2g+1 (first pass); g = 2, 3: rows 2(g-2), +1
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const int l = 2 * (g & 1) + c;
                            half8 x0, x1, x2, x3;
                            if (g < 2) { x0 = T[l]; x1 = T[4 + l]; x2 = T[8 + l]; x3 = T[12 + l]; }
                            else { x0 = T[4 * l]; x1 = T[4 * l + 1]; x2 = T[4 * l + 2]; x3 = T[4 * l + 3]; }
                            const half8 y0 = h8sub(x0, x2), y1 = h8add(x1, x2), y2 = h8sub(x2, x1), y3 = h8sub(x1, x3);
                            if (g < 2) { T[l] = y0; T[4 + l] = y1; T[8 + l] = y2; T[12 + l] = y3; }
                            else { T[4 * l] = y0; T[4 * l + 1] = y1; T[4 * l + 2] = y2; T[4 * l + 3] = y3; }
                        }
                    }
                    if constexpr (P > 0) { for (int q = g; q < P; q += 4) dma(wslot, q); }
                }
                ++sub;
                stage = stage == 2 ? 0 : stage + 1;
            }
        }
        float s2 = 0.f;
        for (int q = 0; q < 16; ++q) s2 += (float)r[0][q][0] + (float)r[1][q][1] + (float)r[2][q][2];
        dst[blockIdx.x * blockDim.x + threadIdx.x] = s2;
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q)
        for (int e = 0; e < 16; ++e) s += acc[q][e];
    dst[blockIdx.x * blockDim.x + threadIdx.x] += s;
}

#undef addrA
#undef addrB
// sub-steps (of 16 MFMAs per wave) per loop iteration of a mix-lab variant
static int mix_lab_substeps(int mode) { return mode >= 2 ? 3 : 2; }

template <int V>
static void lab_launch(const _Float16* src, float* dst, int blocks, int iters, hipStream_t stream) {
    hipLaunchKernelGGL(kstep_lab_kernel<V>, dim3(blocks), dim3(512), 0, stream, src, dst, iters);
}

// k-steps executed per loop iteration of a variant (the three-set variants unroll three)
int kstep_lab_steps_per_iter(int variant) {
    if (variant & 2048) return mix_lab_substeps(variant & 7);
    return ((variant & 32) && !(variant & 512)) ? 3 : 1;
}
// waves per workgroup (each issuing 16 MFMAs of 32x32x16 per step)
int kstep_lab_waves(int variant) { return (variant & 2048) ? 4 : 8; }

hipError_t launch_kstep_lab(const _Float16* src, float* dst, int blocks, int variant, int iters, hipStream_t stream) {
    if (variant & 2048) {
        // mix lab: 2048 | mode (bits 0..2) | barrier mode (bits 4..5) | DMA pieces (bits 6..9)
        switch (variant & 2047) {
#define MIX(m, b, p) case (m) | ((b) << 4) | ((p) << 6): hipLaunchKernelGGL((mix_lab_kernel<m, b, p>), dim3(blocks), dim3(256), 0, stream, src, dst, iters); break;
            MIX(0, 0, 0) MIX(0, 2, 0) MIX(0, 2, 8) MIX(0, 2, 12)
            MIX(1, 0, 0) MIX(1, 2, 0) MIX(1, 2, 8) MIX(1, 2, 12)
            MIX(2, 0, 0) MIX(2, 1, 0) MIX(2, 1, 4) MIX(2, 1, 9)
            MIX(3, 0, 0) MIX(3, 1, 0) MIX(3, 1, 4) MIX(3, 1, 9)
#undef MIX
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (variant) {
#define LABCASE(v) case v: lab_launch<v>(src, dst, blocks, iters, stream); break;
        LABCASE(0) LABCASE(1) LABCASE(2) LABCASE(3) LABCASE(4) LABCASE(5) LABCASE(6) LABCASE(7)
        LABCASE(9) LABCASE(11) LABCASE(17) LABCASE(19) LABCASE(21) LABCASE(23) LABCASE(25) LABCASE(27)
        LABCASE(32) LABCASE(33) LABCASE(34) LABCASE(35) LABCASE(49) LABCASE(51)
        LABCASE(64 * 4 + 1) LABCASE(64 * 4 + 7) LABCASE(64 * 4 + 5) LABCASE(64 * 6 + 1) LABCASE(64 * 6 + 7) LABCASE(64 * 2 + 1) LABCASE(64 * 2 + 7)
        LABCASE(64 * 4 + 0) LABCASE(64 * 3 + 1) LABCASE(64 * 3 + 7)
#undef LABCASE
#define LAB16(v) case 512 + v: hipLaunchKernelGGL(kstep_lab16_kernel<v>, dim3(blocks), dim3(512), 0, stream, src, dst, iters); break;
        LAB16(0) LAB16(1) LAB16(5) LAB16(64 * 4 + 1) LAB16(64 * 4 + 5) LAB16(64 * 4 + 0)
#undef LAB16
#define LAB8(v) case 1024 + v: hipLaunchKernelGGL(kstep_lab8_kernel<v>, dim3(blocks), dim3(512), 0, stream, src, dst, iters); break;
        LAB8(0) LAB8(1) LAB8(64 * 4 + 1) LAB8(64 * 4 + 0)
#undef LAB8
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace frp
