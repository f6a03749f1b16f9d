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

template <int V>
static void lab_launch(const _Float16* src, float* dst, int blocks, int iters, hipStream_t stream) {
    hipLaunchKernelGGL(kstep_lab_kernel<V>, dim3(blocks), dim3(512), 0, stream, src, dst, iters);
}

// k-steps executed per loop iteration of a variant (the three-set variants unroll three)
int kstep_lab_steps_per_iter(int variant) { return ((variant & 32) && !(variant & 512)) ? 3 : 1; }

hipError_t launch_kstep_lab(const _Float16* src, float* dst, int blocks, int variant, int iters, hipStream_t stream) {
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
