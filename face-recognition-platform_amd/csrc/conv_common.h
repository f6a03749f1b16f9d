// Device helpers shared by the MFMA convolution kernels (conv_mfma.hip, conv3x3_rows.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frp_internal.h"

namespace frp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
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
// written to a buffer nothing else reads
__device__ __forceinline__ void stamp(unsigned long long* buf, int slot) {
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

// q = m / d, r = m % d for 0 <= m < 2^24 via a float reciprocal and one correction step
// (an integer division costs ~40 instructions; the prologue needs two per pixel row)
__device__ __forceinline__ void fast_divmod(int m, int d, float inv_d, int& q, int& r) {
    q = (int)((float)m * inv_d);
    r = m - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
}

__device__ __forceinline__ void wait_lgkmcnt0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N == 0, "add the literal");
}

}  // namespace frp
