// K6: embedding-vs-watchlist cosine match with fused top-1 (batched GEMV over the gallery).
//
// Replaces face_recognition.face_distance(stored_encodings, q) + the per-target Python loop
// of FaceService.compare_faces (backend/app/services/face_service.py:409-432) and the caller's
// filter (backend/app/routes/camera.py:246-256) for M query faces at once.  Gallery rows and
// queries are unit vectors, so ranking by cosine == ranking by the reference's Euclidean
// distance (d^2 = 2 - 2cos).
//
// HBM layout: gallery [N][512] fp16 row-major, streamed exactly once per pass; queries
// [Mpad][512] fp16 (L2-resident).  A workgroup owns 128 gallery rows: each of its 4 waves
// loads 32 rows x 512 k straight into registers as 32 MFMA A-fragments (all 32 loads in
// flight at once = 32 KiB per wave), then walks the query tiles (32 queries each, staged by LDS-DMA in a
// swizzled LDS image: two separate buffers, the DMA of tile t+1 issued before the MFMAs of tile t and waited
// for - s_waitcnt vmcnt(0) - only after them) with v_mfma_f32_32x32x16_f16 and reduces the 32x32
// score tile to a running (max, argmin-index-on-ties) per query.  Per-workgroup partial
// winners go to HBM and a second tiny kernel reduces them.  Algorithmic bytes = N*512*2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frp_internal.h"

namespace frp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define MT_G 128           // gallery rows per workgroup
#define MT_Q 32            // queries per tile
#define MD 512

__device__ __forceinline__ int q_lds_off(int row, int chunk) {   // 1 KiB rows, 64 chunks of 16 B
    return row * 1024 + ((chunk ^ (row & 15)) << 4);
}

__global__ __launch_bounds__(256, 2) void match_kernel(MatchParams p) {
    // two DISTINCT arrays: with one runtime-indexed array the compiler cannot tell the DMA target from the tile being
    // read and waits for the prefetch (vmcnt(0)) before the first fragment read of every tile
    __shared__ __attribute__((aligned(16))) unsigned char qs0[MT_Q * 1024];
    __shared__ __attribute__((aligned(16))) unsigned char qs1[MT_Q * 1024];
    __shared__ float red_cos[4][MT_Q];
    __shared__ int red_idx[4][MT_Q];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const long g0 = (long)blockIdx.x * MT_G + wave * 32;

    // ---- gallery rows -> registers (A fragments), rows past N clamped (masked in the epilogue)
    half8 gf[32];
    {
        long row = g0 + fr;
        if (row >= p.N) row = p.N - 1;
        const _Float16* gp = p.gallery + row * MD + fh * 8;
#pragma unroll
        for (int s = 0; s < 32; ++s) gf[s] = *reinterpret_cast<const half8*>(gp + s * 16);
    }

    const int nqt = (p.M + MT_Q - 1) / MT_Q;
    // query tile staging by LDS-DMA (no staging VGPRs: the 128 fragment registers stay resident).
    // One wave-instruction writes one 1-KiB row linearly (lane L -> chunk position L), so the
    // bank swizzle is applied on the per-lane SOURCE address: position L holds chunk L ^ (row&15).
    auto q_dma = [&](int qt, unsigned char* buf) {
        const _Float16* qp = p.q + (long)qt * MT_Q * MD;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = wave * 8 + i;
            const _Float16* src = qp + (long)row * MD + ((lane ^ (row & 15)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(buf + row * 1024), 16, 0, 0);
        }
    };
    q_dma(0, qs0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int qt = 0; qt < nqt; ++qt) {
        const bool odd = qt & 1;
        if (qt + 1 < nqt) { if (odd) q_dma(qt + 1, qs0); else q_dma(qt + 1, qs1); }
        floatx16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        if (odd) {
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const half8 qf = *reinterpret_cast<const half8*>(&qs1[q_lds_off(fr, 2 * s + fh)]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(gf[s], qf, acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const half8 qf = *reinterpret_cast<const half8*>(&qs0[q_lds_off(fr, 2 * s + fh)]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(gf[s], qf, acc, 0, 0, 0);
            }
        }
        // acc[e]: gallery row (e&3) + 8*(e>>2) + 4*fh of this wave's 32, query column fr
        const int q = qt * MT_Q + fr;
        float best = -3.0f;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long g = g0 + (e & 3) + 8 * (e >> 2) + 4 * fh;
            const float v = acc[e];
            if (g < p.N) {
                if (p.all_scores && q < p.M) p.all_scores[(long)q * p.N + g] = v;
                if (v > best) { best = v; bidx = (int)g; }   // rows visited in increasing g: ties keep the lower index
            }
        }
        {   // combine the two half-waves (same query column)
            const float ob = __shfl_xor(best, 32);
            const int oi = __shfl_xor(bidx, 32);
            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        }
        if (lane < 32) { red_cos[wave][fr] = best; red_idx[wave][fr] = bidx; }
        __syncthreads();
        if (t < MT_Q) {
            float bc = red_cos[0][t];
            int bi = red_idx[0][t];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float c = red_cos[w][t];
                const int i = red_idx[w][t];
                if (c > bc || (c == bc && i < bi)) { bc = c; bi = i; }
            }
            const long o = (long)blockIdx.x * p.Mpad + qt * MT_Q + t;
            p.part_cos[o] = bc;
            p.part_idx[o] = bi;
        }
        // the next tile's DMA (issued before this tile's MFMAs) has had the whole tile to land
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// one wave per query: reduce the per-workgroup partials
__global__ __launch_bounds__(256) void match_reduce_kernel(MatchParams p) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= p.M) return;
    float best = -3.0f;
    int bidx = 0x7fffffff;
    for (int w = lane; w < p.n_wg; w += 64) {
        const float c = p.part_cos[(long)w * p.Mpad + q];
        const int i = p.part_idx[(long)w * p.Mpad + q];
        if (c > best || (c == best && i < bidx)) { best = c; bidx = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bidx, o);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    if (lane == 0) {
        p.best_cos[q] = best;
        p.best_idx[q] = bidx == 0x7fffffff ? -1 : bidx;
    }
}

int match_num_workgroups(long N) { return (int)((N + MT_G - 1) / MT_G); }

hipError_t launch_match(const MatchParams& p, hipStream_t stream) {
    if (p.N <= 0 || p.M <= 0 || !p.gallery || !p.q || !p.part_cos || !p.part_idx || !p.best_cos || !p.best_idx)
        return hipErrorInvalidValue;
    if (p.Mpad % MT_Q != 0 || p.Mpad < p.M || p.n_wg != match_num_workgroups(p.N) || p.N > 0x7fffff00L)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(match_kernel, dim3(p.n_wg), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(match_reduce_kernel, dim3((p.M + 3) / 4), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// Top-k per query row of a device score matrix [M x N], ordered by (cosine descending, row ascending):
// the order of the oracle's stable argsort and of the fused top-1.  One workgroup per query; pass r
// takes the best element strictly AFTER the (r-1)-th in that total order, so k coalesced passes over
// the row (N*4 bytes each, from L2/MALL for N <= a few million) and no per-thread candidate lists.
// Replaces np.argpartition + argsort over face_distance (backend/app/services/face_service.py:599-603).
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ scores, long N, int k,
                                                        int32_t* __restrict__ idx_out, float* __restrict__ cos_out) {
    __shared__ float sc[4];
    __shared__ int si[4];
    const float* s = scores + (long)blockIdx.x * N;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float prev_c = __builtin_inff();
    int prev_i = -1;
    for (int r = 0; r < k; ++r) {
        float bc = -__builtin_inff();
        int bi = 0x7fffffff;
        for (long j = t; j < N; j += 256) {
            const float c = s[j];
            const bool after_prev = (c < prev_c) || (c == prev_c && (int)j > prev_i);
            if (after_prev && (c > bc || (c == bc && (int)j < bi))) { bc = c; bi = (int)j; }
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const float oc = __shfl_down(bc, off);
            const int oi = __shfl_down(bi, off);
            if (oc > bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
        }
        if (lane == 0) { sc[wave] = bc; si[wave] = bi; }
        __syncthreads();
        if (t == 0) {
#pragma unroll
            for (int w = 1; w < 4; ++w)
                if (sc[w] > bc || (sc[w] == bc && si[w] < bi)) { bc = sc[w]; bi = si[w]; }
            const bool found = bi != 0x7fffffff;
            idx_out[(long)blockIdx.x * k + r] = found ? bi : -1;
            cos_out[(long)blockIdx.x * k + r] = found ? bc : -2.0f;
            sc[0] = found ? bc : -__builtin_inff();
            si[0] = found ? bi : 0x7fffffff;
        }
        __syncthreads();
        prev_c = sc[0];
        prev_i = si[0];
        __syncthreads();
    }
}

hipError_t launch_topk_rows(const float* scores, int M, long N, int k, int32_t* idx_out, float* cos_out, hipStream_t stream) {
    if (!scores || !idx_out || !cos_out || M <= 0 || N <= 0 || N > 0x7fffff00L || k <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(topk_rows_kernel, dim3(M), dim3(256), 0, stream, scores, N, k, idx_out, cos_out);
    return hipGetLastError();
}

}  // namespace frp
