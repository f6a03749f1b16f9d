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
#include <stdlib.h>
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

// Top-1 only, M <= 512 queries (the streaming path: 320 faces per step): PERSISTENT workgroups of eight waves (one per CU;
// a query tile in LDS serves 256 gallery rows) with the running winner of every query kept in REGISTERS across all the
// gallery blocks a wave visits.  The first kernel above reduces each 32-query tile across its four waves through LDS and
// writes a partial winner per workgroup and tile to HBM (two barriers, ~80 compare/select instructions and 20 MB of
// partials per million rows); here a tile costs its 32 MFMAs, a 16-way max and - only in the lanes whose tile maximum
// beats their running best, which becomes rare after the first few blocks - the index search.  One barrier per tile (the
// query-tile DMA), one cross-wave reduction per launch.  1 M rows x 320 queries: 0.48 ms against 0.64 (2.1 TB/s of
// gallery = 0.69 PFLOP/s; like the conv kernels it reads one LDS fragment per MFMA, and that ratio - not the HBM
// stream, not the accumulator chain, not the query DMA: a second accumulator, a staggered start and four-wave
// workgroups all measured the same - is what bounds it).
// Ties: a lane meets its rows in increasing order (blocks in increasing order, rows of a block in increasing order) and
// replaces its winner only on a strictly greater score; half-waves, waves and workgroups are merged by (score desc,
// row asc) - the same total order as the first kernel and the oracle, and bit-identical results.
#define MT_MAXQT 16
#define MT_NW 8            // waves per workgroup: one query tile in LDS serves 256 gallery rows
static_assert(MT_MAXQT * MT_Q == FRP_MATCH_TOP1_MAX, "frp_internal.h: FRP_MATCH_TOP1_MAX");
__global__ __launch_bounds__(64 * MT_NW, 2) void match_top1_kernel(MatchParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char qs0[MT_Q * 1024];
    __shared__ __attribute__((aligned(16))) unsigned char qs1[MT_Q * 1024];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    int nqt = p.Mpad / MT_Q;                                        // <= MT_MAXQT (launch_match)
    if (p.n_dev) {                                                  // query count known on the device only: tiles that hold
        int n = *p.n_dev;                                           // real queries (p.M / p.Mpad are the capacity and the stride)
        n = n < 0 ? 0 : (n > p.M ? p.M : n);
        nqt = (n + MT_Q - 1) / MT_Q;
    }
    const long nb = (p.N + 31) / 32;                                // 32-row gallery blocks
    const long wstride = (long)gridDim.x * MT_NW;
    const long rounds = (nb + wstride - 1) / wstride;               // the same for every wave: they share the barriers

    const int qbase = q_lds_off(fr, fh);
    float best[MT_MAXQT];
    int bidx[MT_MAXQT];
#pragma unroll
    for (int i = 0; i < MT_MAXQT; ++i) { best[i] = -3.0f; bidx[i] = 0x7fffffff; }

    auto q_dma = [&](int qt, unsigned char* buf) {
        // (opaque tile index: the 16 x 8 source addresses of a round are the same in every round, and the compiler would
        // rather keep 256 registers of them than recompute eight per tile)
        int qto = qt;
        asm volatile("" : "+s"(qto));
        const _Float16* qp = p.q + (long)qto * MT_Q * MD;
#pragma unroll
        for (int i = 0; i < MT_Q / MT_NW; ++i) {
            const int row = wave * (MT_Q / MT_NW) + i;
            const _Float16* src = qp + (long)row * MD + ((lane ^ (row & 15)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(buf + row * 1024), 16, 0, 0);
        }
    };

    for (long r = 0; r < rounds; ++r) {
        const long g0 = (r * wstride + (long)blockIdx.x * MT_NW + wave) * 32;     // may lie beyond N: every row masked then
        const bool ragged = g0 + 32 > p.N;
        half8 gf[32];
        {
            long row = g0 + fr;
            if (row >= p.N) row = p.N - 1;
            const _Float16* gp = p.gallery + row * MD + fh * 8;
#pragma unroll
            for (int s = 0; s < 32; ++s) gf[s] = *reinterpret_cast<const half8*>(gp + s * 16);
        }
        q_dma(0, qs0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int qt = 0; qt < MT_MAXQT; ++qt) {
            if (qt < nqt) {
                const bool odd = qt & 1;
                if (qt + 1 < nqt) { if (odd) q_dma(qt + 1, qs0); else q_dma(qt + 1, qs1); }
                floatx16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                const unsigned char* buf = odd ? qs1 : qs0;
                // fragment s of this lane sits at qbase ^ (s << 5) (q_lds_off: chunk 2s + fh, xor-swizzled by fr & 15).
                // One address register made opaque per tile - else all 32 addresses are hoisted out of the loops and
                // held in registers, which the 32 running winners need
                int qa = qbase;
                asm volatile("" : "+v"(qa));
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    const half8 qf = *reinterpret_cast<const half8*>(&buf[qa ^ (s << 5)]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(gf[s], qf, acc, 0, 0, 0);
                    // (fragment reads stay within their group of 8: left alone the scheduler hoists all 32 - 128 registers -
                    // and the 32 running winners no longer fit)
                    if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
                // acc[e]: gallery row (e&3) + 8*(e>>2) + 4*fh of this wave's 32 (increasing in e), query column fr
                if (ragged) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (g0 + (e & 3) + 8 * (e >> 2) + 4 * fh >= p.N) acc[e] = -3.0f;
                }
                float tm = acc[0];
#pragma unroll
                for (int e = 1; e < 16; ++e) tm = fmaxf(tm, acc[e]);
                if (tm > best[qt]) {                    // rare after the first blocks: most tiles stop here
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (acc[e] > best[qt]) { best[qt] = acc[e]; bidx[qt] = (int)(g0 + (e & 3) + 8 * (e >> 2) + 4 * fh); }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next tile's DMA has had this tile to land
                __builtin_amdgcn_s_barrier();
            }
        }
    }

    // ---- one reduction per launch: half-waves, then the four waves through LDS (the query buffers are free now)
    float* red_cos = reinterpret_cast<float*>(qs0);                  // [MT_NW][Mpad]: 16 KiB at most
    int* red_idx = reinterpret_cast<int*>(qs1);
#pragma unroll
    for (int qt = 0; qt < MT_MAXQT; ++qt) {
        if (qt < nqt) {
            float b = best[qt];
            int bi = bidx[qt];
            const float ob = __shfl_xor(b, 32);
            const int oi = __shfl_xor(bi, 32);
            if (ob > b || (ob == b && oi < bi)) { b = ob; bi = oi; }
            if (lane < 32) { red_cos[wave * p.Mpad + qt * MT_Q + fr] = b; red_idx[wave * p.Mpad + qt * MT_Q + fr] = bi; }
        }
    }
    __syncthreads();
    for (int q = t; q < p.Mpad; q += 64 * MT_NW) {
        float bc = red_cos[q];
        int bi = red_idx[q];
#pragma unroll
        for (int w = 1; w < MT_NW; ++w) {
            const float c = red_cos[w * p.Mpad + q];
            const int i = red_idx[w * p.Mpad + q];
            if (c > bc || (c == bc && i < bi)) { bc = c; bi = i; }
        }
        p.part_cos[(long)blockIdx.x * p.Mpad + q] = bc;
        p.part_idx[(long)blockIdx.x * p.Mpad + q] = bi;
    }
}

// one wave per query: reduce the per-workgroup partials
__global__ __launch_bounds__(256) void match_reduce_kernel(MatchParams p) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= p.M || (p.n_dev && q >= *p.n_dev)) return;
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

static int match_cu_count(int dev) {
    static int cached[64] = {};
    if (dev < 0 || dev >= 64) return 0;
    if (cached[dev] <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cached[dev] = n;
    }
    return cached[dev];
}

int match_num_workgroups(long N) { return (int)((N + MT_G - 1) / MT_G); }

hipError_t launch_match(const MatchParams& p, hipStream_t stream) {
    if (p.N <= 0 || p.M <= 0 || !p.gallery || !p.q || !p.part_cos || !p.part_idx || !p.best_cos || !p.best_idx)
        return hipErrorInvalidValue;
    if (p.Mpad % MT_Q != 0 || p.Mpad < p.M || p.n_wg != match_num_workgroups(p.N) || p.N > 0x7fffff00L)
        return hipErrorInvalidValue;
    MatchParams r = p;
    int dev = 0;
    if (p.n_dev && (p.all_scores || p.Mpad > MT_MAXQT * MT_Q || getenv("FRP_MATCH_V1"))) return hipErrorInvalidValue;   // top-1 kernel only
    if (!p.all_scores && p.Mpad <= MT_MAXQT * MT_Q && !getenv("FRP_MATCH_V1") && hipGetDevice(&dev) == hipSuccess) {
        // top-1 for a streaming batch: persistent workgroups, running winners in registers; its partials are the first
        // `grid` rows of the buffers the caller sized for the per-tile kernel (grid <= n_wg)
        const int ncu = match_cu_count(dev);
        const long want = ((p.N + 31) / 32 + MT_NW - 1) / MT_NW;
        const int grid = (int)(want < (long)ncu ? want : (long)ncu);          // 8 waves x ~220 registers: one workgroup per CU
        if (ncu <= 0 || grid <= 0 || grid > p.n_wg) return hipErrorInvalidValue;
        r.n_wg = grid;
        hipLaunchKernelGGL(match_top1_kernel, dim3(grid), dim3(64 * MT_NW), 0, stream, p);
    } else {
        hipLaunchKernelGGL(match_kernel, dim3(p.n_wg), dim3(256), 0, stream, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(match_reduce_kernel, dim3((p.M + 3) / 4), dim3(256), 0, stream, r);
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


// ---------------------------------------------------------------------------------------------------------------
// Exact rows of the REST-style compat path (frp.h: frp_gallery_exact / frp_gallery_distances).  HBM-bound: 4 KB per row and
// pass (N x 512 float64), the arithmetic - 512 float64 subtract + FMA per row and query - is a few per cent of the fp64 rate.
__global__ void gallery_widen_kernel(const _Float16* in, double* out, long n_elems) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= n_elems) return;
    const half8 v = *reinterpret_cast<const half8*>(in + i);
#pragma unroll
    for (int e = 0; e < 8; ++e) out[i + e] = (double)(float)v[e];
}
hipError_t launch_gallery_widen(const _Float16* in, double* out, long N, int D, hipStream_t stream) {
    const long n = N * D;
    if (n <= 0) return hipSuccess;
    if (D % 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gallery_widen_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, in, out, n);
    return hipGetLastError();
}

// One wave per gallery row (a workgroup of four waves walks rows blockIdx * 4 + wave, + 4 gridDim, ...): lane l holds elements
// l, l + 64, ... of the row (8 coalesced 512-byte loads) and, per query of the chunk in LDS, sums its eight squared differences
// in that order; the 64 partial sums are combined by a butterfly (xor 32, 16, ... 1): a fixed order, so a distance depends on the
// row and the query only - not on N, M or the launch geometry.  sqrt is the correctly rounded one.
#define GD_QCHUNK 8
__global__ __launch_bounds__(256) void gallery_distances_kernel(const double* __restrict__ rows, long N, const double* __restrict__ q, int M,
                                                                 double* __restrict__ out) {
    __shared__ double qs[GD_QCHUNK][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int m0 = 0; m0 < M; m0 += GD_QCHUNK) {
        const int mc = M - m0 < GD_QCHUNK ? M - m0 : GD_QCHUNK;
        __syncthreads();
        for (int i = threadIdx.x; i < mc * 512; i += 256) qs[i >> 9][i & 511] = q[(long)m0 * 512 + i];
        __syncthreads();
        for (long r = (long)blockIdx.x * 4 + wave; r < N; r += (long)gridDim.x * 4) {
            double g[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = rows[r * 512 + lane + 64 * i];
            for (int m = 0; m < mc; ++m) {
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double d = g[i] - qs[m][lane + 64 * i];
                    acc = __builtin_fma(d, d, acc);
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
                if (lane == 0) out[(long)(m0 + m) * N + r] = __builtin_sqrt(acc);
            }
        }
    }
}
hipError_t launch_gallery_distances(const double* rows, long N, const double* q, int M, double* out, hipStream_t stream) {
    if (N <= 0 || M <= 0) return hipSuccess;
    long blocks = (N + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gallery_distances_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rows, N, q, M, out);
    return hipGetLastError();
}

}  // namespace frp
