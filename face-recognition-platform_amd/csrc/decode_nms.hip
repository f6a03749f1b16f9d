// K3: anchor decode + candidate selection + ordering + greedy NMS: a chip-wide logit gather, then one
// workgroup per frame.
//
// Replaces the tail of face_recognition.face_locations (backend/app/routes/camera.py:232,
// backend/app/services/face_service.py:156) and the caller's cap `face_locations[:max_faces]`
// (camera.py:233-235) for the anchor-dense detector: strides {8,16,32}, 2 anchors per
// location, 15 values per anchor (logit, 4 distances, 5 x (dx,dy)).
//
// Ordering is total and deterministic: every anchor gets the unique 36-bit key
//   (sortable fp16 logit bits << 20) | (0xFFFFF - anchor_index)
// so "logit descending, ties by lower anchor index" is a plain integer order.  Candidates are
// the anchors with logit >= logit(threshold); if more than CAP=1024 survive, an exact 3-pass
// radix select keeps the CAP largest keys.  The candidates are bitonic-sorted in LDS and NMS
// runs greedily in that order with a barrier only on kept boxes (<= max_faces of them).
// All box arithmetic is fp32 with one rounding per operation (file is built with
// -ffp-contract=off) so it reproduces the fp32 oracle bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frp_internal.h"

namespace frp {

#define NMS_CAP 1024
#define NT 1024

__device__ __forceinline__ unsigned sortable16(unsigned short h) {
    if ((h & 0x7fffu) > 0x7c00u) return 0u;     // NaN (real fp16 weights can overflow a head): the lowest key, below -inf
    return (h & 0x8000u) ? (unsigned)(unsigned short)(~h) : (unsigned)(h | 0x8000u);
}
// candidate test: logit >= threshold; in forced top-K mode (threshold -inf) EVERY anchor is a candidate, NaN logits
// included (they sort last), so exactly min(K, anchors) faces come out and the compact face list never runs short
__device__ __forceinline__ bool is_candidate(_Float16 lg, float lt) {
    return lt == -INFINITY ? true : (float)lg >= lt;
}

// scalar members + select chains (runtime-indexed arrays would live in scratch)
struct FrameView {
    const _Float16 *h0, *h1, *h2;
    int w0, w1, w2;
    int e0, e1;     // cumulative anchor counts after level 0 / 1
};

__device__ __forceinline__ const _Float16* anchor_ptr(const FrameView& fv, int idx, int& level, int& x, int& y) {
    level = idx < fv.e0 ? 0 : (idx < fv.e1 ? 1 : 2);
    const int li = idx - (level == 0 ? 0 : (level == 1 ? fv.e0 : fv.e1));
    const int wl = level == 0 ? fv.w0 : (level == 1 ? fv.w1 : fv.w2);
    const _Float16* base = level == 0 ? fv.h0 : (level == 1 ? fv.h1 : fv.h2);
    const int pos = li >> 1, a = li & 1;
    y = pos / wl;
    x = pos - y * wl;
    return base + (long)pos * 32 + a * 15;
}

__device__ __forceinline__ FrameView frame_view(const DecodeParams& p, int b, int& A) {
    FrameView fv;
    fv.h0 = p.head[0] + (long)b * p.hl[0] * p.wl[0] * 32;
    fv.h1 = p.head[1] + (long)b * p.hl[1] * p.wl[1] * 32;
    fv.h2 = p.head[2] + (long)b * p.hl[2] * p.wl[2] * 32;
    fv.w0 = p.wl[0]; fv.w1 = p.wl[1]; fv.w2 = p.wl[2];
    fv.e0 = p.hl[0] * p.wl[0] * 2;
    fv.e1 = fv.e0 + p.hl[1] * p.wl[1] * 2;
    A = fv.e1 + p.hl[2] * p.wl[2] * 2;
    return fv;
}

// The logits sit at a 30- or 34-byte stride inside 64-byte location rows: this chip-wide pass gathers
// them once into a dense per-frame array (171 KB at 1080p, L2-resident) that the per-frame workgroup
// below counts / selects / gathers from, instead of one workgroup dragging 5.5 MB of sectors per frame.
__global__ __launch_bounds__(256) void gather_logits_kernel(DecodeParams p) {
    int A;
    const FrameView fv = frame_view(p, blockIdx.y, A);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A) return;
    int lv, x, y;
    p.logits[(long)blockIdx.y * A + i] = anchor_ptr(fv, i, lv, x, y)[0];
}

__global__ __launch_bounds__(NT) void decode_nms_kernel(DecodeParams p) {
    __shared__ unsigned long long keys[NMS_CAP];
    __shared__ int hist[4096];
    __shared__ float bx1[NMS_CAP], by1[NMS_CAP], bx2[NMS_CAP], by2[NMS_CAP], barea[NMS_CAP];
    __shared__ unsigned char supp[NMS_CAP];
    __shared__ int s_cnt, s_digit, s_need, s_keepn;
    __shared__ int s_keep[FRP_MAX_FACES_CAP];
    __shared__ int wsum[NT / 64];

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    int A;
    const FrameView fv = frame_view(p, b, A);
    const float lt = p.logit_thresh;

    // ---- pass 0: count candidates
    if (tid == 0) { s_cnt = 0; s_keepn = 0; }
    __syncthreads();
    const _Float16* dense = p.logits + (long)b * A;     // written by gather_logits_kernel
    int local = 0;
    for (int i = tid; i < A; i += NT) local += is_candidate(dense[i], lt) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((tid & 63) == 0) atomicAdd(&s_cnt, local);
    __syncthreads();
    const int total = s_cnt;
    __syncthreads();

    // ---- exact radix select of the CAP-th largest key when too many candidates
    unsigned long long T = 0ull;   // keep keys >= T
    if (total > NMS_CAP) {
        unsigned long long prefix = 0ull;
        int need = NMS_CAP;
        for (int pass = 0; pass < 3; ++pass) {
            const int shift = 24 - 12 * pass;
            for (int i = tid; i < 4096; i += NT) hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < A; i += NT) {
                const _Float16 lg = dense[i];
                if (is_candidate(lg, lt)) {
                    const unsigned long long k = ((unsigned long long)sortable16(__builtin_bit_cast(unsigned short, lg)) << 20) |
                                                 (unsigned long long)(0xFFFFF - i);
                    if (pass == 0 || (k >> (shift + 12)) == prefix) atomicAdd(&hist[(int)((k >> shift) & 0xFFF)], 1);
                }
            }
            __syncthreads();
            // suffix counts: thread t owns bins 4t..4t+3; find d with cnt(>d) < need <= cnt(>=d)
            const int b0 = tid * 4;
            const int h0 = hist[b0], h1 = hist[b0 + 1], h2 = hist[b0 + 2], h3 = hist[b0 + 3];
            const int mine = h0 + h1 + h2 + h3;
            // inclusive suffix sum over threads (higher tid = higher bins)
            int v = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int n = __shfl_down(v, o);
                if ((tid & 63) + o < 64) v += n;
            }
            if ((tid & 63) == 0) wsum[tid >> 6] = v;
            __syncthreads();
            int above_waves = 0;
            for (int w = (tid >> 6) + 1; w < NT / 64; ++w) above_waves += wsum[w];
            const int suffix_incl = v + above_waves;   // count in bins >= 4t
            const int above = suffix_incl - mine;      // count in bins > 4t+3
            // scan my 4 bins from high to low
            int c = above;
            if (c < need && need <= c + h3) { s_digit = b0 + 3; s_need = need - c; }
            c += h3;
            if (c < need && need <= c + h2) { s_digit = b0 + 2; s_need = need - c; }
            c += h2;
            if (c < need && need <= c + h1) { s_digit = b0 + 1; s_need = need - c; }
            c += h1;
            if (c < need && need <= c + h0) { s_digit = b0; s_need = need - c; }
            __syncthreads();
            prefix = (prefix << 12) | (unsigned long long)s_digit;
            need = s_need;
            __syncthreads();
        }
        T = prefix;
    }

    // ---- gather candidates
    if (tid == 0) s_cnt = 0;
    for (int i = tid; i < NMS_CAP; i += NT) { keys[i] = 0ull; supp[i] = 0; }
    __syncthreads();
    for (int i = tid; i < A; i += NT) {
        const _Float16 lg = dense[i];
        if (is_candidate(lg, lt)) {
            const unsigned long long k = ((unsigned long long)sortable16(__builtin_bit_cast(unsigned short, lg)) << 20) |
                                         (unsigned long long)(0xFFFFF - i);
            if (k >= T) {
                const int s = atomicAdd(&s_cnt, 1);
                if (s < NMS_CAP) keys[s] = k | (1ull << 40);   // bit 40 marks a real entry (> any padding 0)
            }
        }
    }
    __syncthreads();
    const int n = s_cnt < NMS_CAP ? s_cnt : NMS_CAP;

    // ---- bitonic sort, descending
    for (int k2 = 2; k2 <= NMS_CAP; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const int i = tid;
            const int ixj = i ^ j;
            if (ixj > i) {
                const unsigned long long a = keys[i], c = keys[ixj];
                const bool desc = (i & k2) == 0;
                if (desc ? (a < c) : (a > c)) { keys[i] = c; keys[ixj] = a; }
            }
            __syncthreads();
        }
    }

    // ---- decode boxes of the sorted candidates
    if (tid < n) {
        const int idx = 0xFFFFF - (int)(keys[tid] & 0xFFFFF);
        int lv, x, y;
        const _Float16* q = anchor_ptr(fv, idx, lv, x, y);
        const float s = (float)(8 << lv);
        const float cx = (float)(x * (8 << lv)), cy = (float)(y * (8 << lv));
        const float x1 = cx - (float)q[1] * s, y1 = cy - (float)q[2] * s;
        const float x2 = cx + (float)q[3] * s, y2 = cy + (float)q[4] * s;
        bx1[tid] = x1; by1[tid] = y1; bx2[tid] = x2; by2[tid] = y2;
        barea[tid] = (x2 - x1 + 1.0f) * (y2 - y1 + 1.0f);
    }
    __syncthreads();

    // ---- greedy NMS in key order; barrier only when a box is kept
    const int K = p.max_faces;
    for (int i = 0; i < n; ++i) {
        if (supp[i]) continue;            // uniform: written before the last barrier
        if (tid == 0) s_keep[s_keepn] = i;
        const int kept = s_keepn + 1;     // uniform read (updated after the barrier below)
        if (kept >= K) { __syncthreads(); if (tid == 0) s_keepn = kept; __syncthreads(); break; }
        const float ix1 = bx1[i], iy1 = by1[i], ix2 = bx2[i], iy2 = by2[i], ia = barea[i];
        for (int j = i + 1 + tid; j < n; j += NT) {
            const float xx1 = fmaxf(ix1, bx1[j]), yy1 = fmaxf(iy1, by1[j]);
            const float xx2 = fminf(ix2, bx2[j]), yy2 = fminf(iy2, by2[j]);
            const float w = fmaxf(0.0f, xx2 - xx1 + 1.0f), h = fmaxf(0.0f, yy2 - yy1 + 1.0f);
            const float inter = w * h;
            const float ovr = inter / ((ia + barea[j]) - inter);
            if (ovr > p.nms_iou) supp[j] = 1;
        }
        __syncthreads();
        if (tid == 0) s_keepn = kept;
        __syncthreads();
    }
    __syncthreads();
    const int kept = s_keepn;

    // ---- outputs
    if (tid < K) {
        float* ob = p.boxes + ((long)b * K + tid) * 4;
        float* ok = p.kps + ((long)b * K + tid) * 10;
        if (tid < kept) {
            const int c = s_keep[tid];
            const int idx = 0xFFFFF - (int)(keys[c] & 0xFFFFF);
            int lv, x, y;
            const _Float16* q = anchor_ptr(fv, idx, lv, x, y);
            const float s = (float)(8 << lv);
            const float cx = (float)(x * (8 << lv)), cy = (float)(y * (8 << lv));
            ob[0] = bx1[c]; ob[1] = by1[c]; ob[2] = bx2[c]; ob[3] = by2[c];
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                ok[2 * t] = cx + (float)q[5 + 2 * t] * s;
                ok[2 * t + 1] = cy + (float)q[6 + 2 * t] * s;
            }
            p.scores[(long)b * K + tid] = 1.0f / (1.0f + expf(-(float)q[0]));
            if (p.anchor) p.anchor[(long)b * K + tid] = idx;
        } else {
            ob[0] = ob[1] = ob[2] = ob[3] = 0.f;
            for (int t = 0; t < 10; ++t) ok[t] = 0.f;
            p.scores[(long)b * K + tid] = 0.f;
            if (p.anchor) p.anchor[(long)b * K + tid] = -1;
        }
    }
    if (tid == 0) p.counts[b] = kept;
}

// Bilinear down/up-scale of u8 frames for the multi-scale pyramid (BASELINE config 4).  Pixel centres:
// src = (dst + 0.5) * (S / D) - 0.5, clamped to the image; fp32 lerp in x then y with one rounding per
// operation (this file is built with -ffp-contract=off), result floor(v + 0.5) -> u8.
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ src, int B, int H, int W,
                                                        uint8_t* __restrict__ dst, int Hs, int Ws, float ry, float rx) {
    const long total = (long)B * Hs * Ws;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Ws);
        const long r = i / Ws;
        const int y = (int)(r % Hs);
        const int b = (int)(r / Hs);
        float sy = ((float)y + 0.5f) * ry - 0.5f, sx = ((float)x + 0.5f) * rx - 0.5f;
        sy = fminf(fmaxf(sy, 0.0f), (float)(H - 1));
        sx = fminf(fmaxf(sx, 0.0f), (float)(W - 1));
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
        const float fy = sy - (float)y0, fx = sx - (float)x0;
        const uint8_t* p00 = src + (((long)b * H + y0) * W + x0) * 3;
        const uint8_t* p01 = src + (((long)b * H + y0) * W + x1) * 3;
        const uint8_t* p10 = src + (((long)b * H + y1) * W + x0) * 3;
        const uint8_t* p11 = src + (((long)b * H + y1) * W + x1) * 3;
        uint8_t* o = dst + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float top = (float)p00[c] * (1.0f - fx) + (float)p01[c] * fx;
            const float bot = (float)p10[c] * (1.0f - fx) + (float)p11[c] * fx;
            const float v = top * (1.0f - fy) + bot * fy;
            o[c] = (uint8_t)fminf(fmaxf(floorf(v + 0.5f), 0.0f), 255.0f);
        }
    }
}

hipError_t launch_resize_u8(const uint8_t* src, int B, int H, int W, uint8_t* dst, int Hs, int Ws, hipStream_t stream) {
    if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || Hs <= 0 || Ws <= 0) return hipErrorInvalidValue;
    const long total = (long)B * Hs * Ws;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(resize_u8_kernel, dim3(grid), dim3(256), 0, stream, src, B, H, W, dst, Hs, Ws,
                       (float)H / (float)Hs, (float)W / (float)Ws);
    return hipGetLastError();
}

hipError_t launch_decode_nms(const DecodeParams& p, hipStream_t stream) {
    if (p.B <= 0 || p.max_faces <= 0 || p.max_faces > FRP_MAX_FACES_CAP) return hipErrorInvalidValue;
    long A = 0;
    for (int l = 0; l < 3; ++l) {
        if (!p.head[l] || p.hl[l] <= 0 || p.wl[l] <= 0) return hipErrorInvalidValue;
        A += (long)p.hl[l] * p.wl[l] * 2;
    }
    if (A > 0xFFFFF) return hipErrorInvalidValue;   // 20-bit anchor index in the sort key
    if (!p.boxes || !p.kps || !p.scores || !p.counts || !p.logits) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_logits_kernel, dim3((unsigned)((A + 255) / 256), p.B), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(decode_nms_kernel, dim3(p.B), dim3(NT), 0, stream, p);
    return hipGetLastError();
}

}  // namespace frp
