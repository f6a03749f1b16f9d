// HBM-bound helper kernels of the hot path (all coalesced 16-byte stores, fp16 NHWC8 blobs).
//
// K1 preprocess   : replaces cv2.cvtColor(frame, COLOR_BGR2RGB) (backend/app/routes/camera.py:225)
//                   + the detector input blob (insightface scrfd: zero canvas, (rgb-127.5)/128)
// K4 align        : replaces the landmark alignment inside face_recognition.face_encodings
//                   (camera.py:237, face_service.py:179) with the ArcFace 5-point similarity warp
// K5 tail l2norm  : unit-normalises the FC output (embedding)
// gallery upload  : replaces np.array([ENCODINGS[t] ...]) rebuilt per call (face_service.py:409)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frp_internal.h"

namespace frp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------ K1
// one thread per canvas pixel: 3 u8 in, 16 bytes out.  rgb_in: input already RGB.
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ src, int B, int H, int W,
                                                         long row_stride, long frame_stride,
                                                         _Float16* __restrict__ out, int Hc, int Wc, int rgb_in) {
    const long total = (long)B * Hc * Wc;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wc);
        const long r = i / Wc;
        const int y = (int)(r % Hc);
        const int b = (int)(r / Hc);
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;   // letterbox canvas is u8 zero
        if (y < H && x < W) {
            const uint8_t* p = src + b * frame_stride + y * row_stride + 3L * x;
            c0 = (float)p[0]; c1 = (float)p[1]; c2 = (float)p[2];
        }
        const float r_ = rgb_in ? c0 : c2, g_ = c1, b_ = rgb_in ? c2 : c0;
        half8 v;
        v[0] = (_Float16)((r_ - 127.5f) * (1.0f / 128.0f));
        v[1] = (_Float16)((g_ - 127.5f) * (1.0f / 128.0f));
        v[2] = (_Float16)((b_ - 127.5f) * (1.0f / 128.0f));
        v[3] = v[4] = v[5] = v[6] = v[7] = (_Float16)0.f;
        *reinterpret_cast<half8*>(out + i * 8) = v;
    }
}

hipError_t launch_preprocess(const uint8_t* bgr, int B, int H, int W, long row_stride, long frame_stride,
                             _Float16* out, int Hc, int Wc, int rgb_in, hipStream_t stream) {
    if (!bgr || !out || B <= 0 || H <= 0 || W <= 0 || Hc < H || Wc < W) return hipErrorInvalidValue;
    const long total = (long)B * Hc * Wc;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid), dim3(256), 0, stream, bgr, B, H, W, row_stride, frame_stride,
                       out, Hc, Wc, rgb_in);
    return hipGetLastError();
}

// ------------------------------------------------------------------ face list compaction
// counts[B] -> face_slot[n] = b*max_faces + k (frame-major), n_faces.  One block.
__global__ __launch_bounds__(256) void compact_faces_kernel(const int32_t* __restrict__ counts, int B, int max_faces,
                                                            int32_t* __restrict__ face_slot, int32_t* __restrict__ n_faces) {
    __shared__ int offs[1024 + 1];
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < B; ++b) { offs[b] = acc; acc += counts[b]; }
        offs[B] = acc;
        *n_faces = acc;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int c = counts[b], o = offs[b];
        for (int k = 0; k < c; ++k) face_slot[o + k] = b * max_faces + k;
    }
}

hipError_t launch_compact_faces(const int32_t* counts, int B, int max_faces, int32_t* face_slot, int32_t* n_faces,
                                hipStream_t stream) {
    if (B <= 0 || B > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(compact_faces_kernel, dim3(1), dim3(256), 0, stream, counts, B, max_faces, face_slot, n_faces);
    return hipGetLastError();
}

// ------------------------------------------------------------------ K4
// ArcFace 112x112 template (insightface utils/face_align.py arcface_dst)
__constant__ float kTemplate[10] = {38.2946f, 51.6963f, 73.5318f, 51.5014f, 56.0252f, 71.7366f,
                                    41.5493f, 92.3655f, 70.7299f, 92.2041f};

// ALIGN_SPLIT workgroups per face (16 chip rows each: 320 faces alone would put 5 waves on a CU and
// leave the gather latency-bound).  Least-squares similarity (closed form of the 2-D Umeyama
// problem: scaled rotation [[a,-b],[b,a]] + t) in double, inverse map per output pixel in
// fp32, float bilinear with constant-0 border, RGB (x-127.5)/127.5 -> fp16 NHWC8.
#define ALIGN_SPLIT 7
__global__ __launch_bounds__(256) void align_kernel(AlignParams p) {
    const int f = blockIdx.x / ALIGN_SPLIT, part = blockIdx.x - f * ALIGN_SPLIT;
    if (f >= p.n_faces || (p.n_dev && f >= *p.n_dev)) return;
    const int slot = p.face_slot ? p.face_slot[f] : f;
    const int b = p.face_slot ? slot / p.max_faces : 0;
    const float* k = p.kps + (long)slot * 10;
    double msx = 0, msy = 0, mdx = 0, mdy = 0;
    for (int i = 0; i < 5; ++i) { msx += k[2 * i]; msy += k[2 * i + 1]; mdx += kTemplate[2 * i]; mdy += kTemplate[2 * i + 1]; }
    msx /= 5; msy /= 5; mdx /= 5; mdy /= 5;
    double num_a = 0, num_b = 0, den = 0;
    for (int i = 0; i < 5; ++i) {
        const double sx = k[2 * i] - msx, sy = k[2 * i + 1] - msy;
        const double dx = kTemplate[2 * i] - mdx, dy = kTemplate[2 * i + 1] - mdy;
        num_a += sx * dx + sy * dy;
        num_b += sx * dy - sy * dx;
        den += sx * sx + sy * sy;
    }
    const bool ok = den > 1e-12 && (num_a * num_a + num_b * num_b) > 1e-24;
    const double a = ok ? num_a / den : 1.0, bb = ok ? num_b / den : 0.0;
    const double tx = mdx - (a * msx - bb * msy), ty = mdy - (bb * msx + a * msy);
    // inverse: src = Minv * (dst - t)
    const double det = a * a + bb * bb;
    const float i00 = (float)(a / det), i01 = (float)(bb / det), i10 = (float)(-bb / det), i11 = (float)(a / det);
    const float itx = (float)(-(a * tx + bb * ty) / det), ity = (float)(-(-bb * tx + a * ty) / det);
    const uint8_t* img = p.frames + (long)b * p.frame_stride;
    _Float16* out = p.chips + (long)f * (FRP_CHIP_PIX * 8);
    const int c_r = p.rgb_in ? 0 : 2, c_b = p.rgb_in ? 2 : 0;
    constexpr int kPart = FRP_CHIP_PIX / ALIGN_SPLIT;      // 1792 pixels = 16 rows
    static_assert(kPart * ALIGN_SPLIT == FRP_CHIP_PIX, "chip rows must split evenly");
    for (int i = part * kPart + threadIdx.x; i < (part + 1) * kPart; i += blockDim.x) {
        const int v = i / 112, u = i - v * 112;
        const float sx = i00 * (float)u + i01 * (float)v + itx;
        const float sy = i10 * (float)u + i11 * (float)v + ity;
        const float fx0 = floorf(sx), fy0 = floorf(sy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float ax = sx - fx0, ay = sy - fy0;
        // Both taps of an image row are 6 contiguous bytes: ONE unaligned 8-byte load per row from a clamped
        // (always valid) address instead of 6 byte loads (the gather was bound by load instructions: one
        // lane-address per clock).  An out-of-image tap gets weight 0 (adds an exact +0: border value 0, same
        // bits as skipping it); loads are unconditional so both rows are in flight together.
        unsigned char tap[4][3];
        float wgt[4];
        const int bx_max = 3 * p.W - 8;                                     // last byte offset an 8-byte read may start at
        int bx = 3 * x0;
        bx = bx < 0 ? 0 : (bx > bx_max ? bx_max : bx);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int yy = y0 + dy;
            const int yc = yy < 0 ? 0 : (yy >= p.H ? p.H - 1 : yy);
            unsigned long long v;
            __builtin_memcpy(&v, img + (long)yc * p.row_stride + bx, 8);    // unaligned 64-bit global load
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int xx = x0 + dx;
                const bool inside = ok && (unsigned)xx < (unsigned)p.W && (unsigned)yy < (unsigned)p.H;
                int o = 3 * xx - bx;                                        // byte offset of this tap inside v
                o = o < 0 ? 0 : (o > 5 ? 5 : o);                            // (only out-of-image taps get clamped)
                const unsigned pix = (unsigned)(v >> (8 * o));
                tap[dy * 2 + dx][0] = (unsigned char)(pix >> (8 * c_r));
                tap[dy * 2 + dx][1] = (unsigned char)(pix >> 8);
                tap[dy * 2 + dx][2] = (unsigned char)(pix >> (8 * c_b));
                wgt[dy * 2 + dx] = inside ? (dx ? ax : 1.f - ax) * (dy ? ay : 1.f - ay) : 0.f;
            }
        }
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[0] += wgt[k] * (float)tap[k][0];
            acc[1] += wgt[k] * (float)tap[k][1];
            acc[2] += wgt[k] * (float)tap[k][2];
        }
        half8 o;
        o[0] = (_Float16)((acc[0] - 127.5f) * (1.0f / 127.5f));
        o[1] = (_Float16)((acc[1] - 127.5f) * (1.0f / 127.5f));
        o[2] = (_Float16)((acc[2] - 127.5f) * (1.0f / 127.5f));
        o[3] = o[4] = o[5] = o[6] = o[7] = (_Float16)0.f;
        *reinterpret_cast<half8*>(out + (long)i * 8) = o;
    }
}

hipError_t launch_align(const AlignParams& p, hipStream_t stream) {
    if (p.n_faces <= 0) return hipSuccess;
    if (!p.frames || !p.kps || !p.chips || p.H <= 0 || p.W < 3) return hipErrorInvalidValue;   // 8-byte row reads need 3*W >= 8
    hipLaunchKernelGGL(align_kernel, dim3(p.n_faces * ALIGN_SPLIT), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// aligned u8 BGR chips -> blob
__global__ __launch_bounds__(256) void chips_to_blob_kernel(const uint8_t* __restrict__ chips, long total,
                                                            _Float16* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const uint8_t* p = chips + 3 * i;
        half8 o;
        o[0] = (_Float16)(((float)p[2] - 127.5f) * (1.0f / 127.5f));
        o[1] = (_Float16)(((float)p[1] - 127.5f) * (1.0f / 127.5f));
        o[2] = (_Float16)(((float)p[0] - 127.5f) * (1.0f / 127.5f));
        o[3] = o[4] = o[5] = o[6] = o[7] = (_Float16)0.f;
        *reinterpret_cast<half8*>(out + i * 8) = o;
    }
}

hipError_t launch_chips_to_blob(const uint8_t* chips, int M, _Float16* out, hipStream_t stream) {
    if (M <= 0) return hipSuccess;
    const long total = (long)M * FRP_CHIP_PIX;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(chips_to_blob_kernel, dim3(grid), dim3(256), 0, stream, chips, total, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ K5 tail: one workgroup per row
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void l2norm_kernel(float* __restrict__ emb, _Float16* __restrict__ emb16, int M, int D,
                                                      const float* __restrict__ partials, int ksplit, const float* __restrict__ bias,
                                                      const int32_t* __restrict__ n_dev, int fc_ktot, int n_cu) {
    // one workgroup per row (D <= 1024: up to 4 elements per thread in registers)
    __shared__ float wsum[4];
    const int row = blockIdx.x, t = threadIdx.x;
    if (n_dev) {                      // the row count lives on the device: M was the capacity, the slabs are [ks][n][D]
        const int n = *n_dev < M ? *n_dev : M;
        if (row >= n) return;
        M = n;
        if (ksplit < 0) ksplit = conv_pick_ksplit(n, D, fc_ktot, FRP_FLAG_OUT_F32, false, n_cu);
    }
    float* e = emb + (long)row * D;
    float v[4];
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = t + q * 256;
        v[q] = 0.f;
        if (i < D) {
            if (partials) {               // split-K FC: reduce the slabs (+ folded bias) first
                float a = bias[i];
                for (int s = 0; s < ksplit; ++s) a += partials[((long)s * M + row) * D + i];
                v[q] = a;
            } else {
                v[q] = e[i];
            }
            ss += v[q] * v[q];
        }
    }
    ss = wave_sum(ss);
    if ((t & 63) == 0) wsum[t >> 6] = ss;
    __syncthreads();
    ss = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    const float inv = ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = t + q * 256;
        if (i < D) {
            const float o = v[q] * inv;
            e[i] = o;
            if (emb16) emb16[(long)row * D + i] = (_Float16)o;
        }
    }
}

hipError_t launch_l2norm(float* emb, _Float16* emb16, int M, int D, hipStream_t stream, const float* partials, int ksplit,
                         const float* bias, const int32_t* n_dev, int fc_ktot, int n_cu) {
    if (M <= 0) return hipSuccess;
    if (D <= 0 || D > 1024 || (partials && ((ksplit <= 0 && !(ksplit == -1 && n_dev && fc_ktot > 0 && n_cu > 0)) || !bias)))
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(l2norm_kernel, dim3(M), dim3(256), 0, stream, emb, emb16, M, D, partials, ksplit, bias, n_dev, fc_ktot, n_cu);
    return hipGetLastError();
}

// fp32 rows -> unit fp16 rows (gallery upload, query conversion)
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ in, _Float16* __restrict__ out,
                                                             long N, int D) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* e = in + row * D;
    float ss = 0.f;
    for (int i = lane; i < D; i += 64) ss += e[i] * e[i];
    ss = wave_sum(ss);
    const float inv = ss > 0.f ? 1.0f / sqrtf(ss) : 0.f;
    for (int i = lane; i < D; i += 64) out[row * D + i] = (_Float16)(e[i] * inv);
}

hipError_t launch_gallery_normalize(const float* in, _Float16* out, long N, int D, hipStream_t stream) {
    if (N <= 0) return hipSuccess;
    const long grid = (N + 3) / 4;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)grid), dim3(256), 0, stream, in, out, N, D);
    return hipGetLastError();
}

#ifdef FRP_LAB
// ------------------------------------------------------------------ benchmark helper
// uniform [-scale, scale) fp16 fill from a counter hash (tuning runs must use random data:
// zero operands raise the clock and flatter the kernel)
__global__ __launch_bounds__(256) void fill_random_f16_kernel(_Float16* __restrict__ p, long n, unsigned seed, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (_Float16)(((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale);
    }
}

// pure MFMA loop on register operands (no memory traffic): the practical matrix-core ceiling
// of this device at the clock it holds under load (tuning reference only)
typedef float floatx16_ __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_peak_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, int iters) {
    half8 a = *reinterpret_cast<const half8*>(src + (threadIdx.x & 63) * 8);
    half8 b = *reinterpret_cast<const half8*>(src + 512 + (threadIdx.x & 63) * 8);
    floatx16_ c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The conv k-step without memory traffic or barriers: 8 waves per CU (2 per SIMD), each a 64x64 tile = 16 MFMAs
// per step fed by READS ds_read_b128 per group of 4 MFMAs (4 = the conv kernels: one read per MFMA; 3 / 2 =
// what larger per-wave tiles would need), fragments double-buffered as in the kernels.  What this loop sustains
// is the ceiling of that wave layout on this device.
template <int READS>
__global__ __launch_bounds__(512, 2) void mfma_lds_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * 384 * 128];      // the 3-slot ring of the 256x128 tile
    for (int i = threadIdx.x; i < 3 * 384 * 128 / 16; i += 512)
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(src)[i];     // 144 KiB of random fp16
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int prow0 = (wave >> 1) * 64, crow0 = 256 + (wave & 1) * 64;
    floatx16_ acc[2][2] = {};
    half8 f[2][4];
    auto rd = [&](int stage, int kk, int S) {
        const unsigned char* base = lds + stage * (384 * 128);
#pragma unroll
        for (int q = 0; q < READS; ++q) {
            const int row = (q < 2 ? prow0 + q * 32 : crow0 + (q - 2) * 32) + fr;
            f[S][q] = *reinterpret_cast<const half8*>(base + row * 128 + (((2 * kk + fh) ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int q = READS; q < 4; ++q) f[S][q] = f[S][q - READS];                  // fewer reads: reuse a fragment
    };
    auto mm = [&](int S) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[S][2], f[S][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[S][3], f[S][0], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[S][2], f[S][1], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[S][3], f[S][1], acc[1][1], 0, 0, 0);
    };
    int stage = 0;
    for (int it = 0; it < iters; ++it) {
        rd(stage, 0, 0);
        rd(stage, 1, 1); mm(0);
        rd(stage, 2, 0); mm(1);
        rd(stage, 3, 1); mm(0);
        mm(1);
        stage = stage == 2 ? 0 : stage + 1;
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += acc[0][0][e] + acc[0][1][e] + acc[1][0][e] + acc[1][1][e];
    dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

hipError_t launch_mfma_lds(const _Float16* src, float* dst, int blocks, int reads, int iters, hipStream_t stream) {
    if (reads == 4) hipLaunchKernelGGL(mfma_lds_kernel<4>, dim3(blocks), dim3(512), 0, stream, src, dst, iters);
    else if (reads == 3) hipLaunchKernelGGL(mfma_lds_kernel<3>, dim3(blocks), dim3(512), 0, stream, src, dst, iters);
    else if (reads == 2) hipLaunchKernelGGL(mfma_lds_kernel<2>, dim3(blocks), dim3(512), 0, stream, src, dst, iters);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_mfma_peak(const _Float16* src, float* dst, int blocks, int iters, hipStream_t stream) {
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, iters);
    return hipGetLastError();
}

hipError_t launch_fill_random_f16(_Float16* p, long n, unsigned seed, float scale, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_random_f16_kernel, dim3(4096), dim3(256), 0, stream, p, n, seed, scale);
    return hipGetLastError();
}
#endif  // FRP_LAB

// Diagnostic (frp_debug_det_hashes, tools/det_hash_bisect.py): order-independent 64-bit hash of a tensor's 16-byte words, added to *slot.
__global__ __launch_bounds__(256) void tensor_hash_kernel(const uint4* __restrict__ src, long n16, unsigned long long* __restrict__ slot) {
    unsigned long long acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        unsigned long long a = ((unsigned long long)v.x << 32 | v.y) ^ 0x9e3779b97f4a7c15ull * (unsigned long long)(i + 1);
        unsigned long long b = ((unsigned long long)v.z << 32 | v.w) + 0xc2b2ae3d27d4eb4full * (unsigned long long)(i + 1);
        a ^= a >> 29; a *= 0xbf58476d1ce4e5b9ull; a ^= a >> 32;
        b ^= b >> 31; b *= 0x94d049bb133111ebull; b ^= b >> 29;
        acc += a + (b << 1 | b >> 63);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(slot, acc);
}
hipError_t launch_tensor_hash(const void* src, size_t bytes, unsigned long long* slot, hipStream_t stream) {
    const long n16 = (long)(bytes / 16);
    if (n16 <= 0) return hipSuccess;
    hipLaunchKernelGGL(tensor_hash_kernel, dim3(1024), dim3(256), 0, stream, (const uint4*)src, n16, slot);
    return hipGetLastError();
}

}  // namespace frp
