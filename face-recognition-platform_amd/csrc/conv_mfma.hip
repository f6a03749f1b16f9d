// K2: implicit-GEMM convolution (3x3 / 1x1, stride 1 / 2) on CDNA4 matrix cores.
//
// Replaces the arithmetic the reference delegates to dlib / onnxruntime behind
// face_recognition.face_locations / face_encodings (backend/app/routes/camera.py:232,237,
// backend/app/services/face_service.py:156,179) for the RetinaFace-style detector and
// the ArcFace IResNet embedder (layer tables: ../netspec.py).
//
// Data layout (HBM): activations NHWC fp16, weights [Cout][kh][kw][Cin] fp16 (K-major
// for both MFMA operands), bias / PReLU slope fp32, accumulation fp32.
//
// GEMM view: D[cout][pixel] = sum_k Wt[cout][k] * X[pixel][k], k = (kh, kw, cin).  The
// weights are the MFMA A operand and the pixels the B operand, so an accumulator lane
// owns ONE output pixel and 4 consecutive output channels per register group: the
// epilogue (bias / 9-class border bias, residual, ReLU / PReLU) stores 8-byte fp16x4
// runs into the NHWC row of that pixel.
//
// Tile: TP pixels x TC couts x 64 k per step, 4 waves (one per SIMD), v_mfma_f32_32x32x16_f16.
// Staging: 16-byte global loads -> registers -> ds_write_b128 into a double-buffered,
// XOR-swizzled LDS image (chunk ^= (row>>1)&7: conflict-free for ds_read_b128 fragment
// reads of 128-byte rows), loads for step k+1 issued before the MFMAs of step k.
// Zero padding and ragged tiles are predicated in the loader (no padded copies in HBM).
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous range of
// tiles and the cout tiles of one pixel tile run back to back on the same L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frp_internal.h"

namespace frp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// select-to-zero as per-dword AND (a uint4 ?: makes hipcc round-trip through scratch)
__device__ __forceinline__ uint4 mask4(uint4 v, bool ok) {
    const unsigned m = ok ? 0xffffffffu : 0u;
    return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}

template <int TP, int TC, int WP, int WC, bool SMALL>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XB = TP * 128;          // bytes of one X stage
    constexpr int WB = TC * 128;
    constexpr int STAGE = XB + WB;
    constexpr int XR = TP / 32;           // rows per thread per stage
    constexpr int WR = TC / 32;
    constexpr int MP = TP / WP / 32;      // MFMA tiles per wave (pixels)
    constexpr int MC = TC / WC / 32;      // MFMA tiles per wave (couts)
    static_assert(WP * WC == 4, "4 waves");

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    // XCD-aware bijective remap: blocks b, b+8, ... share an XCD -> give them consecutive tiles
    int wid;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, j = b >> 3;
        wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int ptile = wid / p.n_ctiles;
    const int ctile = wid - ptile * p.n_ctiles;
    const int m0 = ptile * TP;
    const int c0 = ctile * TC;

    // ---------------- loader state
    const int lrow = t >> 3, lchunk = t & 7;
    long xbase[XR];
    int iy0[XR], ix0[XR];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (m < p.M) {
            const int n = m / HoWo;
            const int rem = m - n * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            iy0[i] = oy * p.stride - p.pad;
            ix0[i] = ox * p.stride - p.pad;
            xbase[i] = (((long)n * p.H + iy0[i]) * p.W + ix0[i]) * p.Cin;
        } else {
            iy0[i] = -(1 << 24);
            ix0[i] = -(1 << 24);
            xbase[i] = 0;
        }
    }
    long wbase[WR];
    bool wvalid[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int co = c0 + lrow + 32 * i;
        wvalid[i] = co < p.Cout;
        wbase[i] = (long)co * p.Ktot;
    }

    uint4 xreg[XR], wreg[WR];
    bool xok[XR], wok[WR];   // zero-fill masks, applied at ds_write time so the loads stay in flight

    // aligned path: uniform (kh, kw, cblk) walk
    int kh = 0, kw = 0, cb = 0;
    const int cpt = p.Cin >> 6;   // 64-channel blocks per tap (aligned path)

    auto load_stage = [&](int ks) {
        if constexpr (!SMALL) {
            const int off = (kh * p.W + kw) * p.Cin + (cb << 6) + (lchunk << 3);
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const bool ok = (unsigned)(iy0[i] + kh) < (unsigned)p.H && (unsigned)(ix0[i] + kw) < (unsigned)p.W;
                // unconditional load from a safe address + select: no divergent branch per load
                xreg[i] = *reinterpret_cast<const uint4*>(p.x + (ok ? xbase[i] + off : 0L));
                xok[i] = ok;
            }
            const int kg = (ks << 6) + (lchunk << 3);
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                wreg[i] = *reinterpret_cast<const uint4*>(p.w + (wvalid[i] ? wbase[i] + kg : 0L));
                wok[i] = wvalid[i];
            }
            if (++cb == cpt) { cb = 0; if (++kw == p.KS) { kw = 0; ++kh; } }
        } else {
            const int kg = (ks << 6) + (lchunk << 3);
            const bool kok = kg < p.Ktot;
            const int tap = kg >> p.cin_shift;
            const int ci = kg & (p.Cin - 1);
            const int tkh = (p.KS == 3) ? tap / 3 : 0;
            const int tkw = tap - tkh * p.KS;
            const int off = (tkh * p.W + tkw) * p.Cin + ci;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const bool ok = kok && (unsigned)(iy0[i] + tkh) < (unsigned)p.H && (unsigned)(ix0[i] + tkw) < (unsigned)p.W;
                xreg[i] = *reinterpret_cast<const uint4*>(p.x + (ok ? xbase[i] + off : 0L));
                xok[i] = ok;
            }
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                const bool ok = wvalid[i] && kok;
                wreg[i] = *reinterpret_cast<const uint4*>(p.w + (ok ? wbase[i] + kg : 0L));
                wok[i] = ok;
            }
        }
    };
    auto store_stage = [&](int buf) {
        unsigned char* xs = smem + buf * STAGE;
        unsigned char* ws = xs + XB;
#pragma unroll
        for (int i = 0; i < XR; ++i)
            *reinterpret_cast<uint4*>(xs + lds_off(lrow + 32 * i, lchunk)) = mask4(xreg[i], xok[i]);
#pragma unroll
        for (int i = 0; i < WR; ++i)
            *reinterpret_cast<uint4*>(ws + lds_off(lrow + 32 * i, lchunk)) = mask4(wreg[i], wok[i]);
    };

    floatx16 acc[MP][MC];
#pragma unroll
    for (int i = 0; i < MP; ++i)
#pragma unroll
        for (int j = 0; j < MC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wave_p = wave / WC, wave_c = wave - wave_p * WC;
    const int prow0 = wave_p * (TP / WP), crow0 = wave_c * (TC / WC);
    const int fr = lane & 31, fh = lane >> 5;

    load_stage(0);
    store_stage(0);
    __syncthreads();

    const int nk = p.nk;
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) load_stage(ks + 1);
        const unsigned char* xs = smem + cur * STAGE;
        const unsigned char* ws = xs + XB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            half8 bf[MP], af[MC];
#pragma unroll
            for (int i = 0; i < MP; ++i)
                bf[i] = *reinterpret_cast<const half8*>(xs + lds_off(prow0 + i * 32 + fr, 2 * kk + fh));
#pragma unroll
            for (int j = 0; j < MC; ++j)
                af[j] = *reinterpret_cast<const half8*>(ws + lds_off(crow0 + j * 32 + fr, 2 * kk + fh));
#pragma unroll
            for (int i = 0; i < MP; ++i)
#pragma unroll
                for (int j = 0; j < MC; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j], bf[i], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < nk) store_stage(cur ^ 1);
        __syncthreads();
    }

    // ---------------- epilogue
    const bool border = p.flags & FRP_FLAG_BORDER_BIAS;
    const bool out32 = p.flags & FRP_FLAG_OUT_F32;
    const bool up2 = p.flags & FRP_FLAG_RES_UP2;
#pragma unroll
    for (int i = 0; i < MP; ++i) {
        const int m = m0 + prow0 + i * 32 + fr;
        if (m >= p.M) continue;
        int cls = 0;
        long ridx = (long)m * p.Cout;
        if (border || up2) {
            const int n = m / HoWo;
            const int rem = m - n * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            if (border) cls = ((oy == 0) ? 0 : (oy == p.Ho - 1) ? 2 : 1) * 3 + ((ox == 0) ? 0 : (ox == p.Wo - 1) ? 2 : 1);
            if (up2) ridx = (((long)n * p.Hr + (oy >> 1)) * p.Wr + (ox >> 1)) * p.Cout;
        }
        const float* bias = p.bias + (long)cls * p.Cout;
        const long obase = (long)m * p.Cout;
#pragma unroll
        for (int j = 0; j < MC; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = c0 + crow0 + j * 32 + 8 * g + 4 * fh;
                if (co >= p.Cout) continue;
                floatx4 v;
                const floatx4 b4 = *reinterpret_cast<const floatx4*>(bias + co);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + b4[e];
                if (p.res) {
                    const half4 r4 = *reinterpret_cast<const half4*>(p.res + ridx + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
                }
                if (p.act == FRP_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                } else if (p.act == FRP_ACT_PRELU) {
                    const floatx4 s4 = *reinterpret_cast<const floatx4*>(p.slope + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * s4[e];
                }
                if (out32) {
                    *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(p.out) + obase + co) = v;
                } else {
                    half4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[e];
                    *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(p.out) + obase + co) = h;
                }
            }
        }
    }
}

template <int TP, int TC, int WP, int WC, bool SMALL>
static hipError_t launch_cfg(const ConvParams& p0, hipStream_t stream) {
    ConvParams p = p0;
    p.n_ptiles = (p.M + TP - 1) / TP;
    p.n_ctiles = (p.Cout + TC - 1) / TC;
    const int lds = 2 * (TP + TC) * 128;
    static bool attr_set[64] = {};
    auto kern = conv_mfma_kernel<TP, TC, WP, WC, SMALL>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const long nwg = (long)p.n_ptiles * p.n_ctiles;
    if (nwg <= 0 || nwg > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, stream, p);
    return hipGetLastError();
}

// Host-side shape checks + tile selection.  Returns hipErrorInvalidValue on a shape the
// kernel does not cover instead of launching (a faulting kernel can take the node down).
hipError_t launch_conv(const ConvParams& in, hipStream_t stream) {
    ConvParams p = in;
    if (!(p.KS == 1 || p.KS == 3) || !(p.stride == 1 || p.stride == 2)) return hipErrorInvalidValue;
    if (p.Cin < 8 || (p.Cin & 7) || (p.Cout & 3) || p.N <= 0 || p.H <= 0 || p.W <= 0) return hipErrorInvalidValue;
    p.pad = p.KS / 2;
    p.Ho = (p.H + 2 * p.pad - p.KS) / p.stride + 1;
    p.Wo = (p.W + 2 * p.pad - p.KS) / p.stride + 1;
    const long M = (long)p.N * p.Ho * p.Wo;
    if (M <= 0 || M > 0x7fffffffL) return hipErrorInvalidValue;
    p.M = (int)M;
    p.Ktot = p.KS * p.KS * p.Cin;
    p.nk = (p.Ktot + 63) / 64;
    const bool small = (p.Cin & 63) != 0;
    p.cin_shift = 0;
    if (small) {
        if (p.Cin & (p.Cin - 1)) return hipErrorInvalidValue;   // small path needs power-of-two Cin
        while ((1 << p.cin_shift) < p.Cin) ++p.cin_shift;
    }
    if ((p.flags & FRP_FLAG_BORDER_BIAS) && !(p.KS == 3 && p.stride == 1 && p.Ho >= 2 && p.Wo >= 2))
        return hipErrorInvalidValue;
    if ((p.flags & FRP_FLAG_RES_UP2) && (!p.res || p.Hr * 2 != p.Ho || p.Wr * 2 != p.Wo)) return hipErrorInvalidValue;
    if (!p.x || !p.w || !p.bias || !p.out) return hipErrorInvalidValue;
    if (p.act == FRP_ACT_PRELU && !p.slope) return hipErrorInvalidValue;
    if (p.Cout >= 128) {
        return small ? launch_cfg<128, 128, 2, 2, true>(p, stream) : launch_cfg<128, 128, 2, 2, false>(p, stream);
    } else if (p.Cout > 32) {
        return small ? launch_cfg<256, 64, 4, 1, true>(p, stream) : launch_cfg<256, 64, 4, 1, false>(p, stream);
    } else {
        return small ? launch_cfg<256, 32, 4, 1, true>(p, stream) : launch_cfg<256, 32, 4, 1, false>(p, stream);
    }
}

}  // namespace frp
