// K1+K2 fused for the detector stem: u8 frames -> 3x3 stride-2 conv (3 -> 32) + bias + ReLU,
// fp16 NHWC out, without materialising the normalised NHWC8 blob (1.07 GB per 32 x 1080p).
//
// Replaces cv2.cvtColor(frame, COLOR_BGR2RGB) (backend/app/routes/camera.py:225), the SCRFD
// input blob ((rgb - 127.5) / 128 on a zero-u8 letterbox canvas) and the first layer of the
// detector behind face_recognition.face_locations (camera.py:232, face_service.py:156).
//
// One workgroup = 4 output rows x 64 output columns (4 waves, one output row each).  The
// 9 x 129-pixel input patch is loaded with coalesced byte loads, normalised once and kept
// in LDS as fp16 in RGB order (the BGR swap happens at staging time).  For a fixed kernel row
// kh the 9 values (kw, c) of an output pixel are 9 CONSECUTIVE halfwords of a patch row, so K
// is laid out as 3 steps of 16 (k' = kw*3 + c, 9 used, weights zero beyond): one
// v_mfma_f32_32x32x16_f16 per kernel row with the folded weights as the A operand (three
// register fragments per lane, loaded once) and 4 dword LDS reads per lane as the B operand
// (lane stride 12 B: conflict-free; lanes of the upper half-wave read elements 8..15, of which
// only element 8 meets a non-zero weight).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frp_internal.h"

namespace frp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define ST_ROWS 4
#define ST_COLS 64
#define ST_PR (2 * ST_ROWS + 1)          // patch rows
#define ST_PE ((2 * ST_COLS + 1) * 3)    // patch elements per row (387)
#define ST_PITCH 392                     // halfs per patch row

// exchange between the half-waves so that each lane ends up with 16 contiguous output bytes
// (lane<32: couts 16q..16q+7, lane>=32: couts 16q+8..16q+15 of its pixel) instead of two 8-byte runs
__device__ __forceinline__ void swap_halves(unsigned& a, unsigned& b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

__device__ __forceinline__ floatx16 mfma16(half8 a, half8 b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256) void stem_u8_kernel(StemParams p) {
    __shared__ __attribute__((aligned(16))) _Float16 patch[ST_PR * ST_PITCH + 16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tiles_x = (p.Wo + ST_COLS - 1) / ST_COLS;
    const int tiles_y = (p.Ho + ST_ROWS - 1) / ST_ROWS;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int oy0 = ty * ST_ROWS, ox0 = tx * ST_COLS;
    const uint8_t* frame = p.frames + (long)b * p.frame_stride;

    // ---- stage the normalised patch: rows 2*oy0-1 .. +8, columns 2*ox0-1 .. +128
    const int ix_first = 2 * ox0 - 1;
#pragma unroll
    for (int pr = 0; pr < ST_PR; ++pr) {
        const int iy = 2 * oy0 - 1 + pr;
        const bool yin = (unsigned)iy < (unsigned)p.Hc;      // inside the canvas (else conv zero padding)
        const bool yimg = (unsigned)iy < (unsigned)p.H;      // inside the frame (else letterbox u8 0)
        const uint8_t* row = frame + (long)(yimg ? iy : 0) * p.row_stride;
        for (int e = t; e < ST_PE; e += 256) {
            const int px = e / 3;
            const int ix = ix_first + px;
            float v = 0.f;                                    // conv padding: 0 in the normalised domain
            if (yin && (unsigned)ix < (unsigned)p.Wc) {
                float u = 0.f;                                // letterbox canvas: u8 zero
                if (yimg && ix < p.W) u = (float)row[(long)ix * 3 + (e - px * 3)];
                v = (u - 127.5f) * (1.0f / 128.0f);
            }
            const int c = e - px * 3;
            patch[pr * ST_PITCH + px * 3 + (p.rgb_in ? c : 2 - c)] = (_Float16)v;
        }
    }
    // the windows of the last pixels run up to 8 halfwords past a row / the patch: keep them finite
    for (int e = t; e < ST_PR * (ST_PITCH - ST_PE) + 16; e += 256) {
        if (e < ST_PR * (ST_PITCH - ST_PE)) {
            const int pr = e / (ST_PITCH - ST_PE), off = e - pr * (ST_PITCH - ST_PE);
            patch[pr * ST_PITCH + ST_PE + off] = (_Float16)0.f;
        } else {
            patch[ST_PR * ST_PITCH + (e - ST_PR * (ST_PITCH - ST_PE))] = (_Float16)0.f;
        }
    }

    // ---- weights: A fragments of step kh (cout row = lane&31, k' = 8*(lane>>5) + j = kw*3 + c, RGB
    // order; zero for k' >= 9).  wfold is [32][3][3][8] fp16.
    half8 wa[3];
    {
        const int co = lane & 31, h = lane >> 5;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * h + j;
                _Float16 w = (_Float16)0.f;
                if (k < 9) {
                    const int kw = k / 3, c = k - kw * 3;
                    w = p.w[((co * 3 + kh) * 3 + kw) * 8 + c];
                }
                wa[kh][j] = w;
            }
    }
    __syncthreads();

    const int oy = oy0 + wave;
    const int r = lane & 31, h = lane >> 5;
    const floatx4 bias4[4] = {*reinterpret_cast<const floatx4*>(p.bias + 4 * h), *reinterpret_cast<const floatx4*>(p.bias + 8 + 4 * h),
                              *reinterpret_cast<const floatx4*>(p.bias + 16 + 4 * h), *reinterpret_cast<const floatx4*>(p.bias + 24 + 4 * h)};
#pragma unroll
    for (int half_tile = 0; half_tile < ST_COLS / 32; ++half_tile) {
        const int oxl = half_tile * 32 + r;
        // window start of this pixel (+8 elements for the upper half-wave); 12-byte lane stride
        const unsigned* win = reinterpret_cast<const unsigned*>(patch + (2 * wave) * ST_PITCH + 6 * oxl + 8 * h);
        floatx16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            union { unsigned u[4]; half8 v; } bf;
#pragma unroll
            for (int q = 0; q < 4; ++q) bf.u[q] = win[kh * (ST_PITCH / 2) + q];
            acc = mfma16(wa[kh], bf.v, acc);
        }
        const int ox = ox0 + oxl;
        // (all 64 lanes take part in the half-wave exchange; only the stores are predicated)
        union { half4 v; unsigned u[2]; } pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[g].v[e] = (_Float16)fmaxf(acc[4 * g + e] + bias4[g][e], 0.f);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            swap_halves(pk[2 * q].u[0], pk[2 * q + 1].u[0]);
            swap_halves(pk[2 * q].u[1], pk[2 * q + 1].u[1]);
        }
        if (oy < p.Ho && ox < p.Wo) {
            _Float16* o = p.out + (((long)b * p.Ho + oy) * p.Wo + ox) * 32;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                *reinterpret_cast<uint4*>(o + 16 * q + 8 * h) = make_uint4(pk[2 * q].u[0], pk[2 * q].u[1], pk[2 * q + 1].u[0], pk[2 * q + 1].u[1]);
        }
    }
}

hipError_t launch_stem_u8(const StemParams& p, hipStream_t stream) {
    if (!p.frames || !p.w || !p.bias || !p.out || p.B <= 0 || p.H <= 0 || p.W <= 0 || p.Hc < p.H || p.Wc < p.W ||
        (p.Hc & 1) || (p.Wc & 1) || p.Ho != p.Hc / 2 || p.Wo != p.Wc / 2)
        return hipErrorInvalidValue;
    const long tiles = (long)p.B * ((p.Ho + ST_ROWS - 1) / ST_ROWS) * ((p.Wo + ST_COLS - 1) / ST_COLS);
    if (tiles <= 0 || tiles > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stem_u8_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace frp
