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
#include "conv_common.h"

namespace frp {

// (vector typedefs and the half-wave exchange `swap_halves` come from conv_common.h)

#define ST_ROWS 4
#define ST_COLS 64
#define ST_PR (2 * ST_ROWS + 1)          // patch rows
#define ST_PE ((2 * ST_COLS + 1) * 3)    // patch elements per row (387)
#define ST_PITCH 392                     // halfs per patch row

// bias + ReLU + fp16 pack of one accumulator quad, two elements per instruction (v_pk_add_f32, v_cvt_pk_f16_f32,
// v_pk_max_f16; relu(round16(y)) == round16(relu(y))): -> two dwords of packed halfwords
__device__ __forceinline__ void bias_relu_pack(const floatx16& acc, int g, floatx4 b, unsigned (&out)[2]) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const float2v v = float2v{acc[4 * g + 2 * h2], acc[4 * g + 2 * h2 + 1]} + float2v{b[2 * h2], b[2 * h2 + 1]};
        half2v hv = __builtin_convertvector(v, half2v);
        hv = __builtin_elementwise_max(hv, half2v{0, 0});
        out[h2] = __builtin_bit_cast(unsigned, hv);
    }
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

// ------------------------------------------------------------------------------------------------
// K1+K2+K2 fused: u8 frames -> stem1 (3x3 s2, 3->32, ReLU) -> stem2 (3x3 s2, 32->64, ReLU), fp16 NHWC64
// out.  Neither the normalised blob (1.07 GB per 32 x 1080p) nor the stem1 activation (another 1.07 GB
// written and re-read 1.4x) ever reaches HBM: 0.2 GB of frames in, 0.53 GB out.
//
// Persistent workgroups (4 waves) walk stem2 output tiles of 4 rows x 32 columns:
//   A. the 19 x 131-pixel u8 patch is normalised once into LDS (fp16, RGB order), as in stem_u8_kernel;
//   B. stem1 for the 9 x 65 halo'd tile (18 jobs of 32 pixels + one for the 65th column): one MFMA per
//      kernel row (K laid out as 3 x 16, see above),
//      bias + ReLU, positions outside the stem1 map forced to 0 (they are stem2's zero padding), result
//      to LDS as [row][column parity][column/2][32 ch] with an 80-byte pixel pitch: a stem2 tap reads
//      every second stem1 column, so with the parity split the 32 lanes of a fragment read walk
//      consecutive 80-byte slots - conflict-free for ds_read_b128 (64 B pitch: 8-way);
//   C. stem2: wave w owns cout group w&1 and output rows 2*(w>>1)+{0,1}; its 18 A fragments (9 taps x
//      2 channel halves, 72 VGPRs) are loaded once per workgroup straight from the [cout][kh][kw][cin]
//      weights (16 contiguous bytes per lane); per output row 18 x (ds_read_b128, MFMA), bias + ReLU,
//      half-wave exchange, 16-byte stores.
#define S12_R2 4
#define S12_C2 32
#define S12_S1R (2 * S12_R2 + 1)            // 9 stem1 rows
#define S12_S1C (2 * S12_C2 + 1)            // 65 stem1 columns
#define S12_PR (2 * S12_S1R + 1)            // 19 patch rows
#define S12_PE ((2 * S12_S1C + 1) * 3)      // 393 patch elements per row
#define S12_PITCH 400                       // halfs per patch row
#define S12_PATCH_HALFS (S12_PR * S12_PITCH + 16)    // the windows of the last column read up to 8 halfwords past a row
#define S12_PIX 80                          // bytes per stem1 pixel in LDS (64 used)
#define S12_S1BYTES (S12_S1R * 2 * 33 * S12_PIX)

__global__ __launch_bounds__(256, 2) void stem12_u8_kernel(Stem12Params p) {
    __shared__ __attribute__((aligned(16))) _Float16 patch[S12_PATCH_HALFS];
    __shared__ __attribute__((aligned(16))) unsigned char s1[S12_S1BYTES];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_x = (p.Wo2 + S12_C2 - 1) / S12_C2;
    const int tiles_y = (p.Ho2 + S12_R2 - 1) / S12_R2;
    const int n_tiles = p.B * tiles_y * tiles_x;

    // ---- per-workgroup constants: stem1 A fragments (k' = kw*3 + c, RGB order; see stem_u8_kernel),
    // stem2 A fragments of this wave's cout group, biases
    half8 wa[3];
    {
        const int co = r;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * h + j;
                _Float16 w = (_Float16)0.f;
                if (k < 9) {             // the patch is kept in BGR order (the camera path's memory order: its tiles are
                    const int kw = k / 3, c = k - kw * 3;   // copied straight); RGB frames are swapped at staging time, so
                    w = p.w1[((co * 3 + kh) * 3 + kw) * 8 + 2 - c];   // both orders accumulate identically
                }
                wa[kh][j] = w;
            }
    }
    const int g2 = wave & 1;
    half8 wb[18];
#pragma unroll
    for (int m = 0; m < 18; ++m)
        wb[m] = *reinterpret_cast<const half8*>(p.w2 + (((g2 * 32 + r) * 9 + (m >> 1)) * 32 + 16 * (m & 1) + 8 * h));
    floatx4 b1[4], b2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        b1[g] = *reinterpret_cast<const floatx4*>(p.bias1 + 8 * g + 4 * h);
        b2[g] = *reinterpret_cast<const floatx4*>(p.bias2 + 32 * g2 + 8 * g + 4 * h);
    }
    // every halfword of the patch is finite from the start: the windows of the last columns read row tails (and 8
    // halfwords past the last row) that meet zero weights, and 0 x NaN would not be 0
    for (int e = t; e < S12_PATCH_HALFS; e += 256) patch[e] = (_Float16)0.f;
    // ... and the fill is complete before ANY wave stages the first tile: the fill and phase A map threads to patch elements
    // differently, so without this barrier a wave that starts late zeroes halfwords another wave has already staged (round 5:
    // one 4 x 32 tile among the first tiles of the workgroups off by a few per cent once in ~350 passes of 4 x 1080p, whatever
    // kernel family followed - found by tools/det_hash_bisect.py; the cross-family test of round 4 was the first one tight
    // enough to see it)
    __syncthreads();

    // The u8 patch of a tile is fetched as aligned dwords (99 per patch row, 8 per thread) one tile AHEAD:
    // the loads are issued before phase B of the previous tile and unpacked in phase A, so their HBM
    // latency hides under a whole tile of work (19 dependent row round trips per tile without this).
    constexpr int DW_ROW = 99;                                  // dwords covering 393 bytes at any alignment
    constexpr int DW_PER_THREAD = (S12_PR * DW_ROW + 255) / 256;   // 8
    const uint8_t* const buf_lo = p.frames;
    const uint8_t* const buf_hi = p.frames + (long)p.B * p.frame_stride;
    unsigned pre[DW_PER_THREAD];
    // Interior tiles of BGR frames (84 % at 1080p): the patch is 19 x 393 bytes straight out of the frame.  Thread-invariant part of
    // the addresses: dword i of this thread is bytes 4d .. 4d+3 of patch row pr - rel[i] bytes from the patch origin in
    // the frame (one UNALIGNED dword load from a uniform base), ldso[i] bytes into `patch` (one aligned 8-byte write
    // of 4 halfwords).  Border tiles keep the per-element path (aligned dwords + shift, padding / letterbox logic).
    int rel[DW_PER_THREAD], ldso[DW_PER_THREAD];
#pragma unroll
    for (int i = 0; i < DW_PER_THREAD; ++i) {
        const int idx = t + i * 256;
        const int pr = idx / DW_ROW, d = idx - pr * DW_ROW;
        rel[i] = pr < S12_PR ? pr * (int)p.row_stride + 4 * d : -1;
        ldso[i] = (pr * S12_PITCH + 4 * d) * 2;
    }
    bool pre_fast = false;                                      // how `pre` was fetched (uniform)
    auto tile_interior = [&](int y2_0, int x2_0) {
        const int iy0 = 4 * y2_0 - 3, ix0 = 4 * x2_0 - 3;
        // (one pixel of slack on the right: the 99th dword of a row reads 3 bytes past its 393)
        return iy0 >= 0 && iy0 + S12_PR <= p.H && ix0 >= 0 && ix0 + (2 * S12_S1C + 1) + 1 <= p.W;
    };
    auto tile_coords = [&](int tile, int& b, int& y2_0, int& x2_0) {
        int q = tile;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y;
        b = q / tiles_y;
        y2_0 = ty * S12_R2;
        x2_0 = tx * S12_C2;
    };
    // byte address of patch row pr, element 0 (may lie outside the frame / the buffer: masked later)
    auto row_addr = [&](int b, int y2_0, int x2_0, int pr) -> const uint8_t* {
        int iy = 4 * y2_0 - 3 + pr;
        iy = iy < 0 ? 0 : (iy >= p.H ? p.H - 1 : iy);
        return p.frames + (long)b * p.frame_stride + (long)iy * p.row_stride + (long)(4 * x2_0 - 3) * 3;
    };
    auto prefetch = [&](int tile) {
        int b, y2_0, x2_0;
        tile_coords(tile, b, y2_0, x2_0);
        pre_fast = !p.rgb_in && tile_interior(y2_0, x2_0);
        if (pre_fast) {
            const uint8_t* base = p.frames + (long)b * p.frame_stride + (long)(4 * y2_0 - 3) * p.row_stride + (long)(4 * x2_0 - 3) * 3;
#pragma unroll
            for (int i = 0; i < DW_PER_THREAD; ++i) {
                unsigned v = 0u;
                if (rel[i] >= 0) __builtin_memcpy(&v, base + (unsigned)rel[i], 4);       // unaligned global_load_dword
                pre[i] = v;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < DW_PER_THREAD; ++i) {
            const int idx = t + i * 256;
            const int pr = idx / DW_ROW, d = idx - pr * DW_ROW;
            unsigned v = 0u;
            if (pr < S12_PR) {
                const uint8_t* a = row_addr(b, y2_0, x2_0, pr);
                const uint8_t* al = a - ((uintptr_t)a & 3) + 4 * d;        // aligned dword d of this row
                if (al >= buf_lo && al + 4 <= buf_hi) {
                    v = *reinterpret_cast<const unsigned*>(al);
                } else {                                                    // first / last bytes of the whole batch
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (al + j >= buf_lo && al + j < buf_hi) v |= (unsigned)al[j] << (8 * j);
                }
            }
            pre[i] = v;
        }
    };
    if ((int)blockIdx.x < n_tiles) prefetch(blockIdx.x);

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int b, y2_0, x2_0;
        tile_coords(tile, b, y2_0, x2_0);
        const int s1r0 = 2 * y2_0 - 1, s1c0 = 2 * x2_0 - 1;      // stem1 coordinates of the tile's first row / column
        const int iy0 = 2 * s1r0 - 1, ix0 = 2 * s1c0 - 1;        // input coordinates of the patch origin

        // ---- A: unpack the prefetched dwords into the normalised fp16 patch (RGB order).  Tiles whose
        // patch lies inside the frame (all but the border ring) skip the per-element padding logic.
        const bool interior = iy0 >= 0 && iy0 + S12_PR <= p.H && ix0 >= 0 && ix0 + (2 * S12_S1C + 1) <= p.W;
        if (pre_fast) {
            // u8 -> fp16 without a per-byte convert: bytes b0, b1 dropped into (0x64, b) pairs are the halfwords
            // 1024 + b; minus 1024, then b * 2^-7 - 255/256 = (b - 127.5) / 128, every step exact in fp16
            const half2v k1024 = {(_Float16)1024.f, (_Float16)1024.f}, kscale = {(_Float16)0.0078125f, (_Float16)0.0078125f},
                         kbias = {(_Float16)-0.99609375f, (_Float16)-0.99609375f};
#pragma unroll
            for (int i = 0; i < DW_PER_THREAD; ++i) {
                if (rel[i] >= 0) {
                    const unsigned lo = __builtin_amdgcn_perm(0x64646464u, pre[i], 0x04010400u);
                    const unsigned hi = __builtin_amdgcn_perm(0x64646464u, pre[i], 0x04030402u);
                    const half2v a = __builtin_elementwise_fma(__builtin_bit_cast(half2v, lo) - k1024, kscale, kbias);
                    const half2v c = __builtin_elementwise_fma(__builtin_bit_cast(half2v, hi) - k1024, kscale, kbias);
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(patch) + ldso[i]) =
                        make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, c));
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < DW_PER_THREAD; ++i) {
            const int idx = t + i * 256;
            const int pr = idx / DW_ROW, d = idx - pr * DW_ROW;
            if (pr < S12_PR) {
                const int sh = (int)((uintptr_t)row_addr(b, y2_0, x2_0, pr) & 3);
                const int e0 = 4 * d - sh;                            // patch-row element of byte 0 (-3 .. 392)
                int px = (e0 + 3) / 3 - 1, c = e0 - px * 3;           // floor division for e0 >= -3
                _Float16* prow = patch + pr * S12_PITCH;
                const int iy = iy0 + pr;
                const bool yin = (unsigned)iy < (unsigned)p.Hc;
                const bool yimg = (unsigned)iy < (unsigned)p.H;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e0 + j;
                    if (e >= 0 && e < S12_PE) {
                        float v = ((float)((pre[i] >> (8 * j)) & 0xffu) - 127.5f) * (1.0f / 128.0f);
                        if (!interior) {
                            const int ix = ix0 + px;
                            const bool in_canvas = yin && (unsigned)ix < (unsigned)p.Wc;
                            const bool in_frame = yimg && ix < p.W;
                            // conv padding: 0 in the normalised domain; letterbox canvas: u8 zero
                            v = in_canvas ? (in_frame ? v : (0.f - 127.5f) * (1.0f / 128.0f)) : 0.f;
                        }
                        prow[px * 3 + (p.rgb_in ? 2 - c : c)] = (_Float16)v;
                    }
                    if (++c == 3) { c = 0; ++px; }
                }
            }
        }
        // row tails read by the last windows (they meet zero weights: any FINITE value will do; the fast path leaves
        // three normalised bytes of the neighbouring pixel there, this path zeros)
        for (int e = t; e < S12_PR * (S12_PITCH - S12_PE); e += 256) {
            const int pr = e / (S12_PITCH - S12_PE);
            patch[pr * S12_PITCH + S12_PE + (e - pr * (S12_PITCH - S12_PE))] = (_Float16)0.f;
        }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < n_tiles) prefetch(tile + gridDim.x);   // lands during phases B and C

        // ---- B: stem1 on the halo'd tile -> LDS.  18 (row, 32-column group) jobs + one job whose 32
        // "pixels" are column 64 of the 9 rows (lanes r >= 9 idle along), over 4 waves.
        for (int job = wave; job < S12_S1R * 2 + 1; job += 4) {
            const bool extra = job == S12_S1R * 2;
            const int row = extra ? (r < S12_S1R ? r : 0) : job >> 1;
            const int col = extra ? 2 * S12_C2 : (job & 1) * 32 + r;   // tile-local stem1 column of this lane's pixel
            const unsigned* win = reinterpret_cast<const unsigned*>(patch + (2 * row) * S12_PITCH + 6 * col + 8 * h);
            floatx16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                union { unsigned u[4]; half8 v; } bf;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) bf.u[qq] = win[kh * (S12_PITCH / 2) + qq];
                acc = mfma16(wa[kh], bf.v, acc);
            }
            const int gy = s1r0 + row, gx = s1c0 + col;            // global stem1 coordinates
            const bool inside = (unsigned)gy < (unsigned)p.Ho1 && (unsigned)gx < (unsigned)p.Wo1;
            struct { unsigned u[2]; } pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bias_relu_pack(acc, g, b1[g], pk[g].u);
                pk[g].u[0] = inside ? pk[g].u[0] : 0u;
                pk[g].u[1] = inside ? pk[g].u[1] : 0u;
            }
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                swap_halves(pk[2 * qq].u[0], pk[2 * qq + 1].u[0]);
                swap_halves(pk[2 * qq].u[1], pk[2 * qq + 1].u[1]);
            }
            // after the exchange lane (pixel r, half h) holds couts 16*qq + 8*h .. +7 of its pixel.  But the
            // exchange pairs lane l with l+32: `inside` of both lanes is the same pixel, so the zeros agree.
            if (!extra || r < S12_S1R) {
                unsigned char* dst = s1 + ((row * 2 + (col & 1)) * 33 + (col >> 1)) * S12_PIX;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
                    *reinterpret_cast<uint4*>(dst + (2 * qq + h) * 16) =
                        make_uint4(pk[2 * qq].u[0], pk[2 * qq].u[1], pk[2 * qq + 1].u[0], pk[2 * qq + 1].u[1]);
            }
        }
        __syncthreads();

        // ---- C: stem2 for this wave's cout group and two output rows
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
            const int y = 2 * (wave >> 1) + yy;                    // tile-local output row
            floatx16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int m = 0; m < 18; ++m) {
                const int tap = m >> 1, kh = tap / 3, kw = tap - kh * 3;
                const unsigned char* src = s1 + (((2 * y + kh) * 2 + (kw & 1)) * 33 + r + (kw >> 1)) * S12_PIX + (2 * (m & 1) + h) * 16;
                acc = mfma16(wb[m], *reinterpret_cast<const half8*>(src), acc);
            }
            const int oy = y2_0 + y, ox = x2_0 + r;
            struct { unsigned u[2]; } pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bias_relu_pack(acc, g, b2[g], pk[g].u);
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                swap_halves(pk[2 * qq].u[0], pk[2 * qq + 1].u[0]);
                swap_halves(pk[2 * qq].u[1], pk[2 * qq + 1].u[1]);
            }
            if (oy < p.Ho2 && ox < p.Wo2) {
                _Float16* o = p.out + (((long)b * p.Ho2 + oy) * p.Wo2 + ox) * 64 + 32 * g2;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
                    *reinterpret_cast<uint4*>(o + 16 * qq + 8 * h) =
                        make_uint4(pk[2 * qq].u[0], pk[2 * qq].u[1], pk[2 * qq + 1].u[0], pk[2 * qq + 1].u[1]);
            }
        }
        // (no barrier here: the next tile's phase A only writes `patch`, which phase C does not read, and
        // its barrier orders this phase C before the next phase B overwrites `s1`)
    }
}

hipError_t launch_stem12_u8(const Stem12Params& p, hipStream_t stream) {
    if (!p.frames || !p.w1 || !p.bias1 || !p.w2 || !p.bias2 || !p.out || p.B <= 0 || p.H <= 0 || p.W <= 0 || p.Hc < p.H ||
        p.Wc < p.W || (p.Hc & 3) || (p.Wc & 3) || p.Ho1 != p.Hc / 2 || p.Wo1 != p.Wc / 2 || p.Ho2 != p.Ho1 / 2 || p.Wo2 != p.Wo1 / 2)
        return hipErrorInvalidValue;
    const long tiles = (long)p.B * ((p.Ho2 + S12_R2 - 1) / S12_R2) * ((p.Wo2 + S12_C2 - 1) / S12_C2);
    if (tiles <= 0 || tiles > 0x7fffffffL) return hipErrorInvalidValue;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    const int ncu = device_cu_count(dev);             // cached per device
    if (ncu <= 0) return hipErrorInvalidDevice;
    const long slots = 2L * ncu;                      // persistent: two workgroups per CU (63 KiB LDS each)
    hipLaunchKernelGGL(stem12_u8_kernel, dim3((unsigned)(tiles < slots ? tiles : slots)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Embedder stem: aligned chips (fp16 NHWC8, 3 real channels) -> conv3x3 s1 (3 -> 64) + bias + PReLU,
// fp16 NHWC64.  The generic kernel pads K = 27 to 128 and gathers per lane; this layer is bound by its
// 514 MB output (320 faces), so it gets the stem treatment: per kernel row the three taps of a pixel are
// 12 consecutive halfwords of an RGB0-packed LDS patch (k' = kw*4 + c), one MFMA (K = 16) per kernel
// row and cout group, B fragments shared by both cout groups.
// One workgroup = 8 output rows x up to 128 columns of one image (4 waves, 2 rows each).
#define ES_ROWS 8
#define ES_COLS 128
#define ES_PITCH 132        // patch pixels per row: columns -1 .. 130 (the upper half-wave reads two pixels further)

__global__ __launch_bounds__(256) void emb_stem_kernel(EmbStemParams p) {
    __shared__ __attribute__((aligned(16))) uint2 patch[(ES_ROWS + 2) * ES_PITCH];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_x = (p.W + ES_COLS - 1) / ES_COLS;
    const int tiles_y = (p.H + ES_ROWS - 1) / ES_ROWS;
    int q = blockIdx.x;
    const int tx = q % tiles_x; q /= tiles_x;
    const int ty = q % tiles_y;
    const int img = q / tiles_y;
    if (p.n_dev && img >= *p.n_dev) return;                                // chips beyond the device-side count do not exist
    const int y0 = ty * ES_ROWS, x0 = tx * ES_COLS;
    const _Float16* chip = p.x + (long)img * p.H * p.W * 8;

    // ---- patch: rows y0-1 .. y0+8, columns x0-1 .. x0+130; first 8 bytes (R,G,B,0) of each NHWC8 pixel
    for (int e = t; e < (ES_ROWS + 2) * ES_PITCH; e += 256) {
        const int pr = e / ES_PITCH, pc = e - pr * ES_PITCH;
        const int y = y0 - 1 + pr, x = x0 - 1 + pc;
        uint2 v = make_uint2(0u, 0u);
        if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
            v = *reinterpret_cast<const uint2*>(chip + ((long)y * p.W + x) * 8);
        patch[e] = v;
    }
    // ---- weights: A fragments [cout group][kernel row]; lane (cout r, half h): k' = 8h + j = kw*4 + c
    half8 wa[2][3];
#pragma unroll
    for (int cg = 0; cg < 2; ++cg)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const _Float16* w = p.w + (((cg * 32 + r) * 3 + kh) * 3) * 8;        // [kw][8]
            union { uint2 u[2]; half8 v; } f;
            f.u[0] = *reinterpret_cast<const uint2*>(w + (2 * h) * 8);            // kw = 0 (h=0) / 2 (h=1), channels 0..3
            f.u[1] = h ? make_uint2(0u, 0u) : *reinterpret_cast<const uint2*>(w + 8);   // kw = 1 / none
            wa[cg][kh] = f.v;
        }
    floatx4 bias[2][4], slope[2][4];
#pragma unroll
    for (int cg = 0; cg < 2; ++cg)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bias[cg][g] = *reinterpret_cast<const floatx4*>(p.bias + 32 * cg + 8 * g + 4 * h);
            slope[cg][g] = *reinterpret_cast<const floatx4*>(p.slope + 32 * cg + 8 * g + 4 * h);
        }
    __syncthreads();

#pragma unroll
    for (int yy = 0; yy < 2; ++yy) {
        const int yl = 2 * wave + yy, oy = y0 + yl;
#pragma unroll
        for (int cgp = 0; cgp < ES_COLS / 32; ++cgp) {
            const int xl = cgp * 32 + r, ox = x0 + xl;
            if (x0 + cgp * 32 >= p.W) break;                                       // uniform: no column of this group exists
            floatx16 acc[2];
#pragma unroll
            for (int cg = 0; cg < 2; ++cg)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[cg][e] = 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const uint2* src = patch + (yl + kh) * ES_PITCH + xl + 2 * h;      // taps start at column ox-1 = patch column xl
                union { uint2 u[2]; half8 v; } bf;
                bf.u[0] = src[0];
                bf.u[1] = src[1];
                acc[0] = mfma16(wa[0][kh], bf.v, acc[0]);
                acc[1] = mfma16(wa[1][kh], bf.v, acc[1]);
            }
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                union { half4 v; unsigned u[2]; } pk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[cg][4 * g + e] + bias[cg][g][e];
                        pk[g].v[e] = (_Float16)(v > 0.f ? v : v * slope[cg][g][e]);
                    }
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    swap_halves(pk[2 * qq].u[0], pk[2 * qq + 1].u[0]);
                    swap_halves(pk[2 * qq].u[1], pk[2 * qq + 1].u[1]);
                }
                if (oy < p.H && ox < p.W) {
                    _Float16* o = p.out + (((long)img * p.H + oy) * p.W + ox) * 64 + 32 * cg;
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq)
                        *reinterpret_cast<uint4*>(o + 16 * qq + 8 * h) =
                            make_uint4(pk[2 * qq].u[0], pk[2 * qq].u[1], pk[2 * qq + 1].u[0], pk[2 * qq + 1].u[1]);
                }
            }
        }
    }
}

hipError_t launch_emb_stem(const EmbStemParams& p, hipStream_t stream) {
    if (!p.x || !p.w || !p.bias || !p.slope || !p.out || p.M <= 0 || p.H <= 0 || p.W <= 0) return hipErrorInvalidValue;
    const long tiles = (long)p.M * ((p.H + ES_ROWS - 1) / ES_ROWS) * ((p.W + ES_COLS - 1) / ES_COLS);
    if (tiles <= 0 || tiles > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(emb_stem_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, p);
    return hipGetLastError();
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
