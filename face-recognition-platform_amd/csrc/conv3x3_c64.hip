// K2d: 3x3 stride-1 convolution 64 -> 64 channels - the layers that live on the LARGEST maps (the detector's 272 x 480 stage,
// the embedder's 112 x 112 and 56 x 56 stage: 8 launches, 1.5 ms of an 11.5 ms step in round 4, at 650-920 TFLOP/s).
//
// Why its own kernel.  With 64 input channels a tile's whole contraction is 9 k-steps: in the row-patch kernel
// (conv3x3_lean.hip, 512 x 64 tiles) every tile pays a ring hand-over, nine barriers, nine weight stages of LDS-DMA and an
// epilogue for 8.7 us of k-loop, and the 64-cout tile re-reads every weight fragment from LDS.  Here
//   * the WEIGHTS never move after the prologue: a wave owns 32 couts, and the 36 A fragments of their whole K = 576 sit in 144
//     registers for the life of the workgroup - no weight DMA, no weight fragment reads, no weight ring;
//   * the pixel operand is a 2-D tile (8 rows x 32 columns of one image; halo'd patch 10 x 34 pixels) brought in by LDS-DMA into a
//     three-slot ring, TWO tiles ahead; out-of-image pixels are zero-filled by the DMA itself (per-lane source offsets pushed out of
//     the buffer's range), so fragment reads need no masks and no zero block;
//   * a tile is ONE barrier: all 72 MFMAs of a wave (2 pixel blocks x 36 k-slices) run on fragments of one patch, one
//     ds_read_b128 per MFMA, every one of them at the block's base register + an immediate (144-byte pixel pitch: C64_PITCH);
//   * stores are unconditional buffer stores (pixels that do not exist get an out-of-range offset), so the in-order memory
//     counter is exact and the wait in front of a tile's barrier admits the next tile's DMA and the last epilogue's stores.
// Eight waves = 4 pixel groups (tile rows 2g, 2g + 1) x 2 cout groups; accumulation order (kh, kw, 16-channel slice) and
// epilogue arithmetic are those of the other direct kernels: bit-identical results (tests: test_conv_c64_equals_generic_kernel).
//
// Replaces the same reference calls as conv_mfma.hip (face_recognition.face_locations / face_encodings,
// backend/app/routes/camera.py:232,237).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "frp_internal.h"
#include "conv_common.h"

namespace frp {

// Tile geometry (template parameter TW): 32 columns x 8 rows (halo'd patch 10 x 34 = 340 pixels; a pixel block of the MFMA = one tile
// row) or 16 x 16 (patch 18 x 18 = 324; a pixel block = two tile rows of 16) - whichever wastes fewer tile pixels on the map.
#define C64_ROWS_MAX 340
#define C64_PITCH 144                       // LDS bytes per patch pixel: 128 + one 16-byte pad slot.  9 sixteen-byte slots per pixel
                                            // make consecutive pixels walk all 16 slot positions of the 256-byte bank row (9 is odd), so
                                            // a ds_read_b128's 16-lane groups are conflict-free WITHOUT an xor swizzle - and without
                                            // one every (tap, 16-channel slice) of a pixel block is the block's base address + an
                                            // immediate: no vector instruction in the k-loop (with the xor layout of the other kernels a
                                            // tile cost 72 v_xor + 90 address operations per wave, and the loop was VALU-issue-bound)
#define C64_PIECES 48                       // LDS-DMA pieces per patch (1 KiB = 7.1 pixels each; 340 x 144 B = 47.8 KiB: six per wave)
static_assert(C64_ROWS_MAX * 9 <= C64_PIECES * 64, "patch slots");
#define C64_BUF (C64_PIECES * 1024)
#define C64_NBUF 3
#define C64_OFF_PAR (C64_NBUF * C64_BUF)
#define C64_LDS (C64_OFF_PAR + 10 * 64 * 4)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Specialised at compile time on (activation, residual, border-class bias): with the flags as run-time values the epilogue was a
// chain of ~90 uniform branches and the register allocator spilled eleven registers, each reload a vmcnt(0) in the tile loop.
// STEM (round 5): the launch also computes the embedder's stem conv of its own input patch (ConvParams::stem_x): per tile the chips under
// the patch (+ one more halo pixel) arrive by LDS-DMA two tiles ahead (one piece per wave), a short MFMA phase turns them into the
// 64-channel patch in LDS (and stores the tile's own pixels of that map for the block's shortcut) - the patch of tile t + 1 in the
// iteration that multiplies tile t, so the one barrier per tile stays and the phase is not exposed between two barriers (as a phase of
// its own in front of the k-loop the fused launch took as long as the two launches it replaces).  Same instructions on the same operands as emb_stem_kernel and as the unfused k-loop: bit-identical.
#ifndef C64S_ABL
#define C64S_ABL 0                          // lab builds: timing ablations of the fused stem (1: no stem phase, 2: no stores of the stem map, 4: no patch writes, 8: no output stores)
#endif
#define C64_CHIP_SLOT 8192                  // chips under a patch: (TH + 4) x (TW + 4) pixels of 16 B <= 512 pixels
#define C64_OFF_CHIPS (2 * C64_BUF)          // (the patch is double-buffered: the stem phase of tile t + 1 runs beside the k-loop of tile t)
#define C64_OFF_SW (C64_OFF_CHIPS + 3 * C64_CHIP_SLOT)       // stem weight fragments [cout group][kh][lane] (6 KiB)
#define C64_OFF_SPAR (C64_OFF_SW + 6 * 1024)                 // stem bias [64], slope [64]

template <int ACT, bool RES, bool BORDER, int C64_TW = 32, bool STEM = false>
__global__ __launch_bounds__(512, 2) void conv3x3_c64_kernel(ConvParams p_in) {
    constexpr int C64_TH = 256 / C64_TW;              // 8 or 16 tile rows
    constexpr int C64_RB = 32 / C64_TW;               // tile rows per 32-pixel block: 1 or 2
    constexpr int C64_PW = C64_TW + 2;                // patch columns
    constexpr int C64_ROWS = (C64_TH + 2) * C64_PW;   // patch pixels
    static_assert((C64_TW == 32 || C64_TW == 16) && C64_ROWS <= C64_ROWS_MAX, "tile geometry");
    constexpr int C64_CW = C64_TW + 4, C64_CH = C64_TH + 4;   // chips under a patch (STEM)
    constexpr int C64_NBLK = (C64_ROWS + 31) / 32;            // 32-pixel blocks of a patch (STEM): 11
    static_assert(C64_CW * C64_CH <= 512 && C64_NBLK <= 16 && C64_OFF_SPAR + 512 <= C64_OFF_PAR && !(STEM && RES), "stem fusion");
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    ConvParams p = p_in;
    int N = p.N;
    if (p.n_dev) {                             // image count known on the device only (threshold mode)
        const int n = *p.n_dev;
        N = n < 0 ? 0 : (n > p.N ? p.N : n);
    }
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tx_n = (p.W + C64_TW - 1) / C64_TW, ty_n = (p.H + C64_TH - 1) / C64_TH;
    const int per = tx_n * ty_n;
    const int n_tiles = N * per;
    int t0, t1, tstep;
    if ((gridDim.x & 7) == 0) {                // XCD-interleaved tile walk (see conv3x3_lean.hip)
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, pr = gridDim.x >> 3;
        const int cs = (int)((long)x * n_tiles / 8), ce = (int)((long)(x + 1) * n_tiles / 8);
        t0 = cs + j;
        t1 = ce;
        tstep = pr;
    } else {
        t0 = (int)((long)blockIdx.x * n_tiles / gridDim.x);
        t1 = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
        tstep = 1;
    }
    if (t0 >= t1) return;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const unsigned o_bytes = (unsigned)((long)p.N * p.H * p.W * 64 * 2);             // (< 2 GiB: launcher)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, o_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.x), 0, p.res ? o_bytes : 0u, 0x00020000);

    // ---------------- epilogue parameters, once per workgroup (one cout tile): bias [9][64] (class-major), PReLU slope [64]
    float* lds_bias = reinterpret_cast<float*>(smem + C64_OFF_PAR);
    float* lds_slope = lds_bias + 9 * 64;
    constexpr bool border = BORDER;
    for (int i = t; i < 9 * 64; i += 512) lds_bias[i] = (border || i < 64) ? p.bias[i] : 0.f;
    if (t < 64) lds_slope[t] = (ACT == FRP_ACT_PRELU && p.slope) ? p.slope[t] : 0.f;

    // ---------------- STEM: the stem's A fragments (lane (cout r, half h) of cout group cg, kernel row kh: k' = 8h + j = kw*4 + c, as
    // emb_stem_kernel builds them in registers - here they wait in LDS, the registers hold this kernel's own 144), bias and slope
    if (STEM) {
        if (t < 384) {
            const int f = t >> 6, l = t & 63, r_ = l & 31, h_ = l >> 5, cg = f / 3, kh = f - cg * 3;
            const _Float16* w = p.stem_w + (((cg * 32 + r_) * 3 + kh) * 3) * 8;
            union { uint2 u[2]; u32x4 v; } fw;
            fw.u[0] = *reinterpret_cast<const uint2*>(w + (2 * h_) * 8);
            fw.u[1] = h_ ? make_uint2(0u, 0u) : *reinterpret_cast<const uint2*>(w + 8);
            *reinterpret_cast<u32x4*>(smem + C64_OFF_SW + t * 16) = fw.v;
        }
        if (t < 64) {
            reinterpret_cast<float*>(smem + C64_OFF_SPAR)[t] = p.stem_bias[t];
            reinterpret_cast<float*>(smem + C64_OFF_SPAR)[64 + t] = p.stem_slope[t];
        }
    }

    // ---------------- this wave's weights: 32 couts x K = 576 as 36 MFMA A fragments (k-slice s = tap * 4 + kk: channels 16 kk + 8 fh .. + 7)
    const int wave_p = wave >> 1, wave_c = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    half8 A[36];
    {
        const _Float16* wrow = p.w + (long)(wave_c * 32 + fr) * 576 + 8 * fh;
#pragma unroll
        for (int s = 0; s < 36; ++s) A[s] = *reinterpret_cast<const half8*>(wrow + (s >> 2) * 64 + (s & 3) * 16);
    }

    // ---------------- DMA lane geometry: piece j of this wave fills the 16-byte slots 64 (wave + 8 j) + lane of a patch buffer; slot q
    // = chunk q % 9 (8: the pad) of patch pixel R = q / 9 = (R / 34, R % 34)
    // (recomputed per piece from the lane index - a dozen vector operations against twelve registers held through the k-loop,
    // whose 144 weight registers leave no room for tables; lane_e is made opaque per tile so that the compiler does not build the
    // tables after all)
    int lane_e = lane;
    struct Patch { int base, y0, x0; bool live; };          // scalars of a tile's patch (image coordinates of its origin)
    auto patch_of = [&](int tile) -> Patch {
        Patch q{0, 0, 0, tile < t1};
        if (q.live) {
            const int n = tile / per, r = tile - n * per;
            const int ty = r / tx_n, tx = r - ty * tx_n;
            q.y0 = ty * C64_TH - 1;
            q.x0 = tx * C64_TW - 1;
            q.base = ((n * p.H + q.y0) * p.W + q.x0) * 128;
        }
        return q;
    };
    auto issue_piece = [&](const Patch& q, int j, int slot) {
        const int sl = (wave + 8 * j) * 64 + lane_e;
        const int R = (sl * 7282) >> 16, chunk = sl - 9 * R;               // sl / 9, sl % 9 (exact for sl < 3072)
        const int py = C64_TW == 32 ? (R * 241) >> 13 : (R * 3641) >> 16;  // R / 34, R / 18 (exact for R < 384)
        const int px = R - py * C64_PW;
        const int y = q.y0 + py, x = q.x0 + px;
        const bool ok = q.live && R < C64_ROWS && chunk < 8 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        dma16(xrsrc, smem + slot * C64_BUF + (wave + 8 * j) * 1024, ok ? (unsigned)(q.base + (py * p.W + px) * 128 + chunk * 16) : CONV_OOB);
    };
    auto issue_patch = [&](int tile, int slot) {
        const Patch q = patch_of(tile);
#pragma unroll
        for (int j = 0; j < 6; ++j) issue_piece(q, j, slot);
    };

    // ---------------- fragment addresses: pixel block b = tile rows RB (2 wave_p + b) .. + RB - 1, lane = (row fr / TW, column fr % TW); tap
    // (kh, kw), slice kk reads patch pixel (row + kh, column + kw), chunk 2 kk + fh: the block's base + the immediate
    // ((kh * PW + kw) * 144 + kk * 32)
    // 16-column tiles: WHICH 16 lanes of a block take its first row is chosen to match ds_read_b128's lane groups ({0-3, 12-15, 20-27} and
    // {4-11, 16-19, 28-31} of a half-wave are serviced together): one group = 16 consecutive pixels of one patch row, conflict-free like
    // the 32-column form (lanes 0-15 / 16-31 would put 8 pixels of each row into a group: two bank conflicts per read)
    const int fq = fr >> 2;
    const int lrow = C64_TW == 32 ? 0 : (0x96 >> fq) & 1, lcol = C64_TW == 32 ? fr : ((fq >> 1) << 2) | (fr & 3);
    const int fbase = ((2 * wave_p * C64_RB + lrow) * C64_PW + lcol) * C64_PITCH + fh * 16;

    // ---------------- STEM: the chips under a tile's patch, one LDS-DMA piece per wave: slot pixel 64 wave + lane = (cy, cx) of the
    // (TH + 4) x (TW + 4) block whose origin lies two pixels up and left of the tile; outside the chip: zeros (the stem's padding)
    const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(STEM ? p.stem_x : p.x), 0, STEM ? (unsigned)((long)p.N * p.H * p.W * 16) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(STEM ? (void*)p.stem_out : p.out, 0, STEM ? o_bytes : 0u, 0x00020000);
    auto issue_chips = [&](int tile, int cslot) {
        const Patch q = patch_of(tile);                          // (y0, x0: the patch origin = tile origin - 1)
        const int sl = wave * 64 + lane_e;
        const int cy = C64_TW == 32 ? (sl * 1821) >> 16 : (sl * 3277) >> 16, cx = sl - cy * C64_CW;     // sl / 36, sl / 20 (exact for sl < 512)
        const int y = q.y0 - 1 + cy, x = q.x0 - 1 + cx;
        const bool ok = q.live && cy < C64_CH && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const int n_ = q.live ? tile / per : 0;
        dma16(crsrc, smem + C64_OFF_CHIPS + cslot * C64_CHIP_SLOT + wave * 1024, ok ? (unsigned)(((n_ * p.H + y) * p.W + x) * 16) : CONV_OOB);
    };
    const int n_myblk = (wave + 8 < C64_NBLK) ? 2 : 1;           // patch blocks of this wave in the stem phase (wave-uniform)

    // ---- stem phase: the 64-channel patch of `tile` from its chips (slot cslot of the chips ring) into patch buffer pbuf
    auto stem_phase = [&](int tile, int cslot, int pbuf) __attribute__((always_inline)) {
            const Patch cq = patch_of(tile);
            const int n_ = tile / per;
            const unsigned char* cs = smem + C64_OFF_CHIPS + cslot * C64_CHIP_SLOT;
            const float* sb = reinterpret_cast<const float*>(smem + C64_OFF_SPAR);
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {
                if (bi == 1 && n_myblk < 2) break;
                const int R = (wave + 8 * bi) * 32 + fr;
                const bool valid = R < C64_ROWS;
                const int Rc = valid ? R : 0;
                const int py = C64_TW == 32 ? (Rc * 241) >> 13 : (Rc * 3641) >> 16, px = Rc - py * C64_PW;
                floatx16 sacc[2];
#pragma unroll
                for (int cg = 0; cg < 2; ++cg)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sacc[cg][e] = 0.f;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    // lane half h: chips (row py + kh, columns px + 2h, px + 2h + 1), 4 of their 8 halfwords each (R, G, B, 0): kw = 2h, 2h + 1
                    const unsigned char* src = cs + ((py + kh) * C64_CW + px + 2 * fh) * 16;
                    union { uint2 u[2]; half8 v; } bfr;
                    bfr.u[0] = *reinterpret_cast<const uint2*>(src);
                    bfr.u[1] = *reinterpret_cast<const uint2*>(src + 16);
#pragma unroll
                    for (int cg = 0; cg < 2; ++cg) {
                        const half8 wa = *reinterpret_cast<const half8*>(smem + C64_OFF_SW + ((cg * 3 + kh) * 64 + lane) * 16);
                        sacc[cg] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, bfr.v, sacc[cg], 0, 0, 0);
                    }
                }
                const int y = cq.y0 + py, x = cq.x0 + px;
                const bool inimg = valid && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                // a pixel of the tile itself: stored (stem_even_only: if a stride-2 reader will ask for it)
                const bool mine = inimg && py >= 1 && py <= C64_TH && px >= 1 && px <= C64_TW && !(p.stem_even_only && ((y | x) & 1));
                const unsigned goff = mine ? (unsigned)(((n_ * p.H + y) * p.W + x) * 128 + fh * 16) : CONV_OOB;
                unsigned char* prow = smem + pbuf * C64_BUF + R * C64_PITCH + fh * 16;
#pragma unroll
                for (int cg = 0; cg < 2; ++cg) {
                    union { half4 hv; unsigned u[2]; } pk[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const floatx4 b4 = *reinterpret_cast<const floatx4*>(sb + 32 * cg + 8 * g + 4 * fh);
                        const floatx4 s4 = *reinterpret_cast<const floatx4*>(sb + 64 + 32 * cg + 8 * g + 4 * fh);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = sacc[cg][4 * g + e] + b4[e];
                            pk[g].hv[e] = (_Float16)(v > 0.f ? v : v * s4[e]);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        swap_halves(pk[2 * q].u[0], pk[2 * q + 1].u[0]);
                        swap_halves(pk[2 * q].u[1], pk[2 * q + 1].u[1]);
                        u32x4 o = {pk[2 * q].u[0], pk[2 * q].u[1], pk[2 * q + 1].u[0], pk[2 * q + 1].u[1]};     // couts 32 cg + 16 q + 8 fh .. + 7
                        if (C64S_ABL & 2) asm volatile("" ::"v"(o)); else
                        __builtin_amdgcn_raw_buffer_store_b128(o, srsrc, goff + cg * 64 + q * 32, 0, 0);
                        if (!inimg) o = u32x4{0u, 0u, 0u, 0u};                  // the conv's zero padding, not the stem of padded chips
                        if (valid && !(C64S_ABL & 4)) *reinterpret_cast<u32x4*>(prow + cg * 64 + q * 32) = o;
                    }
                }
            }
    };

    __syncthreads();                           // parameters visible (no LDS-DMA in flight yet: the fence costs nothing)
    if (STEM) {
        issue_chips(t0, 0);
        issue_chips(t0 + tstep, 1);
        wait_vmcnt<1>();
        __builtin_amdgcn_s_barrier();          // every wave's piece of the first tile's chips has landed
        stem_phase(t0, 0, 0);
    } else {
        issue_patch(t0, 0);
        issue_patch(t0 + tstep, 1);
    }
    constexpr bool has_res = RES;

    // ---------------- epilogue of a tile: lane = pixel (oy0 + b, ox), registers 4g .. 4g+3 = couts 32 wave_c + 8g + 4 fh .. + 3
    floatx16 acc[2];
    auto epilogue = [&](int tile) __attribute__((always_inline)) {
        const int n = tile / per, r = tile - n * per;
        const int ty = r / tx_n, tx = r - ty * tx_n;
        const int oy0 = ty * C64_TH + 2 * wave_p * C64_RB + lrow, ox = tx * C64_TW + lcol;
        u32x4 rr[2][2];
        unsigned ooff[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int oy = oy0 + b * C64_RB;
            const bool ok = oy < p.H && ox < p.W;
            const int m = (n * p.H + oy) * p.W + ox;
            ooff[b] = ok ? (unsigned)(m * 128 + wave_c * 64 + fh * 16) : CONV_OOB;
            if (has_res) {
#pragma unroll
                for (int q = 0; q < 2; ++q) rr[b][q] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ooff[b] + 32 * q, 0, 0);
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int oy = oy0 + b * C64_RB;
            int cls = 0;
            if (border) cls = ((oy == 0) ? 0 : (oy == p.H - 1) ? 2 : 1) * 3 + ((ox == 0) ? 0 : (ox == p.W - 1) ? 2 : 1);
            half4 r4[4];
            if (has_res) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    u32x4 v = rr[b][q];
                    unsigned x_ = v.x, y_ = v.y, z_ = v.z, w_ = v.w;
                    swap_halves(x_, z_);
                    swap_halves(y_, w_);
                    union { unsigned u[2]; half4 h; } lo, hi;
                    lo.u[0] = x_; lo.u[1] = y_; hi.u[0] = z_; hi.u[1] = w_;
                    r4[2 * q] = lo.h;
                    r4[2 * q + 1] = hi.h;
                }
            }
            union { half4 h; unsigned u[2]; } pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wave_c * 32 + 8 * g + 4 * fh;
                const floatx4 b4 = *reinterpret_cast<const floatx4*>(lds_bias + cls * 64 + cl);
                floatx4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[b][4 * g + e] + b4[e];
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)r4[g][e];
                }
                if (ACT == FRP_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (ACT == FRP_ACT_PRELU) {
                    const floatx4 s4 = *reinterpret_cast<const floatx4*>(lds_slope + cl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * s4[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[g].h[e] = (_Float16)v[e];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                swap_halves(pk[2 * q].u[0], pk[2 * q + 1].u[0]);
                swap_halves(pk[2 * q].u[1], pk[2 * q + 1].u[1]);
                const u32x4 o = {pk[2 * q].u[0], pk[2 * q].u[1], pk[2 * q + 1].u[0], pk[2 * q + 1].u[1]};
                if (STEM && (C64S_ABL & 8)) asm volatile("" ::"v"(o)); else
                __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, ooff[b] + 32 * q, 0, 0);   // couts 32 wave_c + 16 q + 8 fh .. + 7
            }
        }
    };

    // PING-PONG: the two waves of a SIMD (w and w + 4) run the iteration's two halves in opposite order.  Waves 0-3 multiply tile ct, then
    // convert and store it; waves 4-7 first convert and store tile ct - 1 (their accumulators wait across the barrier), then multiply tile
    // ct - so one wave's 72 MFMAs run under its partner's epilogue instead of both waves sharing the matrix core and then both leaving it
    // idle (a tile took 4.9 us with NO memory traffic at all, 2.8 of them MFMAs: profiles/r5/ab_stem_fusion.txt).  One barrier per tile as
    // before; a wave's own operations keep their order, so the counted waits only change for the late waves' first two tiles.
    // Measured (tools/c64_pingpong_probe.py, same process): 272 x 480 with residual 404 -> 383 us, without 344 -> 338, in the pipeline the
    // detector -37 us per step.  The fused-stem variant does NOT take it (its late waves' accumulators next to the stem phase's registers:
    // 13 spilled registers, embedder +35 us): there the stem phase alone is dealt to the two halves of the iteration.
    const bool late = !STEM && wave >= 4 && !(p.dbg & 4096);   // (dbg bit 4096 = flags bit 22 of frp_conv2d_nhwc / frp_conv_bench: all waves in the same order - A/B runs)
    const bool stem_first = STEM && wave < 4;
    int slot = 0, it = 0;
    for (int ct = t0; ct < t1; ct += tstep, ++it) {
        // tile ct has landed: behind its pieces in the in-order counter sit the next tile's six pieces and - from the second tile
        // on - the four stores of the last epilogue (unconditional: the count is exact)
        if (STEM) {
            // the chips of tile ct + 1 (consumed below) have landed once only this wave's stores behind them may be pending: 4 per patch
            // block of the stem phase that followed their issue + - from the second tile on - the last epilogue's 4
            if (C64S_ABL) wait_vmcnt<0>();
            else if (ct == t0) { if (n_myblk == 2) wait_vmcnt<8>(); else wait_vmcnt<4>(); }
            else if (n_myblk == 2) wait_vmcnt<12>();
            else wait_vmcnt<8>();
        } else if (ct == t0 || (late && it == 1)) wait_vmcnt<6>();
        else if (has_res) wait_vmcnt<14>();    // (the epilogue's four residual loads sit in front of its stores)
        else wait_vmcnt<10>();
        retire_lds_reads();
        __builtin_amdgcn_s_barrier();
        asm volatile("" : "+v"(lane_e));
        int nslot = slot + 2;
        nslot = nslot >= C64_NBUF ? nslot - C64_NBUF : nslot;
        if (STEM) {
            issue_chips(ct + 2 * tstep, nslot);                       // into the slot the previous iteration's stem phase read
            // (the next tile's patch: waves 0-3 build their blocks BEFORE this tile's k-loop, waves 4-7 behind its epilogue - the two waves
            // of a SIMD, w and w + 4, convert and pack at different times)
        }
        if (late && it > 0) epilogue(ct - tstep);
        if (STEM && !(C64S_ABL & 1) && stem_first && ct + tstep < t1) stem_phase(ct + tstep, slot + 1 == C64_NBUF ? 0 : slot + 1, (it + 1) & 1);
        // the patch two tiles ahead goes out BETWEEN the MFMAs below (its address arithmetic hides under matrix time; it has two
        // tiles to land); past the end: zero-fill into a slot nobody reads
        const Patch nq = patch_of(ct + 2 * tstep);

        const int boff = STEM ? (it & 1) * C64_BUF : slot * C64_BUF;

#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
        half8 bf[3][2];                        // fragments of k-slices s, s + 1, s + 2 (requested two slices ahead of their MFMAs)
        const unsigned char* pb = smem + (fbase + boff);
        auto rd = [&](int s, int S) {
            const int tap = s >> 2, kk = s & 3;
#pragma unroll
            for (int b = 0; b < 2; ++b)
                bf[S][b] = *reinterpret_cast<const half8*>(pb + ((b * C64_RB + tap / 3) * C64_PW + tap % 3) * C64_PITCH + kk * 32);
        };
        rd(0, 0);
        rd(1, 1);
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + 2 < 36) rd(s + 2, (s + 2) % 3);
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], bf[s % 3][b], acc[b], 0, 0, 0);
            if (!STEM && s % 5 == 4 && s / 5 < 6) issue_piece(nq, s / 5, nslot);
            __builtin_amdgcn_sched_barrier(0);             // (the slices stay in this order: the compiler pulls the reads next to their MFMAs)
        }

        if (!late) {
            epilogue(ct);
            // (the early waves' accumulators are dead here; said in a way the register allocator sees - it does not correlate this branch
            // with the late waves' deferred epilogue and would keep all 32 registers live through the stem phase below)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
        }
        if (STEM && !(C64S_ABL & 1) && !stem_first && ct + tstep < t1) stem_phase(ct + tstep, slot + 1 == C64_NBUF ? 0 : slot + 1, (it + 1) & 1);
        slot = slot + 1 == C64_NBUF ? 0 : slot + 1;
    }
    if (late) epilogue(t0 + (it - 1) * tstep);             // the late waves' last tile
    // nothing of this workgroup's DMA stream may still be in flight when its LDS is handed to the next workgroup
    wait_vmcnt<0>();
}

static int c64_variant(const ConvParams& p);
// tile width for a map: the geometry that wastes fewer tile pixels (32 x 8 on a tie: its halo is the smaller share of the DMA)
static int c64_tile_width(const ConvParams& p, double* fill = nullptr) {
    double best = 0;
    int tw_best = 32;
    for (int tw : {32, 16}) {
        const int th = 256 / tw;
        const double f = (double)p.H * p.W / ((double)((p.H + th - 1) / th * th) * (double)((p.W + tw - 1) / tw * tw));
        if (f > best + 1e-9) { best = f; tw_best = tw; }
    }
    if (fill) *fill = best;
    return tw_best;
}
static long c64_tiles(const ConvParams& p, int tw) {
    const int th = 256 / tw;
    return (long)p.N * ((p.H + th - 1) / th) * ((p.W + tw - 1) / tw);
}
// Shapes the kernel covers (`p` with launch_conv()'s derived fields) - and where it pays: at least two rounds of tiles.
bool conv3x3_c64_eligible(const ConvParams& p) {
    if (p.KS != 3 || p.stride != 1 || p.Cin != 64 || p.Cout != 64 || p.ksplit != 1 || p.x2 || p.out2) return false;
    if (p.flags & (FRP_FLAG_OUT_F32 | FRP_FLAG_F8 | FRP_FLAG_OUT_FP8 | FRP_FLAG_RES_UP2)) return false;
    if (p.Ho != p.H || p.Wo != p.W) return false;
    if ((long)p.N * p.H * p.W * 128 >= 0x7f000000L) return false;                  // signed 32-bit byte offsets (+ a patch of slack)
    if (!c64_variant(p)) return false;
    if (p.act == FRP_ACT_PRELU && !p.slope) return false;
    if (p.stem_x && (c64_variant(p) != 3 || !p.stem_w || !p.stem_bias || !p.stem_slope || !p.stem_out || (long)p.N * p.H * p.W * 16 >= 0x7f000000L))
        return false;
    double fill = 0;
    const int tw = c64_tile_width(p, &fill);
    if (c64_tiles(p, tw) < 2L * (p.n_cu > 0 ? p.n_cu : 256)) return false;
    // where it pays (tools/c64_probe.py, profiles/r5/c64_probe.txt): x1.11 / x1.14 over the row-patch kernel on maps whose tiles are all
    // full (the detector's 272 x 480 in 32 x 8 tiles, the embedder's 112 x 112 in 16 x 16 tiles); x0.99-1.03 on its 56 x 56 maps, where
    // either geometry leaves an eighth of the tile pixels empty - those stay on the row-patch kernel (FRP_C64_ALL=1 takes every
    // eligible shape: the parity tests)
    static const bool all = getenv("FRP_C64_ALL") != nullptr;
    return all || fill >= 0.95;
}

template <int ACT, bool RES, bool BORDER, int TW, bool STEM = false>
static hipError_t launch_c64_geo(const ConvParams& p, hipStream_t stream) {
    static bool attr_set[64] = {};
    auto kern = conv3x3_c64_kernel<ACT, RES, BORDER, TW, STEM>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const int ncu = device_cu_count(dev);
    if (ncu <= 0) return hipErrorInvalidDevice;
    const long tiles = c64_tiles(p, TW);
    if (tiles <= 0 || tiles > 0x7fffffffL) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(tiles < ncu ? tiles : ncu);       // persistent: one workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), C64_LDS, stream, p);
    return hipGetLastError();
}

// maps on which the launch of the embedder's first 64 -> 64 conv also computes the stem in front of it (frp_api.cpp: run_net): where the
// kernel is routed anyway
bool conv3x3_c64_fuses_stem(int N, int H, int W, int n_cu) {
    ConvParams q{};
    q.N = N; q.H = H; q.W = W; q.n_cu = n_cu;
    double fill = 0;
    const int tw = c64_tile_width(q, &fill);
    return (long)N * H * W * 128 < 0x7f000000L && c64_tiles(q, tw) >= 2L * (n_cu > 0 ? n_cu : 256) && fill >= 0.95;
}

template <int ACT, bool RES, bool BORDER>
static hipError_t launch_c64_cfg(const ConvParams& p, hipStream_t stream) {
    return c64_tile_width(p) == 16 ? launch_c64_geo<ACT, RES, BORDER, 16>(p, stream) : launch_c64_geo<ACT, RES, BORDER, 32>(p, stream);
}

// the four (activation, residual, border bias) combinations the two networks have on these layers
static int c64_variant(const ConvParams& p) {
    const bool res = p.res != nullptr, border = (p.flags & FRP_FLAG_BORDER_BIAS) != 0;
    if (p.act == FRP_ACT_RELU && !border) return res ? 2 : 1;          // detector layer1
    if (p.act == FRP_ACT_PRELU && !res && border) return 3;            // IResNet conv1 (folded pre-conv BN: 9 bias classes)
    if (p.act == FRP_ACT_NONE && res && !border) return 4;             // IResNet conv2
    return 0;
}

hipError_t launch_conv3x3_c64(const ConvParams& p, hipStream_t stream) {
    if (!conv3x3_c64_eligible(p)) return hipErrorInvalidValue;
    switch (c64_variant(p)) {
        case 1: return launch_c64_cfg<FRP_ACT_RELU, false, false>(p, stream);
        case 2: return launch_c64_cfg<FRP_ACT_RELU, true, false>(p, stream);
        case 3:
            if (p.stem_x)
                return c64_tile_width(p) == 16 ? launch_c64_geo<FRP_ACT_PRELU, false, true, 16, true>(p, stream)
                                               : launch_c64_geo<FRP_ACT_PRELU, false, true, 32, true>(p, stream);
            return launch_c64_cfg<FRP_ACT_PRELU, false, true>(p, stream);
        case 4: return launch_c64_cfg<FRP_ACT_NONE, true, false>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace frp
