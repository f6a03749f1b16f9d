// K2b: 3x3 stride-1 convolution (Cin a multiple of 64) with ROW PATCHES: the dominant layer
// class of both networks (detector body / FPN / head towers, every IResNet conv1/conv2).
//
// Same GEMM view, data layout, MFMA tiling and epilogue as conv_mfma.hip (weights = A operand,
// pixels = B operand, 256 pixels x TC couts per workgroup, 8 waves, v_mfma_f32_32x32x16_f16,
// k order = (64-channel block, kh, kw)).  The difference is how the pixel operand gets to LDS:
//
//   conv_mfma.hip DMAs a fresh 256-row image for EVERY tap (9 x 32 KiB per channel block).
//   Here one "row patch" per (channel block, kh) holds the 258 consecutive pixels
//   [m0 + (kh-1)*W - 1, m0 + (kh-1)*W + 257) of the flattened NHW index space, and the three
//   taps kw = 0,1,2 read it at row offsets +0,+1,+2: 3 x 33 KiB per channel block, a third of
//   the LDS-DMA traffic (and of the L2 requests), issued two ROW steps (six k-steps) ahead, so
//   an HBM miss on first touch is long gone when the rows are needed.  The weights keep their
//   per-tap 3-slot ring (L2-resident, short latency).
//
//   A flattened neighbour that lies outside the image (padding, or the pixel run wrapped into
//   the next row / image) must read as zero: that is decided when the B fragment is READ - a lane
//   whose tap is invalid reads a 256-byte zero block instead (at the bank offset its real address
//   would have had, so the substitution adds no bank conflicts).  The patch DMA therefore needs no
//   per-row validity at all; rows before / beyond the tensor are zeros by the buffer range check.
//
// Replaces the same reference calls as conv_mfma.hip (face_recognition.face_locations /
// face_encodings, backend/app/routes/camera.py:232,237, backend/app/services/face_service.py:156,179).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "frp_internal.h"
#include "conv_common.h"

namespace frp {

// PF = true: cross-barrier fragment prefetch.  The kk = 0 fragments of k-step s+1 are requested before the last
// MFMA group of step s, so the first MFMAs after the barrier issue at once instead of sitting out an LDS round
// trip with every wave of the CU in the same state (2 waves per SIMD, both behind the barrier: nothing feeds the
// matrix pipe for ~150-250 of a step's ~1100 cycles).  That needs the k = 0..15 slice of the NEXT weight stage
// in LDS one barrier early; instead of a fourth ring slot the weight stages are SHIFTED by a quarter: stage s
// holds chunks 2..7 (kk = 1..3) of step s and, in chunk positions 0..1, kk = 0 of step s+1 - the DMA source
// address is per lane, so lanes carrying logical chunks 0/1 simply run their cursor one step ahead.  The pixel
// side needs nothing: the row patches are resident two row steps ahead anyway.
// ABL: timing-only ablations for tools/conv_bench.py (results are wrong): 1 no MFMA, 2 no fragment reads, 4 no DMA,
// 8 no barrier.
template <int TC, int WP, int WC, bool PF, int ABL = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_rows_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    constexpr int TP = 256, NW = 8;
    constexpr int XROWS = 264;                // 33 groups of 8 rows; rows 0..257 are read
    constexpr int XSLOT = XROWS * 128;        // 33,792 B (a multiple of 256)
    constexpr int WSLOT = TC * 128;
    constexpr int NXS = 3, NWS = 3;           // ring slots: row patches / weight stages
    constexpr int WI = TC / 8 / NW;           // weight DMA pieces per wave per k-step (2 or 1)
    constexpr int MP = TP / WP / 32, MC = TC / WC / 32;
    constexpr int OFF_W = NXS * XSLOT;
    constexpr int OFF_Z = OFF_W + NWS * WSLOT;         // 256 zero bytes (256-aligned)
    constexpr int OFF_PAR = OFF_Z + 256;
    static_assert(WP * WC == NW && (WI == 1 || WI == 2) && MP * MC <= 4 && (OFF_Z & 255) == 0, "layout");

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    const int n_tiles = p.n_ptiles * p.n_ctiles;
    // Tile walk.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  XCD x takes the
    // contiguous chunk [x*T/8, (x+1)*T/8) of the tile list and its G/8 workgroups walk it INTERLEAVED (workgroup
    // j: chunk + j, + j + G/8, ...): at any moment the workgroups of one L2 work on G/8 consecutive tiles, i.e. on
    // a band of neighbouring image rows whose kh = 0/1/2 row patches (and both cout tiles of a pixel tile) are the
    // same lines - fetched once per L2 instead of once per workgroup several tiles apart.
    int t0, t1, tstep;
    if ((gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
        const int cs = (int)((long)x * n_tiles / 8), ce = (int)((long)(x + 1) * n_tiles / 8);
        t0 = cs + j;
        t1 = ce;
        tstep = per;
    } else {                                        // small grids (fewer tiles than CUs): contiguous ranges
        t0 = (int)((long)blockIdx.x * n_tiles / gridDim.x);
        t1 = (int)((long)(blockIdx.x + 1) * n_tiles / gridDim.x);
        tstep = 1;
    }
    if (t0 >= t1) return;
    const int cpt = p.Cin >> 6;                // 64-channel blocks

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    // ---------------- DMA lane geometry.  A piece fills 8 LDS rows x 128 B; lane -> row lane/8, chunk
    // position lane%8, which must hold logical chunk pos ^ ((row>>1)&7) (source-side swizzle).
    // Piece groups g = wave + 8q: (row>>1)&7 = ((wave&1)<<2) | (lane>>4).  Group 32 (rows 256..263,
    // fired by every wave with identical data): (row>>1)&7 = lane>>4.
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
    const int lchunk32 = (lane & 7) ^ (lane >> 4);
    const int rowpitch = p.Cin * 2;            // bytes per pixel
    const int khpitch = p.W * rowpitch;        // bytes per image row

    // X cursor: runs two ROW steps ahead of the MFMAs; one row step = patch (tile, cb, kh)
    int xt = t0, x_cb = 0, x_kh = 0, x_slot = 0;
    bool x_live = true;
    int xg = 0, xg32 = 0;                      // per-lane byte offset of (patch row, chunk) for kh = 0, cb = 0
    auto setup_x_tile = [&](int tile) {
        const int m0i = (tile / p.n_ctiles) * TP;
        xg = (m0i - p.W - 1 + wave * 8 + lrow) * rowpitch + lchunk * 16;
        xg32 = (m0i - p.W - 1 + 256 + lrow) * rowpitch + lchunk32 * 16;
    };
    // piece q (0..3: group wave + 8q, 4: group 32) of the patch under the X cursor
    auto x_piece = [&](int q) {
        if constexpr (ABL & 4) return;
        const int rs = x_kh * khpitch + (x_cb << 7);
        if (q < 4)
            dma16(xrsrc, smem + x_slot * XSLOT + (wave + 8 * q) * 1024, (unsigned)(xg + rs + q * 64 * rowpitch));
        else
            dma16(xrsrc, smem + x_slot * XSLOT + 32 * 1024, (unsigned)(xg32 + rs));
    };
    auto advance_x = [&]() {
        x_slot = x_slot == NXS - 1 ? 0 : x_slot + 1;
        if (++x_kh == 3) {
            x_kh = 0;
            if (++x_cb == cpt) {
                x_cb = 0;
                if ((xt += tstep) < t1) setup_x_tile(xt); else x_live = false;
            }
        }
    };

    // W cursor: runs two k-steps ahead; one stage = TC rows x 64 k of tap (kh, kw), block cb.
    // PF: the cursor is the position of the stage's kk = 0 quarter (one step AHEAD of its kk = 1..3 part);
    // voffC[] keeps the per-lane offsets the cursor had one step ago for the lanes carrying chunks 2..7.
    int wt = t0, w_cb = 0, w_tap = 0, w_slot = 0;
    bool w_live = true;                        // PF: the kk = 0 (ahead) cursor is inside this workgroup's tiles
    bool c_live = !PF;                         // PF: the stage's own step exists (= w_live one step ago)
    const bool is0 = lchunk < 2;               // this lane carries logical chunk 0/1 (kk = 0)
    unsigned woff[WI], voffC[WI], voffN[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) voffC[i] = voffN[i] = CONV_OOB;
    auto setup_w_tile = [&](int tile) {
        const int c0i = (tile % p.n_ctiles) * TC;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int co = c0i + (i * NW + wave) * 8 + lrow;
            woff[i] = co < p.Cout ? (unsigned)(co * p.Ktot + lchunk * 8) * 2u : CONV_OOB;
        }
    };
    // per-lane source offsets of the stage under the cursor (once per k-step, before its pieces are fired)
    auto prep_w = [&]() {
        const unsigned kadd = (unsigned)((w_tap * cpt + w_cb) << 7);
#pragma unroll
        for (int i = 0; i < WI; ++i) voffN[i] = (w_live && woff[i] != CONV_OOB) ? woff[i] + kadd : CONV_OOB;
    };
    auto w_piece = [&](int i) {
        if constexpr (ABL & 4) return;
        if constexpr (PF) {
            dma16(wrsrc, smem + OFF_W + w_slot * WSLOT + (i * NW + wave) * 1024, is0 ? voffN[i] : voffC[i]);
        } else {
            const unsigned kadd = (unsigned)((w_tap * cpt + w_cb) << 7);
            dma16(wrsrc, smem + OFF_W + w_slot * WSLOT + (i * NW + wave) * 1024, woff[i] != CONV_OOB ? woff[i] + kadd : CONV_OOB);
        }
    };
    auto advance_w = [&]() {
        w_slot = w_slot == NWS - 1 ? 0 : w_slot + 1;
        if constexpr (PF) {
#pragma unroll
            for (int i = 0; i < WI; ++i) voffC[i] = voffN[i];
            c_live = w_live;
            if (!w_live) return;
        }
        if (++w_tap == 9) {
            w_tap = 0;
            if (++w_cb == cpt) {
                w_cb = 0;
                if ((wt += tstep) < t1) setup_w_tile(wt); else w_live = false;
            }
        }
    };

    // ---------------- consumer geometry
    const int wave_p = wave / WC, wave_c = wave - wave_p * WC;
    const int prow0 = wave_p * (TP / WP), crow0 = wave_c * (TC / WC);
    const int fr = lane & 31, fh = lane >> 5;
    const int HoWo = p.Ho * p.Wo;
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)p.Wo;
    // B fragment addressing: pixel row prow0 + 32i + fr of the tile reads patch row (that + kw);
    // brow[kw][i] = row * 128, bxor[kw][i] = (fh ^ ((row>>1)&7)) << 4 (the kk part is xor'ed in later)
    int brow[3][MP], bxor[3][MP];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int i = 0; i < MP; ++i) {
            const int row = prow0 + i * 32 + fr + kw;
            brow[kw][i] = row * 128;
            bxor[kw][i] = (fh ^ ((row >> 1) & 7)) << 4;
        }
    int aoff[MC][4];                            // A fragment offsets inside a weight stage
#pragma unroll
    for (int j = 0; j < MC; ++j)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) aoff[j][kk] = lds_off(crow0 + j * 32 + fr, 2 * kk + fh);

    floatx16 acc[MP][MC];
    half8 bf[2][MP], af[2][MC];
    int bbase[MP];                              // per k-step: patch row address or the zero block
    auto read_frags = [&](int wsoff, int kw, int kk, int S) {
        if constexpr (ABL & 2) return;
#pragma unroll
        for (int i = 0; i < MP; ++i)
            bf[S][i] = *reinterpret_cast<const half8*>(smem + bbase[i] + (bxor[kw][i] ^ (kk << 5)));
#pragma unroll
        for (int j = 0; j < MC; ++j)
            af[S][j] = *reinterpret_cast<const half8*>(smem + wsoff + aoff[j][kk]);
    };
    auto mfma_group = [&](int S) {
        if constexpr (ABL & 1) {
#pragma unroll
            for (int i = 0; i < MP; ++i) asm volatile("" ::"v"(bf[S][i]));
#pragma unroll
            for (int j = 0; j < MC; ++j) asm volatile("" ::"v"(af[S][j]));
            return;
        }
#pragma unroll
        for (int i = 0; i < MP; ++i)
#pragma unroll
            for (int j = 0; j < MC; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[S][j], bf[S][i], acc[i][j], 0, 0, 0);
    };
    if constexpr (ABL & 2) {                   // fragments: anything finite, loaded once
#pragma unroll
        for (int S = 0; S < 2; ++S) {
#pragma unroll
            for (int i = 0; i < MP; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) bf[S][i][e] = (_Float16)(0.001f * (lane + e + i));
#pragma unroll
            for (int j = 0; j < MC; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) af[S][j][e] = (_Float16)(0.002f * (lane - e + j));
        }
    }

    // ---------------- per-tile epilogue parameters in LDS (see conv_mfma.hip)
    float* lds_bias = reinterpret_cast<float*>(smem + OFF_PAR);            // [9][TC]
    float* lds_slope = lds_bias + 9 * TC;                                   // [TC]
    constexpr int PPT = (9 * TC + NW * 64 - 1) / (NW * 64);
    float pb[PPT], ps = 0.f;
    const bool border = p.flags & FRP_FLAG_BORDER_BIAS;
    auto fetch_params = [&](int tile) {
        const int c0p = (tile % p.n_ctiles) * TC;
        const int nb = (border ? 9 : 1) * TC;
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int idx = t + q * NW * 64;
            const int cls = idx / TC, co = c0p + (idx - cls * TC);
            pb[q] = (idx < nb && co < p.Cout) ? p.bias[(long)cls * p.Cout + co] : 0.f;
        }
        if (p.act == FRP_ACT_PRELU && t < TC) ps = (c0p + t < p.Cout) ? p.slope[c0p + t] : 0.f;
    };
    auto store_params = [&]() {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int idx = t + q * NW * 64;
            if (idx < 9 * TC) lds_bias[idx] = pb[q];
        }
        if (t < TC) lds_slope[t] = ps;
    };

    // ---------------- prologue: zero block, row patches 0 and 1, weight stages 0 and 1 (a tile has at
    // least 3 row steps and 9 k-steps, so they always exist)
    stamp(p.stamps, 0);
    if (t < 64) reinterpret_cast<unsigned*>(smem + OFF_Z)[t] = 0u;
    __syncthreads();
    setup_x_tile(t0);
    setup_w_tile(t0);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int q = 0; q < 5; ++q) x_piece(q);
        advance_x();
    }
    // PF: three shifted stages (-1: only the kk = 0 quarter of step 0; 0; 1), stage s in slot (s + 1) % 3
#pragma unroll
    for (int s = 0; s < (PF ? 3 : 2); ++s) {
        if constexpr (PF) prep_w();
#pragma unroll
        for (int i = 0; i < WI; ++i) w_piece(i);
        advance_w();
    }
    int cx_slot = 0, cw_slot = PF ? 1 : 0;     // consumer ring positions
    int last_cnt = -1;                         // pieces this wave fired in the previous k-step (-1: drain)
    const bool out32 = p.flags & FRP_FLAG_OUT_F32;
    const bool up2 = p.flags & FRP_FLAG_RES_UP2;
    const bool has_res = p.res != nullptr;
    const bool prelu = p.act == FRP_ACT_PRELU;
    const bool relu = p.act == FRP_ACT_RELU;

    // ---------------- epilogue of tile ct (as in conv_mfma.hip): bias / border-class bias from the
    // LDS parameter cache, residual, activation in fp32, 16-byte fp16 stores after a half-wave
    // exchange; interior tiles take the unpredicated copy.
    // The residual of a tile is requested one k-step EARLY (at the start of the tile's last k-step, right
    // after that step's counted wait, so the loads sit behind nothing the ring waits for): on the big
    // detector maps it comes from HBM, and an epilogue that issued the loads itself sat out their latency.
    uint4 rres[MP][MC][2];
    auto issue_residual_loads = [&](int m0, int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MP; ++i) {
            const int mraw = m0 + prow0 + i * 32 + fr;
            const int m = mraw < p.M ? mraw : 0;
            long ridx = (long)m * p.Cout;
            if (up2) {
                int n, rem, oy, ox;
                fast_divmod(m, HoWo, inv_howo, n, rem);
                fast_divmod(rem, p.Wo, inv_wo, oy, ox);
                ridx = (((long)n * p.Hr + (oy >> 1)) * p.Wr + (ox >> 1)) * p.Cout;
            }
#pragma unroll
            for (int j = 0; j < MC; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    // 16-byte loads in the STORE layout (couts 16q + 8*fh .. +7 of this pixel); clamped, always valid
                    const int co = c0 + crow0 + j * 32 + 16 * q + 8 * fh;
                    rres[i][j][q] = *reinterpret_cast<const uint4*>(p.res + ridx + (co < p.Cout ? co : 0));
                }
        }
    };

    auto epilogue_body = [&](auto FULL_T, int m0, int c0) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(FULL_T)::value;
        half4 r4[MP][MC][4];
        bool mok[MP];
        long obase[MP];
        int cls[MP];
#pragma unroll
        for (int i = 0; i < MP; ++i) {
            const int mraw = m0 + prow0 + i * 32 + fr;
            mok[i] = FULL || mraw < p.M;
            const int m = mok[i] ? mraw : 0;
            cls[i] = 0;
            if (border) {
                int n, rem, oy, ox;
                fast_divmod(m, HoWo, inv_howo, n, rem);
                fast_divmod(rem, p.Wo, inv_wo, oy, ox);
                cls[i] = ((oy == 0) ? 0 : (oy == p.Ho - 1) ? 2 : 1) * 3 + ((ox == 0) ? 0 : (ox == p.Wo - 1) ? 2 : 1);
            }
            obase[i] = (long)m * p.Cout;
            if (has_res) {
                // the same half-wave exchange as for the stores (it is its own inverse) turns the 16-byte
                // store-layout chunks into the accumulator layout
#pragma unroll
                for (int j = 0; j < MC; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        uint4 rr = rres[i][j][q];
                        swap_halves(rr.x, rr.z);
                        swap_halves(rr.y, rr.w);
                        union { unsigned u[2]; half4 h; } lo, hi;
                        lo.u[0] = rr.x; lo.u[1] = rr.y; hi.u[0] = rr.z; hi.u[1] = rr.w;
                        r4[i][j][2 * q] = lo.h;
                        r4[i][j][2 * q + 1] = hi.h;
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < MP; ++i) {
#pragma unroll
            for (int j = 0; j < MC; ++j) {
                floatx4 v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cl = crow0 + j * 32 + 8 * g + 4 * fh;
                    const floatx4 b4 = *reinterpret_cast<const floatx4*>(lds_bias + cls[i] * TC + cl);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] = acc[i][j][4 * g + e] + b4[e];
                    if (has_res) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[g][e] += (float)r4[i][j][g][e];
                    }
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[g][e] = fmaxf(v[g][e], 0.f);
                    } else if (prelu) {
                        const floatx4 s4 = *reinterpret_cast<const floatx4*>(lds_slope + cl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[g][e] = v[g][e] > 0.f ? v[g][e] : v[g][e] * s4[e];
                    }
                }
                if (out32) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = c0 + crow0 + j * 32 + 8 * g + 4 * fh;
                        if (FULL || (mok[i] && co < p.Cout))
                            *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(p.out) + obase[i] + co) = v[g];
                    }
                } else {
                    union { half4 h; unsigned u[2]; } pk[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk[g].h[e] = (_Float16)v[g][e];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        swap_halves(pk[2 * q].u[0], pk[2 * q + 1].u[0]);
                        swap_halves(pk[2 * q].u[1], pk[2 * q + 1].u[1]);
                        const int co = c0 + crow0 + j * 32 + 16 * q + 8 * fh;
                        if (FULL || (mok[i] && co < p.Cout))
                            *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(p.out) + obase[i] + co) =
                                make_uint4(pk[2 * q].u[0], pk[2 * q].u[1], pk[2 * q + 1].u[0], pk[2 * q + 1].u[1]);
                    }
                }
            }
        }
    };
    auto run_epilogue = [&](int m0, int c0) __attribute__((always_inline)) {
        if (m0 + TP <= p.M && c0 + TC <= p.Cout) epilogue_body(std::true_type{}, m0, c0); else epilogue_body(std::false_type{}, m0, c0);
    };

    stamp(p.stamps, 1);
    for (int ct = t0; ct < t1; ct += tstep) {
#pragma unroll
        for (int i = 0; i < MP; ++i)
#pragma unroll
            for (int j = 0; j < MC; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int ptile = ct / p.n_ctiles;
        const int m0 = ptile * TP;
        const int c0 = (ct - ptile * p.n_ctiles) * TC;
        // 9-bit tap validity of this lane's pixels (bit kh*3+kw), 0 for rows beyond M
        unsigned tapmask[MP];
#pragma unroll
        for (int i = 0; i < MP; ++i) {
            const int m = m0 + prow0 + i * 32 + fr;
            unsigned mask = 0;
            if (m < p.M) {
                int n, rem, oy, ox;
                fast_divmod(m, HoWo, inv_howo, n, rem);
                fast_divmod(rem, p.Wo, inv_wo, oy, ox);
                const unsigned ym = (oy > 0 ? 1u : 0u) | 2u | (oy < p.H - 1 ? 4u : 0u);
                const unsigned xm = (ox > 0 ? 1u : 0u) | 2u | (ox < p.W - 1 ? 4u : 0u);
#pragma unroll
                for (int d = 0; d < 3; ++d) mask |= ((ym >> d) & 1u) ? (xm << (3 * d)) : 0u;
            }
            tapmask[i] = mask;
        }
        if (ct == t0) stamp(p.stamps, 2);

        // One k-step: wait for this wave's share of the stage, barrier (every wave's share landed;
        // every wave is done with the slots the next issues overwrite), 4 x (fragment reads, MFMA
        // group) with this step's DMA pieces fired in between.
        // PF, top of a tile: request the kk = 0 fragments of its first k-step (weights: last quarter of the
        // previous stage's slot; pixels: the resident first row patch).  Both are visible since the previous
        // step's barrier; the very first tile needs the prologue's DMA landed and a barrier of its own.
        if constexpr (PF) {
            if (ct == t0) { wait_vmcnt<0>(); __builtin_amdgcn_s_barrier(); }
            const int xs0 = cx_slot * XSLOT;
#pragma unroll
            for (int i = 0; i < MP; ++i) bbase[i] = (tapmask[i] & 1u) ? xs0 + brow[0][i] : OFF_Z + (brow[0][i] & 128);
            read_frags(OFF_W + (cw_slot == 0 ? NWS - 1 : cw_slot - 1) * WSLOT, 0, 0, 0);
        }

#define ROWS_WAIT()                                                                                   \
    do {                                                                                              \
        if (last_cnt == WI + 2) wait_vmcnt<WI + 2>();                                                 \
        else if (last_cnt == WI + 1) wait_vmcnt<WI + 1>();                                            \
        else if (last_cnt == WI) wait_vmcnt<WI>();                                                    \
        else wait_vmcnt<0>();                                                                         \
    } while (0)
#define ROWS_STEP(KW, FIRST, LASTSTEP)                                                                \
    do {                                                                                              \
        ROWS_WAIT();                                                                                  \
        /* PF: the prefetched fragments are in registers, so the slot they came from may be refilled */ \
        if constexpr (PF) wait_lgkmcnt0();                                                            \
        if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();                                       \
        const int xsoff = cx_slot * XSLOT;                                                            \
        const int wsoff = OFF_W + cw_slot * WSLOT;                                                    \
        if constexpr (!PF) {                                                                          \
            _Pragma("unroll") for (int i = 0; i < MP; ++i)                                            \
                bbase[i] = ((tapmask[i] >> (kh * 3 + (KW))) & 1u) ? xsoff + brow[KW][i] : OFF_Z + (brow[KW][i] & 128); \
        }                                                                                             \
        if (FIRST) fetch_params(ct);                                                                  \
        if ((LASTSTEP) && has_res) issue_residual_loads(m0, c0);                                      \
        if constexpr (!PF) read_frags(wsoff, KW, 0, 0);                                               \
        int cnt = 0;                                                                                  \
        const bool wfire = PF ? (c_live || w_live) : w_live;                                          \
        if constexpr (PF) prep_w();                                                                   \
        /* weight pieces first: their slack is two k-steps at best, the row patches have six */      \
        read_frags(wsoff, KW, 1, 1); mfma_group(0);                                                   \
        if (wfire) { w_piece(0); ++cnt; }                                                             \
        if (WI == 2 && wfire) { w_piece(WI - 1); ++cnt; }                                             \
        read_frags(wsoff, KW, 2, 0); mfma_group(1);                                                   \
        if (x_live) { x_piece((KW) * 2 < 4 ? (KW) * 2 : 4); ++cnt; }                                  \
        read_frags(wsoff, KW, 3, 1); mfma_group(0);                                                   \
        if ((KW) < 2 && x_live) { x_piece((KW) * 2 + 1); ++cnt; }                                     \
        if constexpr (PF) {                                                                           \
            if (!(LASTSTEP)) {                                                                        \
                /* kk = 0 of the next k-step: same weight slot (chunks 0/1), next tap of the patch ring */ \
                constexpr int NKW = (KW) == 2 ? 0 : (KW) + 1;                                         \
                const int nkh = (KW) == 2 ? (kh == 2 ? 0 : kh + 1) : kh;                              \
                const int nxs = (KW) == 2 ? (cx_slot == NXS - 1 ? 0 : cx_slot + 1) * XSLOT : xsoff;   \
                _Pragma("unroll") for (int i = 0; i < MP; ++i)                                        \
                    bbase[i] = ((tapmask[i] >> (nkh * 3 + NKW)) & 1u) ? nxs + brow[NKW][i] : OFF_Z + (brow[NKW][i] & 128); \
                read_frags(wsoff, NKW, 0, 0);                                                         \
            }                                                                                         \
        }                                                                                             \
        mfma_group(1);                                                                                \
        if ((KW) == 2 && x_live) advance_x();                                                         \
        if (wfire) advance_w();                                                                       \
        last_cnt = cnt;                                                                               \
        if (FIRST) store_params();                                                                    \
        cw_slot = cw_slot == NWS - 1 ? 0 : cw_slot + 1;                                               \
        if ((KW) == 2) cx_slot = cx_slot == NXS - 1 ? 0 : cx_slot + 1;                                \
    } while (0)

        for (int cb = 0; cb < cpt; ++cb) {
            for (int kh = 0; kh < 3; ++kh) {
                const bool first = (cb == 0 && kh == 0);
                // right after an epilogue its stores are pending too (vmcnt counts them, in issue
                // order, behind the DMAs): drain everything there
                if (first && ct != t0) last_cnt = -1;
                ROWS_STEP(0, first, false);
                if (first && ct == t0) stamp(p.stamps, 3);
                ROWS_STEP(1, false, false);
                ROWS_STEP(2, false, cb == cpt - 1 && kh == 2);
            }
        }
#undef ROWS_WAIT
#undef ROWS_STEP
        if (ct == t0) stamp(p.stamps, 4);

        run_epilogue(m0, c0);
        if (ct == t0) stamp(p.stamps, 5);
    }
    stamp(p.stamps, 6);
}

template <int TC, int WP, int WC, bool PF, int ABL = 0>
static hipError_t launch_rows_cfg(const ConvParams& p0, hipStream_t stream) {
    ConvParams p = p0;
    p.n_ptiles = (p.M + 255) / 256;
    p.n_ctiles = (p.Cout + TC - 1) / TC;
    const int lds = 3 * 264 * 128 + 3 * TC * 128 + 256 + 10 * TC * 4;
    static bool attr_set[64] = {};
    auto kern = conv3x3_rows_kernel<TC, WP, WC, PF, ABL>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const long ntiles = (long)p.n_ptiles * p.n_ctiles;
    if (ntiles <= 0 || ntiles > 0x7fffffffL) return hipErrorInvalidValue;
    const int ncu = device_cu_count(dev);
    if (ncu <= 0) return hipErrorInvalidDevice;
    const unsigned grid = (unsigned)(ntiles < ncu ? ntiles : ncu);     // persistent: one workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_conv3x3_rows(const ConvParams& p, hipStream_t stream) {
    if (!conv3x3_rows_eligible(p)) return hipErrorInvalidValue;
    if (p.out2) return launch_conv3x3_lean(p, stream);   // the fp8-copy epilogue lives in the static-loop generation only
    // dbg bits 2..5: timing-only ablations of the 128-cout kernel (wrong results; conv_bench only)
    if (p.Cout > 64) switch ((p.dbg >> 2) & 15) {
        case 1: return launch_rows_cfg<128, 4, 2, false, 1>(p, stream);
        case 2: return launch_rows_cfg<128, 4, 2, false, 2>(p, stream);
        case 4: return launch_rows_cfg<128, 4, 2, false, 4>(p, stream);
        case 8: return launch_rows_cfg<128, 4, 2, false, 8>(p, stream);
        case 6: return launch_rows_cfg<128, 4, 2, false, 6>(p, stream);
        case 14: return launch_rows_cfg<128, 4, 2, false, 14>(p, stream);
        case 5: return launch_rows_cfg<128, 4, 2, false, 5>(p, stream);
        case 13: return launch_rows_cfg<128, 4, 2, false, 13>(p, stream);
        case 12: return launch_rows_cfg<128, 4, 2, false, 12>(p, stream);
        case 10: return launch_rows_cfg<128, 4, 2, false, 10>(p, stream);
        default: break;
    }
    // dbg bit 2 (value 2): the pre-prefetch k-step (A/B runs, bit-identical results)
    if (p.dbg & 2) {
        if (p.Cout > 64) return launch_rows_cfg<128, 4, 2, false>(p, stream);
        return launch_rows_cfg<64, 8, 1, false>(p, stream);
    }
    if (p.dbg & 64) {                              // first-generation kernel with the cross-barrier prefetch
        if (p.Cout > 64) return launch_rows_cfg<128, 4, 2, true>(p, stream);
        return launch_rows_cfg<64, 8, 1, true>(p, stream);
    }
    return launch_conv3x3_lean(p, stream);       // second generation: static k-loop (conv3x3_lean.hip)
}

}  // namespace frp
