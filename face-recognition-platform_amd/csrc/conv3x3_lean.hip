// K2b (second generation): 3x3 stride-1 convolution with ROW PATCHES and a STATIC k-loop.
//
// Same data path as conv3x3_rows.hip (row patches for the pixel operand, per-tap weight stages, LDS-DMA rings that
// run across tiles, zero padding decided at fragment-read time, identical accumulation order and epilogue - the two
// kernels agree bit for bit), but the k-loop carries no run-time bookkeeping:
//
//   * the nine taps of a 64-channel block are one fully unrolled body.  A tile consumes 9*cpt weight stages and
//     3*cpt row patches - multiples of the ring depth 3 - so every ring slot, tap index and vmcnt count inside the
//     body is a compile-time constant (the first generation advanced cursors, slot counters and "live" flags with
//     ~80 scalar instructions and a dozen branches per k-step: measured 17 % of the loop against the same
//     MFMA / ds_read / DMA stream without them, tools/kstep_lab.py);
//   * the DMA stream never stops or branches: past the last tile of a workgroup the per-lane source offsets are
//     out of range, the hardware range check writes zeros into slots nobody reads any more;
//   * weight pieces take the (channel block, tap) part of their source address in the SGPR offset operand of
//     buffer_load ... lds (not range-checked, always inside the row), so their VGPR offset changes once per tile.
//
// Replaces the same reference calls as conv_mfma.hip (face_recognition.face_locations / face_encodings,
// backend/app/routes/camera.py:232,237, backend/app/services/face_service.py:156,179).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "frp_internal.h"
#include "conv_common.h"

namespace frp {

// one LDS-DMA piece with a scalar byte offset added to the source address (not part of the range check)
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_base, unsigned voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_base, 16, voffset, soffset, 0, 0);
}

// F8 (BASELINE config 5, "fp8 ArcFace weights (CDNA4 fp8 MFMA)"): activations and weights are OCP FP8 E4M3 bytes, the
// matrix instruction is the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales.  A 128-byte LDS row is
// then 128 channels, a k-step one tap x 128 channels = 2 MFMAs of K = 64 per 32x32 block: the same DMA pieces, LDS reads
// and matrix cycles per step as the fp16 kernel for twice the FLOPs - and the chip holds a higher clock on fp8 operands
// (tools/kstep_lab.py: 3.16 vs 1.27 PFLOP/s for the k-step with barrier and DMA).  Per-cout weight scales (and the input
// tensor's scale) multiply the fp32 accumulator in the epilogue; outputs leave as fp16, as fp8, or as both.
typedef int intx8 __attribute__((ext_vector_type(8)));
//
// TP = 512 (Cout <= 64): at 256 pixels a 64-cout tile leaves each wave a 32 x 64 tile - 1.5 fragment reads per MFMA, and
// the LDS read port, not the matrix pipe, sets the pace.  512 pixels x 64 couts gives every wave the 64 x 64 tile of the
// 128-cout configuration (1.0 reads per MFMA).  Two 65 KiB patches are all the LDS holds then, so the patch ring has
// TWO slots: a row's patch is fired ONE row ahead (during the first two k-steps of the row before it: 9 pieces per wave,
// 5 + 4), its slot is the running patch parity (a scalar, not a constant), and the counted waits change accordingly.
//
// TP = 128, TC = 64 (small maps and small batches: fewer 256 x 128 tiles than CUs): quarter tiles, 32 x 32 per wave, 78 KiB of
// LDS and <= 128 registers, so TWO workgroups share a CU - four times the tiles on twice the slots.  2 fragment reads per MFMA,
// which does not matter where a launch is one round of latency-bound k-steps; same k order, bit-identical results.
//
// TCU (used couts of the tile, default TC): the detector's head outputs have 30 (32) couts - half of a 64-cout tile would
// multiply zero rows.  TCU = 32 keeps the 64-row weight stage in LDS (rows beyond Cout arrive as zeros by the range check,
// no HBM traffic) and drops the second 32-cout MFMA block: half the MFMAs, 3 instead of 4 fragment reads per sub-step.
template <int TC, int WP, int WC, bool F8, int TP, int TCU = TC>
__global__ __launch_bounds__(512, TP == 128 ? 4 : 2) void conv3x3_lean_kernel(ConvParams p_in) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    ConvParams p = p_in;
    if (p.n_dev) {                             // image count known on the device only (threshold mode)
        int n = *p.n_dev;
        n = n < 0 ? 0 : (n > p.N ? p.N : n);
        p.M = n * p.Ho * p.Wo;
        p.n_ptiles = (p.M + TP - 1) / TP;
    }
    constexpr int NW = 8;
    constexpr int NXS = TP == 512 ? 2 : 3;    // patch ring slots
    constexpr int NQ = TP / 64;               // row groups (of 8 rows) per wave; one more group is shared
    constexpr int XROWS = TP + 8;             // TP/8 + 1 groups of 8 rows; rows 0..TP+1 are read
    constexpr int XSLOT = XROWS * 128;        // 33,792 / 66,560 B (multiples of 256)
    constexpr int WSLOT = TC * 128;
    constexpr int WI = TC / 8 / NW;           // weight DMA pieces per wave per k-step (2 or 1)
    constexpr int MP = TP / WP / 32, MC = TCU / WC / 32;
    static_assert(TCU <= TC && TCU % (WC * 32) == 0, "used couts: whole 32-cout blocks per wave");
    constexpr int OFF_W = NXS * XSLOT;
    constexpr int OFF_Z = OFF_W + 3 * WSLOT;  // 256 zero bytes (256-aligned)
    constexpr int OFF_PAR = OFF_Z + 256;
    static_assert(WP * WC == NW && (WI == 1 || WI == 2) && MP * MC <= 4 && (OFF_Z & 255) == 0 &&
                  (TP == 256 || (TP == 128 && WI == 1 && !F8) || (TP == 512 && WI == 1 && !F8)), "layout");

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    const int n_tiles = p.n_ptiles * p.n_ctiles;
    const int G = (TP == 128 && p.n_workers) ? p.n_workers : (int)gridDim.x;      // workgroups that walk tiles
    if constexpr (TP == 128) {
        if ((int)blockIdx.x >= G) { conv_prefetch_weights(p, (int)blockIdx.x - G, (int)gridDim.x - G, 512); return; }
    }
    // XCD-interleaved tile walk (see conv3x3_rows.hip)
    int t0, t1, tstep;
    if ((G & 7) == 0) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, per = G >> 3;
        const int cs = (int)((long)x * n_tiles / 8), ce = (int)((long)(x + 1) * n_tiles / 8);
        t0 = cs + j;
        t1 = ce;
        tstep = per;
    } else {
        t0 = (int)((long)blockIdx.x * n_tiles / G);
        t1 = (int)((long)(blockIdx.x + 1) * n_tiles / G);
        tstep = 1;
    }
    if (t0 >= t1) return;
    constexpr int ES = F8 ? 1 : 2;             // bytes per element
    const int cpt = (p.Cin * ES) >> 7;         // channel blocks of one 128-byte LDS row (64 fp16 / 128 fp8 channels)
    const int cin2 = p.Cin * ES;               // bytes per pixel = bytes per tap in a weight row

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    // ---------------- DMA lane geometry (as conv3x3_rows.hip): a piece fills 8 LDS rows x 128 B; lane -> row lane/8,
    // chunk position lane%8 holding logical chunk pos ^ ((row>>1)&7) (source-side swizzle)
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
    const int lchunk32 = (lane & 7) ^ (lane >> 4);
    const int khpitch = p.W * cin2;            // bytes per image row
    constexpr int DEAD = (int)0x80000000;      // per-lane offset that stays out of range whatever is added to it

    // Source descriptors of one tile.  X: per-lane byte offset of (patch row, chunk) for kh = 0, cb = 0 (pieces 0..3
    // and the 33rd row group); W: per-lane byte offset of (cout row, chunk) at tap 0, cb 0 - or DEAD.
    struct Desc { int xg, xg32; unsigned woff[WI]; };
    auto make_desc = [&](int tile, Desc& d) {
        if (tile >= t1) {
            d.xg = d.xg32 = DEAD;
#pragma unroll
            for (int i = 0; i < WI; ++i) d.woff[i] = CONV_OOB;
            return;
        }
        const int pt = tile / p.n_ctiles;
        const int m0i = pt * TP, c0i = (tile - pt * p.n_ctiles) * TC;
        d.xg = (m0i - p.W - 1 + wave * 8 + lrow) * cin2 + lchunk * 16;
        d.xg32 = (m0i - p.W - 1 + TP + lrow) * cin2 + lchunk32 * 16;
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int co = c0i + (i * NW + wave) * 8 + lrow;
            d.woff[i] = co < p.Cout ? (unsigned)(co * p.Ktot * ES + lchunk * 16) : CONV_OOB;
        }
    };
    // piece q (0..NQ-1: row group wave + 8q, NQ: the last group, TP/8) of patch (cb byte offset cbs, kernel row kh) into
    // ring slot `slot`
    auto x_piece = [&](const Desc& d, int q, int kh, int cbs, int slot) {
        const int rs = kh * khpitch + cbs;
        if (q < NQ) dma16(xrsrc, smem + slot * XSLOT + (wave + 8 * q) * 1024, (unsigned)(d.xg + rs + q * 64 * cin2));
        else dma16(xrsrc, smem + slot * XSLOT + (TP / 8) * 1024, (unsigned)(d.xg32 + rs));
    };
    // weight stage (tap, cb byte offset cbs) into ring slot `slot`: this wave's WI pieces
    auto w_stage = [&](const Desc& d, int tap, int cbs, int slot) {
#pragma unroll
        for (int i = 0; i < WI; ++i)
            dma16s(wrsrc, smem + OFF_W + slot * WSLOT + (i * NW + wave) * 1024, d.woff[i], tap * cin2 + cbs);
    };

    // ---------------- consumer geometry
    const int wave_p = wave / WC, wave_c = wave - wave_p * WC;
    const int prow0 = wave_p * (TP / WP), crow0 = wave_c * (TC / WC);
    const int fr = lane & 31, fh = lane >> 5;
    const int HoWo = p.Ho * p.Wo;
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)p.Wo;
    // B fragment addressing: pixel row prow0 + 32i + fr of the tile reads patch row (that + kw), chunk 2kk + fh at
    // 16-byte position (2kk + fh) ^ ((row >> 1) & 7).  pv[kw][i] = row * 128 + ((fh ^ ((row>>1)&7)) << 4) is the kk = 0
    // address inside a patch slot, zv[kw][i] the same bank offset inside the 256-byte zero block (a lane whose tap lies
    // outside the image reads zeros there without adding bank conflicts); kk flips address bits 5..6 (xor, not add,
    // so it cannot ride in the instruction's immediate offset).
    int pv[3][MP], zv[3][MP];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int i = 0; i < MP; ++i) {
            const int row = prow0 + i * 32 + fr + kw;
            const int bx = (((F8 ? 2 : 1) * fh) ^ ((row >> 1) & 7)) << 4;
            pv[kw][i] = row * 128 + bx;
            zv[kw][i] = OFF_Z + ((row * 128) & 128) + bx;
        }
    // fp16: kk = 0..3, a fragment is chunk 2kk + fh (16 B).  fp8: kk = 0..1, a fragment is chunks 4kk + 2fh and + 1 (32 B:
    // the hardware pairs register slot s of A with slot s of B, so any lane -> k assignment works as long as both operands
    // use the same one).  The kk = 0 chunk is (F8 ? 2 : 1) * fh; further kk flip address bits 5..6 (fp16) / bit 6 (fp8).
    constexpr int NKK = F8 ? 2 : 4;
    constexpr int KKSH = F8 ? 6 : 5;
    int aoff[MC][NKK];                          // A fragment offsets inside a weight stage (+ OFF_W)
#pragma unroll
    for (int j = 0; j < MC; ++j)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) aoff[j][kk] = OFF_W + lds_off(crow0 + j * 32 + fr, F8 ? 4 * kk + 2 * fh : 2 * kk + fh);

    floatx16 acc[MP][MC];
    using frag_t = typename std::conditional<F8, intx8, half8>::type;
    constexpr int NFS = F8 ? 1 : 2;             // fragment sets (fp8 fragments are 8 registers each: one set, the SIMD's
    frag_t bf[NFS][MP], af[NFS][MC];            // partner wave covers the read latency)
    int bbase[MP];                              // per k-step: kk = 0 address of this lane's B rows (patch or zero block)
    auto lds_frag = [&](int addr) -> frag_t {
        if constexpr (F8) {
            const uint4 lo = *reinterpret_cast<const uint4*>(smem + addr);
            const uint4 hi = *reinterpret_cast<const uint4*>(smem + (addr ^ 16));
            return intx8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        } else {
            return *reinterpret_cast<const half8*>(smem + addr);
        }
    };
    auto read_frags = [&](int wslot, int kw, int kk, int S) {
#pragma unroll
        for (int i = 0; i < MP; ++i) bf[S][i] = lds_frag(bbase[i] ^ (kk << KKSH));
#pragma unroll
        for (int j = 0; j < MC; ++j) af[S][j] = lds_frag(aoff[j][kk] + wslot * WSLOT);
    };
    auto mfma_group = [&](int S) {
#pragma unroll
        for (int i = 0; i < MP; ++i)
#pragma unroll
            for (int j = 0; j < MC; ++j) {
                if constexpr (F8) {
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[S][j], bf[S][i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                } else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[S][j], bf[S][i], acc[i][j], 0, 0, 0);
            }
    };

    // an empty volatile asm that "modifies" the accumulators: volatile asms keep their order, so the MFMAs before it stay
    // in their k-step
    auto pin_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MP; ++i)
#pragma unroll
            for (int j = 0; j < MC; ++j) asm volatile("" : "+v"(acc[i][j]));
    };

    // ---------------- per-tile epilogue parameters in LDS (see conv_mfma.hip)
    float* lds_bias = reinterpret_cast<float*>(smem + OFF_PAR);            // [9][TC]
    float* lds_slope = lds_bias + 9 * TC;                                   // [TC]
    float* lds_wscale = lds_slope + TC;                                     // [TC] (fp8: weight scale x input scale)
    constexpr int PPT = (9 * TC + NW * 64 - 1) / (NW * 64);
    float pb[PPT], ps = 0.f, pw = 0.f;
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
        if (F8 && t < TC) pw = (c0p + t < p.Cout) ? p.wscale[c0p + t] * p.in_scale : 0.f;
    };
    auto store_params = [&]() {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int idx = t + q * NW * 64;
            if (idx < 9 * TC) lds_bias[idx] = pb[q];
        }
        if (t < TC) lds_slope[t] = ps;
        if (F8 && t < TC) lds_wscale[t] = pw;
    };

    // ---------------- prologue: zero block; row patches (0,0), (0,1) and weight stages tap 0, 1 of the first tile
    stamp(p.stamps, 0);
    if (t < 64) reinterpret_cast<unsigned*>(smem + OFF_Z)[t] = 0u;
    __syncthreads();
    Desc cur, nt;
    make_desc(t0, cur);
    if constexpr (NXS == 3) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int q = 0; q <= NQ; ++q) x_piece(cur, q, r, 0, r);
            w_stage(cur, r, 0, r);
        }
    } else {                                   // two slots: row 0 now, row 1 during row 0's k-steps
#pragma unroll
        for (int q = 0; q <= NQ; ++q) x_piece(cur, q, 0, 0, 0);
        w_stage(cur, 0, 0, 0);
        w_stage(cur, 1, 0, 1);
    }
    int xs = 0;                                // NXS == 2: byte offset of the patch slot the running row reads
    const bool has_res = p.res != nullptr;

    // ---------------- epilogue of a tile (as conv3x3_rows.hip): bias / border-class bias from the LDS parameter cache,
    // residual (requested one k-step early), activation in fp32, 16-byte fp16 stores after a half-wave exchange
    // (fr_e / fh_e: copies of the lane coordinates made opaque per tile, so that the epilogue's address arithmetic is
    // not hoisted to the top of the kernel and kept live - or spilled - across the whole k-loop)
    uint4 rres[MP][MC][2];
    int fr_e = fr, fh_e = fh;
    auto issue_residual_loads = [&](int m0, int c0) __attribute__((always_inline)) {
        conv_residual_loads<MP, MC>(p, rres, m0, c0, prow0, crow0, fr_e, fh_e, HoWo, inv_howo, inv_wo);
    };
    auto run_epilogue = [&](int m0, int c0) __attribute__((always_inline)) {
        if constexpr (F8)
            conv_epilogue8<MP, MC, TC, true>(p, acc, rres, lds_bias, lds_slope, lds_wscale, m0, c0, TP, prow0, crow0, fr_e, fh_e, HoWo, inv_howo, inv_wo);
        else if (p.out2)
            conv_epilogue8<MP, MC, TC, false>(p, acc, rres, lds_bias, lds_slope, lds_wscale, m0, c0, TP, prow0, crow0, fr_e, fh_e, HoWo, inv_howo, inv_wo);
        else
            conv_epilogue<MP, MC, TC>(p, acc, rres, lds_bias, lds_slope, m0, c0, TP, prow0, crow0, fr_e, fh_e, HoWo, inv_howo, inv_wo);
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
        make_desc(ct + tstep, nt);             // the tile after this one (DEAD past the end)
        asm volatile("" : "+v"(fr_e), "+v"(fh_e));
        if (ct == t0) stamp(p.stamps, 2);

        // One k-step, everything about it static: tap KH*3+KW reads weight slot tap % 3 and patch slot KH; it fires the
        // weight stage two taps ahead and - spread over the three steps of a kernel row - the patch two rows ahead.
        // `nx` = the descriptor of what lies beyond this channel block (the next block of this tile, the next tile,
        // or nothing), `ncbs` its channel-block byte offset.
#define LEAN_STEP(KH, KW)                                                                                       \
    do {                                                                                                        \
        constexpr int TAP = (KH) * 3 + (KW);                                                                    \
        if (TAP == 0) { if (cb == 0) wait_vmcnt<0>(); else wait_vmcnt<WI + 1>(); }                              \
        else if ((KW) == 0) wait_vmcnt<WI + 1>();                                                               \
        else wait_vmcnt<WI + (NQ == 4 ? 2 : 1)>();          /* = the pieces the previous step issued */           \
        retire_lds_reads();                                                                                     \
        __builtin_amdgcn_s_barrier();                                                                           \
        _Pragma("unroll") for (int i = 0; i < MP; ++i)                                                          \
            bbase[i] = ((tapmask[i] >> TAP) & 1u) ? (KH) * XSLOT + pv[KW][i] : zv[KW][i];                       \
        if (TAP == 0 && cb == 0) fetch_params(ct);                                                              \
        /* fp8: left alone, the optimiser sinks all 72 MFMAs of the unrolled body below its last barrier (they touch   \
           registers only) and the fragments of nine steps are spilled: pin_acc() ties each group to its step */  \
        read_frags(TAP % 3, KW, 0, 0);                                                                          \
        if constexpr (F8) { mfma_group(0); pin_acc(); }                                                         \
        read_frags(TAP % 3, KW, 1, F8 ? 0 : 1);                                                                 \
        if constexpr (!F8) mfma_group(0);                                                                       \
        if (TAP < 7) w_stage(cur, TAP + 2, cbs, (TAP + 2) % 3); else w_stage(nx, TAP - 7, ncbs, (TAP + 2) % 3); \
        if constexpr (!F8) { read_frags(TAP % 3, KW, 2, 0); mfma_group(1); }                                    \
        /* a row patch is NQ + 1 pieces per wave: 5 spread 2 + 2 + 1 over the row's steps (TP = 256), 3 spread 1 + 1 + 1 (128) */ \
        if ((KH) == 0) x_piece(cur, NQ == 4 ? ((KW) * 2 < 4 ? (KW) * 2 : 4) : (KW), 2, cbs, 2);                 \
        else x_piece(nx, NQ == 4 ? ((KW) * 2 < 4 ? (KW) * 2 : 4) : (KW), (KH) - 1, ncbs, ((KH) + 2) % 3);       \
        if constexpr (!F8) { read_frags(TAP % 3, KW, 3, 1); mfma_group(0); }                                    \
        if (NQ == 4 && (KW) < 2) {                                                                              \
            if ((KH) == 0) x_piece(cur, (KW) * 2 + 1, 2, cbs, 2);                                               \
            else x_piece(nx, (KW) * 2 + 1, (KH) - 1, ncbs, ((KH) + 2) % 3);                                     \
        }                                                                                                       \
        /* the residual of the tile is requested under the last MFMA group (fragment set 0 is dead by then) */   \
        if (TAP == 8 && cb == cpt - 1 && has_res) issue_residual_loads(m0, c0);                                 \
        mfma_group(F8 ? 0 : 1);                                                                                 \
        if constexpr (F8) pin_acc();                                                                            \
        if (TAP == 0 && cb == 0) store_params();                                                                \
    } while (0)

        // Two-slot schedule (TP = 512).  Row KH reads slot xs and fires the NEXT row's patch (row KH + 1 of this channel
        // block, or row 0 of what lies beyond it) into the other slot: 5 pieces in its first k-step, 4 in its second.
        // In-flight budget at the barrier of a step (vmcnt retires in issue order; a step issues its weight stage first):
        //   kw = 0 needs the patch, last fired in the previous kw = 1 step - behind it only the kw = 2 weight stage: 1
        //   kw = 1 needs weight stage TAP, fired in the previous kw = 2 step - behind it this row's kw = 0 issue: 1 + 5
        //   kw = 2 needs weight stage TAP, first issue of this row's kw = 0 step - behind it 5 + (1 + 4): 10
#define LEAN2_STEP(KH, KW)                                                                                      \
    do {                                                                                                        \
        constexpr int TAP = (KH) * 3 + (KW);                                                                    \
        if (TAP == 0 && cb == 0) wait_vmcnt<0>();                                                               \
        else if ((KW) == 0) wait_vmcnt<1>();                                                                    \
        else if ((KW) == 1) wait_vmcnt<6>();                                                                    \
        else wait_vmcnt<10>();                                                                                  \
        retire_lds_reads();                                                                                     \
        __builtin_amdgcn_s_barrier();                                                                           \
        _Pragma("unroll") for (int i = 0; i < MP; ++i)                                                          \
            bbase[i] = ((tapmask[i] >> TAP) & 1u) ? xs + pv[KW][i] : zv[KW][i];                                 \
        if (TAP == 0 && cb == 0) fetch_params(ct);                                                              \
        read_frags(TAP % 3, KW, 0, 0);                                                                          \
        read_frags(TAP % 3, KW, 1, 1);                                                                          \
        mfma_group(0);                                                                                          \
        if (TAP < 7) w_stage(cur, TAP + 2, cbs, (TAP + 2) % 3); else w_stage(nx, TAP - 7, ncbs, (TAP + 2) % 3); \
        read_frags(TAP % 3, KW, 2, 0);                                                                          \
        mfma_group(1);                                                                                          \
        if ((KW) < 2) {                                                                                         \
            const int slot2 = (xs == 0) ? 1 : 0;                                                                \
            _Pragma("unroll") for (int q = (KW) * 5; q < ((KW) == 0 ? 5 : NQ + 1); ++q) {                       \
                if ((KH) < 2) x_piece(cur, q, (KH) + 1, cbs, slot2); else x_piece(nx, q, 0, ncbs, slot2);       \
            }                                                                                                   \
        }                                                                                                       \
        read_frags(TAP % 3, KW, 3, 1);                                                                          \
        mfma_group(0);                                                                                          \
        if (TAP == 8 && cb == cpt - 1 && has_res) issue_residual_loads(m0, c0);                                 \
        mfma_group(1);                                                                                          \
        if ((KW) == 2) xs = (xs == 0) ? XSLOT : 0;                                                              \
        if (TAP == 0 && cb == 0) store_params();                                                                \
    } while (0)

        for (int cb = 0; cb < cpt; ++cb) {
            const int cbs = cb << 7;
            const bool inner = cb + 1 < cpt;
            Desc nx;
            nx.xg = inner ? cur.xg : nt.xg;
            nx.xg32 = inner ? cur.xg32 : nt.xg32;
#pragma unroll
            for (int i = 0; i < WI; ++i) nx.woff[i] = inner ? cur.woff[i] : nt.woff[i];
            const int ncbs = inner ? cbs + 128 : 0;
            // (opaque per iteration: keeps the 9 x MP slot-offset sums and the per-piece source offsets from being
            // hoisted out of the loop and held - or spilled - in registers)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int i = 0; i < MP; ++i) asm volatile("" : "+v"(pv[kw][i]), "+v"(zv[kw][i]));
            asm volatile("" : "+v"(cur.xg), "+v"(cur.xg32), "+v"(nx.xg), "+v"(nx.xg32));
            if constexpr (NXS == 3) {
                LEAN_STEP(0, 0);
                if (cb == 0 && ct == t0) stamp(p.stamps, 3);
                LEAN_STEP(0, 1);
                LEAN_STEP(0, 2);
                LEAN_STEP(1, 0);
                LEAN_STEP(1, 1);
                LEAN_STEP(1, 2);
                LEAN_STEP(2, 0);
                LEAN_STEP(2, 1);
                LEAN_STEP(2, 2);
            } else {
                LEAN2_STEP(0, 0);
                if (cb == 0 && ct == t0) stamp(p.stamps, 3);
                LEAN2_STEP(0, 1);
                LEAN2_STEP(0, 2);
                LEAN2_STEP(1, 0);
                LEAN2_STEP(1, 1);
                LEAN2_STEP(1, 2);
                LEAN2_STEP(2, 0);
                LEAN2_STEP(2, 1);
                LEAN2_STEP(2, 2);
            }
        }
#undef LEAN_STEP
#undef LEAN2_STEP
        if (ct == t0) stamp(p.stamps, 4);

        run_epilogue(m0, c0);
        cur = nt;
        if (ct == t0) stamp(p.stamps, 5);
    }
    stamp(p.stamps, 6);
}

template <int TC, int WP, int WC, bool F8, int TP, int TCU = TC>
static hipError_t launch_lean_cfg(const ConvParams& p0, hipStream_t stream) {
    ConvParams p = p0;
    p.n_ptiles = (p.M + TP - 1) / TP;
    p.n_ctiles = (p.Cout + TC - 1) / TC;
    const int lds = (TP == 512 ? 2 : 3) * (TP + 8) * 128 + 3 * TC * 128 + 256 + 11 * TC * 4;
    static bool attr_set[64] = {};
    auto kern = conv3x3_lean_kernel<TC, WP, WC, F8, TP, TCU>;
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
    const long slots = (long)ncu * (TP == 128 ? 2 : 1);                // persistent: one workgroup per CU (quarter tiles: two)
    unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
    p.n_workers = 0;
    // a launch of at most 128 tiles (a call of up to ~20 faces: where the latency of ONE call is what counts) leaves three quarters
    // of the slots empty: 64 more workgroups warm the L2s for the next launch.  Larger quarter-tile launches (config 4's ~36 faces
    // on two lanes) keep their spare CUs for the other lane's kernels: there the prefetchers cost 4 % of the throughput.
    if (TP == 128 && p.pf_ptr && p.pf_bytes >= 4096 && grid <= 128 && (long)grid + CONV_PF_WGS <= slots && conv_prefetch_enabled()) {
        p.n_workers = (int)grid;
        grid += CONV_PF_WGS;
    }
    if (grid > 256) p.stamps = nullptr;                                // (the diagnostic stamp buffer holds 256 workgroups)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
    return hipGetLastError();
}

// Shapes the row-patch kernel covers; `p` carries the derived fields of launch_conv().
bool conv3x3_rows_eligible(const ConvParams& p) {
    if (p.KS != 3 || p.stride != 1 || (p.Cin & 63) || p.ksplit != 1) return false;
    if (p.Ho != p.H || p.Wo != p.W) return false;
    // signed 32-bit patch offsets: the patch of the last tile runs up to W + 264 pixels past the tensor
    const long reach = ((long)p.M + p.W + 600) * p.Cin * 2;
    return reach < 0x7fffffffL;
}

hipError_t launch_conv3x3_lean(const ConvParams& p, hipStream_t stream) {
    if (!conv3x3_rows_eligible(p)) return hipErrorInvalidValue;
    if (p.flags & FRP_FLAG_F8) {                   // fp8 operands: whole 128-channel rows, 128-cout tiles only
        if ((p.Cin & 127) || !p.wscale) return hipErrorInvalidValue;
        return launch_lean_cfg<128, 4, 2, true, 256>(p, stream);
    }
    {                                              // few tiles (small maps, few faces): quarter tiles, two workgroups per CU
        int ncu = p.n_cu;                          // (set by the engine; standalone callers: ask the runtime)
        if (ncu <= 0) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
            ncu = device_cu_count(dev);
        }
        const long def_tiles = p.Cout > 64 ? (long)((p.M + 255) / 256) * ((p.Cout + 127) / 128) : (long)((p.M + 511) / 512);
        if (conv_small_m(p, def_tiles, ncu)) return launch_lean_cfg<64, 4, 2, false, 128>(p, stream);
    }
    if (p.Cout > 64) return launch_lean_cfg<128, 4, 2, false, 256>(p, stream);
    if (p.dbg & 128) return launch_lean_cfg<64, 8, 1, false, 256>(p, stream);      // A/B: the 256-pixel tile (1.5 reads per MFMA)
    if (p.Cout <= 32 && !(p.dbg & 64)) return launch_lean_cfg<64, 8, 1, false, 512, 32>(p, stream);   // head outputs (dbg 64: A/B, all 64 rows)
    return launch_lean_cfg<64, 8, 1, false, 512>(p, stream);
}

}  // namespace frp
