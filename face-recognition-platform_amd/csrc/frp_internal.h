// Internal declarations shared by the HIP kernels and the C-ABI host code (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FRP_ACT_NONE 0
#define FRP_ACT_RELU 1
#define FRP_ACT_PRELU 2
#define FRP_FLAG_BORDER_BIAS 1
#define FRP_FLAG_OUT_F32 2
#define FRP_FLAG_RES_UP2 4
#define FRP_FLAG_FLATTEN 8
#define FRP_FLAG_F8 32          // conv on fp8 operands: input tensor and weights are OCP E4M3 bytes (blob: FRP_OPFLAG_FP8_MFMA)
#define FRP_FLAG_OUT_FP8 64     // the primary output is fp8 (value / out_scale); an fp8 COPY of an fp16 output goes to `out2`
#define FRP_CHIP_PIX (112 * 112)
#ifndef FRP_MAX_FACES_CAP
#define FRP_MAX_FACES_CAP 128
#endif

#include "jpeg_host.h"

namespace frp {

struct ConvParams {
    const _Float16* x;      // [N,H,W,Cin]
    const _Float16* w;      // [Cout][KS][KS][Cin]
    const float* bias;      // [Cout] or [9][Cout] (FRP_FLAG_BORDER_BIAS)
    const float* slope;     // [Cout] (PReLU) or null
    const _Float16* res;    // [N,Ho,Wo,Cout] (or [N,Hr,Wr,Cout] with FRP_FLAG_RES_UP2) or null
    void* out;              // [N,Ho,Wo,Cout] fp16 (fp32 with FRP_FLAG_OUT_F32, fp8 with FRP_FLAG_OUT_FP8)
    void* out2;             // optional fp8 copy of an fp16 output (value / out_scale), or null
    const _Float16* x2;     // K-concat (generic kernel, 3x3, Cin % 64 == 0): a second K segment = the 1x1 conv of tensor x2 [N,H,W,Cin2]
                            // at the CENTRE tap's position (the block's stride-2 shortcut folded into its second conv);
                            // w rows are then [3][3][Cin] followed by [Cin2], bias the sum of both.  null = none
    int Cin2;               // channels of x2 (Cin2 % 64 == 0, Cin == Cin2 or 2 * Cin2)
    const float* wscale;    // FRP_FLAG_F8: per-cout scale of the fp8 weights
    float in_scale;         // FRP_FLAG_F8: the fp8 input tensor holds value / in_scale
    float out_scale;        // fp8 outputs hold value / out_scale
    int N, H, W, Cin, Cout, KS, stride;
    int act, flags;
    int Hr, Wr;             // residual spatial dims (RES_UP2)
    int ksplit;             // >1: split-K, `out` = fp32 workspace [ksplit][M][Cout] of raw partial sums;
                            // -1 (only with n_dev): the kernel picks conv_pick_ksplit(M) itself
    const _Float16* wino_w; // null, or the Winograd weight image of this layer (conv3x3_wino.hip; built by frp_api.cpp:build_wino_image):
                            // launch_conv() takes the Winograd kernel when the shape is eligible
    const int32_t* n_dev;   // null, or the number of images that really exist (<= N) in device memory: the kernel derives
                            // M and its tile count from it (threshold mode: the face count never visits the host mid-pipeline)
    int n_cu;               // compute units (input of the device-side split-K choice)
    unsigned long long* stamps;   // conv_bench diagnostics: [grid][8] 100 MHz phase stamps, or null
    int dbg;                // A/B switch (tests, conv_bench): 1 = generic kernel also for row-patch shapes
    int wino_wide_only;     // the Winograd kernel only in its 2-D tile form (maps wider than its flattened tiles cover) and only where the
                            // tile arithmetic says it pays (conv3x3_wino.hip: wino_2d_pays) - the detector's layers
    const void* pf_ptr;     // quarter-tile launches with CUs to spare: the NEXT launch's weights (pf_bytes of them), read once by 64 extra
    unsigned pf_bytes;      // workgroups - 8 per XCD - while this launch runs, so that the next one streams them from L2 (a call of a few
    int n_workers;          // faces reads every weight once, cold: its k-steps wait for HBM); n_workers: set by the launcher (0 = the whole grid works)
    int small_m;            // quarter tiles (128 pixels x 64 couts, two workgroups per CU): 0 = when the default tiling leaves half
                            // of the CUs idle (conv_common.h: conv_small_m), 1 = always, -1 = never (tests, A/B runs)
    // fused embedder stem (conv3x3_c64.hip, round 5): this launch is the 3x3 64 -> 64 conv that FOLLOWS the stem - the kernel computes the
    // stem (3x3, 3 real of 8 channels -> 64, + bias + PReLU) of every patch pixel itself, from the chips, and feeds its MFMAs from LDS: the
    // 64-channel map is written (for the block's shortcut) but never read back.  `x` is then only the tensor that WOULD have been read.
    const _Float16* stem_x;     // chips [N,H,W,8] fp16 (R,G,B,0..), or null = no fusion
    const _Float16* stem_w;     // folded stem weights [64][3][3][8]
    const float* stem_bias;     // [64]
    const float* stem_slope;    // [64]
    _Float16* stem_out;         // [N,H,W,64]: the stem's output, written by this launch
    int stem_even_only;         // only the pixels (even y, even x) of stem_out are written: its one reader is a 1x1 stride-2 conv (the block's
                                // shortcut, riding in a later k-loop) - a quarter of the map's bytes
    // derived by launch_conv():
    int pad, Ho, Wo, M, Ktot, nk, cin_shift, n_ptiles, n_ctiles;
    unsigned x_bytes, w_bytes, x2_bytes;   // buffer-descriptor sizes (each < 2 GiB)
    int x2_shift;           // log2(Cin / Cin2)
};

hipError_t launch_conv(const ConvParams& p, hipStream_t stream);
// row-patch variant for 3x3 stride-1 layers with Cin % 64 == 0 (conv3x3_rows.hip); launch_conv()
// routes eligible shapes to it.  `p` must carry launch_conv()'s derived fields.
bool conv3x3_rows_eligible(const ConvParams& p);
#ifdef FRP_LAB
hipError_t launch_conv3x3_rows(const ConvParams& p, hipStream_t stream);   // first generation + ablations (lab build only)
#endif
hipError_t launch_conv3x3_lean(const ConvParams& p, hipStream_t stream);   // static k-loop generation (conv3x3_lean.hip)
// 3x3 stride-1 64 -> 64 layers on large maps (conv3x3_c64.hip): weights resident in registers, 2-D pixel tiles, one barrier per tile
bool conv3x3_c64_eligible(const ConvParams& p);
bool conv3x3_c64_fuses_stem(int N, int H, int W, int n_cu);   // maps on which the kernel takes the embedder's stem into the launch of the conv behind it
hipError_t launch_conv3x3_c64(const ConvParams& p, hipStream_t stream);
// 3x3 stride-2 layers (conv3x3_s2.hip): row patches whose left / right neighbour entries are shared by consecutive output pixels
bool conv3x3_s2_eligible(const ConvParams& p);
hipError_t launch_conv3x3_s2(const ConvParams& p, hipStream_t stream);
// Winograd F(2,3) along the image rows (conv3x3_wino.hip): eligibility of a shape (`p` with launch_conv()'s derived fields),
// the same from the static layer geometry, the size of a layer's weight image, the launch (p.w = the image)
bool conv3x3_wino_eligible(const ConvParams& p);
bool conv3x3_wino_shape_ok(int W, int Cin, int ksize, int stride);
bool conv3x3_wino_wide_pays(int N, int H, int W, int Cin, int Cout, int n_cu, bool has_res);   // maps wider than 30: the 2-D tile form, where its tile arithmetic pays
#ifdef FRP_LAB
bool conv3x3_wino_lab_shape_ok(int W, int Cin, int ksize, int stride);   // + the maps only the lab's row-patch form covers (dbg bit 64)
#endif
size_t conv3x3_wino_image_bytes(int Cin, int Cout);
hipError_t launch_conv3x3_wino(const ConvParams& p, hipStream_t stream);

// K1: u8 BGR frames -> normalised fp16 NHWC8 canvas (top-left letterbox, zero u8 pad)
hipError_t launch_preprocess(const uint8_t* bgr, int B, int H, int W, long row_stride, long frame_stride,
                             _Float16* out, int Hc, int Wc, int rgb_in, hipStream_t stream);

// K1+K2 fused detector stem: u8 frames -> conv3x3 s2 (3->32) + bias + ReLU, fp16 NHWC
struct StemParams {
    const uint8_t* frames;   // [B,H,W,3] u8
    int B, H, W;
    long row_stride, frame_stride;
    int Hc, Wc;              // letterbox canvas (multiples of 32)
    int Ho, Wo;              // Hc/2, Wc/2
    int rgb_in;
    const _Float16* w;       // folded [32][3][3][8] fp16 (channels R,G,B,0..)
    const float* bias;       // [32]
    _Float16* out;           // [B,Ho,Wo,32]
};
hipError_t launch_stem_u8(const StemParams& p, hipStream_t stream);

// K1+K2+K2 fused detector stems: u8 frames -> conv3x3 s2 (3->32) + ReLU -> conv3x3 s2 (32->64) + ReLU
struct Stem12Params {
    const uint8_t* frames;   // [B,H,W,3] u8
    int B, H, W;
    long row_stride, frame_stride;
    int Hc, Wc;              // letterbox canvas (multiples of 32)
    int Ho1, Wo1;            // stem1 map: Hc/2, Wc/2
    int Ho2, Wo2;            // stem2 map: Hc/4, Wc/4
    int rgb_in;
    const _Float16* w1;      // folded [32][3][3][8] fp16 (channels R,G,B,0..)
    const float* bias1;      // [32]
    const _Float16* w2;      // folded [64][3][3][32] fp16
    const float* bias2;      // [64]
    _Float16* out;           // [B,Ho2,Wo2,64]
};
hipError_t launch_stem12_u8(const Stem12Params& p, hipStream_t stream);

// embedder stem: chips fp16 NHWC8 (3 real channels) -> conv3x3 s1 (3 -> 64) + bias + PReLU, fp16 NHWC64
struct EmbStemParams {
    const _Float16* x;       // [M,H,W,8]
    int M, H, W;
    const _Float16* w;       // folded [64][3][3][8] fp16 (channels R,G,B,0..)
    const float* bias;       // [64]
    const float* slope;      // [64]
    _Float16* out;           // [M,H,W,64]
    const int32_t* n_dev;    // null, or the number of chips that really exist (<= M), read on the device
};
hipError_t launch_emb_stem(const EmbStemParams& p, hipStream_t stream);

// K3: decode + candidate select + sort + NMS, one workgroup per frame
struct DecodeParams {
    const _Float16* head[3];   // per stride [B, H_l, W_l, 32]
    int hl[3], wl[3];
    int B, max_faces;
    float logit_thresh, nms_iou;
    float* boxes;     // [B, max_faces, 4]
    float* kps;       // [B, max_faces, 10]
    float* scores;    // [B, max_faces]
    int32_t* anchor;  // [B, max_faces]
    int32_t* counts;  // [B]
    _Float16* logits; // scratch [B, anchors]: dense copy of the anchor logits
};
hipError_t launch_decode_nms(const DecodeParams& p, hipStream_t stream);

// JPEG ingest, device half (jpeg_kernels.hip): quantised coefficients -> BGR frames
struct JpegParams {
    const int16_t* coef;     // [B][blocks_per_image][64], natural order, quantised; per image the components back to back
    const uint16_t* qtab;    // [B][3][64]
    uint8_t* planes;         // scratch: per image the sample planes of the components ([by * 8][bx * 8] u8 each)
    uint8_t* frames;         // out: [B, H, W, 3] u8 BGR
    int B, W, H, components;
    int bx[3], by[3];        // blocks per row / column of each component (whole MCUs)
    int blocks_per_image;
    long plane_off[3], plane_img;   // byte offsets of the component planes inside an image's scratch, bytes per image (multiples of 8)
    int hs, vs;              // luma sampling factors (chroma 1 x 1)
    int cw, ch;              // real extent of the chroma planes: ceil(W / hs), ceil(H / vs)
};
hipError_t launch_jpeg_decode(const JpegParams& p, hipStream_t stream);

// Entropy decoding ON THE DEVICE for streams with restart intervals (round 5): the DC predictors reset at every RSTn marker, so the
// intervals of a scan are independent bit streams - one thread each (jpeg_kernels.hip: jpeg_huffman_kernel).  Canonical Huffman
// tables: jpeg_host.h: JpegHuffTableDev.
struct JpegHuffParams {
    const uint8_t* scan;             // the entropy-coded segments of all images, back to back
    const uint32_t* int_off;         // [B][n_int + 1]: byte offset (into `scan`) of every interval's first byte; [n_int] = one past the image's data
    const JpegHuffTableDev* tables;  // [B][6]: component c's DC table at 2c, its AC table at 2c + 1
    int16_t* coef;                   // out: [B][coef_per_image], natural order (zeroed before the launch)
    int32_t* err;                    // out: [B], non-zero = the image's bit stream is corrupt / ends early
    long coef_per_image;
    int B, n_int, ri;                // intervals per image, MCUs per interval
    int mcus_x, mcus_y, components;
    int hs[3], vs[3], bx[3];         // sampling factors and blocks per row of each component
    long comp_off[3];                // first coefficient of each component inside an image
};
hipError_t launch_jpeg_huffman(const JpegHuffParams& p, hipStream_t stream);

// u8 bilinear resize for the detection pyramid (frames [B,H,W,3] tightly packed)
hipError_t launch_tensor_hash(const void* src, size_t bytes, unsigned long long* slot, hipStream_t stream);
hipError_t launch_resize_u8(const uint8_t* src, int B, int H, int W, uint8_t* dst, int Hs, int Ws, hipStream_t stream);

// K4: 5-point similarity + bilinear warp to 112x112 -> normalised fp16 NHWC8 chips
struct AlignParams {
    const uint8_t* frames;  // [B,H,W,3] u8 BGR
    int B, H, W;
    long row_stride, frame_stride;
    const float* kps;       // [B, max_faces, 10]
    const int32_t* counts;  // [B]
    int max_faces;
    const int32_t* face_slot;  // [n_faces] -> b*max_faces + k   (compacted face list)
    int n_faces;
    int rgb_in;             // frames are RGB instead of BGR
    _Float16* chips;        // [n_faces,112,112,8]
    const int32_t* n_dev;   // null, or the real face count (<= n_faces, which then is the capacity), read on the device
};
hipError_t launch_align(const AlignParams& p, hipStream_t stream);
// raw aligned u8 chips (BGR [M,112,112,3]) -> normalised fp16 NHWC8
hipError_t launch_chips_to_blob(const uint8_t* chips, int M, _Float16* out, hipStream_t stream);
// build the compact face list from counts (device side)
hipError_t launch_compact_faces(const int32_t* counts, int B, int max_faces, int32_t* face_slot, int32_t* n_faces,
                                hipStream_t stream);

// K5 tail: row-wise L2 normalisation of [M,512] fp32 (in place) + fp16 copy for the matcher.
// With `partials` (split-K FC): emb[m][c] = sum_s partials[s][m][c] + bias[c] first.
// `n_dev` (optional): the real row count lives in device memory (M is then the capacity); with ksplit == -1 the split
// factor of the FC that wrote `partials` is re-derived from it (conv_pick_ksplit with fc_ktot, n_cu).
hipError_t launch_l2norm(float* emb, _Float16* emb16, int M, int D, hipStream_t stream,
                         const float* partials = nullptr, int ksplit = 0, const float* bias = nullptr,
                         const int32_t* n_dev = nullptr, int fc_ktot = 0, int n_cu = 0);
// Split-K factor for a skinny GEMM-shaped conv (the 25088 -> 512 FC: 8 output tiles but 392 k-steps): the largest divisor
// of nk that keeps >= 8 k-steps per slice and does not exceed the CU count in (tile, slice) pairs.  1 = no split.
// Host and device evaluate the same function (device: when the image count is only known there).
__host__ __device__ inline int conv_pick_ksplit(int M, int Cout, int Ktot, int flags, bool has_res, int n_cu) {
    if (!(flags & FRP_FLAG_OUT_F32) || has_res || (Ktot & 63) || M <= 0) return 1;
    const int nk = Ktot / 64;
    const long tiles = (long)((M + 255) / 256) * ((Cout + 127) / 128);
    if (tiles * 4 > n_cu || nk < 64) return 1;
    int best = 1;
    for (int s = 2; s <= nk / 8; ++s)
        if (nk % s == 0 && tiles * s <= n_cu) best = s;
    return best;
}
// gallery upload: fp32 rows -> unit fp16 rows
hipError_t launch_gallery_normalize(const float* in, _Float16* out, long N, int D, hipStream_t stream);
// exact compat rows (frp.h: frp_gallery_exact): unit fp16 rows widened to float64; Euclidean distances of M float64 queries to the
// N float64 rows, out[m * N + row] (face_recognition.face_distance, face_service.py:410,465,599)
hipError_t launch_gallery_widen(const _Float16* in, double* out, long N, int D, hipStream_t stream);
hipError_t launch_gallery_distances(const double* rows, long N, const double* q, int M, double* out, hipStream_t stream);

#ifdef FRP_LAB
hipError_t launch_mfma_peak(const _Float16* src, float* dst, int blocks, int iters, hipStream_t stream);
// the conv k-step's MFMA + ds_read_b128 mix without memory traffic or barriers (reads per 4 MFMAs: 4, 3 or 2)
hipError_t launch_mfma_lds(const _Float16* src, float* dst, int blocks, int reads, int iters, hipStream_t stream);
// tuning lab: schedules of the conv k-step's inner loop in isolation (kstep_lab.hip)
hipError_t launch_kstep_lab(const _Float16* src, float* dst, int blocks, int variant, int iters, hipStream_t stream);
int kstep_lab_steps_per_iter(int variant);
int kstep_lab_waves(int variant);
hipError_t launch_fill_random_f16(_Float16* p, long n, unsigned seed, float scale, hipStream_t stream);
#endif

// K6: cosine match, top-1 (and optional full score matrix)
struct MatchParams {
    const _Float16* gallery;  // [N, 512] unit rows
    long N;
    const _Float16* q;        // [Mpad, 512] (rows >= M zero)
    int M, Mpad;
    float* part_cos;          // [n_wg, Mpad]
    int32_t* part_idx;        // [n_wg, Mpad]
    float* best_cos;          // [M]
    int32_t* best_idx;        // [M]
    float* all_scores;        // [M, N] or null
    int n_wg;
    const int32_t* n_dev;     // null, or the real query count (<= M) in device memory (top-1 path only)
};
int match_num_workgroups(long N);
#define FRP_MATCH_TOP1_MAX 512   // queries per launch the persistent top-1 kernel covers (the only one that takes n_dev)
hipError_t launch_match(const MatchParams& p, hipStream_t stream);
// top-k of each row of a device score matrix [M x N] by (cosine desc, row asc); -1 / -2.0 beyond N entries
hipError_t launch_topk_rows(const float* scores, int M, long N, int k, int32_t* idx_out, float* cos_out, hipStream_t stream);

}  // namespace frp
