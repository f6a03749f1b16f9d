/* frp.h -- C ABI of the MI355X-native face-recognition hot path (libfrp.so).
 *
 * Drop-in boundary for the detect -> align -> embed -> match loop behind
 * backend/app/services/face_service.py of achiever04/face-recognition-platform.
 * The reference is pure Python and calls into un-vendored native packages; each
 * entry point below names the reference interface (file:line under /root/reference)
 * it replaces.  Plain C types only, no exceptions cross this boundary, every
 * function returns 0 on success or a negative frp_status; the message for the last
 * failure on a handle is available from frp_last_error().
 *
 * Ownership: the caller allocates and owns every in/out buffer (host pointers unless
 * the name says _device); the library owns what frp_create / frp_load_weights /
 * frp_gallery_* allocate and frees it in frp_destroy.
 * Threading: one handle = one HIP device + one private stream; calls on a handle are
 * serialised by an internal mutex (the reference fans out over a 4-thread pool,
 * backend/app/routes/camera.py:30,277-279 -- use one handle per GPU instead).
 * Gallery updates build a new device snapshot and swap it in, so a reader never sees
 * a half-updated matrix (the reference mutates state.ENCODINGS unlocked,
 * face_service.py:374,522).
 */
#ifndef FRP_H
#define FRP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRP_EMB_DIM 512      /* embedding width (reference: 128-d dlib, face_service.py:179) */
#define FRP_CHIP 112         /* aligned face chip edge */
#define FRP_KPS 5            /* landmarks per face */
#define FRP_MAX_FACES_CAP 128
#define FRP_MAX_TOPK 64/* upper bound for max_faces per frame */

typedef enum frp_status {
    FRP_OK = 0,
    FRP_ERR_INVALID = -1,     /* bad argument / shape the kernels do not cover */
    FRP_ERR_HIP = -2,         /* a HIP runtime call failed (message has the hipError string) */
    FRP_ERR_NO_WEIGHTS = -3,  /* frp_load_weights has not succeeded on this handle */
    FRP_ERR_NO_GALLERY = -4,  /* match requested with an empty gallery */
    FRP_ERR_BLOB = -5,        /* malformed weight blob */
    FRP_ERR_OOM = -6
} frp_status;

typedef enum frp_dtype { FRP_F32 = 0, FRP_F16 = 1, FRP_F64 = 2 } frp_dtype;

/* process flags */
#define FRP_FLAG_FORCED_K 1u   /* take the top max_faces anchors by score: no threshold, no NMS
                                  (benchmark mode) */
#define FRP_FLAG_RGB 2u        /* input frames are RGB (face_recognition.load_image_file order,
                                  face_service.py:139) instead of BGR (cv2 capture order, camera.py:205) */
#define FRP_FLAG_NO_MATCH 4u   /* skip the gallery match (encode_face path, face_service.py:87-219) */

typedef struct frp_handle frp_handle;

typedef struct frp_config {
    int32_t struct_size;   /* sizeof(frp_config) */
    int32_t max_batch;     /* frames per call, default 32 */
    int32_t max_faces;     /* faces kept per frame, default 10 (camera.py:182,233-235) */
    int32_t max_h, max_w;  /* largest frame, default 1080 x 1920 */
    int32_t profile;       /* 1: time every stage with HIP events on the handle's stream */
    int32_t reserved[10];
} frp_config;

/* Per-stage GPU time (HIP events on the handle's stream, accumulated over calls since
 * the last frp_reset_counters; only filled when frp_config.profile = 1) and algorithmic work. */
typedef struct frp_counters {
    int32_t struct_size;
    int32_t calls;
    int64_t frames, faces;
    double ms_h2d, ms_preprocess, ms_det_conv, ms_decode, ms_align, ms_emb_conv, ms_l2norm, ms_match, ms_d2h;
    double ms_total;          /* first to last event of each call, summed */
    double det_conv_flops;    /* algorithmic 2*MAC of the detector conv launches (padding excluded) */
    double emb_conv_flops;    /* same for the embedder */
    int64_t det_conv_launches, emb_conv_launches;
    double match_bytes;       /* algorithmic gallery bytes streamed by the match kernel */
    int64_t match_launches;
    int64_t gallery_rows;
    double f8_conv_flops;     /* part of det/emb_conv_flops that ran on fp8 operands (BASELINE config 5) */
    int64_t f8_conv_launches;
    double reserved[6];
} frp_counters;

/* ---- lifecycle ------------------------------------------------------------------
 * replaces: module singleton construction `face_service = FaceService()`
 * (face_service.py:54-82,769) and the lazy model registry (state.py:135-259). */
int frp_create(int device, const frp_config* cfg, frp_handle** out);
void frp_destroy(frp_handle* h);
const char* frp_last_error(const frp_handle* h); /* valid until the next call on h */
const char* frp_version(void);

/* Weight blob (layout: include/frp_blob.h; produced by weights.pack_blob):
 * folded fp16 conv programs of the detector and the embedder.
 * replaces: insightface FaceAnalysis model-pack loading (deepfake_utils.py:39-51).
 * A failed load leaves the handle WITHOUT weights (FRP_ERR_NO_WEIGHTS on later compute calls), never with a
 * half-replaced program. */
int frp_load_weights(frp_handle* h, const void* blob, size_t bytes);

/* ---- gallery (watchlist embedding matrix) -----------------------------------------
 * replaces: state.ENCODINGS dict name -> list[float] (state.py:78) and its per-call
 * rebuild np.array([ENCODINGS[t] ...]) (face_service.py:409,461,558,595).
 * Rows are L2-normalised and stored as fp16 [N x 512] in HBM; names stay on the host. */
int frp_gallery_set(frp_handle* h, const void* emb, int64_t n, int32_t d, int32_t dtype);
/* rows already unit-norm fp16 in device memory of this handle's GPU (e.g. the output
 * of an RCCL all-gather); copied into a library-owned snapshot */
int frp_gallery_set_device(frp_handle* h, const void* dev_f16, int64_t n, int32_t d);
/* Zero-copy import of a matrix produced ON this GPU (the RCCL all-gather of the watch-list shards):
 * frp_gallery_reserve allocates a fresh, not yet visible snapshot of `capacity_rows` x 512 fp16 and returns its device
 * address; the caller (a collective, a kernel) fills rows [0, n) with UNIT fp16 rows and finishes its stream work;
 * frp_gallery_commit(n) makes it the gallery (n <= capacity; the old snapshot is released).  While a reservation is pending -
 * someone else (RCCL) may be writing into it - the other gallery updates (set, set_device, update_row, remove_row) FAIL with
 * FRP_ERR_INVALID and change nothing; frp_gallery_cancel discards the reservation (no-op without one); a second reserve
 * replaces the first. */
int frp_gallery_reserve(frp_handle* h, int64_t capacity_rows, void** dev_f16);
int frp_gallery_commit(frp_handle* h, int64_t n_rows);
int frp_gallery_cancel(frp_handle* h);
/* device address of the current snapshot (rows [0, frp_gallery_size)), valid until the next gallery update on this handle:
 * the source from which a second handle on the same GPU copies its own snapshot (frp_gallery_set_device) */
const void* frp_gallery_device_ptr(frp_handle* h);
int frp_gallery_update_row(frp_handle* h, int64_t row, const void* emb, int32_t d, int32_t dtype); /* row == size appends */
int frp_gallery_remove_row(frp_handle* h, int64_t row); /* last row moves into `row` (store/delete: face_service.py:374,522) */
int64_t frp_gallery_size(const frp_handle* h);
/* copy the normalised fp16 gallery back to the host (n_rows x 512 uint16) */
int frp_gallery_get(frp_handle* h, void* out_f16, int64_t first_row, int64_t n_rows);

/* ---- exact rows for the REST-style compat path ---------------------------------------
 * replaces: the float64 arithmetic of face_recognition.face_distance on the rows AS ENROLLED
 * (np.linalg.norm(np.array([ENCODINGS[t] ...]) - q, axis=1): face_service.py:409-410,461-465,595-599, the 1-vs-1 calls of the
 * duplicate scan :357 and of cluster_faces :576).  The streaming loop matches on the unit fp16 rows (frp_match, the fused top-1 of
 * frp_process_*); compare_faces / find_k_nearest / batch_compare_faces / the duplicate scan report distances that must agree with
 * the reference's to 1e-6 - also for an exact copy (d = 0, where sqrt(2 - 2 cos) of fp16 rows reads 0.01-0.03), for d == tolerance,
 * for rows that are not unit vectors and for 128-d rows.
 * frp_gallery_exact(h, 1): from now on every row is ALSO kept as float64 [N x 512], exactly as handed to frp_gallery_set /
 * frp_gallery_update_row (dtype FRP_F64: bit for bit; FRP_F32 / FRP_F16: widened; not normalised; rows narrower than 512 are
 * zero-padded by the caller), 4 KB per row; rows that exist when it is switched on, and rows installed from device fp16 data
 * (frp_gallery_set_device, frp_gallery_commit), are the unit fp16 rows widened.  frp_gallery_exact(h, 0) frees the copy.
 * frp_gallery_distances: dist[m * n_cols + row] = ||row - q_m||_2 in float64 (differences, squares and sums in float64, one
 * correctly rounded sqrt; the summation order is fixed, so a result does not depend on N or M) for M queries of 512 doubles;
 * n_cols as in frp_match_scores.  frp_gallery_get_exact copies rows back (n_rows x 512 doubles). */
int frp_gallery_exact(frp_handle* h, int32_t on);
int frp_gallery_distances(frp_handle* h, const double* q, int32_t M, double* dist, int64_t n_cols);
int frp_gallery_get_exact(frp_handle* h, double* out, int64_t first_row, int64_t n_rows);

/* ---- the hot path -----------------------------------------------------------------
 * replaces, per frame: cv2.cvtColor(BGR2RGB) (camera.py:225), face_recognition.face_locations
 * (camera.py:232, face_service.py:156), the max_faces cap (camera.py:233-235),
 * face_recognition.face_encodings (camera.py:237, face_service.py:179), and per face
 * FaceService.compare_faces + the caller's filter loop reduced to top-1
 * (face_service.py:395-443, camera.py:243-259).
 *
 * bgr: B frames of H x W x 3 u8, row_stride bytes between rows, frames contiguous
 * (frame b starts at bgr + b*H*row_stride).  Outputs (any may be NULL):
 *   boxes  [B*K*4]  x1,y1,x2,y2 in frame pixels     kps [B*K*10]  5 x (x,y)
 *   scores [B*K]    sigmoid of the anchor logit      counts [B]   faces kept (<= K = max_faces)
 *   emb    [B*K*512] unit embeddings                 match_idx [B*K] gallery row of the best
 *   match_cos [B*K] its cosine (distance = sqrt(max(0, 2-2cos)))      cosine, -1 if no gallery
 * Slots k >= counts[b] are zero-filled (match_idx -1). */
int frp_process_frames(frp_handle* h, const uint8_t* bgr, int32_t B, int32_t H, int32_t W, int64_t row_stride,
                       int32_t max_faces, float det_thresh, float nms_iou, uint32_t flags,
                       float* boxes, float* kps, float* scores, int32_t* counts,
                       float* emb, int32_t* match_idx, float* match_cos);

/* Same pipeline split for callers that keep frames resident in HBM (the benchmark's timed
 * region starts with inputs already on the device): upload once, process many, fetch. */
int frp_upload_frames(frp_handle* h, const uint8_t* bgr, int32_t B, int32_t H, int32_t W, int64_t row_stride);
int frp_process_resident(frp_handle* h, int32_t max_faces, float det_thresh, float nms_iou, uint32_t flags);
/* B, max_faces: the shape the caller's buffers were sized for; FRP_ERR_INVALID when the handle's last results have
 * another shape (a concurrent caller replaced them) -- nothing is written then */
int frp_fetch_results(frp_handle* h, int32_t B, int32_t max_faces, float* boxes, float* kps, float* scores, int32_t* counts,
                      float* emb, int32_t* match_idx, float* match_cos);
int frp_synchronize(frp_handle* h);

/* Overlapped ingest for streaming callers (the camera loop keeps producing frames while the previous
 * batch is on the GPU, camera.py:277-305): the NEXT batch is copied host -> device on a private copy
 * stream into a staging buffer while the resident batch is being processed; frp_swap_frames makes the
 * staged batch the resident one (stream-ordered, no host wait).  The copy only overlaps when `bgr` is
 * page-locked: frp_host_alloc hands out such memory (freed by frp_host_free or with the handle).
 *     upload_async(t+1); process_resident(t); fetch_results(t); swap_frames(); ...                  */
void* frp_host_alloc(frp_handle* h, size_t bytes);
void frp_host_free(frp_handle* h, void* p);
int frp_upload_frames_async(frp_handle* h, const uint8_t* bgr, int32_t B, int32_t H, int32_t W, int64_t row_stride);
int frp_swap_frames(frp_handle* h);

/* ---- encoded stills ------------------------------------------------------
 * replaces: the image decode behind the upload routes - face_recognition.load_image_file (PIL) at
 * backend/app/services/face_service.py:139 and backend/app/routes/face.py:177-185,216,404,976.  Baseline JPEG (8-bit, Huffman,
 * one interleaved scan; grayscale, 4:4:4, 4:2:2, 4:2:0): the bit stream - the one serial part - is decoded on host threads
 * into quantised DCT coefficients in page-locked memory; dequantisation, the inverse DCT (the integer "slow" algorithm of
 * libjpeg, bit for bit), the triangle-filter chroma upsampling and YCbCr -> BGR run on the GPU, on the handle's copy stream,
 * and write the staging frame buffer of frp_upload_frames_async.  Progressive / arithmetic-coded / 12-bit files are refused. */
typedef struct frp_jpeg_info {
    int32_t width, height, components;      /* 1 (grayscale) or 3 (YCbCr) */
    int32_t h_samp[3], v_samp[3];           /* sampling factors per component (chroma always 1 x 1) */
    int32_t mcus_x, mcus_y;                 /* MCU grid (MCU = 8 h_samp[0] x 8 v_samp[0] pixels) */
    int32_t restart_interval, progressive;
} frp_jpeg_info;
/* header only; needs no handle.  FRP_ERR_INVALID for what the decoder does not cover. */
int frp_jpeg_info_get(const uint8_t* data, size_t size, frp_jpeg_info* info);
/* host entropy decode alone (parity tests, no handle): coef = per component [blocks_y][blocks_x][64] int16 in natural order,
 * quantised, components back to back (blocks_x = mcus_x * h_samp[c], blocks_y = mcus_y * v_samp[c]); qtab [3][64] */
int frp_jpeg_coefficients(const uint8_t* data, size_t size, int16_t* coef, size_t coef_elems, uint16_t* qtab, frp_jpeg_info* info);
/* B stills of identical geometry -> the staging frame buffer [B, height, width, 3] u8 BGR (grayscale: replicated), as
 * frp_upload_frames_async does for raw frames: follow with frp_swap_frames.  Returns when the coefficients are staged; the
 * device half runs asynchronously on the copy stream. */
int frp_upload_jpeg_async(frp_handle* h, const uint8_t* const* jpegs, const size_t* sizes, int32_t B);
/* diagnostic: batches of this handle whose ENTROPY decode ran on the device too (every frame carries restart intervals of at most 32
 * MCUs - one thread per interval; longer intervals: the host decoder, unless FRP_JPEG_DEVICE_HUFFMAN=1; =0: always the host) */
/* network passes replayed from a captured hipGraph so far (round 5: a detector / embedder pass asked for a second time with the same shapes,
 * buffers and switches is captured and from then on replayed by one call; FRP_NO_GRAPH=1 turns that off) - tests and diagnosis */
int64_t frp_debug_graph_replays(frp_handle* h);
int64_t frp_debug_jpeg_device_batches(frp_handle* h);

/* ---- multi-GPU: one process per GPU, ONE collective ----------------------------------------------------
 * Frames are sharded one stream per GPU and never exchanged.  The watch list is: every rank builds (decrypts) rows
 * [rank * ceil(N / R), ...) and the unit fp16 matrix is all-gathered over RCCL / xGMI straight into a reserved snapshot of
 * every rank's handle, then committed - replaces the single in-process dict of the reference (backend/app/state.py:78,
 * rebuilt per call at face_service.py:409).  The library owns the collective (librccl opened at first use); the launcher
 * only carries the 128-byte unique id from rank 0 to the other ranks (any control channel: a file, MPI, torch.distributed
 * object broadcast - frp_amd/dist.py). */
#define FRP_DIST_ID_BYTES 128
/* rank 0: a fresh communicator id */
int frp_dist_unique_id(void* id128);
/* collective over `world` ranks (each on its own GPU): creates this handle's communicator */
int frp_dist_init(frp_handle* h, const void* id128, int32_t rank, int32_t world);
int frp_dist_destroy(frp_handle* h);
/* collective: `shard` = this rank's rows [shard_rows, 512] in HOST memory (dtype FRP_F32 / FRP_F16 / FRP_F64 as
 * frp_gallery_set; normalised on upload), shard_rows = min(block, n_total - rank * block) with block = ceil(n_total / world).
 * On return every rank's gallery is the full [n_total, 512] matrix, rows in rank order. */
int frp_gallery_allgather(frp_handle* h, const void* shard, int64_t shard_rows, int32_t dtype, int64_t n_total);

/* ---- stage entry points (REST paths and parity tests) ------------------------------- */
/* detection only -> face_recognition.face_locations (camera.py:232) */
int frp_detect(frp_handle* h, const uint8_t* bgr, int32_t B, int32_t H, int32_t W, int64_t row_stride,
               int32_t max_faces, float det_thresh, float nms_iou, uint32_t flags,
               float* boxes, float* kps, float* scores, int32_t* counts, int32_t* anchor_idx);
/* Multi-scale pyramid building blocks (BASELINE config 4: "RetinaFace multi-scale pyramid"):
 * detect on the RESIDENT frames bilinearly resized to det_h x det_w (== frame size: no resize); boxes and
 * landmarks come back in the coordinates of the resized image.  The caller merges the scales
 * (pyramid.merge_scales) and hands the merged landmarks to frp_finish_faces, which aligns from the
 * full-resolution resident frames, embeds and matches.  counts[b] <= max_faces faces per frame. */
int frp_detect_resident(frp_handle* h, int32_t B, int32_t det_h, int32_t det_w, int32_t max_faces, float det_thresh, float nms_iou,
                        uint32_t flags, float* boxes, float* kps, float* scores, int32_t* counts, int32_t* anchor_idx);
/* the u8 frames the detector last read (the resident frames or their resize), [B, hs, ws, 3] (parity tests) */
int frp_get_det_source(frp_handle* h, uint8_t* out, int64_t out_bytes, int32_t* hs, int32_t* ws);
/* B: the resident batch the face list and the output buffers were sized for (checked under the handle mutex) */
int frp_finish_faces(frp_handle* h, int32_t B, const float* boxes, const float* kps, const float* scores, const int32_t* counts,
                     int32_t max_faces, uint32_t flags, float* emb, int32_t* match_idx, float* match_cos);

/* raw detector head maps of the last detect/process call, per stride level 0..2:
 * [B, H/stride, W/stride, 32] fp16 (parity tests) */
int frp_get_head_map(frp_handle* h, int32_t level, void* out_f16, int64_t out_bytes, int32_t* hl, int32_t* wl);
/* diagnostic: runs the detector program on the resident frames up to and including op n_ops - 1 and returns that op's
 * output [B, th, tw, tc] fp16 (out_f16 NULL: dimensions only, the prefix still runs).  Buffers are shared between tensors,
 * so an inner tensor is only readable from a prefix run (tools/det_bisect.py: per-layer determinism / parity bisection) */
int frp_debug_det_prefix(frp_handle* h, int32_t n_ops, void* out_f16, int64_t out_bytes, int32_t* th, int32_t* tw, int32_t* tc);
/* diagnostic: enable != 0 - every later detector pass hashes each op's output right behind the op (one 64-bit slot per op,
 * 64 slots); out64 != NULL receives the hashes of the last pass (tools/det_hash_bisect.py: which op of a FULL pass differed) */
int frp_debug_det_hashes(frp_handle* h, int32_t enable, uint64_t* out64);
/* decode + NMS on caller-supplied head maps [B,H_l,W_l,32] fp16 (parity tests) */
int frp_decode_heads(frp_handle* h, const void* head8, const void* head16, const void* head32,
                     int32_t B, int32_t canvas_h, int32_t canvas_w, int32_t max_faces, float det_thresh, float nms_iou,
                     uint32_t flags, float* boxes, float* kps, float* scores, int32_t* counts, int32_t* anchor_idx);
/* 5-landmark similarity warp -> normalised chips [M,112,112,8] fp16 (channels R,G,B,0..)
 * -> insightface norm_crop behind face_encodings (camera.py:237) */
int frp_align(frp_handle* h, const uint8_t* bgr, int32_t H, int32_t W, int64_t row_stride,
              const float* kps, int32_t M, uint32_t flags, void* chips_f16);
/* aligned u8 BGR chips [M,112,112,3] -> unit embeddings [M,512] */
int frp_embed_aligned(frp_handle* h, const uint8_t* chips, int32_t M, float* emb);
/* landmarks on one frame -> unit embeddings [M,512] (face_encodings with known faces) */
int frp_embed_faces(frp_handle* h, const uint8_t* bgr, int32_t H, int32_t W, int64_t row_stride,
                    const float* kps, int32_t M, uint32_t flags, float* emb);
/* cosine top-k (1 <= topk <= FRP_MAX_TOPK) of M queries vs the gallery, ordered by (cosine descending, row
 * ascending): idx / cos are [M x topk]; entries beyond the gallery size are -1 / -2.0
 * -> face_recognition.face_distance + argmin / argpartition (face_service.py:410,599-603) */
int frp_match(frp_handle* h, const float* q, int32_t M, int32_t topk, int32_t* idx, float* cos);
/* all cosines [M x N] (the N-dict compat path of compare_faces, face_service.py:409-432) */
/* n_cols: the gallery size cos_all was sized for; FRP_ERR_INVALID (nothing written) when the gallery has another size
 * by the time the call holds the handle -- re-read frp_gallery_size and retry */
int frp_match_scores(frp_handle* h, const float* q, int32_t M, float* cos_all, int64_t n_cols);

/* one convolution through the MFMA kernel on host tensors (kernel parity tests):
 * x [N,H,W,Cin] fp16, w [Cout][k][k][Cin] fp16, bias fp32 [Cout] or [9][Cout], out fp16 or fp32 */
int frp_conv2d_nhwc(frp_handle* h, const void* x, int32_t N, int32_t H, int32_t W, int32_t Cin,
                    const void* w, int32_t Cout, int32_t ksize, int32_t stride,
                    const float* bias, const float* slope, const void* res, int32_t res_h, int32_t res_w,
                    int32_t act, int32_t flags, void* out);

/* one 3x3 stride-1 convolution on fp8 operands through the block-scaled fp8 MFMA kernel (BASELINE config 5; kernel parity
 * tests): x8 [N,H,W,Cin] and w8 [Cout][3][3][Cin] are OCP E4M3 bytes (Cin a multiple of 128), value = code * scale with
 * one scale per output channel (wscale) and one for the input tensor (in_scale); res16 fp16 or NULL; `out` fp16, or E4M3
 * bytes (value / out_scale) with FRP_FLAG_OUT_FP8 = 64 in `flags`; out2_f8: optional E4M3 copy of an fp16 output */
int frp_conv2d_f8(frp_handle* h, const void* x8, int32_t N, int32_t H, int32_t W, int32_t Cin, const void* w8, int32_t Cout,
                  const float* wscale, const float* bias, const float* slope, const void* res16, int32_t act, int32_t flags,
                  float in_scale, float out_scale, void* out, void* out2_f8);

/* ---- observability ---------------------------------------------------------------
 * replaces: FaceService._metrics / get_performance_metrics (face_service.py:69-77,636-656) */
int frp_get_counters(frp_handle* h, frp_counters* out);
int frp_reset_counters(frp_handle* h);
/* switch the per-stage HIP-event timing (frp_config.profile) on or off.  frp_process_resident leaves its events unread; they
 * are read by the next call that waits for the stream anyway (frp_fetch_results, frp_synchronize) or, with a wait of its own,
 * by the first other call that would re-record them.  frp_upload_frames_async / frp_swap_frames never wait for them. */
int frp_set_profile(frp_handle* h, int32_t on);

#ifdef __cplusplus
}
#endif
#endif /* FRP_H */
