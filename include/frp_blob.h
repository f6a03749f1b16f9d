/* frp_blob.h -- binary layout of the weight/program blob consumed by frp_load_weights.
 * Little-endian.  Produced by face-recognition-platform_amd/weights.py (pack_blob).
 * Replaces the model-pack files insightface downloads at run time
 * (backend/app/utils/deepfake_utils.py:39-51); layer tables: netspec.py. */
#ifndef FRP_BLOB_H
#define FRP_BLOB_H
#include <stdint.h>

#define FRP_BLOB_MAGIC "FRPBLOB1"
#define FRP_BLOB_VERSION 2u
#define FRP_OPFLAG_W_FP8 16
#define FRP_OPFLAG_FP8_MFMA 32 /* with W_FP8: the op runs on fp8 operands (fp8 input tensor, weights stay E4M3 on the device) */
#define FRP_OPFLAG_OUT_FP8 64  /* the op's primary output tensor is fp8 */

#pragma pack(push, 1)
typedef struct frp_blob_header {      /* 128 bytes */
    char magic[8];
    uint32_t version;
    uint32_t header_bytes;
    uint32_t n_det_ops, n_det_bufs, det_in_buf, det_in_ch;
    uint32_t det_head_buf[3];         /* stride 8, 16, 32 */
    uint32_t det_num_anchors;         /* per location */
    uint32_t n_emb_ops, n_emb_bufs, emb_in_buf, emb_in_ch;
    uint32_t emb_out_buf, emb_size, emb_dim, reserved0;
    uint64_t det_ops_offset, emb_ops_offset, data_offset, data_bytes;
    uint64_t det_macs_1080p, emb_macs; /* informational */
} frp_blob_header;

typedef struct frp_conv_op {          /* 80 bytes (blob version 2) */
    int32_t in_buf, out_buf, res_buf; /* physical activation buffer ids; res_buf -1 = none */
    int32_t cin, cout, ksize, stride; /* pad = ksize/2 */
    int32_t act;                      /* 0 none, 1 ReLU, 2 PReLU */
    int32_t flags;                    /* 1 border-class bias [9][cout], 2 fp32 output,
                                         4 residual read at (y>>1,x>>1), 8 input viewed as 1x1x(H*W*C),
                                         16 (FRP_OPFLAG_W_FP8) weights stored as OCP FP8 E4M3FN bytes
                                         [cout][k][k][cin] followed, at the next 16-byte boundary, by cout
                                         fp32 per-output-channel scales: expanded to fp16 at load
                                         (value = fp16(fp32(e4m3) * scale)) unless 32 is set; BASELINE config 5,
                                         32 (FRP_OPFLAG_FP8_MFMA) the conv runs on the block-scaled fp8 MFMA: its input
                                         tensor is fp8 (value / in_scale), the E4M3 weights are used as stored and the
                                         per-cout scales multiply the accumulator in the epilogue,
                                         64 (FRP_OPFLAG_OUT_FP8) the primary output tensor is fp8 (value / out_scale) */
    int32_t real_ch;                  /* cin_real | cout_real << 16 (unpadded channel counts, for flop accounting) */
    int64_t w_off, bias_off, slope_off; /* byte offsets into the data section; slope_off -1 = none */
    int32_t out2_buf;                 /* -1 = none: buffer of an fp8 COPY (value / out_scale) of an fp16 primary output */
    float in_scale;                   /* FP8_MFMA: scale of the fp8 input tensor */
    float out_scale;                  /* scale of the fp8 output(s) of this op */
    int32_t reserved;
} frp_conv_op;
#pragma pack(pop)

#endif
