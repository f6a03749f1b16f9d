/* frp_lab.h -- tuning hooks of the FRP_LAB build (libfrp_lab.so: `make -C face-recognition-platform_amd/csrc lab`).
 * NOT part of the shipped library: libfrp.so exports include/frp.h only.  The lab library is the product library plus
 * the k-step / instruction-mix lab kernels (csrc/kstep_lab.hip), the first generation of the row-patch conv kernel kept
 * as an A/B partner (csrc/conv3x3_rows.hip) and the three entry points below; tools/ select it through FRP_LIB. */
#ifndef FRP_LAB_H
#define FRP_LAB_H
#include "frp.h"
#ifdef __cplusplus
extern "C" {
#endif

/* tuning hook: average milliseconds of `iters` back-to-back launches of one conv shape on
 * random device-resident operands (HIP events on the handle's stream) */
int frp_conv_bench(frp_handle* h, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride,
                   int32_t act, int32_t flags, int32_t with_res, int32_t iters, float* ms_avg,
                   uint64_t* stamps_out /* NULL, or [256][8] per-workgroup 100 MHz phase stamps of the last launch */);

/* tuning hook: sustained v_mfma_f32_32x32x16_f16 rate of this device on register operands (waves_per_simd 1..8),
 * or - waves_per_simd = 16*r + 2, r in {4,3,2} - of the conv k-step's mix: 8 waves per CU, r ds_read_b128 per 4 MFMAs */
int frp_mfma_peak(frp_handle* h, int32_t waves_per_simd, int32_t iters, float* tflops);

/* tuning hook: the conv k-step's inner loop in isolation under different schedules (csrc/kstep_lab.hip) */
int frp_kstep_lab(frp_handle* h, int32_t variant, int32_t iters, float* tflops);

#ifdef __cplusplus
}
#endif
#endif /* FRP_LAB_H */
