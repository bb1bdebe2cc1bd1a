/*
 * asm_hip_debug.h -- TEST-ONLY entry points of libasm_hip.so.  Not part of the drop-in boundary (include/asm_hip.h):
 * nothing in the product package calls them; tests/ uses them to cross-check the MFMA kernels at sizes the CPU
 * oracle cannot reach and to assert which kernel plan a shape gets.
 */
#ifndef ASM_HIP_DEBUG_H_
#define ASM_HIP_DEBUG_H_

#include "asm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Direct convolutions (one thread per output element, fp32 accumulate, same descriptor / layouts as the MFMA
 * kernels): the definition of conv2d_fixed_padding (nets/model_helper.py:67-78) and of its two gradients written
 * as plain loops. */
int asm_conv2d_fprop_naive(const asm_conv_desc* d, const void* x, const void* w, void* y, void* stream);
int asm_conv2d_dgrad_naive(const asm_conv_desc* d, const void* dy, const void* w_krsc, void* dx, void* stream);
int asm_conv2d_wgrad_naive(const asm_conv_desc* d, const void* x, const void* dy, float* dw, void* stream);

/* Each lane l of one wave reads ds_read_b64_tr_b16 at LDS element 4*l of lds[i] = i; out[l*4+j]. */
int asm_debug_tr_probe(void* out256_i16, void* stream);

/* The weight-gradient plan asm_conv2d_wgrad would use for d (with the current ASM_WGRAD_* knobs):
 * plan = {dy-tile rows (32/64/128/256), column-tile width (128/256), tiles_n, tiles_c, pixel splits, pixels per split}. */
int asm_conv2d_wgrad_plan(const asm_conv_desc* d, int32_t plan[6]);

/* Kernel family of the calling thread's last asm_conv2d_fprop* / asm_conv2d_dgrad* launch: 0 igemm_kernel (general fallback),
 * 1 igemm1_kernel (1x1 ring GEMM), 2 igemm2_kernel, 3 igemm3_kernel (rows resident across the taps), 4 conv_halo_kernel,
 * 5 dgrad_s2_kernel, 8 igemm8_kernel (wave-staggered multi-phase loop); -1 before the first call. */
int asm_debug_last_conv_kernel(void);


#ifdef __cplusplus
}
#endif
#endif /* ASM_HIP_DEBUG_H_ */
