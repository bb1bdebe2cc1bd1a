/*
 * asm_hip.h -- C ABI of libasm_hip.so: the MI355X (gfx950) kernels behind the Assemble-ResNet
 * training path of clovaai/assembled-cnn.
 *
 * The reference has no FFI / plugin interface (it is straight-line Python over TensorFlow 1.14,
 * SURVEY.md section 8b); every arithmetic op it runs lives inside TF.  Each entry point below names
 * the reference call site whose TF op it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - every function returns int: 0 = ASM_OK, <0 = error; asm_last_error() gives a thread-local message.
 *       ASM_EINVAL  (bad shape / flag)   ~ the reference's ValueError / assert
 *       ASM_ENOTSUP (unsupported combo)  ~ the reference's NotImplementedError
 *       ASM_EHIP    (a hipError_t was raised by the launch)
 *   - all pointers are DEVICE pointers owned by the caller, 16-byte aligned.  The library never
 *     allocates device memory; scratch is passed in (see *_workspace_bytes).
 *   - activations: NHWC bfloat16 (C % 8 == 0); conv filters: KRSC bfloat16 ([Cout][R][S][Cin]);
 *     statistics, BN parameters, master weights, gradients of parameters: float32.
 *   - every launch takes the hipStream_t to enqueue on (as void*); calls are asynchronous and
 *     re-entrant across streams/threads.  The library reads no environment variable and keeps ONE piece
 *     of process-wide state, set explicitly by the caller: the kernel-selection overrides of asm_tuning
 *     (below; defaults = the shipped heuristics).  Set it before launching from other threads.
 */
#ifndef ASM_HIP_H_
#define ASM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASM_OK 0
#define ASM_EINVAL (-1)
#define ASM_ENOTSUP (-2)
#define ASM_EHIP (-3)

#define ASM_ABI_VERSION 5

const char* asm_last_error(void);
int asm_abi_version(void);
/* kernels this library has launched so far in this process (every stream, every thread): a training step's launch count is
 * the difference across it.  hipMemcpyAsync / hipMemsetAsync fills of the strided 1x1 input gradient are not kernels. */
unsigned long long asm_launch_count(void);

/* ------------------------------------------------------------------------------------------------
 * Launch tape: a sequence of this library's device operations recorded once and issued again by one call.
 * Between asm_tape_begin and asm_tape_end every kernel launch THIS host thread makes through the library is also written
 * down (kernel, geometry, stream, a copy of the arguments), and so is every asm_stream_join; the operations execute as
 * usual while they are recorded (or are captured, when the streams are in HIP stream capture -- which is how
 * train.Trainer.capture records a training step: the capture's private memory pool keeps every buffer of the step at its
 * address, the tape replays the launches without hipGraphLaunch's per-node host cost).  asm_tape_replay issues the recorded
 * operations again, on the recorded streams, with the recorded arguments: the caller guarantees that every pointer in them
 * is still what it was.  asm_tape_mark closes a segment and opens the next; asm_tape_replay(tape, s) replays segment s
 * only (s = -1: all of them), so that the host can act between segments (hand a gradient bucket to RCCL).
 *   asm_tape_begin   -> tape id (> 0); one recording per thread at a time
 *   asm_tape_mark    -> index of the segment that starts here (1, 2, ...)
 *   asm_tape_end     -> tape id; the tape can be replayed from now on
 *   asm_tape_info    -> info = {nodes, kernel launches, joins, fills, segments, argument bytes}
 *   asm_tape_free    -> drops the tape and its events (not while another thread replays it; tapes alive at process exit are
 *                       abandoned, not torn down)
 * asm_stream_join(dst, src): dst waits for everything enqueued on src so far (event record + stream wait; raw
 * hipStream_t handles).  The host layer uses it for every cross-stream edge of a step so that a tape sees them. */
int asm_stream_join(void* dst_stream, void* src_stream);
/* device-to-device copy on `stream`, seen by a tape that is being recorded (a fill node): row slices / concatenations of
 * the input side of a step (the KD label split, run_loop_classification.py:90-96) without a framework kernel in between */
int asm_memcpy_async(void* dst, const void* src, size_t bytes, void* stream);

/* Gradient exchange of one bucket for a caller that owns its RCCL communicator -- MirroredStrategy's all-reduce
 * (official/utils/misc/distribution_utils.py:24-45): in-place SUM over `count` elements (dtype: ASM_F32 or ASM_BF16) of the flat
 * gradient arena, issued on comm_stream (not the null stream) behind everything producer_stream has enqueued so far (the
 * backward kernels that wrote the bucket).  nccl_comm is the caller's ncclComm_t; librccl.so is resolved on first use.  The
 * caller joins comm_stream back into its consumer stream (asm_stream_join) before asm_sgd_momentum, whose grad_scale carries
 * the 1 / replicas.  The Python host of this repository hands the same buckets to torch.distributed instead
 * (dp.GradSync). */
int asm_allreduce_bucket(void* buf, size_t count, int dtype, void* nccl_comm, void* comm_stream, void* producer_stream);
int asm_tape_begin(void);
int asm_tape_mark(void);
int asm_tape_end(void);
int asm_tape_info(int tape, int64_t info[6]);
int asm_tape_replay(int tape, int segment);
int asm_tape_free(int tape);

/* ------------------------------------------------------------------------------------------------
 * Kernel-selection overrides (tests, same-box A/B runs, tuning).  The defaults are the heuristics the
 * benchmarked binary uses; nothing here changes results beyond the summation order a different tile
 * implies.  asm_set_tuning copies the struct (NULL restores the defaults) and is process-wide: it is
 * not synchronised against concurrent launches from other threads.
 * ---------------------------------------------------------------------------------------------- */
typedef struct asm_tuning {
  int32_t igemm_mode;      /* 0: per-layer choice; 1: every forward / input-gradient layer on the general kernel (igemm_kernel) */
  int32_t igemm_tile;      /* 0: per-layer choice; 1: 128-row tiles only; 3: 256 x 256 where Ci % 64 == 0                  */
  int32_t igemm_pfa;       /* -1: per-layer choice; 0 / 1: addend-prefetching epilogue off / on                           */
  int32_t dgrad_parity;    /* stride-2 input gradients: 2 (default) 3x3 with 64 -> 64 channels in ONE launch (filter slice in
                              registers, the four parity classes side by side: dgrad_s2_kernel), the others as four
                              parity-class launches; 1: always four launches; 0: the generic gather                      */
  int32_t wgrad_halo;      /* 0: off; 1: resident-halo weight gradient on the large maps; 2: wherever the shape allows    */
  int32_t wgrad_big;       /* -1: per-layer choice; 0 / 1: 256 x 256 weight-gradient tile off / on                        */
  int32_t wgrad_splits;    /* 0: cost model; n > 0: force n pixel splits                                                   */
  int32_t bn_rows;         /* partial rows (= workgroups) of the batch-norm reducers                                       */
  int32_t igemm3;          /* 3x3 stride-1 layers with >= 128 input channels on maps up to 30 wide with the activation rows
                              resident across the nine taps (igemm3_kernel): 1: where it measured faster than igemm2's
                              tile for the layer; 2: wherever the shape allows; 0: never; 3: as 1, plus the layers with ONE
                              64-channel chunk (Ci = 64, maps up to 62 wide) with a single row buffer (default)        */
  int32_t gemm1;           /* 1x1 convolutions (forward / input gradient) as a GEMM with a ring of LDS stages, the loads of a
                              K step requested several steps ahead (igemm1_kernel): -1 per layer (default), 0 never
                              (igemm2_kernel), n > 0 force tile / depth n of the table in csrc/conv_gemm1.hip            */
  int32_t wgrad_ring;      /* 1x1 stride-1 weight gradients with their tiles by LDS-DMA into two LDS stages (wgrad_kernel<.., 2>):
                              -1 per layer (default), 0 never, n > 0 always                                              */
  int32_t igemm8;          /* 3x3 stride-1 layers (forward / input gradient) with >= 256 output channels on the wave-staggered
                              multi-phase 256 x 256 kernel (igemm8_kernel, csrc/conv_igemm8.hip; bit-identical to igemm2's
                              256 x 256 tile): 1 where the layer took the 256 x 256 tile, the ragged last round on the
                              128-row kernel (default); 2 wherever the shape allows, unsplit; 0 never                     */
} asm_tuning;
void asm_tuning_defaults(asm_tuning* t);
int asm_set_tuning(const asm_tuning* t);
void asm_get_tuning(asm_tuning* t);

/* ------------------------------------------------------------------------------------------------
 * Convolution -- replaces tf.layers.conv2d inside conv2d_fixed_padding (nets/model_helper.py:67-78),
 * the 1x1 "fc" convs of sk_conv2d / se_block (nets/blocks.py:139-147,172-181), the embedding conv
 * (nets/resnet_model.py:576-580) and tf.layers.dense (nets/resnet_model.py:595-597; a 1x1 conv on a
 * 1x1 image).  No bias (use_bias=False everywhere on this path; dense bias is added by asm_bias_add).
 *
 * Geometry: input  x [N, H, W, C] with element pitches (img_pitch, row_pitch, pix_pitch) so that a
 * pre-padded or channel-padded buffer can be addressed (the 7x7x3 stem runs as R=7,S=1,C=32 over a
 * [N][H+6][W+6][4] halo buffer: 8 pixels x 4 channels are one contiguous 32-element row).
 * Output y [N*Ho*Wo, K] row-major with row stride ldy.
 *   out(n, ho, wo, k) = sum_{r,s,c} x(n, ho*stride + r - pad, wo*stride + s - pad, c) * w(k, r, s, c)
 * which is exactly fixed_padding((k-1)//2 before) + VALID for stride>1 and SAME for stride 1.
 * ---------------------------------------------------------------------------------------------- */
typedef struct asm_conv_desc {
  int32_t N, H, W, C;       /* input tensor logical dims                                           */
  int32_t K, R, S;          /* filter dims                                                         */
  int32_t stride, pad;      /* pad = rows/cols of zeros BEFORE (top/left); after is implied by Ho  */
  int32_t Ho, Wo;           /* output spatial dims                                                 */
  int64_t x_img_pitch;      /* elements between images     (0 => H*W*C)                            */
  int32_t x_row_pitch;      /* elements between rows       (0 => W*C)                              */
  int32_t x_pix_pitch;      /* elements between pixels     (0 => C)                                */
  int32_t ldy;              /* output row stride in elements (0 => K)                              */
  int32_t out_f32;          /* 0: bf16 output, 1: float32 output                                   */
} asm_conv_desc;

/* y = conv(x, w).  If stats_partial != NULL (bf16 output only) the kernel also writes per-channel
 * partial sums of the bf16-rounded outputs: stats_partial[mb][0][k] = sum, [mb][1][k] = sum of
 * squares over the rows of M-block mb (128 rows each); asm_conv2d_stats_blocks() gives the number
 * of M-blocks.  This is the first half of tf.layers.batch_normalization(fused=True,training=True)
 * (nets/model_helper.py:26-37) fused into the producing conv. */
int asm_conv2d_fprop(const asm_conv_desc* d, const void* x, const void* w, void* y,
                     float* stats_partial, void* stream);
int asm_conv2d_stats_blocks(const asm_conv_desc* d);

/* Inference (training=False): y = [relu](conv(x, w) * scale[k] + shift[k] [+ residual]) with the moving-statistics
 * coefficients of asm_bn_infer_coeffs folded into the conv epilogue -- tf.layers.batch_normalization(training=False)
 * after conv2d_fixed_padding plus the bottleneck's shortcut add / ReLU (nets/model_helper.py:26-37,
 * nets/resnet_model.py:92-95) without a separate pass.  residual: optional bf16 tensor of the output shape.
 * bf16 output, K % 8 == 0.  Numerically the two-pass form: the conv tile is rounded to bf16 (it crosses LDS as bf16),
 * normalised in fp32 and rounded again. */
int asm_conv2d_fprop_bn(const asm_conv_desc* d, const void* x, const void* w, void* y, const float* scale,
                        const float* shift, const void* residual, int relu, void* stream);

/* dx = conv_transpose(dy, w).  wt is the filter in [C][R][S][K] layout (asm_filter_transpose).
 * Gradient of tf.layers.conv2d w.r.t. its input, which TF autodiff provides to
 * optimizer.compute_gradients (nets/optimizer_setting.py:30).  If addend != NULL (bf16, dx's shape) the kernel
 * writes dx = conv_transpose(dy, w) + addend: the gradient fan-in add (a tensor consumed by two ops, e.g. a block
 * input feeding conv1 and the shortcut) fused into the epilogue; addend may alias dx. */
int asm_conv2d_dgrad(const asm_conv_desc* d, const void* dy, const void* wt, const void* addend, void* dx,
                     void* stream);

/* The same with an addend that is a NOT-YET-MASKED gradient: addend_mask is the packed ReLU mask written by asm_bn_apply
 * ([N*H*W][C/8] bytes) and addend lanes whose bit is 0 count as 0 -- the masked gradient dz = dy * [y > 0] that
 * tf.nn.relu's backward would have materialised for the shortcut branch (nets/resnet_model.py:92-95) is consumed
 * straight from (dy, mask).  Not for the 1x1 stride-2 form (ASM_ENOTSUP). */
int asm_conv2d_dgrad_masked(const asm_conv_desc* d, const void* dy, const void* wt, const void* addend,
                            const uint8_t* addend_mask, void* dx, void* stream);
/* The same with the backward of an average pool folded in: dx = conv_transpose(dy, w) [+ addend [where mask]] +
 * avgpool_bwd(pool_dy) -- the block input of a projection bottleneck is read by conv1 (1x1, stride 1) and by the shortcut's
 * average pool (nets/resnet_model.py:123-141); with the shortcut branch's backward run first, its pooled gradient
 * pool_dy [N][pool_Ho][pool_Wo][C] is gathered in conv1's epilogue instead of being scattered by asm_avgpool_bwd into
 * a full-resolution tensor that conv1's input gradient is then added to.  1x1 / stride 1 only (ASM_ENOTSUP otherwise);
 * pooling geometry and the count_valid divisor rule as asm_avgpool_bwd. */
int asm_conv2d_dgrad_pooled(const asm_conv_desc* d, const void* dy, const void* wt, const void* addend,
                            const uint8_t* addend_mask, const void* pool_dy, int pool_k, int pool_stride, int pool_pad,
                            int pool_Ho, int pool_Wo, int count_valid, void* dx, void* stream);

/* asm_conv2d_dgrad[_masked] + the REDUCE pass of the batch-norm backward of the layer that produced this convolution's input:
 * the tensor written here, dx [N*H*W][C], is the complete gradient of an activation z = [relu](bn(y) [+ shortcut])
 * (nets/resnet_model.py:50-55, 84-95), and that batch norm's backward starts by reducing dbeta = sum dz and
 * dgamma = sum dz * xhat over dz = dx * [z > 0] (what tf.gradients builds for tf.layers.batch_normalization,
 * nets/model_helper.py:26-37) -- asm_bn_bwd_reduce reads dx and y again for it.  Here the epilogue that writes dx also reads
 * bn_y (that layer's pre-BN convolution output, bf16 [N*H*W][C]) and bn_relu_mask (its packed ReLU mask as written by
 * asm_bn_apply, or NULL: no ReLU) and emits partial [asm_conv2d_dgrad_bnred_blocks(d)][2][C] = per 128 rows (sum dz, sum dz * y)
 * of the bf16-ROUNDED dx; asm_bn_bwd_finalize_raw (below) takes these partials (asm_bn_partials_compact applies).
 * 1x1 stride-1 convolutions with C % 8 == 0 (ASM_ENOTSUP otherwise: the block-final batch norms, whose gradient a conv1 input
 * gradient completes, are 4 x the channels of the others; in the MFMA-bound 3x3 input gradients the extra epilogue reads cost
 * what the reduce pass they replace costs); no float atomics: fixed summation order. */
int asm_conv2d_dgrad_bnred_blocks(const asm_conv_desc* d);
int asm_conv2d_dgrad_bnred(const asm_conv_desc* d, const void* dy, const void* wt, const void* addend,
                           const uint8_t* addend_mask, const void* bn_y, const uint8_t* bn_relu_mask, float* partial,
                           void* dx, void* stream);

/* dw[k][r][s][c] (float32) = sum_{n,ho,wo} dy(n,ho,wo,k) * x(n, ho*stride+r-pad, wo*stride+s-pad, c).
 * Split-K over output pixels; `workspace` holds the per-split slabs. */
size_t asm_conv2d_wgrad_workspace_bytes(const asm_conv_desc* d);
int asm_conv2d_wgrad(const asm_conv_desc* d, const void* x, const void* dy, float* dw,
                     void* workspace, size_t workspace_bytes, void* stream);

/* KRSC bf16 -> CRSK bf16 (the dgrad operand).  ldk = K-stride of the output rows (0 => K); columns
 * k >= K of a padded output are left untouched (keep them zero). */
int asm_filter_transpose(const void* w_krsc, void* w_crsk, int K, int R, int S, int C, int ldk, void* stream);

/* Every KRSC -> CRSK copy of a model in one launch.  table: nlayers rows of 8 int32
 * {src_off, dst_off, K, R*S, C, ldk, elem_begin, 0}: offsets in elements into the flat bf16 arenas, elem_begin =
 * running sum of K*R*S*C (row 0 starts at 0); total_elems = the final running sum. */
int asm_filter_transpose_batched(const void* w_arena, void* wt_arena, const int32_t* table, int nlayers,
                                 long long total_elems, void* stream);
/* Same result through 64x64 LDS tiles (16-byte loads along c, 16-byte stores along k): table[l][7] = index of layer
 * l's first tile, a layer has RS * ceil(K/64) * ceil(C/64) tiles; columns K..ldk-1 of the copies are zero filled. */
int asm_filter_transpose_tiled(const void* w_arena, void* wt_arena, const int32_t* table, int nlayers,
                               int total_tiles, void* stream);

/* Stem packing for 3-channel first convs (7x7/2 stem nets/resnet_model.py:359-367; 3x3/2 ResNet-D stem
 * :328-333,:344-347): master [K][k][k][3] float32 -> bf16 [K][k][L] rows with L = round_up(4k, 8)
 * (element s*4+c, zero padded), and the gradient unpack [K][k][L] float32 -> [K][k][k][3]. */
int asm_stem_pack_filter(const float* w_krsc3, void* w_packed, int K, int ksize, void* stream);
int asm_stem_unpack_grad(const float* dw_packed, float* dw_krsc3, int K, int ksize, void* stream);
/* [N,H,W,3] (float32 or bf16) -> zero-haloed [N][H+6][W+6][4] bf16. */
int asm_stem_pad_input(const void* x, int x_is_f32, void* xp, int N, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Batch normalisation -- tf.layers.batch_normalization(fused=True) via batch_norm
 * (nets/model_helper.py:26-37), plus the tf.nn.relu / residual add / UpSampling2D+add that follow
 * it in _bottleneck_block_v1 (nets/resnet_model.py:50-55,92-95) and the BL merge (:499-501).
 * x: [M, C] bf16 (M = N*H*W).  All per-channel arrays float32.
 * ---------------------------------------------------------------------------------------------- */
int asm_bn_stats_blocks(int M, int C);
/* stand-alone partial statistics (same [blocks][2][C] layout as the conv epilogue) */
int asm_bn_stats(const void* x, int M, int C, float* stats_partial, void* stream);
/* [blocks][2][C] partials -> [groups][2][C] (group g sums blocks [g*ceil(blocks/groups), ...)); groups must
 * equal ceil(blocks / ceil(blocks/groups)).  Used before the finalize kernels when blocks is large. */
int asm_bn_partials_compact(const float* partial, int blocks, int C, float* out, int groups, void* stream);
/* partials -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale; moving stats update
 * m <- m*momentum + (1-momentum)*batch  (variance Bessel-corrected for the moving average). */
int asm_bn_finalize(const float* stats_partial, int blocks, int M, int C, const float* gamma,
                    const float* beta, float eps, float momentum, float* moving_mean,
                    float* moving_var, float* mean, float* invstd, float* scale, float* shift,
                    void* stream);
/* eval mode: scale/shift from the moving statistics */
int asm_bn_infer_coeffs(int C, const float* gamma, const float* beta, const float* moving_mean,
                        const float* moving_var, float eps, float* scale, float* shift, void* stream);
/* y = [relu]( x*scale[c] + shift[c] [+ residual] ).  res_mode: 0 none, 1 same shape,
 * 2 residual is [N, H/2, W/2, C] and is nearest-upsampled 2x (needs H, W).  With relu, relu_mask_out (optional,
 * [M][C/8] bytes) receives the packed ReLU mask (bit e of byte v = output channel 8v+e > 0) that the backward
 * kernels can read instead of the 16x larger bf16 output. */
int asm_bn_apply(const void* x, void* y, int M, int C, const float* scale, const float* shift,
                 const void* residual, int res_mode, int relu, int H, int W, uint8_t* relu_mask_out,
                 void* stream);
/* backward.  dy: grad of the output y; relu: 0 none, 1 yout is the bf16 forward output, 2 yout is the packed
 * mask written by asm_bn_apply (yout may be NULL when relu == 0).  Pass 1 reduces dgamma/dbeta partials, finalize makes the per-channel
 * coefficients, pass 2 writes dx (and, if dz_out != NULL, the masked gradient dz = dy*[y>0] that
 * also flows to the residual branch). */
int asm_bn_bwd_reduce(const void* dy, const void* x, const void* yout, int relu, int M, int C,
                      const float* mean, const float* invstd, float* partial, void* stream);
int asm_bn_bwd_finalize(const float* partial, int blocks, int M, int C, const float* gamma,
                        const float* mean, const float* invstd, float* dgamma, float* dbeta,
                        float* coefA, float* coefB, float* coefC, void* stream);
/* asm_bn_bwd_finalize for partials of (sum dz, sum dz * y) -- the raw second moment an input-gradient epilogue emits
 * (asm_conv2d_dgrad_bnred): sum dz * xhat = invstd * (sum dz * y - mean * sum dz), taken in fp64. */
int asm_bn_bwd_finalize_raw(const float* partial, int blocks, int M, int C, const float* gamma,
                            const float* mean, const float* invstd, float* dgamma, float* dbeta,
                            float* coefA, float* coefB, float* coefC, void* stream);
int asm_bn_bwd_apply(const void* dy, const void* x, const void* yout, int relu, int M, int C,
                     const float* coefA, const float* coefB, const float* coefC, void* dx,
                     void* dz_out, void* stream);

/* Two batch norms behind one ReLU -- out = relu(bn_a(xa) + bn_b(xb)), the block-final and the projection-shortcut batch
 * norm of a bottleneck (nets/resnet_model.py:92-96) -- share the masked gradient g = dy * [mask bit]: one reduce and one
 * apply for both.  partial_a / partial_b: [asm_bn_stats_blocks(M, C)][2][C] (sum g, sum g * xhat), each finished by
 * asm_bn_bwd_finalize; coef6 = [6][C]: coefA, coefB, coefC of a, then of b. */
/* forward twin: y = [relu](xa * scale_a + shift_a + bf16(xb * scale_b + shift_b)) -- the shortcut's normalised tensor is
 * evaluated on the fly (rounded to bf16 where the separate asm_bn_apply pass stored it: bit-identical results) */
int asm_bn_apply2(const void* xa, const void* xb, void* y, int M, int C, const float* scale_a, const float* shift_a,
                  const float* scale_b, const float* shift_b, int relu, uint8_t* relu_mask_out, void* stream);
int asm_bn_bwd_reduce2(const void* dy, const void* xa, const void* xb, const uint8_t* relu_mask, int M, int C,
                       const float* mean_a, const float* invstd_a, const float* mean_b, const float* invstd_b,
                       float* partial_a, float* partial_b, void* stream);
int asm_bn_bwd_apply2(const void* dy, const void* xa, const void* xb, const uint8_t* relu_mask, int M, int C,
                      const float* coef6, void* dxa, void* dxb, void* stream);

/* Small tensors (M <= asm_bn_small_max_rows(), e.g. the [N,1,1,d] squeeze layers of sk_conv2d / se_block,
 * nets/blocks.py:139-146): the whole training-mode batch norm in one launch per direction.
 * fwd: batch statistics of x (bf16-rounded, as above), moving-statistics update, mean / invstd out, y = bn(x) [relu],
 *      optional packed ReLU mask [M][C/8].   bwd: dgamma, dbeta and dx from dy, x and the mask (NULL = no ReLU). */
int asm_bn_small_max_rows(void);
int asm_bn_small_fwd(const void* x, void* y, int M, int C, const float* gamma, const float* beta, float eps,
                     float momentum, float* moving_mean, float* moving_var, float* mean, float* invstd, int relu,
                     uint8_t* relu_mask_out, void* stream);
int asm_bn_small_bwd(const void* dy, const void* x, const uint8_t* relu_mask, int M, int C, const float* gamma,
                     const float* mean, const float* invstd, float* dgamma, float* dbeta, void* dx, void* stream);

/* Small dense layers (csrc/dense_small.hip): the [N,1,1,C] squeeze / excite / classifier layers -- sk_fc_1 / sk_fc_2
 * (nets/blocks.py:136-148), se_block's dense pair (:165-178), tf.layers.dense of the head (nets/resnet_model.py:595-597)
 * -- and their input gradients, as plain row-major products with the reduction index contiguous in both operands:
 *   asm_dense_small:        out[m][n] = sum_k p[m][k] * q[n][k] (+ addend[m][n], bf16 [M][ldo]); bf16 operands, fp32
 *                           accumulation, out f32 or bf16 with row stride ldo.  K % 16 == 0; ldp, ldq % 8 == 0.
 *                           fprop: p = x [M][Cin], q = kernel [Cout][Cin]; input gradient: p = dy [M][ldy], q = the
 *                           CRSK copy [Cin][ldk] (asm_filter_transpose), K = ldk.
 *                           `out` owns whole rows: columns N .. ldo-1 of a padded output row (the classifier's 1001 ->
 *                           1008) are WRITTEN as zeros, so whole-row readers (isfinite checks, taps) never see
 *                           uninitialised memory; ldo <= 32 * ceil(N / 32) (ASM_EINVAL otherwise: a column slice of a
 *                           wider matrix is not a valid output).
 *   asm_dense_small_wgrad:  dw[n][k] = sum_m dy[m][n] * x[m][k]  (fp32 [Cout][ldw]; x [M][ldx], dy [M][ldy]).
 * (The one-launch dense + batch-norm forms, measured slower than these + asm_bn_small_*, are in asm_hip_debug.h.) */
int asm_dense_small(const void* p, int ldp, const void* q, int ldq, int M, int N, int K, void* out, int ldo,
                    int out_f32, const void* addend, void* stream);
int asm_dense_small_wgrad(const void* x, int ldx, const void* dy, int ldy, int M, int Cin, int Cout, float* dw, int ldw,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pooling / resampling (NHWC bf16)
 * ---------------------------------------------------------------------------------------------- */
/* tf.layers.max_pooling2d(3, 2, 'SAME') (nets/resnet_model.py:421-424): pad 0 before / 1 after. */
int asm_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* argmax, int N, int H, int W, int C, void* stream);
int asm_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int N, int H, int W, int C, void* stream);
/* zero-pad + tf.layers.average_pooling2d VALID (nets/resnet_model.py:123-141): k in {2,3}, pad
 * before = pad, divisor k*k (count_valid = 0) or number of in-range taps (count_valid = 1, the
 * stride-1 SAME ResNet-D case).  scale multiplies the result (1.0 for pooling; used with k=2,s=2,
 * divisor forced to 1 for the UpSampling2D backward). */
int asm_avgpool_fwd(const void* x, void* y, int N, int H, int W, int C, int k, int stride, int pad,
                    int Ho, int Wo, int count_valid, void* stream);
/* addend (optional, may alias dx): bf16 [N][H][W][C] added to the result -- the gradient fan-in add of the block
 * input (main path + pooled shortcut) fused into the pool backward instead of a separate add pass */
int asm_avgpool_bwd(const void* dy, void* dx, int N, int H, int W, int C, int k, int stride, int pad,
                    int Ho, int Wo, int count_valid, const void* addend, void* stream);
/* gradient of UpSampling2D((2,2)): dx[n,i,j,c] = sum of the 2x2 block of dy */
int asm_upsample2x_bwd(const void* dy, void* dx, int N, int Hs, int Ws, int C, void* stream);
/* the same over dy * [mask bit]: dy is the un-masked gradient behind the merge's ReLU, relu_mask its packed mask
 * ([N*2Hs*2Ws][C/8] bytes, asm_bn_apply): the masked full-resolution gradient is never materialised */
int asm_upsample2x_bwd_masked(const void* dy, const uint8_t* relu_mask, void* dx, int N, int Hs, int Ws, int C,
                              void* stream);
/* blocks.anti_aliased_downsample (nets/blocks.py:45-107): REFLECT pad (k-1)/2, binomial k x k, stride 2 */
int asm_blurpool_fwd(const void* x, void* y, int N, int H, int W, int C, int k, int stride, void* stream);
int asm_blurpool_bwd(const void* dy, void* dx, int N, int H, int W, int C, int k, int stride, void* stream);
/* tf.reduce_mean over H,W (nets/resnet_model.py:561, nets/blocks.py:170): [N,HW,C] -> [N,C] bf16 */
int asm_gap_fwd(const void* x, void* y, int N, int HW, int C, void* stream);
int asm_gap_bwd(const void* dy, void* dx, int N, int HW, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Selective-kernel attention (blocks.sk_conv2d, nets/blocks.py:128-152) and SE (nets/blocks.py:156-184)
 * f: [N, HW, 2F] bf16 (branch 0 = channels [0,F), branch 1 = [F,2F)).
 * ---------------------------------------------------------------------------------------------- */
/* s[n][c] = mean_hw (f0 + f1)  -> bf16 [N, F] */
int asm_sk_gap(const void* f, void* s, int N, int HW, int F, void* stream);
/* V = a0*f0 + a1*f1 with a = softmax over the 2 branches of att (float32 [N, 2F]) -> bf16 [N,HW,F] */
int asm_sk_select_fwd(const void* f, const float* att, void* v, int N, int HW, int F, void* stream);
/* da_b[n][c] = sum_hw f_b*dV ; datt = softmax-pair backward of da  -> datt bf16 [N, 2F] */
int asm_sk_select_bwd_att(const void* f, const void* dv, const float* att, void* datt, int N, int HW, int F, void* stream);
/* df_b = a_b*dV + ds[n][c]/HW  -> bf16 [N,HW,2F]  (ds: bf16 [N,F], gradient of the pooled vector) */
int asm_sk_select_bwd_f(const void* dv, const float* att, const void* ds, void* df, int N, int HW, int F, void* stream);
/* The same unit with the batch norm + ReLU of its 3x3 convolution applied on the fly (training path): y is the
 * convolution output [N, HW, 2F] bf16, scale / shift the per-channel coefficients of asm_bn_finalize, and
 * f = bf16(relu(y * scale + shift)) is recomputed wherever blocks.sk_conv2d reads it (nets/blocks.py:126-152) instead
 * of being written by asm_bn_apply and re-read; its ReLU mask and the gradient df = a_b dV + ds/HW are never
 * materialised either.  gap: s = mean_hw(f0 + f1); select: V = a0 f0 + a1 f1; bwd_att: as asm_sk_select_bwd_att. */
int asm_sk_gap_bn(const void* y, const float* scale, const float* shift, void* s, int N, int HW, int F, void* stream);
int asm_sk_select_bn_fwd(const void* y, const float* scale, const float* shift, const float* att, void* v, int N,
                         int HW, int F, void* stream);
int asm_sk_select_bn_bwd_att(const void* y, const float* scale, const float* shift, const void* dv, const float* att,
                             void* datt, int N, int HW, int F, void* stream);
/* Backward of that 2F-channel batch norm fed with dV directly: dz = (a_b dV + ds/HW) * [y*scale+shift > 0].
 * reduce writes asm_sk_bn_bwd_blocks(N, HW, F) partial rows [rows][2][2F] (sum dz, sum dz*xhat) for
 * asm_bn_bwd_finalize; apply writes dy = A dz + B y + C (bf16 [N, HW, 2F]) -- the gradient of the 3x3 convolution. */
int asm_sk_bn_bwd_blocks(int N, int HW, int F);
int asm_sk_bn_bwd_reduce(const void* dv, const float* att, const void* ds, const void* y, const float* scale,
                         const float* shift, const float* mean, const float* invstd, int N, int HW, int F,
                         float* partial, void* stream);
int asm_sk_bn_bwd_apply(const void* dv, const float* att, const void* ds, const void* y, const float* scale,
                        const float* shift, const float* coefA, const float* coefB, const float* coefC, void* dy,
                        int N, int HW, int F, void* stream);
/* Factorised form of that reduce (the training path's default since round 4; the reduce pass above stays for
 * ASM_SK_FACTOR=0): a_b and ds are constant over an image, so
 *   sum dz = sum_n a_b[n] G0[n] + (ds[n]/HW) M0[n],  sum dz*y = sum_n a_b[n] G1[n] + (ds[n]/HW) M1[n]
 * with per-image statistics [N][2][2F] (fp32): mask_stats = (sum_hw [f>0], sum_hw [f>0] y) out of the pooled-sum pass
 * (asm_sk_gap_bn_stats) and grad_stats = (sum_hw [f>0] dV, sum_hw [f>0] dV y) out of the gate-gradient pass
 * (asm_sk_select_bn_bwd_att_stats), both of which read y (and dV) anyway.  asm_sk_bn_bwd_finalize turns them into
 * dgamma, dbeta and the apply coefficients (xhat is affine in y): the reduce pass over the whole tensor disappears. */
int asm_sk_gap_bn_stats(const void* y, const float* scale, const float* shift, const float* mean, const float* invstd,
                        void* s, float* mask_stats, int N, int HW, int F, void* stream);
int asm_sk_select_bn_bwd_att_stats(const void* y, const float* scale, const float* shift, const float* mean,
                                   const float* invstd, const void* dv, const float* att, void* datt, float* grad_stats,
                                   int N, int HW, int F, void* stream);
int asm_sk_bn_bwd_finalize(const float* grad_stats, const float* mask_stats, const float* att, const void* ds, int N,
                           int HW, int F, const float* gamma, const float* mean, const float* invstd, float* dgamma,
                           float* dbeta, float* coefA, float* coefB, float* coefC, void* stream);
/* SE: y = x * sigmoid(e[n][c]);  e float32 [N, C] (pre-sigmoid) */
int asm_se_scale_fwd(const void* x, const float* e, void* y, int N, int HW, int C, void* stream);
/* de[n][c] = sigmoid'(e) * sum_hw x*dy (bf16 out) */
int asm_se_scale_bwd_e(const void* x, const void* dy, const float* e, void* de, int N, int HW, int C, void* stream);
/* dx = sigmoid(e)*dy + dsq[n][c]/HW (dsq: bf16 [N,C]) */
int asm_se_scale_bwd_x(const void* dy, const float* e, const void* dsq, void* dx, int N, int HW, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise helpers
 * ---------------------------------------------------------------------------------------------- */
int asm_relu_fwd(const void* x, void* y, size_t n, void* stream);
int asm_relu_bwd(const void* dy, const void* y, void* dx, size_t n, void* stream);
int asm_add_bf16(const void* a, const void* b, void* out, size_t n, void* stream);   /* out = a + b */
/* dx = dy where the packed ReLU mask (asm_bn_apply) has a 1 bit, else 0; n elements, n / 8 mask bytes */
int asm_mask_apply(const void* dy, const uint8_t* mask, void* dx, size_t n, void* stream);
int asm_bias_add_f32(float* y, const float* bias, int M, int C, int ldy, void* stream);
int asm_bias_grad_bf16(const void* dz, int M, int C, int ld, float* dbias, void* stream);
int asm_cast_f32_to_bf16(const float* x, void* y, size_t n, void* stream);
int asm_cast_bf16_to_f32(const void* x, float* y, size_t n, void* stream);   /* bf16 gradient buckets back into the fp32 arena */

/* ------------------------------------------------------------------------------------------------
 * Loss -- tf.losses.softmax_cross_entropy(label_smoothing) (losses/cls_losses.py:31-33) + the KD term
 * T^2 * CE(logits/T, teacher) (nets/run_loop_classification.py:156-162), forward and backward fused.
 * logits float32 [B][ld]; targets float32 dense [B][C] (one-hot, mixed, ...), teacher float32 dense
 * probabilities [B][C] or NULL.  loss_rows[b] = CE_b + KD_b (caller averages over B);
 * dlogits bf16 [B][ld_out] = loss_scale/B * (...), zero in the padding columns.
 * ---------------------------------------------------------------------------------------------- */
int asm_softmax_ce(const float* logits, int ld, const float* targets, const float* teacher, int B,
                   int C, float label_smoothing, float kd_temp, float loss_scale, float* loss_rows,
                   void* dlogits, int ld_out, void* stream);
int asm_onehot(const int32_t* labels, float* out, int B, int C, void* stream);
/* teacher = softmax(teacher_logits / T) (nets/run_loop_classification.py:93-94) */
int asm_softmax_rows(const float* x, float* y, int B, int C, float inv_temp, void* stream);
int asm_mean_f32(const float* x, int n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input: mean subtraction + cast (preprocessing/imagenet_preprocessing.py:122-155, utils/data_util.py:382)
 * fused with data_util.mixup (utils/data_util.py:97-158) and the stem halo padding.
 * images: [Bin, H, W, 3] uint8 or float32 in 0..255.  mixup_type 0: out[b] = img[b]-mean.
 * type 1 (2B->B): out[b] = lam[b]*img[b] + (1-lam[b])*img[B+b] - mean.
 * type 2 (B->B): first half as type 1 with lam1, second half lam2[b]*img[b] + (1-lam2[b])*img[Bin-1-b].
 * out: [Bout][H+6][W+6][4] bf16 (zero halo / zero 4th channel).
 * ---------------------------------------------------------------------------------------------- */
int asm_mixup_meansub(const void* images, int is_u8, int Bin, int H, int W, int mixup_type,
                      const float* lam1, const float* lam2, void* out, void* stream);
/* labels: dense float32 [Bin][C] -> mixed [Bout][C] (same lambda rule) */
int asm_mixup_labels(const float* y, int Bin, int C, int mixup_type, const float* lam1,
                     const float* lam2, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser -- tf.train.MomentumOptimizer + the L2 term folded in (nets/optimizer_setting.py:23-38,
 * nets/run_loop_classification.py:166-178):  g' = g*grad_scale + wd*w ; a = mom*a + g' ; w -= lr*a ;
 * w_bf16 = bf16(w).  One launch over a flat arena.
 * ---------------------------------------------------------------------------------------------- */
int asm_sgd_momentum(float* w, float* accum, const float* grad, void* w_bf16, size_t n, float lr,
                     float momentum, float weight_decay, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Remaining loss / pooling variants and the "next" rows of the scope table
 * ---------------------------------------------------------------------------------------------- */
/* cls_loss_type == 'sigmoid' (losses/cls_losses.py:34-38): loss_out[0] = sum(sigmoid_ce) / sum(onehot),
 * loss_out[1] = sum(onehot); dlogits bf16 [B][ld_out] = loss_scale * (sigmoid(z) - y) / sum(onehot).
 * rows_ws: float32 [B][2] scratch. */
int asm_sigmoid_ce(const float* logits, int ld, const float* targets, int B, int C, float loss_scale,
                   float* rows_ws, float* loss_out, void* dlogits, int ld_out, void* stream);
/* blocks.generalized_mean_pooling (nets/blocks.py:22-42): [N,HW,C] -> [N,C]; ssum [N,C] float32 is kept for
 * the backward pass. */
int asm_gem_fwd(const void* x, void* y, float* ssum, int N, int HW, int C, float p, void* stream);
int asm_gem_bwd(const void* x, const void* dy, const float* ssum, void* dx, int N, int HW, int C, float p,
                void* stream);
/* blocks.dropblock (nets/blocks.py:191-251).  uniform: float32 [H-bs+1, W-bs+1, C] draws of tf.random_uniform
 * (ONE mask for the whole batch); keep: float32 [H,W,C] = 1 - maxpool_bs(pad(relu(sign(gamma - u))));
 * scale[0] = H*W*C / (sum(keep) + 1e-8).  apply: y = [relu](x * keep * scale); for the backward pass call it
 * on dy with relu_mask_from = the forward output (dx = dy * keep * scale * [y > 0]) or NULL (no fused ReLU). */
int asm_dropblock_mask(const float* uniform, float gamma, int H, int W, int C, int block_size, float* keep,
                       float* scale, void* stream);
/* The same with the Bernoulli mean in DEVICE memory (gamma_dev[0]): a recorded training step (asm_tape_replay) issues
 * this launch with the arguments it was recorded with while keep_prob follows its schedule
 * (functions/model_fns.py:26-33, 221-226), so the host rewrites gamma_dev[0] before each replay. */
int asm_dropblock_mask_dev(const float* uniform, const float* gamma_dev, int H, int W, int C, int block_size, float* keep,
                           float* scale, void* stream);
int asm_dropblock_apply(const void* x, const float* keep, const float* scale, const void* relu_mask_from, int relu,
                        void* y, int N, int HWC, void* stream);
/* Evaluation metrics (nets/run_loop_classification.py:208-219): per row arg-max, max softmax probability,
 * top-1 hit and tf.nn.in_top_k(k=5) hit; accumulate adds a batch into the 33-float running state
 * {sum top1, sum top5, count, correct[10], confidence[10], count[10]} of metric/ece_metric.py:171-298. */
int asm_eval_rows(const float* logits, int ld, const int32_t* labels, int B, int C, int32_t* pred, float* conf,
                  float* top1, float* top5, void* stream);
int asm_eval_accumulate(const float* conf, const float* top1, const float* top5, int B, float* state33,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input-pipeline tail: the tensor work of imagenet_preprocessing.preprocess_image after
 * JPEG decode (preprocessing/imagenet_preprocessing.py:269-313), for a ragged batch of decoded uint8 images
 * packed back to back in `src`:
 *   window  = image[crop_y : crop_y+crop_h, crop_x : crop_x+crop_w]      (train: the sampled box of :57-90 /
 *             tf.image.decode_and_crop_jpeg; eval: the whole image)
 *   window  = flip_left_right(window) if flip                            (:94-96)
 *   resized = tf.image.resize_images(window, [resize_h, resize_w], BILINEAR, align_corners=False)   (:210-225;
 *             TF-1.14 legacy sampling without half-pixel centres, float32, no fused multiply-add)
 *   out     = resized[out_y : out_y+out_h, out_x : out_x+out_w]          (central_crop :97-120; train: 0,0)
 *   out    -= CHANNEL_MEANS if subtract_mean                             (mean_image_subtraction :122-155)
 * out: float32 [N][out_h][out_w][3].  descs: DEVICE array of N descriptors.  A descriptor whose windows do not
 * fit (the host mirror rejects those with ValueError) produces zeros for that image, never an out-of-range read.
 * ---------------------------------------------------------------------------------------------- */
typedef struct asm_image_desc {
  int64_t src_offset;                      /* byte offset of the image's [Hs][Ws][3] uint8 pixels in src   */
  int32_t Hs, Ws;                          /* decoded size                                                   */
  int32_t crop_y, crop_x, crop_h, crop_w;  /* window cut from the decoded image before the resize            */
  int32_t resize_h, resize_w;              /* size the window is resized to                                  */
  int32_t out_y, out_x;                    /* top-left of the output inside the resized window               */
  int32_t flip;                            /* 1: window mirrored left-right before the resize                */
  int32_t reserved;
} asm_image_desc;                          /* 56 bytes */
int asm_resize_crop_flip(const uint8_t* src, int64_t src_bytes, const asm_image_desc* descs, int N,
                         int out_h, int out_w, int subtract_mean, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Flag surface and topology planner.  asm_model_cfg carries the flags of nets/hparams_config.py:30-292 (+
 * official/utils/flags/_performance.py:29-42) that reach the hot path, with the reference's names; asm_model_plan walks
 * functions/model_fns.Model + nets/resnet_model.Model.__call__ (:305-599) for that configuration and an input of
 * N x H x W x 3 and returns the layers in creation order: every variable-owning layer with its TensorFlow scope name
 * ("resnet_model/stage1/big1/conv2d_3": + "/kernel", or "/gamma" "/beta" "/moving_mean" "/moving_variance", or
 * "/kernel" "/bias" for the dense layer), shapes, and its element offset in tf.trainable_variables() order, plus the
 * parameter-free ops between them.  Host code only -- usable (and tested) without a GPU.
 * Errors mirror the reference: bad resnet_version / unknown resnet_size -> ASM_EINVAL (ValueError), resnet_size < 50,
 * unknown pool_type, dtype other than bf16 -> ASM_ENOTSUP (NotImplementedError; the reference's fp16 / fp32 graph
 * dtypes are not computed by this library).
 * ---------------------------------------------------------------------------------------------- */
#define ASM_F32 0
#define ASM_BF16 1
#define ASM_F16 2
#define ASM_AA_SCONV 1          /* 'sconv' in anti_alias_type */
#define ASM_AA_PROJ 2           /* 'proj' in anti_alias_type  */
#define ASM_POOL_GAP 0
#define ASM_POOL_GEM 1
#define ASM_POOL_FLATTEN 2
typedef struct asm_model_cfg {
  int32_t resnet_size, resnet_version, num_classes;
  int32_t use_se_block, use_sk_block, use_resnet_d;
  int32_t anti_alias_filter_size, anti_alias_type;   /* bitmask ASM_AA_* */
  int32_t bl_alpha, bl_beta;
  int32_t zero_gamma, no_downsample, pool_type, embedding_size;
  int32_t dtype;                                     /* ASM_BF16 */
  int32_t mixup_type;
  float bn_momentum, bn_eps, loss_scale, label_smoothing, kd_temp, weight_decay, momentum;
} asm_model_cfg;

#define ASM_PLAN_CONV 0        /* conv2d_fixed_padding: 1 trainable tensor [R,S,C,K] (stored [K][R][S][C])   */
#define ASM_PLAN_BN 1          /* batch_norm: gamma, beta (+ moving_mean, moving_variance), C channels        */
#define ASM_PLAN_DENSE 2       /* tf.layers.dense: kernel [C,K] + bias [K]                                    */
#define ASM_PLAN_MAXPOOL 3
#define ASM_PLAN_AVGPOOL 4
#define ASM_PLAN_BLURPOOL 5
#define ASM_PLAN_GAP 6
#define ASM_PLAN_GEM 7
#define ASM_PLAN_FLATTEN 8
#define ASM_PLAN_SK_GAP 9
#define ASM_PLAN_SK_SELECT 10
#define ASM_PLAN_SE_SCALE 11
#define ASM_PLAN_ADD 12
#define ASM_PLAN_RELU 1                  /* flags */
#define ASM_PLAN_RESIDUAL 2              /* + shortcut / other branch before the ReLU                          */
#define ASM_PLAN_UPSAMPLED_RESIDUAL 4    /* that operand is UpSampling2D((2,2)) of a half-resolution tensor    */
#define ASM_PLAN_ZERO_GAMMA 8            /* gamma initialised to 0 (zero_gamma, nets/resnet_model.py:84)       */
#define ASM_PLAN_COUNT_VALID 16          /* average pool divides by the in-range taps (stride-1 SAME)          */
typedef struct asm_plan_entry {
  int32_t kind;
  int32_t N, H, W, C;          /* input activation                                                            */
  int32_t K, R, S, stride;     /* output channels, window, stride                                             */
  int32_t Ho, Wo;
  int32_t flags;
  int32_t trainable;           /* trainable tensors owned (conv 1, bn 2, dense 2, ops 0)                      */
  int32_t reserved;
  int64_t param_offset;        /* first element in tf.trainable_variables() order (-1 for parameter-free ops) */
  int64_t param_elems;
  char name[112];              /* variable scope of the layer; enclosing scope for parameter-free ops         */
} asm_plan_entry;
typedef struct asm_plan_summary {
  int32_t n_entries, trainable_tensors;
  int64_t trainable_elems;            /* e.g. 25 559 081 for ResNet-50, 1001 classes                          */
  int64_t forward_macs_per_image;
  int64_t wgrad_workspace_bytes;      /* largest asm_conv2d_wgrad_workspace_bytes over the layers at this N   */
} asm_plan_summary;
/* entries may be NULL to query summary->n_entries first. */
int asm_model_plan(const asm_model_cfg* cfg, int N, int H, int W, asm_plan_entry* entries, int capacity,
                   asm_plan_summary* summary);

#ifdef __cplusplus
}
#endif
#endif /* ASM_HIP_H_ */
