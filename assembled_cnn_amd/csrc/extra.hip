// Kernels of the "next" rows of the scope table and the remaining loss / pooling variants:
// DropBlock, GeM pooling, sigmoid cross-entropy, on-device evaluation metrics (top-1 / top-5 / ECE bins).
#include "common.h"

namespace {

__device__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  return r;
}

// ---- sigmoid cross-entropy (losses/cls_losses.py:34-38): sum(ce) / sum(onehot) ---------------------------
__global__ __launch_bounds__(256) void sigmoid_ce_rows_kernel(const float* __restrict__ logits, int ld,
                                                              const float* __restrict__ y, int C,
                                                              float* __restrict__ rows) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  float ce = 0.f, ys = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float z = logits[(size_t)b * ld + c], t = y[(size_t)b * C + c];
    ce += fmaxf(z, 0.f) - z * t + log1pf(__expf(-fabsf(z)));
    ys += t;
  }
  ce = block_sum(ce, sh);
  ys = block_sum(ys, sh);
  if (threadIdx.x == 0) {
    rows[b * 2] = ce;
    rows[b * 2 + 1] = ys;
  }
}
__global__ __launch_bounds__(256) void sigmoid_ce_total_kernel(const float* __restrict__ rows, int B,
                                                               float* __restrict__ out) {
  __shared__ float sh[4];
  float ce = 0.f, ys = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    ce += rows[b * 2];
    ys += rows[b * 2 + 1];
  }
  ce = block_sum(ce, sh);
  ys = block_sum(ys, sh);
  if (threadIdx.x == 0) {
    out[0] = ce / ys;
    out[1] = ys;
  }
}
__global__ void sigmoid_ce_grad_kernel(const float* __restrict__ logits, int ld, const float* __restrict__ y, int B,
                                       int C, const float* __restrict__ tot, float loss_scale,
                                       bf16_t* __restrict__ dz, int ld_out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)B * ld_out) return;
  const int b = (int)(i / ld_out), c = (int)(i - (size_t)b * ld_out);
  float g = 0.f;
  if (c < C) {
    const float z = logits[(size_t)b * ld + c];
    g = (1.0f / (1.0f + __expf(-z)) - y[(size_t)b * C + c]) * loss_scale / tot[1];
  }
  dz[i] = f2bf(g);
}

// ---- generalised-mean pooling (nets/blocks.py:22-42), p = 3 ---------------------------------------------
// y[n][c] = HW^(-1/p) * max(sum_hw clip(x, eps, 1e12)^p, eps)^(1/p)
__global__ __launch_bounds__(256) void gem_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                      float* __restrict__ ssum, int HW, int C, float p) {
  const int c = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < HW; ++r) {
    const float v = fminf(fmaxf(bf2f(x[((size_t)n * HW + r) * C + c]), 1e-6f), 1e12f);
    s += powf(v, p);
  }
  s = fmaxf(s, 1e-6f);
  ssum[(size_t)n * C + c] = s;
  y[(size_t)n * C + c] = f2bf(powf((float)HW, -1.0f / p) * powf(s, 1.0f / p));
}
__global__ void gem_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                               const float* __restrict__ ssum, bf16_t* __restrict__ dx, int N, int HW, int C,
                               float p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)N * HW * C) return;
  const int c = (int)(i % C);
  const int n = (int)(i / ((size_t)HW * C));
  const float xv = bf2f(x[i]);
  const float s = ssum[(size_t)n * C + c];
  float g = 0.f;
  if (xv > 1e-6f && xv < 1e12f && s > 1e-6f)
    g = bf2f(dy[(size_t)n * C + c]) * powf((float)HW, -1.0f / p) * powf(s, 1.0f / p - 1.0f) * powf(xv, p - 1.0f);
  dx[i] = f2bf(g);
}

// ---- DropBlock (nets/blocks.py:191-251) ---------------------------------------------------------------------
// keep[h][w][c] = 1 - any(seed(i,j,c) for |h-(i+tl)| <= .. ) : seeds live on the (H-bs+1)x(W-bs+1) grid, are
// zero-padded by (tl, br) and dilated by a bs x bs SAME max-pool.  The mask is shared by the whole batch.
// gamma_dev != nullptr: the Bernoulli mean is read from device memory (a RECORDED training step replays this launch with
// the arguments it was recorded with, while keep_prob follows its schedule: functions/model_fns.py:26-33)
__global__ void dropblock_mask_kernel(const float* __restrict__ uniform, float gamma_arg, const float* __restrict__ gamma_dev,
                                      int H, int W, int C, int bs, float* __restrict__ keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W * C) return;
  const float gamma = gamma_dev ? *gamma_dev : gamma_arg;
  const int c = i % C, w = (i / C) % W, h = i / (C * W);
  const int br = (bs - 1) / 2, tl = (bs - 1) - br;
  const int hs = H - bs + 1, ws = W - bs + 1;
  // SAME max-pool of size bs, stride 1: window rows [h - pb, h - pb + bs) of the padded seed image with
  // pb = (bs - 1) / 2; padded(y, x) = seed(y - tl, x - tl)
  const int pb = (bs - 1) / 2;
  bool dropped = false;
  for (int dy = 0; dy < bs && !dropped; ++dy) {
    const int sy = h - pb + dy - tl;
    if ((unsigned)sy >= (unsigned)hs) continue;
    for (int dx = 0; dx < bs; ++dx) {
      const int sx = w - pb + dx - tl;
      if ((unsigned)sx >= (unsigned)ws) continue;
      if (gamma - uniform[((size_t)sy * ws + sx) * C + c] > 0.f) {  // relu(sign(gamma - u)) (:187-188)
        dropped = true;
        break;
      }
    }
  }
  keep[i] = dropped ? 0.f : 1.f;
}
__global__ __launch_bounds__(256) void dropblock_norm_kernel(const float* __restrict__ keep, int n,
                                                             float* __restrict__ scale) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += keep[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) *scale = (float)n / (s + 1e-8f);   // mask size / (sum + 1e-8) (:245-250)
}
// y = [relu]( x * keep[hw,c] * scale ) ; backward: dx = dy * keep * scale * [y > 0]
__global__ __launch_bounds__(256) void dropblock_apply_kernel(const bf16_t* __restrict__ x,
                                                              const float* __restrict__ keep,
                                                              const float* __restrict__ scale,
                                                              const bf16_t* __restrict__ ymask, int relu,
                                                              bf16_t* __restrict__ y, size_t nvec, unsigned hwc8) {
  const float sc = *scale;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    const unsigned k = (unsigned)(i % hwc8);
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + i * 8), f);
    const f32x4 k0 = *reinterpret_cast<const f32x4*>(keep + (size_t)k * 8);
    const f32x4 k1 = *reinterpret_cast<const f32x4*>(keep + (size_t)k * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[e] *= k0[e] * sc;
      f[e + 4] *= k1[e] * sc;
    }
    if (ymask) {  // backward through the fused ReLU
      float m[8];
      unpack8(*reinterpret_cast<const u32x4*>(ymask + i * 8), m);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = m[e] > 0.f ? f[e] : 0.f;
    } else if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
    }
    *reinterpret_cast<u32x4*>(y + i * 8) = pack8(f);
  }
}

// ---- evaluation metrics (nets/run_loop_classification.py:208-219, metric/ece_metric.py:171-298) ------------
// per row: argmax, max softmax probability, top-1 hit, in_top_k(k=5) hit
__global__ __launch_bounds__(256) void eval_rows_kernel(const float* __restrict__ logits, int ld,
                                                        const int32_t* __restrict__ labels, int C,
                                                        int32_t* __restrict__ pred, float* __restrict__ conf,
                                                        float* __restrict__ top1, float* __restrict__ top5) {
  __shared__ float shv[4];
  __shared__ int shi[4];
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const float* z = logits + (size_t)b * ld;
  const int lab = labels[b];
  // a label outside [0, C) can never be hit (tf.nn.in_top_k yields False for an out-of-range target): no stray read
  const bool lab_ok = (unsigned)lab < (unsigned)C;
  const float zl = lab_ok ? z[lab] : INFINITY;
  float mx = -INFINITY;
  int arg = 0x7fffffff;
  float higher = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float v = z[c];
    if (v > mx) {
      mx = v;
      arg = c;
    }
    higher += (v > zl) ? 1.f : 0.f;   // tf.nn.in_top_k: the target is in the top k unless k others are larger
  }
  // block arg-max (lowest index wins ties, like tf.argmax)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ov > mx || (ov == mx && oa < arg)) {
      mx = ov;
      arg = oa;
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    shv[w] = mx;
    shi[w] = arg;
  }
  __syncthreads();
  mx = shv[0];
  arg = shi[0];
  for (int i = 1; i < 4; ++i)
    if (shv[i] > mx || (shv[i] == mx && shi[i] < arg)) {
      mx = shv[i];
      arg = shi[i];
    }
  float se = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) se += __expf(z[c] - mx);
  se = block_sum(se, sh);
  higher = block_sum(higher, sh);
  if (threadIdx.x == 0) {
    pred[b] = arg;
    conf[b] = 1.0f / se;                      // max of softmax
    top1[b] = arg == lab ? 1.f : 0.f;
    top5[b] = (lab_ok && higher < 5.f) ? 1.f : 0.f;
  }
}
// state[0..2] = sum top1, sum top5, count ; state[3..12] correct per bin, [13..22] conf per bin, [23..32] count per bin
__global__ __launch_bounds__(64) void eval_accumulate_kernel(const float* __restrict__ conf,
                                                             const float* __restrict__ top1,
                                                             const float* __restrict__ top5, int B,
                                                             float* __restrict__ state) {
  const int t = threadIdx.x;
  if (t < 3) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += (t == 0) ? top1[b] : (t == 1 ? top5[b] : 1.f);
    state[t] += s;
  } else if (t < 13) {  // ECE bin t-3: (lo, hi] with the reference's epsilon-widened outer edges
    const int bin = t - 3;
    const float eps = 1e-7f;
    const float lo = bin == 0 ? -eps : (float)bin / 10.f;
    const float hi = bin == 9 ? 1.f + eps : (float)(bin + 1) / 10.f;
    float cor = 0.f, cs = 0.f, cnt = 0.f;
    for (int b = 0; b < B; ++b) {
      const float c = conf[b];
      if (c > lo && c <= hi) {
        cor += top1[b];
        cs += c;
        cnt += 1.f;
      }
    }
    state[3 + bin] += cor;
    state[13 + bin] += cs;
    state[23 + bin] += cnt;
  }
}

inline unsigned ew_grid(size_t n) {
  size_t b = cdivz(n, 256);
  return (unsigned)(b < 8192 ? (b ? b : 1) : 8192);
}

}  // namespace

extern "C" int asm_sigmoid_ce(const float* logits, int ld, const float* targets, int B, int C, float loss_scale,
                              float* rows_ws, float* loss_out, void* dlogits, int ld_out, void* stream) {
  ASM_REQUIRE(logits && targets && rows_ws && loss_out && B > 0 && C > 0 && ld >= C, "sigmoid_ce: bad arguments");
  ASM_REQUIRE(!dlogits || ld_out >= C, "sigmoid_ce: bad ld_out");
  hipStream_t st = (hipStream_t)stream;
  ASM_LAUNCH(sigmoid_ce_rows_kernel, dim3(B), dim3(256), 0, st, logits, ld, targets, C, rows_ws);
  ASM_LAUNCH(sigmoid_ce_total_kernel, dim3(1), dim3(256), 0, st, rows_ws, B, loss_out);
  if (dlogits)
    ASM_LAUNCH(sigmoid_ce_grad_kernel, dim3((unsigned)cdivz((size_t)B * ld_out, 256)), dim3(256), 0, st, logits,
                       ld, targets, B, C, loss_out, loss_scale, (bf16_t*)dlogits, ld_out);
  ASM_CHECK_LAUNCH("sigmoid_ce");
  return ASM_OK;
}

extern "C" int asm_gem_fwd(const void* x, void* y, float* ssum, int N, int HW, int C, float p, void* stream) {
  ASM_REQUIRE(x && y && ssum && N > 0 && HW > 0 && C > 0 && p > 0.f, "gem_fwd: bad arguments");
  ASM_LAUNCH(gem_fwd_kernel, dim3(cdiv(C, 256), N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)y, ssum, HW, C, p);
  ASM_CHECK_LAUNCH("gem_fwd");
  return ASM_OK;
}
extern "C" int asm_gem_bwd(const void* x, const void* dy, const float* ssum, void* dx, int N, int HW, int C, float p,
                           void* stream) {
  ASM_REQUIRE(x && dy && ssum && dx && N > 0 && HW > 0 && C > 0 && p > 0.f, "gem_bwd: bad arguments");
  ASM_LAUNCH(gem_bwd_kernel, dim3((unsigned)cdivz((size_t)N * HW * C, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)dy, ssum, (bf16_t*)dx, N, HW, C, p);
  ASM_CHECK_LAUNCH("gem_bwd");
  return ASM_OK;
}

extern "C" int asm_dropblock_mask(const float* uniform, float gamma, int H, int W, int C, int block_size, float* keep,
                                  float* scale, void* stream) {
  ASM_REQUIRE(uniform && keep && scale && H >= block_size && W >= block_size && C > 0 && block_size >= 1,
              "dropblock_mask: bad arguments (H=%d W=%d block=%d)", H, W, block_size);
  hipStream_t st = (hipStream_t)stream;
  ASM_LAUNCH(dropblock_mask_kernel, dim3(cdiv(H * W * C, 256)), dim3(256), 0, st, uniform, gamma, (const float*)nullptr, H, W,
                     C, block_size, keep);
  ASM_LAUNCH(dropblock_norm_kernel, dim3(1), dim3(256), 0, st, keep, H * W * C, scale);
  ASM_CHECK_LAUNCH("dropblock_mask");
  return ASM_OK;
}
extern "C" int asm_dropblock_mask_dev(const float* uniform, const float* gamma_dev, int H, int W, int C, int block_size,
                                      float* keep, float* scale, void* stream) {
  ASM_REQUIRE(uniform && gamma_dev && keep && scale && H >= block_size && W >= block_size && C > 0 && block_size >= 1,
              "dropblock_mask_dev: bad arguments (H=%d W=%d block=%d)", H, W, block_size);
  hipStream_t st = (hipStream_t)stream;
  ASM_LAUNCH(dropblock_mask_kernel, dim3(cdiv(H * W * C, 256)), dim3(256), 0, st, uniform, 0.f, gamma_dev, H, W, C,
                     block_size, keep);
  ASM_LAUNCH(dropblock_norm_kernel, dim3(1), dim3(256), 0, st, keep, H * W * C, scale);
  ASM_CHECK_LAUNCH("dropblock_mask_dev");
  return ASM_OK;
}
extern "C" int asm_dropblock_apply(const void* x, const float* keep, const float* scale, const void* relu_mask_from,
                                   int relu, void* y, int N, int HWC, void* stream) {
  ASM_REQUIRE(x && keep && scale && y && N > 0 && HWC > 0 && HWC % 8 == 0, "dropblock_apply: bad arguments");
  const size_t nvec = (size_t)N * (HWC / 8);
  ASM_LAUNCH(dropblock_apply_kernel, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     keep, scale, (const bf16_t*)relu_mask_from, relu, (bf16_t*)y, nvec, (unsigned)(HWC / 8));
  ASM_CHECK_LAUNCH("dropblock_apply");
  return ASM_OK;
}

extern "C" int asm_eval_rows(const float* logits, int ld, const int32_t* labels, int B, int C, int32_t* pred,
                             float* conf, float* top1, float* top5, void* stream) {
  ASM_REQUIRE(logits && labels && pred && conf && top1 && top5 && B > 0 && C > 0 && ld >= C, "eval_rows: bad arguments");
  ASM_LAUNCH(eval_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, C, pred, conf,
                     top1, top5);
  ASM_CHECK_LAUNCH("eval_rows");
  return ASM_OK;
}
extern "C" int asm_eval_accumulate(const float* conf, const float* top1, const float* top5, int B, float* state33,
                                   void* stream) {
  ASM_REQUIRE(conf && top1 && top5 && state33 && B > 0, "eval_accumulate: bad arguments");
  ASM_LAUNCH(eval_accumulate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, conf, top1, top5, B, state33);
  ASM_CHECK_LAUNCH("eval_accumulate");
  return ASM_OK;
}
