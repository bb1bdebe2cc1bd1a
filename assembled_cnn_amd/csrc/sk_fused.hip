// Selective-kernel unit with the batch-norm + ReLU of its 3x3 convolution applied ON THE FLY (training path).
//
// blocks.sk_conv2d (nets/blocks.py:110-154) is  conv3x3(F -> 2F) -> BN -> ReLU -> f ; s = mean_hw(f0 + f1) ; gates ;
// V = a0 f0 + a1 f1.  The normalised tensor f (2F channels, the widest activation of the bottleneck) has exactly three
// readers -- the pooled sum, the select and their backward twins -- and all of them are element-wise in (pixel, channel),
// so f = relu(y * scale[c] + shift[c]) is recomputed from the convolution output y wherever it is needed instead of being
// written by a BN-apply pass and re-read: forward moves 7F instead of 11.1F bytes-equivalents per pixel, backward 11F
// instead of 16.2F (f, its 1-bit ReLU mask and the gradient df = a_b dV + ds/HW are never materialised; the BN backward
// reducer and apply take dV and rebuild df in registers).
//
// Numerics: f is rounded to bf16 exactly where the un-fused path stores it, so pooled sums, V and the gate gradients are
// bit-compatible with bn_apply -> sk_gap / sk_select; df is NOT rounded to bf16 (the un-fused path stored it), which
// is strictly closer to the fp32 reference.  HBM-bound, 16-byte vectors, no atomics (fixed-order partials).
#include "common.h"

namespace {

__device__ __forceinline__ u32x4 ldv(const bf16_t* p, size_t off) { return *reinterpret_cast<const u32x4*>(p + off); }
__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }

// 8 channels of f = bf16(relu(y * sc + sh)) from one 16-byte vector of y
struct Coef8 {
  float sc[8], sh[8];
};
__device__ __forceinline__ void load_coef(const float* __restrict__ scale, const float* __restrict__ shift, int ch0, Coef8& k) {
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(scale + ch0), s1 = *reinterpret_cast<const f32x4*>(scale + ch0 + 4);
  const f32x4 h0 = *reinterpret_cast<const f32x4*>(shift + ch0), h1 = *reinterpret_cast<const f32x4*>(shift + ch0 + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    k.sc[e] = s0[e]; k.sc[e + 4] = s1[e];
    k.sh[e] = h0[e]; k.sh[e + 4] = h1[e];
  }
}
__device__ __forceinline__ void bnrelu8(const u32x4& vy, const Coef8& k, float* f) {
  float y[8];
  unpack8(vy, y);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = rbf(fmaxf(y[e] * k.sc[e] + k.sh[e], 0.f));
}

__device__ __forceinline__ void gate8(const float* __restrict__ att, int n, int F, int c0, float* a0) {
  // a0 = softmax_0(l0, l1) = 1 / (1 + exp(l1 - l0)); a1 = 1 - a0   (nets/blocks.py:150-151)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float l0 = att[(size_t)n * 2 * F + c0 + e];
    const float l1 = att[(size_t)n * 2 * F + F + c0 + e];
    a0[e] = 1.0f / (1.0f + __expf(l1 - l0));
  }
}

// ---- s[n][c] = mean_hw( f0 + f1 ) ----------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void sk_gap_bn_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, bf16_t* __restrict__ s, int HW,
                                                       int F, int vcb) {
  __shared__ float red[NT][9];
  const int vcols = F >> 3;
  const int vcl = threadIdx.x % vcb, rl = threadIdx.x / vcb, nrl = NT / vcb;
  const int vc = blockIdx.x * vcb + vcl;
  const int n = blockIdx.y;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (vc < vcols && rl < nrl) {
    Coef8 k0, k1;
    load_coef(scale, shift, vc * 8, k0);
    load_coef(scale, shift, F + vc * 8, k1);
    constexpr int U = 4;
    const bf16_t* base = y + (size_t)n * HW * 2 * F + vc * 8;
    int r = rl;
    for (; r + (U - 1) * nrl < HW; r += U * nrl) {
      u32x4 v[U], w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t off = (size_t)(r + u * nrl) * 2 * F;
        v[u] = ldv(base, off);
        w[u] = ldv(base, off + F);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f0[8], f1[8];
        bnrelu8(v[u], k0, f0);
        bnrelu8(w[u], k1, f1);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f0[e] + f1[e];
      }
    }
    for (; r < HW; r += nrl) {
      const size_t off = (size_t)r * 2 * F;
      float f0[8], f1[8];
      bnrelu8(ldv(base, off), k0, f0);
      bnrelu8(ldv(base, off + F), k1, f1);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f0[e] + f1[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  if (rl == 0 && vc < vcols) {
    float o[8];
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
      for (int r = 0; r < nrl; ++r) t += red[r * vcb + vcl][e];
      o[e] = t * inv;
    }
    *reinterpret_cast<u32x4*>(s + (size_t)n * F + vc * 8) = pack8(o);
  }
}

// ---- V = a0 f0 + a1 f1 -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sk_select_bn_fwd_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ att, bf16_t* __restrict__ v,
                                                               int N, int HW, int F) {
  const int vcols = F >> 3;
  const size_t nvec = (size_t)N * HW * vcols;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const unsigned iu = (unsigned)i;
  const unsigned mu = iu / (unsigned)vcols;
  const int vc = (int)(iu - mu * (unsigned)vcols);
  const size_t m = mu;
  const int n = (int)(mu / (unsigned)HW);
  Coef8 k0, k1;
  load_coef(scale, shift, vc * 8, k0);
  load_coef(scale, shift, F + vc * 8, k1);
  float a0[8], f0[8], f1[8], o[8];
  gate8(att, n, F, vc * 8, a0);
  bnrelu8(ldv(y, m * 2 * F + vc * 8), k0, f0);
  bnrelu8(ldv(y, m * 2 * F + F + vc * 8), k1, f1);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = a0[e] * f0[e] + (1.0f - a0[e]) * f1[e];
  *reinterpret_cast<u32x4*>(v + i * 8) = pack8(o);
}

// ---- datt[n][c] = a0 a1 sum_hw (f0 - f1) dV ;  datt[n][F + c] = -that ----------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void sk_bn_bwd_att_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const bf16_t* __restrict__ dv,
                                                           const float* __restrict__ att, bf16_t* __restrict__ datt,
                                                           int HW, int F, int vcb) {
  __shared__ float red[NT][9];
  const int vcols = F >> 3;
  const int vcl = threadIdx.x % vcb, rl = threadIdx.x / vcb, nrl = NT / vcb;
  const int vc = blockIdx.x * vcb + vcl;
  const int n = blockIdx.y;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (vc < vcols && rl < nrl) {
    Coef8 k0, k1;
    load_coef(scale, shift, vc * 8, k0);
    load_coef(scale, shift, F + vc * 8, k1);
    constexpr int U = 2;
    int r = rl;
    for (; r + (U - 1) * nrl < HW; r += U * nrl) {
      u32x4 v0[U], v1[U], vg[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t m = (size_t)n * HW + r + u * nrl;
        v0[u] = ldv(y, m * 2 * F + vc * 8);
        v1[u] = ldv(y, m * 2 * F + F + vc * 8);
        vg[u] = ldv(dv, m * F + vc * 8);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f0[8], f1[8], g[8];
        bnrelu8(v0[u], k0, f0);
        bnrelu8(v1[u], k1, f1);
        unpack8(vg[u], g);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (f0[e] - f1[e]) * g[e];
      }
    }
    for (; r < HW; r += nrl) {
      const size_t m = (size_t)n * HW + r;
      float f0[8], f1[8], g[8];
      bnrelu8(ldv(y, m * 2 * F + vc * 8), k0, f0);
      bnrelu8(ldv(y, m * 2 * F + F + vc * 8), k1, f1);
      unpack8(ldv(dv, m * F + vc * 8), g);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (f0[e] - f1[e]) * g[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  if (rl == 0 && vc < vcols) {
    float a0[8], d0[8], d1[8];
    gate8(att, n, F, vc * 8, a0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
      for (int r = 0; r < nrl; ++r) t += red[r * vcb + vcl][e];
      d0[e] = a0[e] * (1.0f - a0[e]) * t;
      d1[e] = -d0[e];
    }
    *reinterpret_cast<u32x4*>(datt + (size_t)n * 2 * F + vc * 8) = pack8(d0);
    *reinterpret_cast<u32x4*>(datt + (size_t)n * 2 * F + F + vc * 8) = pack8(d1);
  }
}

// ---- BN backward of the 2F-channel batch norm, fed with dV ---------------------------------------------------------------
// df[m][b][c] = a_b[n][c] dV[m][c] + ds[n][c] / HW ;  dz = df * [y * scale + shift > 0] ;  xhat = (y - mean) * invstd
// pass 1: partial[(n * chunks + chunk)][0][ch] = sum dz, [1][ch] = sum dz * xhat over the chunk's pixels of image n
// grid (chunks, N); block = vcb (= 2F/8 <= 256) vector columns x rpb row lanes.
struct SkBnGeom {
  int HW, F, vcols, rpb, rows_per_chunk, chunks;
};

__device__ __forceinline__ void sk_thread_setup(const SkBnGeom& g, int n, int vc, const float* __restrict__ att,
                                                const bf16_t* __restrict__ ds, float* ab, float* u) {
  const int fv = g.F >> 3;
  const int b = vc >= fv ? 1 : 0;
  const int c0 = (vc - b * fv) * 8;
  float a0[8], q[8];
  gate8(att, n, g.F, c0, a0);
  unpack8(ldv(ds, (size_t)n * g.F + c0), q);
  const float inv = 1.0f / (float)g.HW;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ab[e] = b ? (1.0f - a0[e]) : a0[e];
    u[e] = q[e] * inv;
  }
}

__global__ __launch_bounds__(256) void sk_bn_bwd_reduce_kernel(const bf16_t* __restrict__ dv, const float* __restrict__ att,
                                                               const bf16_t* __restrict__ ds, const bf16_t* __restrict__ y,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, SkBnGeom g,
                                                               float* __restrict__ partial) {
  __shared__ float red[4096];   // [2][rpb][vcols * 8]
  const int tid = threadIdx.x;
  const int vc = tid % g.vcols;
  const int rr = tid / g.vcols;
  const bool active = rr < g.rpb;
  const int n = blockIdx.y;
  const int C2 = 2 * g.F;
  const int fv = g.F >> 3;
  const int cv = vc >= fv ? vc - fv : vc;   // vector column inside dV
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
  if (active) {
    Coef8 k;
    load_coef(scale, shift, vc * 8, k);
    float mu[8], is[8], ab[8], u[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mu[e] = mean[vc * 8 + e];
      is[e] = invstd[vc * 8 + e];
    }
    sk_thread_setup(g, n, vc, att, ds, ab, u);
    const int r_begin = blockIdx.x * g.rows_per_chunk;
    const int r_end = min(g.HW, r_begin + g.rows_per_chunk);
    constexpr int U = 2;
    for (int r = r_begin + rr; r < r_end; r += U * g.rpb) {
      u32x4 vy[U], vg[U];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int rq = r + q * g.rpb;
        const size_t m = (size_t)n * g.HW + (rq < r_end ? rq : r);
        vy[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(y + m * C2 + vc * 8));
        vg[q] = ldv(dv, m * g.F + cv * 8);
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        if (r + q * g.rpb >= r_end) break;
        float fy[8], fg[8];
        unpack8(vy[q], fy);
        unpack8(vg[q], fg);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dz = (fy[e] * k.sc[e] + k.sh[e] > 0.f) ? (ab[e] * fg[e] + u[e]) : 0.f;
          s[e] += dz;
          ss[e] += dz * ((fy[e] - mu[e]) * is[e]);
        }
      }
    }
  }
  const int ncol = g.vcols * 8;
  if (active) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(0 * g.rpb + rr) * ncol + vc * 8 + e] = s[e];
      red[(1 * g.rpb + rr) * ncol + vc * 8 + e] = ss[e];
    }
  }
  __syncthreads();
  const size_t prow = (size_t)n * g.chunks + blockIdx.x;
  for (int i = tid; i < 2 * ncol; i += 256) {
    const int which = i / ncol, col = i - which * ncol;
    float acc = 0.f;
    for (int r = 0; r < g.rpb; ++r) acc += red[(which * g.rpb + r) * ncol + col];
    partial[(prow * 2 + which) * C2 + col] = acc;
  }
}

// pass 2: dy = A * dz + B * y + C   (coefficients from asm_bn_bwd_finalize)
// Same (chunk, image) x (vector column, row lane) decomposition as the reducer: a thread keeps its 8 channels, so the
// gates, ds / HW and the five per-channel coefficient vectors are loaded ONCE and the loop streams y, dV -> dy (the
// one-vector-per-thread form re-derived 56 scalars for every 32 bytes moved and ran at 3.3 TB/s).
__global__ __launch_bounds__(256) void sk_bn_bwd_apply_kernel(const bf16_t* __restrict__ dv, const float* __restrict__ att,
                                                              const bf16_t* __restrict__ ds, const bf16_t* __restrict__ y,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ cA,
                                                              const float* __restrict__ cB, const float* __restrict__ cC,
                                                              bf16_t* __restrict__ dy, SkBnGeom g) {
  const int tid = threadIdx.x;
  const int vc = tid % g.vcols;
  const int rr = tid / g.vcols;
  if (rr >= g.rpb) return;
  const int n = blockIdx.y;
  const int C2 = 2 * g.F;
  const int fv = g.F >> 3;
  const int cv = vc >= fv ? vc - fv : vc;
  Coef8 k;
  load_coef(scale, shift, vc * 8, k);
  float ab[8], u[8], kA[8], kB[8], kC[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    kA[e] = cA[vc * 8 + e];
    kB[e] = cB[vc * 8 + e];
    kC[e] = cC[vc * 8 + e];
  }
  sk_thread_setup(g, n, vc, att, ds, ab, u);
  const int r_begin = blockIdx.x * g.rows_per_chunk;
  const int r_end = min(g.HW, r_begin + g.rows_per_chunk);
  constexpr int U = 2;
  for (int r = r_begin + rr; r < r_end; r += U * g.rpb) {
    u32x4 vy[U], vg[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int rq = r + q * g.rpb;
      const size_t m = (size_t)n * g.HW + (rq < r_end ? rq : r);
      vy[q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(y + m * C2 + vc * 8));
      vg[q] = ldv(dv, m * g.F + cv * 8);
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int rq = r + q * g.rpb;
      if (rq >= r_end) break;
      float fy[8], fg[8], o[8];
      unpack8(vy[q], fy);
      unpack8(vg[q], fg);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dz = (fy[e] * k.sc[e] + k.sh[e] > 0.f) ? (ab[e] * fg[e] + u[e]) : 0.f;
        o[e] = kA[e] * dz + kB[e] * fy[e] + kC[e];
      }
      const size_t m = (size_t)n * g.HW + rq;
      __builtin_nontemporal_store(pack8(o), reinterpret_cast<u32x4*>(dy + m * C2 + vc * 8));
    }
  }
}

SkBnGeom make_geom(int N, int HW, int F) {
  SkBnGeom g;
  g.HW = HW;
  g.F = F;
  g.vcols = 2 * F / 8;
  g.rpb = 256 / g.vcols;
  // ~1024 partial rows in total (4 workgroups per CU keep the reducer bandwidth-bound, see bn.hip), at least 4 trips
  int chunks = 1024 / N;
  if (chunks < 1) chunks = 1;
  int rows = cdiv(HW, chunks);
  rows = cdiv(rows, g.rpb) * g.rpb;
  if (rows < g.rpb * 4) rows = g.rpb * 4;
  g.rows_per_chunk = rows;
  g.chunks = cdiv(HW, rows);
  return g;
}

}  // namespace

#define SKF_OK(name)                                                                                        \
  ASM_REQUIRE(N > 0 && HW > 0 && F > 0 && F % 8 == 0 && 2 * F / 8 <= 256, name ": bad shape (N=%d HW=%d F=%d)", N, HW, F); \
  ASM_REQUIRE((size_t)N * HW * (2 * F / 8) < 0x7fffffffull, name ": tensor too large for 32-bit indexing")

extern "C" int asm_sk_gap_bn(const void* y, const float* scale, const float* shift, void* s, int N, int HW, int F,
                             void* stream) {
  SKF_OK("sk_gap_bn");
  ASM_REQUIRE(y && scale && shift && s, "sk_gap_bn: null pointer");
  const int vcb = F / 8 < 32 ? F / 8 : 32;
  if (HW >= 512)
    hipLaunchKernelGGL(sk_gap_bn_kernel<1024>, dim3(cdiv(F / 8, vcb), N), dim3(1024), 0, (hipStream_t)stream,
                       (const bf16_t*)y, scale, shift, (bf16_t*)s, HW, F, vcb);
  else
    hipLaunchKernelGGL(sk_gap_bn_kernel<256>, dim3(cdiv(F / 8, vcb), N), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)y, scale, shift, (bf16_t*)s, HW, F, vcb);
  ASM_CHECK_LAUNCH("sk_gap_bn");
  return ASM_OK;
}

extern "C" int asm_sk_select_bn_fwd(const void* y, const float* scale, const float* shift, const float* att, void* v,
                                    int N, int HW, int F, void* stream) {
  SKF_OK("sk_select_bn_fwd");
  ASM_REQUIRE(y && scale && shift && att && v, "sk_select_bn_fwd: null pointer");
  const size_t nvec = (size_t)N * HW * (F / 8);
  hipLaunchKernelGGL(sk_select_bn_fwd_kernel, dim3((unsigned)cdivz(nvec, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)y, scale, shift, att, (bf16_t*)v, N, HW, F);
  ASM_CHECK_LAUNCH("sk_select_bn_fwd");
  return ASM_OK;
}

extern "C" int asm_sk_select_bn_bwd_att(const void* y, const float* scale, const float* shift, const void* dv,
                                        const float* att, void* datt, int N, int HW, int F, void* stream) {
  SKF_OK("sk_select_bn_bwd_att");
  ASM_REQUIRE(y && scale && shift && dv && att && datt, "sk_select_bn_bwd_att: null pointer");
  const int vcb = F / 8 < 32 ? F / 8 : 32;
  if (HW >= 512)
    hipLaunchKernelGGL(sk_bn_bwd_att_kernel<1024>, dim3(cdiv(F / 8, vcb), N), dim3(1024), 0, (hipStream_t)stream,
                       (const bf16_t*)y, scale, shift, (const bf16_t*)dv, att, (bf16_t*)datt, HW, F, vcb);
  else
    hipLaunchKernelGGL(sk_bn_bwd_att_kernel<256>, dim3(cdiv(F / 8, vcb), N), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)y, scale, shift, (const bf16_t*)dv, att, (bf16_t*)datt, HW, F, vcb);
  ASM_CHECK_LAUNCH("sk_select_bn_bwd_att");
  return ASM_OK;
}

extern "C" int asm_sk_bn_bwd_blocks(int N, int HW, int F) {
  if (N <= 0 || HW <= 0 || F <= 0 || F % 8 || 2 * F / 8 > 256) return ASM_EINVAL;
  return N * make_geom(N, HW, F).chunks;
}

extern "C" int asm_sk_bn_bwd_reduce(const void* dv, const float* att, const void* ds, const void* y, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, int N, int HW, int F,
                                    float* partial, void* stream) {
  SKF_OK("sk_bn_bwd_reduce");
  ASM_REQUIRE(dv && att && ds && y && scale && shift && mean && invstd && partial, "sk_bn_bwd_reduce: null pointer");
  const SkBnGeom g = make_geom(N, HW, F);
  hipLaunchKernelGGL(sk_bn_bwd_reduce_kernel, dim3(g.chunks, N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dv, att,
                     (const bf16_t*)ds, (const bf16_t*)y, scale, shift, mean, invstd, g, partial);
  ASM_CHECK_LAUNCH("sk_bn_bwd_reduce");
  return ASM_OK;
}

extern "C" int asm_sk_bn_bwd_apply(const void* dv, const float* att, const void* ds, const void* y, const float* scale,
                                   const float* shift, const float* coefA, const float* coefB, const float* coefC,
                                   void* dy, int N, int HW, int F, void* stream) {
  SKF_OK("sk_bn_bwd_apply");
  ASM_REQUIRE(dv && att && ds && y && scale && shift && coefA && coefB && coefC && dy, "sk_bn_bwd_apply: null pointer");
  const SkBnGeom g = make_geom(N, HW, F);
  hipLaunchKernelGGL(sk_bn_bwd_apply_kernel, dim3(g.chunks, N), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dv, att, (const bf16_t*)ds, (const bf16_t*)y, scale, shift, coefA, coefB, coefC,
                     (bf16_t*)dy, g);
  ASM_CHECK_LAUNCH("sk_bn_bwd_apply");
  return ASM_OK;
}
