// Selective-kernel (SK) select and squeeze-excite (SE) scale kernels, NHWC bf16.
//   SK:  V = a0*f0 + a1*f1,  a = softmax over the two branches of att[n][b*F + c]   (nets/blocks.py:147-152)
//   SE:  y = x * sigmoid(e[n][c])                                                    (nets/blocks.py:182-183)
// Per-(n,c) reductions over H*W use the same (row-lanes x vector-columns) block shape as the GAP kernel.
#include "common.h"

namespace {

inline unsigned grid_for(size_t nvec) {
  size_t b = cdivz(nvec, 256);
  return (unsigned)(b ? b : 1);
}
__device__ __forceinline__ u32x4 ldv(const bf16_t* p, size_t off) { return *reinterpret_cast<const u32x4*>(p + off); }

__device__ __forceinline__ void gate8(const float* att, int n, int F, int c0, float* a0) {
  // a0 = softmax_0(l0, l1) = 1 / (1 + exp(l1 - l0)); a1 = 1 - a0
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float l0 = att[(size_t)n * 2 * F + c0 + e];
    const float l1 = att[(size_t)n * 2 * F + F + c0 + e];
    a0[e] = 1.0f / (1.0f + __expf(l1 - l0));
  }
}

__global__ __launch_bounds__(256) void sk_select_fwd_kernel(const bf16_t* __restrict__ f, const float* __restrict__ att,
                                                            bf16_t* __restrict__ v, int N, int HW, int F) {
  const int vcols = F >> 3;
  const size_t nvec = (size_t)N * HW * vcols;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const unsigned iu = (unsigned)i;
  const unsigned mu = iu / (unsigned)vcols;
  const int vc = (int)(iu - mu * (unsigned)vcols);
  const size_t m = mu;
  const int n = (int)(mu / (unsigned)HW);
  float a0[8], f0[8], f1[8], o[8];
  gate8(att, n, F, vc * 8, a0);
  unpack8(ldv(f, m * 2 * F + vc * 8), f0);
  unpack8(ldv(f, m * 2 * F + F + vc * 8), f1);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = a0[e] * f0[e] + (1.0f - a0[e]) * f1[e];
  *reinterpret_cast<u32x4*>(v + i * 8) = pack8(o);
}

// datt[n][c] = a0*a1*(da0 - da1), datt[n][F+c] = -that ;  da_b = sum_hw f_b * dV
// NT = 1024 with 2 rows in flight per trip for the large maps (one block per (image, 32 vector columns) walks all HW rows)
template <int NT>
__global__ __launch_bounds__(NT) void sk_bwd_att_kernel(const bf16_t* __restrict__ f, const bf16_t* __restrict__ dv,
                                                        const float* __restrict__ att, bf16_t* __restrict__ datt,
                                                        int HW, int F, int vcb) {
  __shared__ float red[NT][9];
  const int vcols = F >> 3;
  const int vcl = threadIdx.x % vcb, rl = threadIdx.x / vcb, nrl = NT / vcb;
  const int vc = blockIdx.x * vcb + vcl;
  const int n = blockIdx.y;
  float acc[8];  // sum_hw (f0 - f1) * dV
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (vc < vcols && rl < nrl) {
    constexpr int U = 2;
    int r = rl;
    for (; r + (U - 1) * nrl < HW; r += U * nrl) {
      u32x4 v0[U], v1[U], vg[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t m = (size_t)n * HW + r + u * nrl;
        v0[u] = ldv(f, m * 2 * F + vc * 8);
        v1[u] = ldv(f, m * 2 * F + F + vc * 8);
        vg[u] = ldv(dv, m * F + vc * 8);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f0[8], f1[8], g[8];
        unpack8(v0[u], f0);
        unpack8(v1[u], f1);
        unpack8(vg[u], g);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (f0[e] - f1[e]) * g[e];
      }
    }
    for (; r < HW; r += nrl) {
      const size_t m = (size_t)n * HW + r;
      float f0[8], f1[8], g[8];
      unpack8(ldv(f, m * 2 * F + vc * 8), f0);
      unpack8(ldv(f, m * 2 * F + F + vc * 8), f1);
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

__global__ __launch_bounds__(256) void sk_bwd_f_kernel(const bf16_t* __restrict__ dv, const float* __restrict__ att,
                                                       const bf16_t* __restrict__ ds, bf16_t* __restrict__ df, int N,
                                                       int HW, int F) {
  const int vcols = F >> 3;
  const size_t nvec = (size_t)N * HW * vcols;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const unsigned iu = (unsigned)i;
  const unsigned mu = iu / (unsigned)vcols;
  const int vc = (int)(iu - mu * (unsigned)vcols);
  const size_t m = mu;
  const int n = (int)(mu / (unsigned)HW);
  float a0[8], g[8], s[8], o0[8], o1[8];
  gate8(att, n, F, vc * 8, a0);
  unpack8(ldv(dv, m * F + vc * 8), g);
  unpack8(ldv(ds, (size_t)n * F + vc * 8), s);
  const float inv = 1.0f / (float)HW;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float u = s[e] * inv;
    o0[e] = a0[e] * g[e] + u;
    o1[e] = (1.0f - a0[e]) * g[e] + u;
  }
  *reinterpret_cast<u32x4*>(df + m * 2 * F + vc * 8) = pack8(o0);
  *reinterpret_cast<u32x4*>(df + m * 2 * F + F + vc * 8) = pack8(o1);
}

// ---- SE -------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

__global__ __launch_bounds__(256) void se_scale_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ e,
                                                           bf16_t* __restrict__ y, int N, int HW, int C) {
  const int vcols = C >> 3;
  const size_t nvec = (size_t)N * HW * vcols;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const unsigned iu = (unsigned)i;
  const int vc = (int)(iu % (unsigned)vcols);
  const int n = (int)(iu / ((unsigned)HW * (unsigned)vcols));
  float f[8];
  unpack8(ldv(x, i * 8), f);
#pragma unroll
  for (int k = 0; k < 8; ++k) f[k] *= sigmoidf(e[(size_t)n * C + vc * 8 + k]);
  *reinterpret_cast<u32x4*>(y + i * 8) = pack8(f);
}

__global__ __launch_bounds__(256) void se_bwd_e_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                       const float* __restrict__ e, bf16_t* __restrict__ de, int HW,
                                                       int C, int vcb) {
  __shared__ float red[256][9];
  const int vcols = C >> 3;
  const int vcl = threadIdx.x % vcb, rl = threadIdx.x / vcb, nrl = 256 / vcb;
  const int vc = blockIdx.x * vcb + vcl;
  const int n = blockIdx.y;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  if (vc < vcols && rl < nrl) {
    for (int r = rl; r < HW; r += nrl) {
      const size_t off = ((size_t)n * HW + r) * C + vc * 8;
      float f[8], g[8];
      unpack8(ldv(x, off), f);
      unpack8(ldv(dy, off), g);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += f[k] * g[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = acc[k];
  __syncthreads();
  if (rl == 0 && vc < vcols) {
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = 0.f;
      for (int r = 0; r < nrl; ++r) t += red[r * vcb + vcl][k];
      const float s = sigmoidf(e[(size_t)n * C + vc * 8 + k]);
      o[k] = t * s * (1.0f - s);
    }
    *reinterpret_cast<u32x4*>(de + (size_t)n * C + vc * 8) = pack8(o);
  }
}

__global__ __launch_bounds__(256) void se_bwd_x_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ e,
                                                       const bf16_t* __restrict__ dsq, bf16_t* __restrict__ dx, int N,
                                                       int HW, int C) {
  const int vcols = C >> 3;
  const size_t nvec = (size_t)N * HW * vcols;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const unsigned iu = (unsigned)i;
  const int vc = (int)(iu % (unsigned)vcols);
  const int n = (int)(iu / ((unsigned)HW * (unsigned)vcols));
  float g[8], q[8];
  unpack8(ldv(dy, i * 8), g);
  unpack8(ldv(dsq, (size_t)n * C + vc * 8), q);
  const float inv = 1.0f / (float)HW;
#pragma unroll
  for (int k = 0; k < 8; ++k) g[k] = g[k] * sigmoidf(e[(size_t)n * C + vc * 8 + k]) + q[k] * inv;
  *reinterpret_cast<u32x4*>(dx + i * 8) = pack8(g);
}

}  // namespace

#define SK_OK(name) ASM_REQUIRE(N > 0 && HW > 0 && F > 0 && F % 8 == 0, name ": bad shape")

extern "C" int asm_sk_select_fwd(const void* f, const float* att, void* v, int N, int HW, int F, void* stream) {
  SK_OK("sk_select_fwd");
  ASM_REQUIRE(f && att && v, "sk_select_fwd: null pointer");
  const size_t nvec = (size_t)N * HW * (F / 8);
  ASM_REQUIRE(nvec < 0x7fffffffull, "sk/se: tensor too large for 32-bit indexing");
  ASM_LAUNCH(sk_select_fwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)f,
                     att, (bf16_t*)v, N, HW, F);
  ASM_CHECK_LAUNCH("sk_select_fwd");
  return ASM_OK;
}

extern "C" int asm_sk_select_bwd_att(const void* f, const void* dv, const float* att, void* datt, int N, int HW,
                                     int F, void* stream) {
  SK_OK("sk_select_bwd_att");
  ASM_REQUIRE(f && dv && att && datt, "sk_select_bwd_att: null pointer");
  const int vcb = F / 8 < 32 ? F / 8 : 32;
  if (HW >= 512)
    ASM_LAUNCH(sk_bwd_att_kernel<1024>, dim3(cdiv(F / 8, vcb), N), dim3(1024), 0, (hipStream_t)stream,
                       (const bf16_t*)f, (const bf16_t*)dv, att, (bf16_t*)datt, HW, F, vcb);
  else
    ASM_LAUNCH(sk_bwd_att_kernel<256>, dim3(cdiv(F / 8, vcb), N), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)f, (const bf16_t*)dv, att, (bf16_t*)datt, HW, F, vcb);
  ASM_CHECK_LAUNCH("sk_select_bwd_att");
  return ASM_OK;
}

extern "C" int asm_sk_select_bwd_f(const void* dv, const float* att, const void* ds, void* df, int N, int HW, int F,
                                   void* stream) {
  SK_OK("sk_select_bwd_f");
  ASM_REQUIRE(dv && att && ds && df, "sk_select_bwd_f: null pointer");
  const size_t nvec = (size_t)N * HW * (F / 8);
  ASM_REQUIRE(nvec < 0x7fffffffull, "sk/se: tensor too large for 32-bit indexing");
  ASM_LAUNCH(sk_bwd_f_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dv, att,
                     (const bf16_t*)ds, (bf16_t*)df, N, HW, F);
  ASM_CHECK_LAUNCH("sk_select_bwd_f");
  return ASM_OK;
}

extern "C" int asm_se_scale_fwd(const void* x, const float* e, void* y, int N, int HW, int C, void* stream) {
  ASM_REQUIRE(x && e && y && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "se_scale_fwd: bad arguments");
  const size_t nvec = (size_t)N * HW * (C / 8);
  ASM_REQUIRE(nvec < 0x7fffffffull, "sk/se: tensor too large for 32-bit indexing");
  ASM_LAUNCH(se_scale_fwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, e,
                     (bf16_t*)y, N, HW, C);
  ASM_CHECK_LAUNCH("se_scale_fwd");
  return ASM_OK;
}

extern "C" int asm_se_scale_bwd_e(const void* x, const void* dy, const float* e, void* de, int N, int HW, int C,
                                  void* stream) {
  ASM_REQUIRE(x && dy && e && de && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "se_scale_bwd_e: bad arguments");
  const int vcb = C / 8 < 32 ? C / 8 : 32;
  ASM_LAUNCH(se_bwd_e_kernel, dim3(cdiv(C / 8, vcb), N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (const bf16_t*)dy, e, (bf16_t*)de, HW, C, vcb);
  ASM_CHECK_LAUNCH("se_scale_bwd_e");
  return ASM_OK;
}

extern "C" int asm_se_scale_bwd_x(const void* dy, const float* e, const void* dsq, void* dx, int N, int HW, int C,
                                  void* stream) {
  ASM_REQUIRE(dy && e && dsq && dx && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "se_scale_bwd_x: bad arguments");
  const size_t nvec = (size_t)N * HW * (C / 8);
  ASM_REQUIRE(nvec < 0x7fffffffull, "sk/se: tensor too large for 32-bit indexing");
  ASM_LAUNCH(se_bwd_x_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, e,
                     (const bf16_t*)dsq, (bf16_t*)dx, N, HW, C);
  ASM_CHECK_LAUNCH("se_scale_bwd_x");
  return ASM_OK;
}
