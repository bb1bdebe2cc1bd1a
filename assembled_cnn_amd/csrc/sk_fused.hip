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

// Cross-row-lane sum of 8 per-thread values -> `out` in the rl == 0 thread of each vector column.  Two levels (round 4):
// the row lanes that share a wave are summed by xor-shuffles (lanes vcb, 2 vcb, ... apart hold the same vector column),
// then one row of LDS per wave and a 16-term sum.  The one-level form had the vcb leader lanes walk all NT / vcb row lanes:
// 1024 LDS reads on 8 lanes for the 56 x 56 maps, ~10 % of the kernel -- and five such sums in the statistics variants.
template <int NT>
__device__ __forceinline__ void lane_sum8(float (*red)[9], const float* v, float* out, int vcb, int vcl, int nrl, bool leader) {
  if ((vcb & (vcb - 1)) == 0 && vcb <= 64) {
    float w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = v[e];
    for (int off = vcb; off < 64; off <<= 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] += __shfl_xor(w[e], off, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane < vcb) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave * vcb + lane][e] = w[e];
    }
    __syncthreads();
    if (leader) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < NT / 64; ++r) t += red[r * vcb + vcl][e];
        out[e] = t;
      }
    }
    return;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = v[e];
  __syncthreads();
  if (leader) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
      for (int r = 0; r < nrl; ++r) t += red[r * vcb + vcl][e];
      out[e] = t;
    }
  }
}

// f, its ReLU mask ([y * sc + sh > 0], as in the backward kernels) and the masked raw value of 8 channels of y.  The
// statistics are sums of mask and mask * y (and the same times dV); xhat = (y - mean) / sigma is affine in y, so the
// finalize turns them into the xhat sums per image -- the streaming loops carry no mean / sigma registers.
__device__ __forceinline__ void bnrelu8m(const u32x4& vy, const Coef8& k, float* f, float* m, float* my) {
  float y[8];
  unpack8(vy, y);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float t = y[e] * k.sc[e] + k.sh[e];
    f[e] = rbf(fmaxf(t, 0.f));
    m[e] = t > 0.f ? 1.f : 0.f;
    my[e] = t > 0.f ? y[e] : 0.f;
  }
}

// ---- s[n][c] = mean_hw( f0 + f1 ) ----------------------------------------------------------------------------------
// ST: also the per-image mask statistics the factorised batch-norm backward needs (see sk_bn_bwd_finalize_kernel):
//     stats[n][0][ch] = sum_hw [f > 0],  stats[n][1][ch] = sum_hw [f > 0] * y     (ch over the 2F conv channels)
template <int NT, bool ST>
__global__ __launch_bounds__(NT, NT == 256 ? 4 : 1) void sk_gap_bn_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, bf16_t* __restrict__ s, int HW,
                                                       int F, int vcb, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, float* __restrict__ stats) {
  __shared__ float red[NT][9];
  const int vcols = F >> 3;
  const int vcl = threadIdx.x % vcb, rl = threadIdx.x / vcb, nrl = NT / vcb;
  const int vc = blockIdx.x * vcb + vcl;
  const int n = blockIdx.y;
  float acc[8], c0[8], x0[8], c1[8], x1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = c0[e] = x0[e] = c1[e] = x1[e] = 0.f;
  if (vc < vcols && rl < nrl) {
    Coef8 k0, k1;
    load_coef(scale, shift, vc * 8, k0);
    load_coef(scale, shift, F + vc * 8, k1);
    auto consume = [&](const u32x4& v, const u32x4& w) {
      if (ST) {
        float f[8], m[8], my[8];
        bnrelu8m(v, k0, f, m, my);
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc[e] += f[e]; c0[e] += m[e]; x0[e] += my[e]; }
        bnrelu8m(w, k1, f, m, my);
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc[e] += f[e]; c1[e] += m[e]; x1[e] += my[e]; }
      } else {
        float f0[8], f1[8];
        bnrelu8(v, k0, f0);
        bnrelu8(w, k1, f1);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f0[e] + f1[e];
      }
    };
    constexpr int U = ST ? 2 : 4;   // the statistics variant carries 40 accumulators: fewer vectors in flight, same occupancy
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
      for (int u = 0; u < U; ++u) consume(v[u], w[u]);
    }
    for (; r < HW; r += nrl) {
      const size_t off = (size_t)r * 2 * F;
      consume(ldv(base, off), ldv(base, off + F));
    }
  }
  const bool leader = rl == 0 && vc < vcols;
  float o[8];
  lane_sum8<NT>(red, acc, o, vcb, vcl, nrl, leader);
  if (leader) {
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= inv;
    *reinterpret_cast<u32x4*>(s + (size_t)n * F + vc * 8) = pack8(o);
  }
  if (ST) {
    float* st = stats + (size_t)n * 4 * F;     // [2][2F]
    const float* src[4] = {c0, c1, x0, x1};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      lane_sum8<NT>(red, src[q], o, vcb, vcl, nrl, leader);
      if (leader) {
        float* dst = st + (q >> 1) * 2 * F + (q & 1) * F + vc * 8;
        *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
      }
    }
  }
}

// ---- V = a0 f0 + a1 f1 -----------------------------------------------------------------------------------------------
// grid (chunks, N); block = F/8 vector columns x rpb row lanes: a thread keeps its 8 channels of one image, so the two
// coefficient vectors and the gates (8 exponentials) are derived ONCE and the loop streams y -> V.  (Round 4: the
// one-vector-per-thread form re-derived 48 scalars and 8 exponentials for every 48 bytes moved: 4.9 TB/s.)
__global__ __launch_bounds__(256) void sk_select_bn_fwd_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ att, bf16_t* __restrict__ v,
                                                               int HW, int F, int rpb, int rows_per_chunk) {
  const int fv = F >> 3;
  const int vc = threadIdx.x % fv, rr = threadIdx.x / fv;
  if (rr >= rpb) return;
  const int n = blockIdx.y;
  Coef8 k0, k1;
  load_coef(scale, shift, vc * 8, k0);
  load_coef(scale, shift, F + vc * 8, k1);
  float a0[8];
  gate8(att, n, F, vc * 8, a0);
  const int r_begin = blockIdx.x * rows_per_chunk;
  const int r_end = min(HW, r_begin + rows_per_chunk);
  constexpr int U = 2;
  for (int r = r_begin + rr; r < r_end; r += U * rpb) {
    u32x4 y0[U], y1[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int rq = r + q * rpb;
      const size_t m = (size_t)n * HW + (rq < r_end ? rq : r);
      y0[q] = ldv(y, m * 2 * F + vc * 8);
      y1[q] = ldv(y, m * 2 * F + F + vc * 8);
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int rq = r + q * rpb;
      if (rq >= r_end) break;
      float f0[8], f1[8], o[8];
      bnrelu8(y0[q], k0, f0);
      bnrelu8(y1[q], k1, f1);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = a0[e] * f0[e] + (1.0f - a0[e]) * f1[e];
      *reinterpret_cast<u32x4*>(v + ((size_t)n * HW + rq) * F + vc * 8) = pack8(o);
    }
  }
}

// ---- datt[n][c] = a0 a1 sum_hw (f0 - f1) dV ;  datt[n][F + c] = -that ----------------------------------------------------
// ST: also the per-image gradient statistics of the factorised batch-norm backward:
//     stats[n][0][ch] = sum_hw [f > 0] dV,  stats[n][1][ch] = sum_hw [f > 0] dV y    (ch over the 2F conv channels)
template <int NT, bool ST>
__global__ __launch_bounds__(NT, NT == 256 ? 4 : 1) void sk_bn_bwd_att_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const bf16_t* __restrict__ dv,
                                                           const float* __restrict__ att, bf16_t* __restrict__ datt,
                                                           int HW, int F, int vcb, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, float* __restrict__ stats) {
  __shared__ float red[NT][9];
  const int vcols = F >> 3;
  const int vcl = threadIdx.x % vcb, rl = threadIdx.x / vcb, nrl = NT / vcb;
  const int vc = blockIdx.x * vcb + vcl;
  const int n = blockIdx.y;
  float acc[8], g0[8], x0[8], g1[8], x1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = g0[e] = x0[e] = g1[e] = x1[e] = 0.f;
  if (vc < vcols && rl < nrl) {
    Coef8 k0, k1;
    load_coef(scale, shift, vc * 8, k0);
    load_coef(scale, shift, F + vc * 8, k1);
    auto consume = [&](const u32x4& v0, const u32x4& v1, const u32x4& vg) {
      float g[8];
      unpack8(vg, g);
      if (ST) {
        float f[8], m[8], my[8];
        bnrelu8m(v0, k0, f, m, my);
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc[e] += f[e] * g[e]; g0[e] += m[e] * g[e]; x0[e] += my[e] * g[e]; }
        bnrelu8m(v1, k1, f, m, my);
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc[e] -= f[e] * g[e]; g1[e] += m[e] * g[e]; x1[e] += my[e] * g[e]; }
      } else {
        float f0[8], f1[8];
        bnrelu8(v0, k0, f0);
        bnrelu8(v1, k1, f1);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (f0[e] - f1[e]) * g[e];
      }
    };
    constexpr int U = ST ? 1 : 2;
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
      for (int u = 0; u < U; ++u) consume(v0[u], v1[u], vg[u]);
    }
    for (; r < HW; r += nrl) {
      const size_t m = (size_t)n * HW + r;
      consume(ldv(y, m * 2 * F + vc * 8), ldv(y, m * 2 * F + F + vc * 8), ldv(dv, m * F + vc * 8));
    }
  }
  const bool leader = rl == 0 && vc < vcols;
  float o[8];
  lane_sum8<NT>(red, acc, o, vcb, vcl, nrl, leader);
  if (leader) {
    float a0[8], d0[8], d1[8];
    gate8(att, n, F, vc * 8, a0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      d0[e] = a0[e] * (1.0f - a0[e]) * o[e];
      d1[e] = -d0[e];
    }
    *reinterpret_cast<u32x4*>(datt + (size_t)n * 2 * F + vc * 8) = pack8(d0);
    *reinterpret_cast<u32x4*>(datt + (size_t)n * 2 * F + F + vc * 8) = pack8(d1);
  }
  if (ST) {
    float* st = stats + (size_t)n * 4 * F;     // [2][2F]
    const float* src[4] = {g0, g1, x0, x1};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      lane_sum8<NT>(red, src[q], o, vcb, vcl, nrl, leader);
      if (leader) {
        float* dst = st + (q >> 1) * 2 * F + (q & 1) * F + vc * 8;
        *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
      }
    }
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
        vy[q] = *reinterpret_cast<const u32x4*>(y + m * C2 + vc * 8);   // plain: the apply pass behind re-reads y (bn.hip)
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

// ---- factorised reduce + finalize ---------------------------------------------------------------------------------------
// dz = [f > 0] (a_b dV + ds / HW) with a_b and ds constant over an image, so the two batch-norm backward sums are
//   sum dz      = sum_n  a_b[n] * G0[n] + (ds[n] / HW) * M0[n]          G0 = sum_hw [f>0] dV        M0 = sum_hw [f>0]
//   sum dz xhat = sum_n  a_b[n] * G1[n] + (ds[n] / HW) * M1[n]          G1 = sum_hw [f>0] dV xhat   M1 = sum_hw [f>0] xhat
// (the passes accumulate y in place of xhat = (y - mean) / sigma, which is affine in y: mean and sigma enter here)
// G comes out of the gate-gradient pass (which reads y and dV anyway), M out of the forward pooled-sum pass: the reduce
// pass over the whole tensor (y and dV once more) disappears.  One block = 16 channels x 64 image lanes, fp64 sums.
__global__ __launch_bounds__(1024) void sk_bn_bwd_finalize_kernel(const float* __restrict__ gst, const float* __restrict__ mst,
                                                                  const float* __restrict__ att, const bf16_t* __restrict__ ds,
                                                                  int N, int HW, int F, const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, float* dgamma,
                                                                  float* dbeta, float* coefA, float* coefB, float* coefC) {
  __shared__ double red[2][64][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int C2 = 2 * F;
  const int ch = blockIdx.x * 16 + cx;
  double s0 = 0.0, s1 = 0.0;
  if (ch < C2) {
    const int b = ch >= F ? 1 : 0, c = ch - b * F;
    const float inv = 1.0f / (float)HW;
    for (int n = ry; n < N; n += 64) {
      const float l0 = att[(size_t)n * C2 + c], l1 = att[(size_t)n * C2 + F + c];
      const float a0 = 1.0f / (1.0f + __expf(l1 - l0));
      const float ab = b ? (1.0f - a0) : a0;
      const float u = bf2f(ds[(size_t)n * F + c]) * inv;
      const float* gs = gst + (size_t)n * 2 * C2;
      const float* ms = mst + (size_t)n * 2 * C2;
      const double g0 = gs[ch], m0 = ms[ch];
      const double w = (double)ab * g0 + (double)u * m0;                               // sum_hw dz of this image
      s0 += w;
      s1 += (double)ab * (double)gs[C2 + ch] + (double)u * (double)ms[C2 + ch];        // sum_hw dz * y
      s1 -= (double)mean[ch] * w;                                                       // -> sum_hw dz * (y - mean)
    }
    s1 *= (double)invstd[ch];     // xhat = (y - mean) * invstd
  }
  red[0][ry][cx] = s0;
  red[1][ry][cx] = s1;
  __syncthreads();
  if (ry == 0 && ch < C2) {
    double db = 0.0, dg = 0.0;
    for (int r = 0; r < 64; ++r) {
      db += red[0][r][cx];
      dg += red[1][r][cx];
    }
    dbeta[ch] = (float)db;
    dgamma[ch] = (float)dg;
    const double M = (double)N * (double)HW;
    const double g = gamma[ch], is = invstd[ch], mu = mean[ch];
    const double A = g * is;
    const double B = -g * is * is * dg / M;
    coefA[ch] = (float)A;
    coefB[ch] = (float)B;
    coefC[ch] = (float)(-g * is * db / M - B * mu);
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

// Vector columns per workgroup of the per-image passes (pooled sum, gate gradient; grid = column groups x images).  Until
// round 4: min(F / 8, 32), i.e. ONE workgroup per image for every layer of the workload -- 256 workgroups of 4 waves on the
// 14 x 14 and 7 x 7 maps, 12 KB in flight per CU: 2.1 - 2.8 TB/s.  Now the columns are cut so that ~1024 workgroups exist
// (8 columns = one 128-byte line per row segment is the floor).
int image_pass_vcb(int N, int F) {
  const int vcols = F / 8;
  int groups = cdiv(1024, N > 0 ? N : 1);
  if (groups < 1) groups = 1;
  int vcb = 32;
  while (vcb > 8 && cdiv(vcols, vcb) < groups) vcb >>= 1;
  return vcols < vcb ? vcols : vcb;
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

static int sk_gap_bn_impl(const void* y, const float* scale, const float* shift, void* s, int N, int HW, int F,
                          const float* mean, const float* invstd, float* stats, void* stream) {
  SKF_OK("sk_gap_bn");
  ASM_REQUIRE(y && scale && shift && s, "sk_gap_bn: null pointer");
  const int vcb = image_pass_vcb(N, F);
  const dim3 grid(cdiv(F / 8, vcb), N);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_GAP(NT, ST)                                                                                       \
  ASM_LAUNCH((sk_gap_bn_kernel<NT, ST>), grid, dim3(NT), 0, st, (const bf16_t*)y, scale, shift, (bf16_t*)s, HW, F, \
                     vcb, mean, invstd, stats)
  if (stats) { if (HW >= 512) LAUNCH_GAP(1024, true); else LAUNCH_GAP(256, true); }
  else { if (HW >= 512) LAUNCH_GAP(1024, false); else LAUNCH_GAP(256, false); }
#undef LAUNCH_GAP
  ASM_CHECK_LAUNCH("sk_gap_bn");
  return ASM_OK;
}

extern "C" int asm_sk_gap_bn(const void* y, const float* scale, const float* shift, void* s, int N, int HW, int F,
                             void* stream) {
  return sk_gap_bn_impl(y, scale, shift, s, N, HW, F, nullptr, nullptr, nullptr, stream);
}

extern "C" int asm_sk_gap_bn_stats(const void* y, const float* scale, const float* shift, const float* mean,
                                   const float* invstd, void* s, float* mask_stats, int N, int HW, int F, void* stream) {
  ASM_REQUIRE(mean && invstd && mask_stats, "sk_gap_bn_stats: null pointer");
  return sk_gap_bn_impl(y, scale, shift, s, N, HW, F, mean, invstd, mask_stats, stream);
}

extern "C" int asm_sk_select_bn_fwd(const void* y, const float* scale, const float* shift, const float* att, void* v,
                                    int N, int HW, int F, void* stream) {
  SKF_OK("sk_select_bn_fwd");
  ASM_REQUIRE(y && scale && shift && att && v, "sk_select_bn_fwd: null pointer");
  // the (chunk, image) decomposition of the backward passes (make_geom) over the F / 8 vector columns of V
  const int fv = F / 8, rpb = 256 / fv;
  int chunks = 2048 / N;
  if (chunks < 1) chunks = 1;
  int rows = cdiv(HW, chunks);
  rows = cdiv(rows, rpb) * rpb;
  if (rows < rpb * 4) rows = rpb * 4;
  ASM_LAUNCH(sk_select_bn_fwd_kernel, dim3(cdiv(HW, rows), N), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)y, scale, shift, att, (bf16_t*)v, HW, F, rpb, rows);
  ASM_CHECK_LAUNCH("sk_select_bn_fwd");
  return ASM_OK;
}

static int sk_att_impl(const void* y, const float* scale, const float* shift, const void* dv, const float* att, void* datt,
                       int N, int HW, int F, const float* mean, const float* invstd, float* stats, void* stream) {
  SKF_OK("sk_select_bn_bwd_att");
  ASM_REQUIRE(y && scale && shift && dv && att && datt, "sk_select_bn_bwd_att: null pointer");
  const int vcb = image_pass_vcb(N, F);
  const dim3 grid(cdiv(F / 8, vcb), N);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_ATT(NT, ST)                                                                                          \
  ASM_LAUNCH((sk_bn_bwd_att_kernel<NT, ST>), grid, dim3(NT), 0, st, (const bf16_t*)y, scale, shift, (const bf16_t*)dv, \
                     att, (bf16_t*)datt, HW, F, vcb, mean, invstd, stats)
  if (stats) { if (HW >= 512) LAUNCH_ATT(1024, true); else LAUNCH_ATT(256, true); }
  else { if (HW >= 512) LAUNCH_ATT(1024, false); else LAUNCH_ATT(256, false); }
#undef LAUNCH_ATT
  ASM_CHECK_LAUNCH("sk_select_bn_bwd_att");
  return ASM_OK;
}

extern "C" int asm_sk_select_bn_bwd_att(const void* y, const float* scale, const float* shift, const void* dv,
                                        const float* att, void* datt, int N, int HW, int F, void* stream) {
  return sk_att_impl(y, scale, shift, dv, att, datt, N, HW, F, nullptr, nullptr, nullptr, stream);
}

extern "C" int asm_sk_select_bn_bwd_att_stats(const void* y, const float* scale, const float* shift, const float* mean,
                                              const float* invstd, const void* dv, const float* att, void* datt,
                                              float* grad_stats, int N, int HW, int F, void* stream) {
  ASM_REQUIRE(mean && invstd && grad_stats, "sk_select_bn_bwd_att_stats: null pointer");
  return sk_att_impl(y, scale, shift, dv, att, datt, N, HW, F, mean, invstd, grad_stats, stream);
}

extern "C" int asm_sk_bn_bwd_finalize(const float* grad_stats, const float* mask_stats, const float* att, const void* ds,
                                      int N, int HW, int F, const float* gamma, const float* mean, const float* invstd,
                                      float* dgamma, float* dbeta, float* coefA, float* coefB, float* coefC, void* stream) {
  SKF_OK("sk_bn_bwd_finalize");
  ASM_REQUIRE(grad_stats && mask_stats && att && ds && gamma && mean && invstd && dgamma && dbeta && coefA && coefB && coefC,
              "sk_bn_bwd_finalize: null pointer");
  ASM_LAUNCH(sk_bn_bwd_finalize_kernel, dim3(cdiv(2 * F, 16)), dim3(1024), 0, (hipStream_t)stream, grad_stats,
                     mask_stats, att, (const bf16_t*)ds, N, HW, F, gamma, mean, invstd, dgamma, dbeta, coefA, coefB, coefC);
  ASM_CHECK_LAUNCH("sk_bn_bwd_finalize");
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
  ASM_LAUNCH(sk_bn_bwd_reduce_kernel, dim3(g.chunks, N), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dv, att,
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
  ASM_LAUNCH(sk_bn_bwd_apply_kernel, dim3(g.chunks, N), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dv, att, (const bf16_t*)ds, (const bf16_t*)y, scale, shift, coefA, coefB, coefC,
                     (bf16_t*)dy, g);
  ASM_CHECK_LAUNCH("sk_bn_bwd_apply");
  return ASM_OK;
}
