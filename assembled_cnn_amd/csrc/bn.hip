// Batch-norm kernels for NHWC bf16 activations viewed as [M, C] (HBM-bound, 16-byte vectors).
// Reductions are two-level and atomics-free: per-block partials [blocks][2][C] (the same layout the
// conv epilogue emits) followed by a tiny finalize that sums the partials in fp64.
#include "common.h"
#include <stdlib.h>


// Cache policy of the streaming kernels (same-box A/B, whole step): the APPLY passes read and write with non-temporal
// hints (their operands are dead afterwards / the result is consumed by a convolution that streams it once): -0.3 ms per
// step in round 1.  The backward REDUCE passes load dy / x plainly (round 4): the apply pass behind them re-reads exactly
// these bytes, and what still sits in L2 / the memory-side cache then is served from there -- 26.42 -> 26.17 ms per step;
// plain stores in the apply passes measured +0.1 .. 0.2 ms.
#define BN_LD_RED(p) (*(p))
#define BN_ST_BWD(v, p) __builtin_nontemporal_store(v, p)
#define BN_ST_FWD(v, p) __builtin_nontemporal_store(v, p)

namespace {

constexpr int RED_FLOATS = 4096;  // 2 stats x 256 threads x 8 lanes

struct RowTiling {
  int vcols;      // C / 8
  int vcb;        // vector columns handled concurrently (<= 256)
  int rpb;        // row lanes per block
  int rows_per_block;
  int blocks;     // row blocks = partial rows per channel
  int slices;     // channel slices: a workgroup reduces rows_per_block rows of ONE slice (grid = blocks * slices)
  int vc_slice;   // vector columns per slice
};

RowTiling make_tiling(int M, int C) {
  RowTiling t;
  t.vcols = C / 8;
  // Channel slices (round 4).  The reducers want ~1024 workgroups (4 per CU: with 512 they ran at 4.4 TB/s, with 1024 at
  // 5.7 TB/s -- occupancy-bound streaming), and until now every workgroup covered ALL channels of its rows: 1024 partial
  // rows per channel, which the finalize kernel behind the reducer walks as a chain of dependent L2 round trips (16 rows
  // per lane: 6.5 us for a launch that moves a few hundred KB, ~80 such launches per backward pass).  With the channels
  // cut into slices of >= 64 (a row segment stays a whole 128-byte line) the same 1024 workgroups leave 1024 / slices
  // partial rows per channel.
  const int want = 8;
  int sl = t.vcols / 8;
  if (sl > want) sl = want;
  if (sl < 1) sl = 1;
  t.slices = sl;
  t.vc_slice = cdiv(t.vcols, sl);
  t.vcb = t.vc_slice < 256 ? t.vc_slice : 256;
  t.rpb = 256 / t.vcb;
  const int cap = asm_tune().bn_rows / sl > 0 ? asm_tune().bn_rows / sl : 1;
  int rows = cdiv(M, cap);
  rows = cdiv(rows, t.rpb) * t.rpb;
  if (rows < t.rpb * 4) rows = t.rpb * 4;
  t.rows_per_block = rows;
  t.blocks = cdiv(M, rows);
  // Which rows a workgroup takes: workgroup b takes the row tiles b, b + blocks, b + 2 blocks, ... (a tile = rpb x 2 rows), the
  // way the apply passes' grid-stride loops sweep ONE compact window over the tensor.  (One contiguous block of rows each --
  // rounds 1 - 3 -- put ~1000 streams M / blocks rows apart: 1.6 MB for the 56 x 56 x 256 tensor, a multiple of 32 KB, so in lock
  // step they asked the same HBM channels for different DRAM pages: 4.3 - 4.5 TB/s against 6.2.)
  return t;
}

// ---- generic two-stat reduction over rows ------------------------------------------------------
// MODE 0: (sum x, sum x^2)            inputs: a = x
// MODE 1: (sum dz, sum dz * xhat)     inputs: a = dy, b = x, c = yout (ReLU mask, optional)
template <int MODE>
__global__ __launch_bounds__(256) void rowreduce_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                        const bf16_t* __restrict__ c, int relu, int M, int C,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, RowTiling t,
                                                        float* __restrict__ partial) {
  __shared__ float red[RED_FLOATS];
  const int tid = threadIdx.x;
  const int vc0 = tid % t.vcb;
  const int rr = tid / t.vcb;
  const bool active = rr < t.rpb;
  const int rdisp = blockIdx.x / t.slices, slice = blockIdx.x - rdisp * t.slices;
  const int rblk = rdisp;
  const int vc_lo = slice * t.vc_slice, vc_hi = min(t.vcols, vc_lo + t.vc_slice);
  constexpr int U = 2;
  const int row_begin = rblk * (U * t.rpb);
  const int row_end = M;
  const int row_step = t.blocks * (U * t.rpb);
  for (int vcbase = vc_lo; vcbase < vc_hi; vcbase += t.vcb) {
    const int vc = vcbase + vc0;
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
    if (active && vc < vc_hi) {
      float mu[8], is[8];
      if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          mu[e] = mean[vc * 8 + e];
          is[e] = invstd[vc * 8 + e];
        }
      }
      // 2 rows per trip (4 was ~5 % slower, 8 ~35 %: registers cost occupancy): all loads of a trip are issued before any is consumed (the reduction is
      // bandwidth-bound only if enough bytes are in flight per CU)
      for (int row = row_begin + rr; row < row_end; row += row_step) {
        u32x4 va[U], vb[U], vy[U];
        unsigned mk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int r = row + u * t.rpb;
          const bool ok = r < row_end;
          const size_t off = (size_t)(ok ? r : row) * C + vc * 8;
          va[u] = BN_LD_RED(reinterpret_cast<const u32x4*>(a + off));
          if (MODE == 1) {
            vb[u] = BN_LD_RED(reinterpret_cast<const u32x4*>(b + off));
            if (relu == 1) vy[u] = *reinterpret_cast<const u32x4*>(c + off);
            if (relu == 2) mk[u] = reinterpret_cast<const uint8_t*>(c)[(size_t)(ok ? r : row) * t.vcols + vc];
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (row + u * t.rpb >= row_end) break;
          float fa[8];
          unpack8(va[u], fa);
          if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              s[e] += fa[e];
              ss[e] += fa[e] * fa[e];
            }
          } else {
            float fb[8];
            unpack8(vb[u], fb);
            if (relu == 1) {
              float fy[8];
              unpack8(vy[u], fy);
#pragma unroll
              for (int e = 0; e < 8; ++e) fa[e] = fy[e] > 0.f ? fa[e] : 0.f;
            } else if (relu == 2) {  // packed ReLU mask: one byte per 8-channel vector
#pragma unroll
              for (int e = 0; e < 8; ++e) fa[e] = ((mk[u] >> e) & 1u) ? fa[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              s[e] += fa[e];
              ss[e] += fa[e] * ((fb[e] - mu[e]) * is[e]);
            }
          }
        }
      }
    }
    // cross row-lane reduction through LDS: red[stat][rr][vc0*8+e]
    __syncthreads();
    if (active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(0 * t.rpb + rr) * (t.vcb * 8) + vc0 * 8 + e] = s[e];
        red[(1 * t.rpb + rr) * (t.vcb * 8) + vc0 * 8 + e] = ss[e];
      }
    }
    __syncthreads();
    const int ncol = t.vcb * 8;
    for (int i = tid; i < 2 * ncol; i += 256) {
      const int which = i / ncol, col = i - which * ncol;
      float acc = 0.f;
      for (int r = 0; r < t.rpb; ++r) acc += red[(which * t.rpb + r) * ncol + col];
      const int ch = vcbase * 8 + col;
      if (ch < vc_hi * 8) partial[((size_t)rblk * 2 + which) * C + ch] = acc;
    }
  }
}

// ---- finalize: partials -> per-channel statistics / coefficients --------------------------------
// block = 16 channels x FL partial-lanes.  These kernels are pure latency (one per BN layer and direction, ~200 per
// step): 64 lanes keep the dependent trip count at <= 4 for the <= 512 partial rows the reducers emit.
constexpr int FL = 64;
// Rows b0 + lane, b0 + lane + nl, ... < b1 of the [blocks][2][C] partials for channel ch, summed in fp64.  SU row pairs
// (2 * SU loads) are issued before any is consumed: the serial form was a chain of dependent L2 / HBM round trips -- ~5-6 us
// per finalize launch, ~190 such launches per training step.  The summation order is fixed (bit-reproducible).
constexpr int SU = 8;
__device__ __forceinline__ void sum_rows(const float* __restrict__ partial, int b0, int b1, int lane, int nl, int C, int ch,
                                         double& s0, double& s1) {
  double a[SU], c[SU];
#pragma unroll
  for (int u = 0; u < SU; ++u) a[u] = c[u] = 0.0;
  int b = b0 + lane;
  for (; b + (SU - 1) * nl < b1; b += SU * nl) {
    float v0[SU], v1[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      v0[u] = partial[((size_t)(b + u * nl) * 2 + 0) * C + ch];
      v1[u] = partial[((size_t)(b + u * nl) * 2 + 1) * C + ch];
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      a[u] += (double)v0[u];
      c[u] += (double)v1[u];
    }
  }
  {   // tail: up to SU - 1 rows, still issued together
    float v0[SU], v1[SU];
#pragma unroll
    for (int u = 0; u < SU - 1; ++u) {
      const bool ok = b + u * nl < b1;
      v0[u] = ok ? partial[((size_t)(b + u * nl) * 2 + 0) * C + ch] : 0.f;
      v1[u] = ok ? partial[((size_t)(b + u * nl) * 2 + 1) * C + ch] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < SU - 1; ++u) {
      a[u] += (double)v0[u];
      c[u] += (double)v1[u];
    }
  }
  s0 = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  s1 = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
}
// finalize kernels: two independent chains per lane.  (Issuing all 16 rows of a lane at once, as sum_rows does, was
// measured SLOWER here -- 12.5 vs 6.5 us for the 1024-row backward partials -- while it helps the compaction, whose lanes
// walk hundreds of rows.)
__device__ __forceinline__ void sum_partials(const float* partial, int blocks, int C, int ch, int ry,
                                             double& s0, double& s1) {
  s0 = 0.0;
  s1 = 0.0;
  if (ch < C) {
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    int b = ry;
    for (; b + FL < blocks; b += 2 * FL) {   // two independent chains: loads overlap
      a0 += (double)partial[((size_t)b * 2 + 0) * C + ch];
      b0 += (double)partial[((size_t)b * 2 + 1) * C + ch];
      a1 += (double)partial[((size_t)(b + FL) * 2 + 0) * C + ch];
      b1 += (double)partial[((size_t)(b + FL) * 2 + 1) * C + ch];
    }
    if (b < blocks) {
      a0 += (double)partial[((size_t)b * 2 + 0) * C + ch];
      b0 += (double)partial[((size_t)b * 2 + 1) * C + ch];
    }
    s0 = a0 + a1;
    s1 = b0 + b1;
  }
}

// partial compaction: [blocks][2][C] -> [groups][2][C]; each output group sums a contiguous range of
// input blocks (fp64 accumulate).  Keeps the finalize kernels short when a conv epilogue emitted
// thousands of 128-row partials (25088 for the 112x112 stem at batch 256).
__global__ __launch_bounds__(16 * FL) void partials_compact_kernel(const float* __restrict__ in, int blocks, int C,
                                                                   float* __restrict__ out, int per_group) {
  __shared__ double red[2][FL][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cx;
  const int b0 = blockIdx.y * per_group;
  const int b1 = min(blocks, b0 + per_group);
  double s0 = 0.0, s1 = 0.0;
  if (ch < C) sum_rows(in, b0, b1, ry, FL, C, ch, s0, s1);
  red[0][ry][cx] = s0;
  red[1][ry][cx] = s1;
  __syncthreads();
  if (ry < 2 && ch < C) {
    double t = 0.0;
    for (int r = 0; r < FL; ++r) t += red[ry][r][cx];
    out[((size_t)blockIdx.y * 2 + ry) * C + ch] = (float)t;
  }
}

__global__ __launch_bounds__(16 * FL) void bn_finalize_kernel(const float* __restrict__ partial, int blocks, int M, int C,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum,
                                                          float* moving_mean, float* moving_var, float* mean,
                                                          float* invstd, float* scale, float* shift) {
  __shared__ double red[2][FL][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cx;
  double s0, s1;
  sum_partials(partial, blocks, C, ch, ry, s0, s1);
  red[0][ry][cx] = s0;
  red[1][ry][cx] = s1;
  __syncthreads();
  if (ry == 0 && ch < C) {
    double a = 0.0, b = 0.0;
    for (int r = 0; r < FL; ++r) {
      a += red[0][r][cx];
      b += red[1][r][cx];
    }
    const double mu = a / (double)M;
    double var = b / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[ch] * is;
    mean[ch] = (float)mu;
    invstd[ch] = is;
    scale[ch] = sc;
    shift[ch] = beta[ch] - (float)mu * sc;
    if (moving_mean) {
      const double unbiased = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
      moving_mean[ch] = moving_mean[ch] * momentum + (float)mu * (1.f - momentum);
      moving_var[ch] = moving_var[ch] * momentum + (float)unbiased * (1.f - momentum);
    }
  }
}

__global__ __launch_bounds__(16 * FL) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int blocks, int M,
                                                              int C, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, float* dgamma,
                                                              float* dbeta, float* coefA, float* coefB,
                                                              float* coefC, int raw) {
  __shared__ double red[2][FL][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cx;
  double s0, s1;
  sum_partials(partial, blocks, C, ch, ry, s0, s1);
  red[0][ry][cx] = s0;
  red[1][ry][cx] = s1;
  __syncthreads();
  if (ry == 0 && ch < C) {
    double db = 0.0, dg = 0.0;
    for (int r = 0; r < FL; ++r) {
      db += red[0][r][cx];
      dg += red[1][r][cx];
    }
    const double g = gamma[ch], is = invstd[ch], mu = mean[ch];
    if (raw) dg = is * (dg - mu * db);      // partials of (sum dz, sum dz * y) from an input-gradient epilogue: -> sum dz * xhat
    dbeta[ch] = (float)db;
    dgamma[ch] = (float)dg;
    const double A = g * is;
    const double B = -g * is * is * dg / (double)M;
    const double Cc = -g * is * db / (double)M - B * mu;
    coefA[ch] = (float)A;
    coefB[ch] = (float)B;
    coefC[ch] = (float)Cc;
  }
}

__global__ void bn_infer_coeffs_kernel(int C, const float* gamma, const float* beta, const float* mm,
                                       const float* mv, float eps, float* scale, float* shift) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch < C) {
    const float is = 1.0f / sqrtf(mv[ch] + eps);
    const float sc = gamma[ch] * is;
    scale[ch] = sc;
    shift[ch] = beta[ch] - mm[ch] * sc;
  }
}

// ---- apply: y = [relu](x*scale + shift [+ residual]) ---------------------------------------------
template <int RES, bool RELU>
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                       size_t nvec, int C, FastDiv fd_vcols,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const bf16_t* __restrict__ res, FastDiv fd_w, FastDiv fd_h,
                                                       int H, int W, uint8_t* __restrict__ mask) {
  const int vcols = C >> 3;
  // When the grid stride is a multiple of C/8 a thread keeps the same 8 channels on every trip: their coefficients
  // are loaded once (they were 4 extra 16-byte loads per 16-byte data load, all L1 hits but all TA cycles).
  const size_t stride = (size_t)gridDim.x * 256;
  const bool fixed = (stride % (size_t)vcols) == 0;
  f32x4 s0, s1, h0, h1;
  if (fixed) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int vc0 = (int)(i0 % (size_t)vcols);
    s0 = *reinterpret_cast<const f32x4*>(scale + vc0 * 8);
    s1 = *reinterpret_cast<const f32x4*>(scale + vc0 * 8 + 4);
    h0 = *reinterpret_cast<const f32x4*>(shift + vc0 * 8);
    h1 = *reinterpret_cast<const f32x4*>(shift + vc0 * 8 + 4);
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    const unsigned m = fd_div((unsigned)i, fd_vcols);
    const int vc = (int)((unsigned)i - m * (unsigned)vcols);
    const u32x4 vx = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x + i * 8));
    float f[8];
    unpack8(vx, f);
    if (!fixed) {
      s0 = *reinterpret_cast<const f32x4*>(scale + vc * 8);
      s1 = *reinterpret_cast<const f32x4*>(scale + vc * 8 + 4);
      h0 = *reinterpret_cast<const f32x4*>(shift + vc * 8);
      h1 = *reinterpret_cast<const f32x4*>(shift + vc * 8 + 4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[e] = f[e] * s0[e] + h0[e];
      f[e + 4] = f[e + 4] * s1[e] + h1[e];
    }
    if (RES != 0) {
      size_t ri;
      if (RES == 1) {
        ri = i * 8;
      } else {
        const unsigned nh = fd_div(m, fd_w);
        const unsigned w = m - nh * (unsigned)W;
        const unsigned n = fd_div(nh, fd_h);
        const unsigned h = nh - n * (unsigned)H;
        ri = ((((size_t)n * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1)) * vcols + vc) * 8;
      }
      const u32x4 vr = *reinterpret_cast<const u32x4*>(res + ri);
      float r[8];
      unpack8(vr, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += r[e];
    }
    if (RELU) {
      if (mask) {
        unsigned mk = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) mk |= (f[e] > 0.f ? 1u : 0u) << e;
        mask[i] = (uint8_t)mk;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
    }
    BN_ST_FWD(pack8(f), reinterpret_cast<u32x4*>(y + i * 8));
  }
}

// ---- backward apply: dx = A*dz + B*x + C ; dz = dy * [yout > 0] -----------------------------------
template <int RELU, bool WRITE_DZ>  // RELU: 0 none, 1 mask from the bf16 forward output, 2 packed bitmask
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                           const bf16_t* __restrict__ yout, size_t nvec, int C,
                                                           FastDiv fd_vcols, const float* __restrict__ cA,
                                                           const float* __restrict__ cB,
                                                           const float* __restrict__ cC, bf16_t* __restrict__ dx,
                                                           bf16_t* __restrict__ dz) {
  const int vcols = C >> 3;
  const size_t stride = (size_t)gridDim.x * 256;
  const bool fixed = (stride % (size_t)vcols) == 0;   // see bn_apply_kernel
  float kA[8], kB[8], kC[8];
  if (fixed) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int vc0 = (int)(i0 % (size_t)vcols);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      kA[e] = cA[vc0 * 8 + e];
      kB[e] = cB[vc0 * 8 + e];
      kC[e] = cC[vc0 * 8 + e];
    }
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    float g[8], fx[8];
    unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(dy + i * 8)), g);
    unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x + i * 8)), fx);
    if (RELU == 1) {
      float fy[8];
      unpack8(*reinterpret_cast<const u32x4*>(yout + i * 8), fy);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = fy[e] > 0.f ? g[e] : 0.f;
    } else if (RELU == 2) {
      const unsigned mk = reinterpret_cast<const uint8_t*>(yout)[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = ((mk >> e) & 1u) ? g[e] : 0.f;
    }
    if (WRITE_DZ) *reinterpret_cast<u32x4*>(dz + i * 8) = pack8(g);
    if (!fixed) {
      const unsigned m = fd_div((unsigned)i, fd_vcols);
      const int vc = (int)((unsigned)i - m * (unsigned)vcols);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        kA[e] = cA[vc * 8 + e];
        kB[e] = cB[vc * 8 + e];
        kC[e] = cC[vc * 8 + e];
      }
    }
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = kA[e] * g[e] + kB[e] * fx[e] + kC[e];
    BN_ST_BWD(pack8(o), reinterpret_cast<u32x4*>(dx + i * 8));
  }
}

// ---- two batch norms behind ONE ReLU: the block-final BN and the projection-shortcut BN of a bottleneck ---------------
// out = relu(bn_a(ya) + bn_b(yb)): both backward passes consume the same masked gradient g = dout * [out > 0].  Run as two
// independent BN backwards that is 2 x (reduce: dout, y, mask; apply: dout, y, mask -> dy) = 20.5 B per element; here dout
// and the mask are read once per pass for both: 16.25 B per element and two launches fewer per projection block.
// Partials: pa = [blocks][2][C] (sum g, sum g * xhat_a), pb likewise for b (sum g repeated, so the finalize is unchanged).
__global__ __launch_bounds__(256) void rowreduce2_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ xa,
                                                         const bf16_t* __restrict__ xb, const uint8_t* __restrict__ mask,
                                                         int M, int C, const float* __restrict__ mean_a,
                                                         const float* __restrict__ invstd_a, const float* __restrict__ mean_b,
                                                         const float* __restrict__ invstd_b, RowTiling t,
                                                         float* __restrict__ pa, float* __restrict__ pb) {
  __shared__ float red[3 * 2048];
  const int tid = threadIdx.x;
  const int vc0 = tid % t.vcb;
  const int rr = tid / t.vcb;
  const bool active = rr < t.rpb;
  const int rdisp = blockIdx.x / t.slices, slice = blockIdx.x - rdisp * t.slices;
  const int rblk = rdisp;
  const int vc_lo = slice * t.vc_slice, vc_hi = min(t.vcols, vc_lo + t.vc_slice);
  constexpr int U = 2;
  const int row_begin = rblk * (U * t.rpb);
  const int row_end = M;
  const int row_step = t.blocks * (U * t.rpb);
  for (int vcbase = vc_lo; vcbase < vc_hi; vcbase += t.vcb) {
    const int vc = vcbase + vc0;
    float s[8], sa[8], sb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = sa[e] = sb[e] = 0.f;
    if (active && vc < vc_hi) {
      float ma[8], ia[8], mb[8], ib[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ma[e] = mean_a[vc * 8 + e];
        ia[e] = invstd_a[vc * 8 + e];
        mb[e] = mean_b[vc * 8 + e];
        ib[e] = invstd_b[vc * 8 + e];
      }
      for (int row = row_begin + rr; row < row_end; row += row_step) {
        u32x4 vg[U], va[U], vb[U];
        unsigned mk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int r = row + u * t.rpb;
          const size_t rq = (size_t)(r < row_end ? r : row);
          const size_t off = rq * C + vc * 8;
          vg[u] = BN_LD_RED(reinterpret_cast<const u32x4*>(dy + off));
          va[u] = BN_LD_RED(reinterpret_cast<const u32x4*>(xa + off));
          vb[u] = BN_LD_RED(reinterpret_cast<const u32x4*>(xb + off));
          mk[u] = mask[rq * t.vcols + vc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (row + u * t.rpb >= row_end) break;
          float g[8], fa[8], fb[8];
          unpack8(vg[u], g);
          unpack8(va[u], fa);
          unpack8(vb[u], fb);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float ge = ((mk[u] >> e) & 1u) ? g[e] : 0.f;
            s[e] += ge;
            sa[e] += ge * ((fa[e] - ma[e]) * ia[e]);
            sb[e] += ge * ((fb[e] - mb[e]) * ib[e]);
          }
        }
      }
    }
    __syncthreads();
    const int ncol = t.vcb * 8;
    if (active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(0 * t.rpb + rr) * ncol + vc0 * 8 + e] = s[e];
        red[(1 * t.rpb + rr) * ncol + vc0 * 8 + e] = sa[e];
        red[(2 * t.rpb + rr) * ncol + vc0 * 8 + e] = sb[e];
      }
    }
    __syncthreads();
    for (int i = tid; i < 3 * ncol; i += 256) {
      const int which = i / ncol, col = i - which * ncol;
      float acc = 0.f;
      for (int r = 0; r < t.rpb; ++r) acc += red[(which * t.rpb + r) * ncol + col];
      const int ch = vcbase * 8 + col;
      if (ch < vc_hi * 8) {
        if (which == 0) {
          pa[((size_t)rblk * 2 + 0) * C + ch] = acc;
          pb[((size_t)rblk * 2 + 0) * C + ch] = acc;
        } else if (which == 1) {
          pa[((size_t)rblk * 2 + 1) * C + ch] = acc;
        } else {
          pb[((size_t)rblk * 2 + 1) * C + ch] = acc;
        }
      }
    }
  }
}

// dxa = A_a * g + B_a * xa + C_a ; dxb likewise ; g = dy * [mask bit].  co = [6][C]: A_a, B_a, C_a, A_b, B_b, C_b.
__global__ __launch_bounds__(256) void bn_bwd_apply2_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ xa,
                                                            const bf16_t* __restrict__ xb, const uint8_t* __restrict__ mask,
                                                            size_t nvec, int C, FastDiv fd_vcols,
                                                            const float* __restrict__ co, bf16_t* __restrict__ dxa,
                                                            bf16_t* __restrict__ dxb) {
  const int vcols = C >> 3;
  const size_t stride = (size_t)gridDim.x * 256;
  const bool fixed = (stride % (size_t)vcols) == 0;   // see bn_apply_kernel
  float k[6][8];
  if (fixed) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int vc0 = (int)(i0 % (size_t)vcols);
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) k[q][e] = co[(size_t)q * C + vc0 * 8 + e];
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    float g[8], fa[8], fb[8];
    unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(dy + i * 8)), g);
    unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xa + i * 8)), fa);
    unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xb + i * 8)), fb);
    const unsigned mk = mask[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = ((mk >> e) & 1u) ? g[e] : 0.f;
    if (!fixed) {
      const unsigned m = fd_div((unsigned)i, fd_vcols);
      const int vc = (int)((unsigned)i - m * (unsigned)vcols);
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) k[q][e] = co[(size_t)q * C + vc * 8 + e];
    }
    float oa[8], ob[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      oa[e] = k[0][e] * g[e] + k[1][e] * fa[e] + k[2][e];
      ob[e] = k[3][e] * g[e] + k[4][e] * fb[e] + k[5][e];
    }
    BN_ST_BWD(pack8(oa), reinterpret_cast<u32x4*>(dxa + i * 8));
    BN_ST_BWD(pack8(ob), reinterpret_cast<u32x4*>(dxb + i * 8));
  }
}

// ---- forward twin of the pair above: out = [relu](bn_a(xa) + bf16(bn_b(xb))) in one pass -----------------------------
// The projection-shortcut batch norm has no consumer but this add, so its normalised tensor need not exist in HBM: it is
// evaluated on the fly (and rounded to bf16 exactly where the separate pass stored it: results are bit-identical).
// Saves the 4 B / element the shortcut's own apply pass moved.  co = [4][C]: scale_a, shift_a, scale_b, shift_b.
template <bool RELU>
__global__ __launch_bounds__(256) void bn_apply2_kernel(const bf16_t* __restrict__ xa, const bf16_t* __restrict__ xb,
                                                        bf16_t* __restrict__ y, size_t nvec, int C, FastDiv fd_vcols,
                                                        const float* __restrict__ sa, const float* __restrict__ ha,
                                                        const float* __restrict__ sb, const float* __restrict__ hb,
                                                        uint8_t* __restrict__ mask) {
  const int vcols = C >> 3;
  const size_t stride = (size_t)gridDim.x * 256;
  const bool fixed = (stride % (size_t)vcols) == 0;   // see bn_apply_kernel
  float k[4][8];
  auto load_coef = [&](int vc) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      k[0][e] = sa[vc * 8 + e];
      k[1][e] = ha[vc * 8 + e];
      k[2][e] = sb[vc * 8 + e];
      k[3][e] = hb[vc * 8 + e];
    }
  };
  if (fixed) load_coef((int)(((size_t)blockIdx.x * 256 + threadIdx.x) % (size_t)vcols));
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    float fa[8], fb[8];
    unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xa + i * 8)), fa);
    unpack8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xb + i * 8)), fb);
    if (!fixed) {
      const unsigned m = fd_div((unsigned)i, fd_vcols);
      load_coef((int)((unsigned)i - m * (unsigned)vcols));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) fb[e] = fb[e] * k[2][e] + k[3][e];
    float rb[8];
    unpack8(pack8(fb), rb);                 // the bf16 rounding of the materialised shortcut tensor
    unsigned mk = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      fa[e] = fa[e] * k[0][e] + k[1][e] + rb[e];
      if (RELU) {
        mk |= (fa[e] > 0.f ? 1u : 0u) << e;
        fa[e] = fmaxf(fa[e], 0.f);
      }
    }
    if (RELU && mask) mask[i] = (uint8_t)mk;
    BN_ST_FWD(pack8(fa), reinterpret_cast<u32x4*>(y + i * 8));
  }
}

// ---- small-tensor batch norm: statistics + finalize + apply in ONE launch (and reduce + finalize + apply backward) ----
// The SK / SE squeeze layers normalise [N, 1, 1, d] tensors (256 x 32..256 elements): three to four ~5 us launches
// of pure latency per direction in the general path.  Here one workgroup owns 64 channels x all M rows (M <= 4096):
// 256 threads = 8 vector columns x 32 row lanes, two passes over data that stays in L1 / L2.
constexpr int SMALL_BN_MAX_ROWS = 4096;

template <bool RELU>
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int M,
                                                           int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float momentum,
                                                           float* moving_mean, float* moving_var, float* mean,
                                                           float* invstd, uint8_t* __restrict__ mask) {
  // [stat][channel][row lane], rows padded to 33 floats: a wave's 64 lanes (8 vector columns x 8 row lanes) write 32 banks
  // two-way (channel-major rows of 64 floats put them on 4 banks: every store was a 16-way conflict)
  __shared__ float red[2][64][33];
  __shared__ float coef[2][64];
  const int vcols = C >> 3;
  const int vcl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int vc = blockIdx.x * 8 + vcl;
  const bool live = vc < vcols;
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
  // M <= 256 rows (the workload: one row per image): a lane's 8 rows are loaded in ONE batch and stay in registers for the
  // apply pass.  (Round 4.  The plain loops below issue a load, wait, accumulate -- 8 + 8 dependent round trips to L2 for a
  // kernel that moves 64 KB: 9 us.)  Same accumulation order, same bits.
  constexpr int KU = 8;
  const bool keep = M <= 32 * KU;
  u32x4 kv[KU];
  if (live && keep) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int r = rl + 32 * u;
      kv[u] = *reinterpret_cast<const u32x4*>(x + (size_t)(r < M ? r : 0) * C + vc * 8);   // (row 0 always exists)
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (rl + 32 * u >= M) break;
      float f[8];
      unpack8(kv[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s[e] += f[e];
        ss[e] += f[e] * f[e];
      }
    }
  } else if (live)
    for (int r = rl; r < M; r += 32) {
      float f[8];
      unpack8(*reinterpret_cast<const u32x4*>(x + (size_t)r * C + vc * 8), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s[e] += f[e];
        ss[e] += f[e] * f[e];
      }
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][vcl * 8 + e][rl] = s[e];
    red[1][vcl * 8 + e][rl] = ss[e];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch < C) {
      double a = 0.0, b = 0.0;
      for (int r = 0; r < 32; ++r) {
        a += (double)red[0][threadIdx.x][r];
        b += (double)red[1][threadIdx.x][r];
      }
      const double mu = a / (double)M;
      double var = b / (double)M - mu * mu;
      if (var < 0.0) var = 0.0;
      const float is = (float)(1.0 / sqrt(var + (double)eps));
      const float sc = gamma[ch] * is;
      mean[ch] = (float)mu;
      invstd[ch] = is;
      coef[0][threadIdx.x] = sc;
      coef[1][threadIdx.x] = beta[ch] - (float)mu * sc;
      if (moving_mean) {
        const double unbiased = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
        moving_mean[ch] = moving_mean[ch] * momentum + (float)mu * (1.f - momentum);
        moving_var[ch] = moving_var[ch] * momentum + (float)unbiased * (1.f - momentum);
      }
    }
  }
  __syncthreads();
  auto apply_row = [&](int r, const u32x4& vx) {
    const size_t i = (size_t)r * vcols + vc;
    float f[8];
    unpack8(vx, f);
    unsigned mk = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      f[e] = f[e] * coef[0][vcl * 8 + e] + coef[1][vcl * 8 + e];
      if (RELU) {
        mk |= (f[e] > 0.f ? 1u : 0u) << e;
        f[e] = fmaxf(f[e], 0.f);
      }
    }
    if (RELU && mask) mask[i] = (uint8_t)mk;
    *reinterpret_cast<u32x4*>(y + i * 8) = pack8(f);
  };
  if (live && keep) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (rl + 32 * u >= M) break;
      apply_row(rl + 32 * u, kv[u]);
    }
  } else if (live) {
    for (int r = rl; r < M; r += 32) apply_row(r, *reinterpret_cast<const u32x4*>(x + ((size_t)r * vcols + vc) * 8));
  }
}

template <int RELU>   // 0 none, 2 packed bitmask
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                           const uint8_t* __restrict__ mask, int M, int C,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, float* dgamma,
                                                           float* dbeta, bf16_t* __restrict__ dx) {
  __shared__ float red[2][64][33];   // [stat][channel][row lane] (see bn_small_fwd_kernel)
  __shared__ float coef[3][64];
  const int vcols = C >> 3;
  const int vcl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int vc = blockIdx.x * 8 + vcl;
  const bool live = vc < vcols;
  float s[8], ss[8], mu[8], is[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s[e] = ss[e] = 0.f;
    mu[e] = live ? mean[vc * 8 + e] : 0.f;
    is[e] = live ? invstd[vc * 8 + e] : 0.f;
  }
  // as in bn_small_fwd_kernel: M <= 256 -> one batch of loads, rows kept in registers for the apply pass
  constexpr int KU = 8;
  const bool keep = M <= 32 * KU;
  u32x4 kg[KU], kx[KU];
  unsigned km[KU];
  auto masked = [&](const u32x4& vg, unsigned mk, float* g) {
    unpack8(vg, g);
    if (RELU == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = ((mk >> e) & 1u) ? g[e] : 0.f;
    }
  };
  auto reduce_row = [&](const u32x4& vg, const u32x4& vx, unsigned mk) {
    float g[8], fx[8];
    masked(vg, mk, g);
    unpack8(vx, fx);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[e] += g[e];
      ss[e] += g[e] * ((fx[e] - mu[e]) * is[e]);
    }
  };
  if (live && keep) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int r = rl + 32 * u;
      const size_t i = (size_t)(r < M ? r : 0) * vcols + vc;
      kg[u] = *reinterpret_cast<const u32x4*>(dy + i * 8);
      kx[u] = *reinterpret_cast<const u32x4*>(x + i * 8);
      km[u] = RELU == 2 ? mask[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (rl + 32 * u >= M) break;
      reduce_row(kg[u], kx[u], km[u]);
    }
  } else if (live) {
    for (int r = rl; r < M; r += 32) {
      const size_t i = (size_t)r * vcols + vc;
      reduce_row(*reinterpret_cast<const u32x4*>(dy + i * 8), *reinterpret_cast<const u32x4*>(x + i * 8),
                 RELU == 2 ? mask[i] : 0u);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][vcl * 8 + e][rl] = s[e];
    red[1][vcl * 8 + e][rl] = ss[e];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch < C) {
      double db = 0.0, dg = 0.0;
      for (int r = 0; r < 32; ++r) {
        db += (double)red[0][threadIdx.x][r];
        dg += (double)red[1][threadIdx.x][r];
      }
      dbeta[ch] = (float)db;
      dgamma[ch] = (float)dg;
      const double g = gamma[ch], isd = invstd[ch], m = mean[ch];
      const double A = g * isd;
      const double B = -g * isd * isd * dg / (double)M;
      coef[0][threadIdx.x] = (float)A;
      coef[1][threadIdx.x] = (float)B;
      coef[2][threadIdx.x] = (float)(-g * isd * db / (double)M - B * m);
    }
  }
  __syncthreads();
  auto apply_row = [&](int r, const u32x4& vg, const u32x4& vx, unsigned mk) {
    float g[8], fx[8], o[8];
    masked(vg, mk, g);
    unpack8(vx, fx);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = coef[0][vcl * 8 + e] * g[e] + coef[1][vcl * 8 + e] * fx[e] + coef[2][vcl * 8 + e];
    *reinterpret_cast<u32x4*>(dx + ((size_t)r * vcols + vc) * 8) = pack8(o);
  };
  if (live && keep) {
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (rl + 32 * u >= M) break;
      apply_row(rl + 32 * u, kg[u], kx[u], km[u]);
    }
  } else if (live) {
    for (int r = rl; r < M; r += 32) {
      const size_t i = (size_t)r * vcols + vc;
      apply_row(r, *reinterpret_cast<const u32x4*>(dy + i * 8), *reinterpret_cast<const u32x4*>(x + i * 8),
                RELU == 2 ? mask[i] : 0u);
    }
  }
}

inline unsigned ew_grid(size_t nvec) {
  size_t b = cdivz(nvec, 256);
  return (unsigned)(b < 4096 ? (b ? b : 1) : 4096);
}

}  // namespace

extern "C" int asm_bn_stats_blocks(int M, int C) {
  if (M <= 0 || C <= 0 || C % 8) return ASM_EINVAL;
  return make_tiling(M, C).blocks;
}

extern "C" int asm_bn_stats(const void* x, int M, int C, float* stats_partial, void* stream) {
  ASM_REQUIRE(x && stats_partial && M > 0 && C > 0 && C % 8 == 0, "bn_stats: bad arguments (M=%d C=%d)", M, C);
  RowTiling t = make_tiling(M, C);
  ASM_LAUNCH((rowreduce_kernel<0>), dim3(t.blocks * t.slices), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, nullptr, nullptr, 0, M, C, nullptr, nullptr, t, stats_partial);
  ASM_CHECK_LAUNCH("bn_stats");
  return ASM_OK;
}

extern "C" int asm_bn_partials_compact(const float* partial, int blocks, int C, float* out, int groups,
                                       void* stream) {
  ASM_REQUIRE(partial && out && blocks > 0 && C > 0 && groups > 0 && groups <= blocks, "bn_partials_compact: bad arguments");
  const int per_group = cdiv(blocks, groups);
  ASM_REQUIRE(cdiv(blocks, per_group) == groups, "bn_partials_compact: groups=%d does not tile blocks=%d", groups, blocks);
  ASM_LAUNCH(partials_compact_kernel, dim3(cdiv(C, 16), groups), dim3(16 * FL), 0, (hipStream_t)stream, partial,
                     blocks, C, out, per_group);
  ASM_CHECK_LAUNCH("bn_partials_compact");
  return ASM_OK;
}

extern "C" int asm_bn_finalize(const float* stats_partial, int blocks, int M, int C, const float* gamma,
                               const float* beta, float eps, float momentum, float* moving_mean,
                               float* moving_var, float* mean, float* invstd, float* scale, float* shift,
                               void* stream) {
  ASM_REQUIRE(stats_partial && gamma && beta && mean && invstd && scale && shift && blocks > 0 && M > 0 && C > 0,
              "bn_finalize: bad arguments");
  ASM_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "bn_finalize: moving stats must both be given");
  ASM_LAUNCH(bn_finalize_kernel, dim3(cdiv(C, 16)), dim3(16 * FL), 0, (hipStream_t)stream, stats_partial,
                     blocks, M, C, gamma, beta, eps, momentum, moving_mean, moving_var, mean, invstd, scale, shift);
  ASM_CHECK_LAUNCH("bn_finalize");
  return ASM_OK;
}

extern "C" int asm_bn_infer_coeffs(int C, const float* gamma, const float* beta, const float* moving_mean,
                                   const float* moving_var, float eps, float* scale, float* shift, void* stream) {
  ASM_REQUIRE(C > 0 && gamma && beta && moving_mean && moving_var && scale && shift, "bn_infer_coeffs: bad arguments");
  ASM_LAUNCH(bn_infer_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, C, gamma, beta,
                     moving_mean, moving_var, eps, scale, shift);
  ASM_CHECK_LAUNCH("bn_infer_coeffs");
  return ASM_OK;
}

extern "C" int asm_bn_apply(const void* x, void* y, int M, int C, const float* scale, const float* shift,
                            const void* residual, int res_mode, int relu, int H, int W, uint8_t* relu_mask_out,
                            void* stream) {
  ASM_REQUIRE(x && y && scale && shift && M > 0 && C > 0 && C % 8 == 0, "bn_apply: bad arguments");
  ASM_REQUIRE(res_mode >= 0 && res_mode <= 2 && (res_mode == 0 || residual), "bn_apply: bad residual mode");
  ASM_REQUIRE((size_t)M * (C / 8) < 0x7fffffffull, "bn_apply: tensor too large");
  if (res_mode == 2)
    ASM_REQUIRE(H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && M % (H * W) == 0, "bn_apply: bad upsample geometry");
  const size_t nvec = (size_t)M * (C / 8);
  const FastDiv fv = make_fastdiv((unsigned)(C / 8));
  const FastDiv fw = make_fastdiv((unsigned)(W > 0 ? W : 1)), fh = make_fastdiv((unsigned)(H > 0 ? H : 1));
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(ew_grid(nvec)), block(256);
#define LAUNCH_APPLY(RES, RELU)                                                                          \
  ASM_LAUNCH((bn_apply_kernel<RES, RELU>), grid, block, 0, st, (const bf16_t*)x, (bf16_t*)y, nvec, C, fv, \
                     scale, shift, (const bf16_t*)residual, fw, fh, H, W, relu_mask_out)
  if (res_mode == 0) { if (relu) LAUNCH_APPLY(0, true); else LAUNCH_APPLY(0, false); }
  else if (res_mode == 1) { if (relu) LAUNCH_APPLY(1, true); else LAUNCH_APPLY(1, false); }
  else { if (relu) LAUNCH_APPLY(2, true); else LAUNCH_APPLY(2, false); }
#undef LAUNCH_APPLY
  ASM_CHECK_LAUNCH("bn_apply");
  return ASM_OK;
}

extern "C" int asm_bn_bwd_reduce(const void* dy, const void* x, const void* yout, int relu, int M, int C,
                                 const float* mean, const float* invstd, float* partial, void* stream) {
  ASM_REQUIRE(dy && x && mean && invstd && partial && M > 0 && C > 0 && C % 8 == 0, "bn_bwd_reduce: bad arguments");
  ASM_REQUIRE(relu >= 0 && relu <= 2 && (!relu || yout), "bn_bwd_reduce: relu mask needs the forward output / bitmask");
  RowTiling t = make_tiling(M, C);
  ASM_LAUNCH((rowreduce_kernel<1>), dim3(t.blocks * t.slices), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     (const bf16_t*)x, (const bf16_t*)yout, relu, M, C, mean, invstd, t, partial);
  ASM_CHECK_LAUNCH("bn_bwd_reduce");
  return ASM_OK;
}

extern "C" int asm_bn_bwd_finalize(const float* partial, int blocks, int M, int C, const float* gamma,
                                   const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                   float* coefA, float* coefB, float* coefC, void* stream) {
  ASM_REQUIRE(partial && gamma && mean && invstd && dgamma && dbeta && coefA && coefB && coefC && blocks > 0,
              "bn_bwd_finalize: bad arguments");
  ASM_LAUNCH(bn_bwd_finalize_kernel, dim3(cdiv(C, 16)), dim3(16 * FL), 0, (hipStream_t)stream, partial, blocks,
                     M, C, gamma, mean, invstd, dgamma, dbeta, coefA, coefB, coefC, 0);
  ASM_CHECK_LAUNCH("bn_bwd_finalize");
  return ASM_OK;
}

extern "C" int asm_bn_bwd_finalize_raw(const float* partial, int blocks, int M, int C, const float* gamma,
                                       const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                       float* coefA, float* coefB, float* coefC, void* stream) {
  ASM_REQUIRE(partial && gamma && mean && invstd && dgamma && dbeta && coefA && coefB && coefC && blocks > 0,
              "bn_bwd_finalize_raw: bad arguments");
  ASM_LAUNCH(bn_bwd_finalize_kernel, dim3(cdiv(C, 16)), dim3(16 * FL), 0, (hipStream_t)stream, partial, blocks,
                     M, C, gamma, mean, invstd, dgamma, dbeta, coefA, coefB, coefC, 1);
  ASM_CHECK_LAUNCH("bn_bwd_finalize_raw");
  return ASM_OK;
}

extern "C" int asm_bn_bwd_apply(const void* dy, const void* x, const void* yout, int relu, int M, int C,
                                const float* coefA, const float* coefB, const float* coefC, void* dx,
                                void* dz_out, void* stream) {
  ASM_REQUIRE(dy && x && dx && coefA && coefB && coefC && M > 0 && C > 0 && C % 8 == 0, "bn_bwd_apply: bad arguments");
  ASM_REQUIRE(relu >= 0 && relu <= 2 && (!relu || yout), "bn_bwd_apply: relu mask needs the forward output / bitmask");
  ASM_REQUIRE((size_t)M * (C / 8) < 0x7fffffffull, "bn_bwd_apply: tensor too large");
  const size_t nvec = (size_t)M * (C / 8);
  const FastDiv fv = make_fastdiv((unsigned)(C / 8));
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(ew_grid(nvec)), block(256);
#define LAUNCH_BWD(RELU, DZ)                                                                              \
  ASM_LAUNCH((bn_bwd_apply_kernel<RELU, DZ>), grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)x, \
                     (const bf16_t*)yout, nvec, C, fv, coefA, coefB, coefC, (bf16_t*)dx, (bf16_t*)dz_out)
  if (relu == 1) { if (dz_out) LAUNCH_BWD(1, true); else LAUNCH_BWD(1, false); }
  else if (relu == 2) { if (dz_out) LAUNCH_BWD(2, true); else LAUNCH_BWD(2, false); }
  else { if (dz_out) LAUNCH_BWD(0, true); else LAUNCH_BWD(0, false); }
#undef LAUNCH_BWD
  ASM_CHECK_LAUNCH("bn_bwd_apply");
  return ASM_OK;
}

extern "C" int asm_bn_small_max_rows(void) { return SMALL_BN_MAX_ROWS; }

extern "C" int asm_bn_small_fwd(const void* x, void* y, int M, int C, const float* gamma, const float* beta, float eps,
                                float momentum, float* moving_mean, float* moving_var, float* mean, float* invstd,
                                int relu, uint8_t* relu_mask_out, void* stream) {
  ASM_REQUIRE(x && y && gamma && beta && mean && invstd && M > 0 && M <= SMALL_BN_MAX_ROWS && C > 0 && C % 8 == 0,
              "bn_small_fwd: bad arguments (M=%d C=%d)", M, C);
  ASM_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "bn_small_fwd: moving stats must both be given");
  const dim3 grid(cdiv(C, 64)), block(256);
  if (relu)
    ASM_LAUNCH(bn_small_fwd_kernel<true>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, M, C,
                       gamma, beta, eps, momentum, moving_mean, moving_var, mean, invstd, relu_mask_out);
  else
    ASM_LAUNCH(bn_small_fwd_kernel<false>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, M, C,
                       gamma, beta, eps, momentum, moving_mean, moving_var, mean, invstd, nullptr);
  ASM_CHECK_LAUNCH("bn_small_fwd");
  return ASM_OK;
}

extern "C" int asm_bn_small_bwd(const void* dy, const void* x, const uint8_t* relu_mask, int M, int C,
                                const float* gamma, const float* mean, const float* invstd, float* dgamma,
                                float* dbeta, void* dx, void* stream) {
  ASM_REQUIRE(dy && x && gamma && mean && invstd && dgamma && dbeta && dx && M > 0 && M <= SMALL_BN_MAX_ROWS && C > 0 &&
                  C % 8 == 0, "bn_small_bwd: bad arguments (M=%d C=%d)", M, C);
  const dim3 grid(cdiv(C, 64)), block(256);
  if (relu_mask)
    ASM_LAUNCH(bn_small_bwd_kernel<2>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x,
                       relu_mask, M, C, gamma, mean, invstd, dgamma, dbeta, (bf16_t*)dx);
  else
    ASM_LAUNCH(bn_small_bwd_kernel<0>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x,
                       nullptr, M, C, gamma, mean, invstd, dgamma, dbeta, (bf16_t*)dx);
  ASM_CHECK_LAUNCH("bn_small_bwd");
  return ASM_OK;
}

extern "C" int asm_bn_bwd_reduce2(const void* dy, const void* xa, const void* xb, const uint8_t* relu_mask, int M, int C,
                                  const float* mean_a, const float* invstd_a, const float* mean_b, const float* invstd_b,
                                  float* partial_a, float* partial_b, void* stream) {
  ASM_REQUIRE(dy && xa && xb && relu_mask && mean_a && invstd_a && mean_b && invstd_b && partial_a && partial_b && M > 0 &&
                  C > 0 && C % 8 == 0, "bn_bwd_reduce2: bad arguments");
  RowTiling t = make_tiling(M, C);
  ASM_LAUNCH(rowreduce2_kernel, dim3(t.blocks * t.slices), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     (const bf16_t*)xa, (const bf16_t*)xb, relu_mask, M, C, mean_a, invstd_a, mean_b, invstd_b, t, partial_a,
                     partial_b);
  ASM_CHECK_LAUNCH("bn_bwd_reduce2");
  return ASM_OK;
}

extern "C" int asm_bn_bwd_apply2(const void* dy, const void* xa, const void* xb, const uint8_t* relu_mask, int M, int C,
                                 const float* coef6, void* dxa, void* dxb, void* stream) {
  ASM_REQUIRE(dy && xa && xb && relu_mask && coef6 && dxa && dxb && M > 0 && C > 0 && C % 8 == 0, "bn_bwd_apply2: bad arguments");
  ASM_REQUIRE((size_t)M * (C / 8) < 0x7fffffffull, "bn_bwd_apply2: tensor too large");
  const size_t nvec = (size_t)M * (C / 8);
  ASM_LAUNCH(bn_bwd_apply2_kernel, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     (const bf16_t*)xa, (const bf16_t*)xb, relu_mask, nvec, C, make_fastdiv((unsigned)(C / 8)), coef6,
                     (bf16_t*)dxa, (bf16_t*)dxb);
  ASM_CHECK_LAUNCH("bn_bwd_apply2");
  return ASM_OK;
}

extern "C" int asm_bn_apply2(const void* xa, const void* xb, void* y, int M, int C, const float* scale_a,
                             const float* shift_a, const float* scale_b, const float* shift_b, int relu,
                             uint8_t* relu_mask_out, void* stream) {
  ASM_REQUIRE(xa && xb && y && scale_a && shift_a && scale_b && shift_b && M > 0 && C > 0 && C % 8 == 0, "bn_apply2: bad arguments");
  ASM_REQUIRE((size_t)M * (C / 8) < 0x7fffffffull, "bn_apply2: tensor too large");
  const size_t nvec = (size_t)M * (C / 8);
  const FastDiv fv = make_fastdiv((unsigned)(C / 8));
  const dim3 grid(ew_grid(nvec)), block(256);
  if (relu)
    ASM_LAUNCH(bn_apply2_kernel<true>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)xa, (const bf16_t*)xb,
                       (bf16_t*)y, nvec, C, fv, scale_a, shift_a, scale_b, shift_b, relu_mask_out);
  else
    ASM_LAUNCH(bn_apply2_kernel<false>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)xa, (const bf16_t*)xb,
                       (bf16_t*)y, nvec, C, fv, scale_a, shift_a, scale_b, shift_b, nullptr);
  ASM_CHECK_LAUNCH("bn_apply2");
  return ASM_OK;
}
