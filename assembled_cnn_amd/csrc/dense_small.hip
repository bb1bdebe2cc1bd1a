// Small dense layers for gfx950: the [N,1,1,C] squeeze / excite / classifier layers (sk_conv2d's sk_fc_1 / sk_fc_2,
// nets/blocks.py:136-148; se_block's two dense layers, :165-178; the final tf.layers.dense, nets/resnet_model.py:595-597)
// and their input gradients.  M = batch rows (256 at the benchmark), reductions of 32..2048.
//
// The implicit-GEMM convolution launches ONE or TWO 128-row workgroups for such a layer and walks K serially with a
// barrier per 32 channels: 8-16 us of pure latency per launch, ~110 launches per training step of Assemble-ResNet-50.
// Here the operands are so small (<= 1 MiB, L2-resident) that no LDS staging is needed at all: both MFMA operands have
// their reduction index contiguous in memory, so a lane's fragment (8 consecutive k of one row) is one 16-byte global
// load.  A workgroup owns a 32 x 32 output tile and splits the reduction over its 4 waves (dense_small_kernel), or owns
// ALL rows of 32 output channels so that the training-mode batch norm that follows / precedes the layer runs in the same
// launch (dense_bn_fwd_kernel: fc + BN statistics + finalize + apply + ReLU; dense_dgrad_bn_bwd_kernel: input gradient
// of the NEXT layer + ReLU mask + BN backward reduce + finalize + apply).
//
// Fragment convention (as csrc/conv_igemm.hip): acc = mfma_32x32x16(A = q rows, B = p rows): lane l holds row (l & 31),
// reduction elements (l >> 5) * 8 .. +8; acc[r] = out[m = l & 31][n = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)].
#include "common.h"

namespace {

constexpr int DENSE_BN_MAX_ROWS = 256;   // 8 waves x one 32-row tile

__device__ __forceinline__ bf16x8 ldg8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

// Reduction steps (16 elements each) s, s + STRIDE, ... are consumed in batches of B with TWO batches of loads in flight:
// the operands come from L2 (~200 cycles), an MFMA takes 32, so the loop is a latency chain unless many loads overlap.
// Steps past the end are clamped for the load (a valid address) and skipped for the multiply (wave-uniform).
template <int B, int STRIDE>
__device__ __forceinline__ void load_batch(const bf16_t* qp, const bf16_t* pp, int s, int steps, bf16x8 (&q)[B], bf16x8 (&p)[B]) {
#pragma unroll
  for (int j = 0; j < B; ++j) {
    const int t = min(s + j * STRIDE, steps - 1);
    q[j] = ldg8(qp + t * 16);
    p[j] = ldg8(pp + t * 16);
  }
}
// even / odd steps go to two accumulators: back-to-back MFMAs on ONE accumulator wait out the 16-pass result latency
template <int B, int STRIDE>
__device__ __forceinline__ void mma_batch(int s, int steps, const bf16x8 (&q)[B], const bf16x8 (&p)[B], f32x16& acc,
                                          f32x16& acc2) {
#pragma unroll
  for (int j = 0; j < B; ++j)
    if (s + j * STRIDE < steps) {
      if (j & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q[j], p[j], acc2, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q[j], p[j], acc, 0, 0, 0);
    }
}
template <int B, int STRIDE>
__device__ __forceinline__ void gemm_steps(const bf16_t* qp, const bf16_t* pp, int s0, int steps, f32x16& acc) {
  bf16x8 qa[B], pa[B], qb[B], pb[B];
  // MFMA ignores EXEC: every branch around one must be a SCALAR branch.  The step cursor derives from the wave index
  // (a VGPR as far as the compiler knows), so pin it to SGPRs -- with a per-lane condition the compiler may predicate
  // the block by EXEC and drop the skip branch, and the "skipped" MFMA still executes.
  s0 = __builtin_amdgcn_readfirstlane(s0);
  steps = __builtin_amdgcn_readfirstlane(steps);
  if (s0 >= steps) return;
  f32x16 acc2;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
  load_batch<B, STRIDE>(qp, pp, s0, steps, qa, pa);
  for (int s = s0; s < steps; s += 2 * B * STRIDE) {
    load_batch<B, STRIDE>(qp, pp, s + B * STRIDE, steps, qb, pb);
    mma_batch<B, STRIDE>(s, steps, qa, pa, acc, acc2);
    load_batch<B, STRIDE>(qp, pp, s + 2 * B * STRIDE, steps, qa, pa);
    mma_batch<B, STRIDE>(s + B * STRIDE, steps, qb, pb, acc, acc2);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
}

struct DenseArgs {
  const bf16_t* p;       // [M][ldp]  rows m, reduction contiguous
  const bf16_t* q;       // [N][ldq]  rows n, reduction contiguous
  void* out;             // [M][ldo]  f32 or bf16
  const bf16_t* addend;  // optional bf16 [M][ldo]
  int ldp, ldq, ldo, M, N, K, out_f32;
};

// out[m][n] = sum_k p[m][k] * q[n][k] (+ addend).  grid (ceil(N/32), ceil(M/32)), 256 threads: wave w takes the
// reduction steps w, w+4, ... (16 elements each); the four partial tiles meet in LDS.
__global__ __launch_bounds__(256) void dense_small_kernel(DenseArgs a) {
  __shared__ float red[4][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int nrow = min(n0 + l31, a.N - 1), mrow = min(m0 + l31, a.M - 1);   // tail rows re-read a valid row, never stored
  const bf16_t* qp = a.q + (size_t)nrow * a.ldq + lhi * 8;
  const bf16_t* pp = a.p + (size_t)mrow * a.ldp + lhi * 8;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  gemm_steps<4, 4>(qp, pp, wave, a.K >> 4, acc);   // wave w: steps w, w + 4, ...
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][l31][(r & 3) + 8 * (r >> 2) + 4 * lhi] = acc[r];
  __syncthreads();
  const int m = tid >> 3, nq = (tid & 7) * 4;
  const int gm = m0 + m, gn = n0 + nq;
  if (gm >= a.M || gn >= a.ldo) return;     // pad columns N .. ldo-1 of a padded output row are written as zeros
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    v[j] = gn + j < a.N ? (red[0][m][nq + j] + red[1][m][nq + j]) + (red[2][m][nq + j] + red[3][m][nq + j]) : 0.f;
  const size_t o = (size_t)gm * a.ldo + gn;
  const bool full = gn + 3 < a.N;
  if (a.addend) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (gn + j < a.N) v[j] += bf2f(a.addend[o + j]);
  }
  if (a.out_f32) {
    float* out = reinterpret_cast<float*>(a.out);
    if (full) {
      f32x4 w = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(out + o) = w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (gn + j < a.ldo) out[o + j] = v[j];
    }
  } else {
    bf16_t* out = reinterpret_cast<bf16_t*>(a.out);
    if (full) {
      u32x2 w;
      w.x = pack2bf(v[0], v[1]);
      w.y = pack2bf(v[2], v[3]);
      *reinterpret_cast<u32x2*>(out + o) = w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (gn + j < a.ldo) out[o + j] = f2bf(v[j]);
    }
  }
}

// ---- the all-rows forms: one workgroup = every row (M <= 256) of 32 output channels ----------------------------------
// 512 threads = 8 waves, wave w owns rows 32w .. 32w+31 over the whole reduction; the bf16-rounded tile lands in LDS
// ([256][32] + 8 pad columns: 16-byte vector reads at any row), then 128 row lanes x 4 vector columns run the batch norm
// exactly as bn_small_fwd_kernel / bn_small_bwd_kernel do on a tensor in HBM.
constexpr int YS_LD = 40;   // bf16 elements per LDS row (80 bytes)

__device__ __forceinline__ void tile_gemm_rows(const bf16_t* __restrict__ p, int ldp, const bf16_t* __restrict__ q, int ldq,
                                               int M, int N, int K, int n0, int wave, int l31, int lhi, f32x16& acc) {
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int m0 = wave * 32;
  if (m0 >= M) return;   // wave-uniform
  const int nrow = min(n0 + l31, N - 1), mrow = min(m0 + l31, M - 1);
  const bf16_t* qp = q + (size_t)nrow * ldq + lhi * 8;
  const bf16_t* pp = p + (size_t)mrow * ldp + lhi * 8;
  gemm_steps<8, 1>(qp, pp, 0, K >> 4, acc);
}

// acc -> bf16 -> ys[m][n_local]
__device__ __forceinline__ void tile_to_lds(const f32x16& acc, bf16_t (*ys)[YS_LD], int M, int wave, int l31, int lhi) {
  const int m = wave * 32 + l31;
  if (m < M) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x2 v;
      v.x = pack2bf(acc[4 * g], acc[4 * g + 1]);
      v.y = pack2bf(acc[4 * g + 2], acc[4 * g + 3]);
      *reinterpret_cast<u32x2*>(&ys[m][8 * g + 4 * lhi]) = v;
    }
  }
}

// per-channel sums over the 128 row lanes.  red is [stat][channel][row lane] with rows padded to 129 floats: the 64 lanes of
// a wave (4 vector columns x 16 row lanes) then write 64 different banks (channel-major rows of 32 floats put 16 lanes
// on each of 4 banks: every store was a 16-way conflict).  32 x 16 threads sum 8 lanes each into red2.
constexpr int RED_LD = 129;
__device__ __forceinline__ void reduce_lanes(float (*red)[32][RED_LD], float (*red2)[16][32], int tid) {
  const int col = tid & 31, grp = tid >> 5;   // 16 groups of 8 row lanes
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[which][col][grp * 8 + r];
    red2[which][grp][col] = t;
  }
}

struct DenseBnFwdArgs {
  const bf16_t* x; const bf16_t* w; int ldx, ldw, M, K, N;
  const float* gamma; const float* beta; float eps, momentum; float* moving_mean; float* moving_var;
  bf16_t* ypre; bf16_t* z; float* mean; float* invstd; uint8_t* mask;
};

template <bool RELU>
__global__ __launch_bounds__(512) void dense_bn_fwd_kernel(DenseBnFwdArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t ys[DENSE_BN_MAX_ROWS][YS_LD];
  __shared__ float red[2][32][RED_LD];
  __shared__ float red2[2][16][32];
  __shared__ float coef[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int n0 = blockIdx.x * 32;
  f32x16 acc;
  tile_gemm_rows(a.x, a.ldx, a.w, a.ldw, a.M, a.N, a.K, n0, wave, l31, lhi, acc);
  tile_to_lds(acc, ys, a.M, wave, l31, lhi);
  __syncthreads();
  const int vc = tid & 3, rl = tid >> 2;      // 4 vector columns x 128 row lanes
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
  for (int r = rl; r < a.M; r += 128) {
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(&ys[r][vc * 8]), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[e] += f[e];
      ss[e] += f[e] * f[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][vc * 8 + e][rl] = s[e];
    red[1][vc * 8 + e][rl] = ss[e];
  }
  __syncthreads();
  reduce_lanes(red, red2, tid);
  __syncthreads();
  if (tid < 32) {
    const int ch = n0 + tid;
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      sa += (double)red2[0][g][tid];
      sb += (double)red2[1][g][tid];
    }
    float sc = 0.f, sh = 0.f;
    if (ch < a.N) {
      const double mu = sa / (double)a.M;
      double var = sb / (double)a.M - mu * mu;
      if (var < 0.0) var = 0.0;
      const float is = (float)(1.0 / sqrt(var + (double)a.eps));
      sc = a.gamma[ch] * is;
      sh = a.beta[ch] - (float)mu * sc;
      a.mean[ch] = (float)mu;
      a.invstd[ch] = is;
      if (a.moving_mean) {
        const double unbiased = var * ((double)a.M / (double)(a.M > 1 ? a.M - 1 : 1));
        a.moving_mean[ch] = a.moving_mean[ch] * a.momentum + (float)mu * (1.f - a.momentum);
        a.moving_var[ch] = a.moving_var[ch] * a.momentum + (float)unbiased * (1.f - a.momentum);
      }
    }
    coef[0][tid] = sc;
    coef[1][tid] = sh;
  }
  __syncthreads();
  const int c0 = n0 + vc * 8;
  if (c0 >= a.N) return;
  const int vcols = a.N >> 3;
  for (int r = rl; r < a.M; r += 128) {
    const u32x4 raw = *reinterpret_cast<const u32x4*>(&ys[r][vc * 8]);
    float f[8];
    unpack8(raw, f);
    unsigned mk = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      f[e] = f[e] * coef[0][vc * 8 + e] + coef[1][vc * 8 + e];
      if (RELU) {
        mk |= (f[e] > 0.f ? 1u : 0u) << e;
        f[e] = fmaxf(f[e], 0.f);
      }
    }
    const size_t o = (size_t)r * a.N + c0;
    *reinterpret_cast<u32x4*>(a.ypre + o) = raw;
    *reinterpret_cast<u32x4*>(a.z + o) = pack8(f);
    if (RELU && a.mask) a.mask[(size_t)r * vcols + (c0 >> 3)] = (uint8_t)mk;
  }
}

struct DenseBnBwdArgs {
  const bf16_t* dy; const bf16_t* wt; int lddy, ldwt, M, K, N;     // g[M][N] = dy[M][K] . wt[N][K]^T
  const bf16_t* ypre; const uint8_t* mask; const float* gamma; const float* mean; const float* invstd;
  float* dgamma; float* dbeta; bf16_t* dx;
};

template <bool RELU>
__global__ __launch_bounds__(512) void dense_dgrad_bn_bwd_kernel(DenseBnBwdArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t ys[DENSE_BN_MAX_ROWS][YS_LD];
  __shared__ float red[2][32][RED_LD];
  __shared__ float red2[2][16][32];
  __shared__ float coef[3][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int n0 = blockIdx.x * 32;
  f32x16 acc;
  tile_gemm_rows(a.dy, a.lddy, a.wt, a.ldwt, a.M, a.N, a.K, n0, wave, l31, lhi, acc);
  tile_to_lds(acc, ys, a.M, wave, l31, lhi);      // the gradient is rounded to bf16 like the separate input-gradient launch did
  __syncthreads();
  const int vc = tid & 3, rl = tid >> 2;
  const int c0 = n0 + vc * 8;
  const bool live = c0 < a.N;
  const int vcols = a.N >> 3;
  float s[8], ss[8], mu[8], is[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s[e] = ss[e] = 0.f;
    mu[e] = live ? a.mean[c0 + e] : 0.f;
    is[e] = live ? a.invstd[c0 + e] : 0.f;
  }
  if (live)
    for (int r = rl; r < a.M; r += 128) {
      float g[8], fx[8];
      unpack8(*reinterpret_cast<const u32x4*>(&ys[r][vc * 8]), g);
      unpack8(*reinterpret_cast<const u32x4*>(a.ypre + (size_t)r * a.N + c0), fx);
      if (RELU) {
        const unsigned mk = a.mask[(size_t)r * vcols + (c0 >> 3)];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = ((mk >> e) & 1u) ? g[e] : 0.f;
        *reinterpret_cast<u32x4*>(&ys[r][vc * 8]) = pack8(g);    // exact: masked lanes become 0, the rest are unchanged bf16 values
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s[e] += g[e];
        ss[e] += g[e] * ((fx[e] - mu[e]) * is[e]);
      }
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][vc * 8 + e][rl] = s[e];
    red[1][vc * 8 + e][rl] = ss[e];
  }
  __syncthreads();
  reduce_lanes(red, red2, tid);
  __syncthreads();
  if (tid < 32) {
    const int ch = n0 + tid;
    double db = 0.0, dg = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      db += (double)red2[0][g][tid];
      dg += (double)red2[1][g][tid];
    }
    float cA = 0.f, cB = 0.f, cC = 0.f;
    if (ch < a.N) {
      a.dbeta[ch] = (float)db;
      a.dgamma[ch] = (float)dg;
      const double gm = a.gamma[ch], isd = a.invstd[ch], m = a.mean[ch];
      const double A = gm * isd;
      const double B = -gm * isd * isd * dg / (double)a.M;
      cA = (float)A;
      cB = (float)B;
      cC = (float)(-gm * isd * db / (double)a.M - B * m);
    }
    coef[0][tid] = cA;
    coef[1][tid] = cB;
    coef[2][tid] = cC;
  }
  __syncthreads();
  if (!live) return;
  for (int r = rl; r < a.M; r += 128) {
    float g[8], fx[8], o[8];
    unpack8(*reinterpret_cast<const u32x4*>(&ys[r][vc * 8]), g);
    const size_t off = (size_t)r * a.N + c0;
    unpack8(*reinterpret_cast<const u32x4*>(a.ypre + off), fx);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = coef[0][vc * 8 + e] * g[e] + coef[1][vc * 8 + e] * fx[e] + coef[2][vc * 8 + e];
    *reinterpret_cast<u32x4*>(a.dx + off) = pack8(o);
  }
}

// ---- weight gradient: dw[n][k] = sum_m dy[m][n] * x[m][k] --------------------------------------------------------------
// Both operands are reduction-MAJOR here (rows m), so a lane's fragment (8 consecutive m of one column) is 8 two-byte
// loads, coalesced across the 32 lanes that hold consecutive columns; at M <= a few hundred rows that is ~64 loads per
// lane and launch.  grid (ceil(Cin/32), ceil(Cout/32)), 4 waves split the rows; fp32 out [Cout][ldw].
struct DenseWgradArgs {
  const bf16_t* x; const bf16_t* dy; float* dw; int ldx, ldy, ldw, M, Cin, Cout;
};

__device__ __forceinline__ bf16x8 gather8(const bf16_t* base, int ld, int r0, int M, bool col_ok) {
  unsigned short v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (col_ok && r0 + j < M) ? base[(size_t)(r0 + j) * ld] : (unsigned short)0;
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (short)v[j];
  return f;
}

__global__ __launch_bounds__(256) void dense_small_wgrad_kernel(DenseWgradArgs a) {
  __shared__ float red[4][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const bool k_ok = k0 + l31 < a.Cin, n_ok = n0 + l31 < a.Cout;
  const bf16_t* xp = a.x + (k_ok ? k0 + l31 : 0);
  const bf16_t* yp = a.dy + (n_ok ? n0 + l31 : 0);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int steps = (a.M + 15) >> 4;
  for (int s = wave; s < steps; s += 8) {          // two steps (32 loads) in flight
    const int r0 = s * 16 + lhi * 8, r1 = (s + 4) * 16 + lhi * 8;
    const bf16x8 fx0 = gather8(xp, a.ldx, r0, a.M, k_ok), fy0 = gather8(yp, a.ldy, r0, a.M, n_ok);
    const bf16x8 fx1 = gather8(xp, a.ldx, r1, a.M, k_ok), fy1 = gather8(yp, a.ldy, r1, a.M, n_ok);   // rows >= M read as 0
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx0, fy0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx1, fy1, acc, 0, 0, 0);
  }
  // acc[r] = dw[n = n0 + l31][k = k0 + (r & 3) + 8 * (r >> 2) + 4 * lhi]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][l31][(r & 3) + 8 * (r >> 2) + 4 * lhi] = acc[r];
  __syncthreads();
  const int n = tid >> 3, kq = (tid & 7) * 4;
  const int gn = n0 + n, gk = k0 + kq;
  if (gn >= a.Cout || gk >= a.Cin) return;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (red[0][n][kq + j] + red[1][n][kq + j]) + (red[2][n][kq + j] + red[3][n][kq + j]);
  float* o = a.dw + (size_t)gn * a.ldw + gk;
  if (gk + 3 < a.Cin) {
    f32x4 w = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(o) = w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (gk + j < a.Cin) o[j] = v[j];
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int asm_dense_bn_max_rows(void) { return DENSE_BN_MAX_ROWS; }

extern "C" int asm_dense_small(const void* p, int ldp, const void* q, int ldq, int M, int N, int K, void* out, int ldo,
                               int out_f32, const void* addend, void* stream) {
  ASM_REQUIRE(p && q && out && M > 0 && N > 0 && K > 0, "dense_small: bad arguments (M=%d N=%d K=%d)", M, N, K);
  ASM_REQUIRE(K % 16 == 0 && ldp % 8 == 0 && ldq % 8 == 0 && ldp >= K && ldq >= K,
              "dense_small: the reduction must be a multiple of 16 and rows 16-byte aligned (K=%d ldp=%d ldq=%d)", K, ldp, ldq);
  ASM_REQUIRE(ldo >= N && ldo % 4 == 0, "dense_small: bad output row stride %d", ldo);
  // `out` owns its whole rows: the pad columns N .. ldo-1 are WRITTEN (zeros) by the 32-column tile that holds column
  // N-1, so the padding must end inside that tile -- a wider row (a column slice of a larger matrix) is refused rather
  // than half-zeroed or clobbered
  ASM_REQUIRE(ldo <= cdiv(N, 32) * 32, "dense_small: row stride %d pads N=%d past its last 32-column tile (out must own whole rows "
              "with fewer than 32 pad columns)", ldo, N);
  ASM_REQUIRE(aligned16(p) && aligned16(q) && aligned16(out) && (!addend || aligned16(addend)), "dense_small: unaligned pointer");
  DenseArgs a;
  a.p = (const bf16_t*)p; a.q = (const bf16_t*)q; a.out = out; a.addend = (const bf16_t*)addend;
  a.ldp = ldp; a.ldq = ldq; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.out_f32 = out_f32 ? 1 : 0;
  ASM_LAUNCH(dense_small_kernel, dim3(cdiv(N, 32), cdiv(M, 32)), dim3(256), 0, (hipStream_t)stream, a);
  ASM_CHECK_LAUNCH("dense_small");
  return ASM_OK;
}

extern "C" int asm_dense_bn_fwd(const void* x, int ldx, const void* w, int ldw, int M, int K, int N, const float* gamma,
                                const float* beta, float eps, float momentum, float* moving_mean, float* moving_var,
                                void* ypre, void* z, float* mean, float* invstd, int relu, uint8_t* relu_mask_out,
                                void* stream) {
  ASM_REQUIRE(x && w && gamma && beta && ypre && z && mean && invstd, "dense_bn_fwd: null pointer");
  ASM_REQUIRE(M > 0 && M <= DENSE_BN_MAX_ROWS && N > 0 && N % 8 == 0 && K > 0 && K % 16 == 0,
              "dense_bn_fwd: bad shape (M=%d K=%d N=%d)", M, K, N);
  ASM_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K, "dense_bn_fwd: bad row strides");
  ASM_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "dense_bn_fwd: moving stats must both be given");
  ASM_REQUIRE(aligned16(x) && aligned16(w) && aligned16(ypre) && aligned16(z), "dense_bn_fwd: unaligned pointer");
  DenseBnFwdArgs a;
  a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.ldx = ldx; a.ldw = ldw; a.M = M; a.K = K; a.N = N;
  a.gamma = gamma; a.beta = beta; a.eps = eps; a.momentum = momentum; a.moving_mean = moving_mean; a.moving_var = moving_var;
  a.ypre = (bf16_t*)ypre; a.z = (bf16_t*)z; a.mean = mean; a.invstd = invstd; a.mask = relu_mask_out;
  const dim3 grid(cdiv(N, 32)), block(512);
  if (relu) ASM_LAUNCH(dense_bn_fwd_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
  else ASM_LAUNCH(dense_bn_fwd_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
  ASM_CHECK_LAUNCH("dense_bn_fwd");
  return ASM_OK;
}

extern "C" int asm_dense_dgrad_bn_bwd(const void* dy, int lddy, const void* wt, int ldwt, int M, int K, int N,
                                      const void* ypre, const uint8_t* relu_mask, const float* gamma, const float* mean,
                                      const float* invstd, float* dgamma, float* dbeta, void* dx, void* stream) {
  ASM_REQUIRE(dy && wt && ypre && gamma && mean && invstd && dgamma && dbeta && dx, "dense_dgrad_bn_bwd: null pointer");
  ASM_REQUIRE(M > 0 && M <= DENSE_BN_MAX_ROWS && N > 0 && N % 8 == 0 && K > 0 && K % 16 == 0,
              "dense_dgrad_bn_bwd: bad shape (M=%d K=%d N=%d)", M, K, N);
  ASM_REQUIRE(lddy % 8 == 0 && ldwt % 8 == 0 && lddy >= K && ldwt >= K, "dense_dgrad_bn_bwd: bad row strides");
  ASM_REQUIRE(aligned16(dy) && aligned16(wt) && aligned16(ypre) && aligned16(dx), "dense_dgrad_bn_bwd: unaligned pointer");
  DenseBnBwdArgs a;
  a.dy = (const bf16_t*)dy; a.wt = (const bf16_t*)wt; a.lddy = lddy; a.ldwt = ldwt; a.M = M; a.K = K; a.N = N;
  a.ypre = (const bf16_t*)ypre; a.mask = relu_mask; a.gamma = gamma; a.mean = mean; a.invstd = invstd;
  a.dgamma = dgamma; a.dbeta = dbeta; a.dx = (bf16_t*)dx;
  const dim3 grid(cdiv(N, 32)), block(512);
  if (relu_mask) ASM_LAUNCH(dense_dgrad_bn_bwd_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
  else ASM_LAUNCH(dense_dgrad_bn_bwd_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
  ASM_CHECK_LAUNCH("dense_dgrad_bn_bwd");
  return ASM_OK;
}

extern "C" int asm_dense_small_wgrad(const void* x, int ldx, const void* dy, int ldy, int M, int Cin, int Cout, float* dw,
                                     int ldw, void* stream) {
  ASM_REQUIRE(x && dy && dw && M > 0 && Cin > 0 && Cout > 0, "dense_small_wgrad: bad arguments");
  ASM_REQUIRE(ldx >= Cin && ldy >= Cout && ldw >= Cin && ldw % 4 == 0, "dense_small_wgrad: bad row strides");
  ASM_REQUIRE(aligned16(dw), "dense_small_wgrad: unaligned pointer");
  DenseWgradArgs a;
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.dw = dw; a.ldx = ldx; a.ldy = ldy; a.ldw = ldw;
  a.M = M; a.Cin = Cin; a.Cout = Cout;
  ASM_LAUNCH(dense_small_wgrad_kernel, dim3(cdiv(Cin, 32), cdiv(Cout, 32)), dim3(256), 0, (hipStream_t)stream, a);
  ASM_CHECK_LAUNCH("dense_small_wgrad");
  return ASM_OK;
}
