// Small dense layers for gfx950: the [N,1,1,C] squeeze / excite / classifier layers (sk_conv2d's sk_fc_1 / sk_fc_2,
// nets/blocks.py:136-148; se_block's two dense layers, :165-178; the final tf.layers.dense, nets/resnet_model.py:595-597)
// and their input gradients.  M = batch rows (256 at the benchmark), reductions of 32..2048.
//
// The implicit-GEMM convolution launches ONE or TWO 128-row workgroups for such a layer and walks K serially with a
// barrier per 32 channels: 8-16 us of pure latency per launch, ~110 launches per training step of Assemble-ResNet-50.
// Here the operands are so small (<= 1 MiB, L2-resident) that no LDS staging is needed at all: both MFMA operands have
// their reduction index contiguous in memory, so a lane's fragment (8 consecutive k of one row) is one 16-byte global
// load.  A workgroup owns a 32 x 32 output tile and splits the reduction over its 4 waves (dense_small_kernel).  (A form that
// owned ALL rows of 32 output channels and ran the neighbouring training-mode batch norm in the same launch measured 0.2 ms
// per step slower -- one CU pulls the whole [256 x K] operand through its own L1 -- and was removed in round 6.)
//
// Fragment convention (as csrc/conv_igemm.hip): acc = mfma_32x32x16(A = q rows, B = p rows): lane l holds row (l & 31),
// reduction elements (l >> 5) * 8 .. +8; acc[r] = out[m = l & 31][n = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)].
#include "common.h"

namespace {

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
