// MFMA implicit-GEMM convolution for gfx950: fprop and dgrad share one gather-GEMM kernel.
//
//   Y[m][n] = sum_{tap t=(r,s)} sum_c  X[gather(m, t)][c] * Wt[n][t][c]
//
//   fprop: m = (img, ho, wo), X = activations NHWC, Wt = filter KRSC, n = output channel
//   dgrad: m = (img, h, w) of dx, X = dy NHWC, Wt = filter CRSK, n = input channel
//
// Block tile 128(M) x BN(N) x BK(K), 256 threads = 4 waves (2 x 2), v_mfma_f32_32x32x16_bf16.
// The MFMA "A" operand is the FILTER tile and the "B" operand the ACTIVATION tile, so an
// accumulator register quad holds 4 consecutive output channels of one pixel: the epilogue packs
// them to bf16 and stages the tile through LDS for 16-byte, fully coalesced NHWC stores.  The
// same pass accumulates per-channel sum / sum-of-squares of the bf16-ROUNDED outputs (the first
// half of the following batch-norm), written as deterministic per-M-block partials (no atomics).
//
// igemm_kernel below is the GENERAL form (any filter size, stride, channel count that is a multiple of 8): global -> LDS by
// LDS-DMA (global_load_lds_dwordx4), an out-of-image tap or a channel tail reads 16 zero bytes instead.  LDS rows are
// XOR-swizzled at 16-byte granularity so the ds_read_b128 fragment reads are bank-conflict free.  The workloads' own layers
// run on the specialised kernels further down (igemm2 / igemm3 / igemm8 / conv_halo, conv_gemm1.hip); the register-staged
// forms of this kernel (rounds 1 - 5) measured slower than its LDS-DMA form wherever it is still used and were removed.
#include "common.h"
#include "igemm_common.h"

using namespace asm_igemm;

// conv_gemm1.hip: the 1x1 layers as a GEMM with a ring of LDS stages (returns 1 when it does not take the layer)
int asm_gemm1_try(IGemmArgs& a, bool out_f32, bool stats, hipStream_t st);
// conv_igemm8.hip: the wide 3x3 stride-1 layers on the wave-staggered multi-phase main loop (returns 1 when it does not take the layer)
int asm_igemm8_try(IGemmArgs& a, bool out_f32, bool stats, hipStream_t st);
bool asm_igemm8_covers(const IGemmArgs& a, bool out_f32);

namespace {

// 16 zero bytes: the source of every masked lane of an LDS-DMA load (global_load_lds has no bounds check)
__device__ __attribute__((aligned(16))) unsigned g_zero16[4] = {0u, 0u, 0u, 0u};


template <int BM, int BN, int BK, int WGM, int WGN, bool OUT_F32, bool STATS, int MODE, bool BNRED = false>
__global__ __launch_bounds__(64 * WGM * WGN) void igemm_kernel(IGemmArgs p) {
  using C = Cfg<BM, BN, BK, WGM, WGN, OUT_F32, STATS, MODE>;
  constexpr int NT = C::NT, CPR = C::CPR, RPP = C::RPP, XP = C::XP, WP = C::WP, ROWB = C::ROWB, STAGE = C::STAGE;
  constexpr int TM = C::TM, TN = C::TN, WTM = C::WTM, WTN = C::WTN;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  // XCD-aware bijective remap: each XCD (bid % 8) gets a contiguous range of logical tiles, and
  // the N-tiles of one M-tile are adjacent, so the activation tile is re-read from that XCD's L2.
  int logical;
  {
    const int nb = p.n_blocks, q = nb >> 3, r = nb & 7;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = logical / p.n_tiles_n;
  const int tile_n = logical - tile_m * p.n_tiles_n;

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);

  const int chunk = tid % CPR;
  const int r0 = tid / CPR;

  // per-thread activation rows
  int xb[XP], bh[XP], bw[XP];
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int m = tile_m * BM + r0 + j * RPP;
    if (m < p.M) {
      const int img = m / p.HoWo;
      const int rem = m - img * p.HoWo;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      xb[j] = img * p.x_img_pitch;
      bh[j] = ho * p.so - p.pad;
      bw[j] = wo * p.so - p.pad;
    } else {
      xb[j] = 0;
      bh[j] = -(1 << 24);
      bw[j] = -(1 << 24);
    }
  }
  int wb[WP];
  bool wrow_ok[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const int row = r0 + j * RPP;
    const int n = tile_n * BN + row;
    wrow_ok[j] = row < BN;
    wb[j] = (wrow_ok[j] && n < p.Co) ? n * p.w_row_pitch : -1;
  }

  int kt_r = 0, kt_s = 0, kt_c = 0;  // tap / channel-chunk cursor of the NEXT tile to load (taps innermost, chunk outermost)
  // LDS-DMA staging: the thread -> (row, chunk) map above is already lane-linear per wave and pass (a wave
  // covers 64/CPR whole rows = 1 KiB), which is what global_load_lds requires of its destination; the XOR
  // swizzle therefore moves to the SOURCE side: the lane sitting at LDS chunk position `chunk` of row r fetches
  // global chunk (chunk ^ swz(r)).  Masked lanes read 16 zero bytes.
  auto issue_tile = [&](int stage, bool live) {
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    unsigned char* xs = smem + stage * STAGE;
    unsigned char* ws = xs + BM * ROWB;
    const int th = p.tsign * kt_r, tw = p.tsign * kt_s;
    const int wrow0 = (tid >> 6) * (64 / CPR);
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int row = r0 + j * RPP;
      const int c = kt_c * BK + ((chunk ^ swz<BK>(row)) << 3);
      int nh = bh[j] + th, nw = bw[j] + tw;
      bool ok = live && c < p.Ci;
      if (p.sd == 2) {
        ok = ok && (((nh | nw) & 1) == 0);
        nh >>= 1;
        nw >>= 1;
      }
      ok = ok && ((unsigned)nh < (unsigned)p.Hi) && ((unsigned)nw < (unsigned)p.Wi);
      const unsigned off = ((unsigned)xb[j] + (unsigned)nh * (unsigned)p.x_row_pitch +
                            (unsigned)nw * (unsigned)p.x_pix_pitch + (unsigned)c) * 2u;
      const unsigned char* src = ok ? reinterpret_cast<const unsigned char*>(p.x) + off
                                    : reinterpret_cast<const unsigned char*>(g_zero16);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xs + (j * RPP + wrow0) * ROWB), 16, 0, 0);
    }
    const int tap0 = (kt_r * p.S + kt_s) * p.Ci + kt_c * BK;
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const int row = r0 + j * RPP;
      if (j * RPP + wrow0 < BN) {  // wave-uniform: rows past the filter tile are never written
        const int c = ((chunk ^ swz<BK>(row)) << 3);
        const bool ok = live && (kt_c * BK + c < p.Ci) && (wb[j] >= 0);
        const unsigned off = (unsigned)(wb[j] + tap0 + c) * 2u;
        const unsigned char* src = ok ? reinterpret_cast<const unsigned char*>(p.w) + off
                                      : reinterpret_cast<const unsigned char*>(g_zero16);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ws + (j * RPP + wrow0) * ROWB), 16, 0, 0);
      }
    }
    if (++kt_s == p.S) {
      kt_s = 0;
      if (++kt_r == p.R) {
        kt_r = 0;
        ++kt_c;
      }
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  const int KT = p.R * p.S * p.kchunks;
  const int l31 = lane & 31, lhi = lane >> 5;

  auto compute = [&](int stage) {
    const unsigned char* xs = smem + stage * STAGE;
    const unsigned char* ws = xs + BM * ROWB;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const int ch = kk * 2 + lhi;
      bf16x8 fw[TN], fx[TM];
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        const int row = wn * WTN + a * 32 + l31;
        fw[a] = *reinterpret_cast<const bf16x8*>(ws + row * ROWB + ((ch ^ swz<BK>(row)) << 4));
      }
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int row = wm * WTM + b * 32 + l31;
        fx[b] = *reinterpret_cast<const bf16x8*>(xs + row * ROWB + ((ch ^ swz<BK>(row)) << 4));
      }
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[a], fx[b], acc[a][b], 0, 0, 0);
    }
  };

  issue_tile(0, true);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll 1
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) issue_tile(cur ^ 1, true);   // DMA of tile kt+1 runs under the MFMAs of tile kt
    compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  igemm_epilogue<C, BM, BN, WTM, WTN, TM, TN, OUT_F32, STATS, false, false, false, BNRED>(p, acc, smem, tile_m, tile_n, tid, wm, wn, l31, lhi);
}

// ------------------------------------------------------------------------------------------------------------
// igemm2_kernel: the same tiling and epilogue with a main loop stripped of per-step address arithmetic.
// PMC on the register-staged kernel showed ~10 VALU + 5 SALU instructions per MFMA (tap decode, bounds checks,
// offset selects, exec-mask branches): the waves were issue-bound, not MFMA- or LDS-bound.  Here
//   * every (staging row, filter tap) byte offset -- zero padding, stride-2 parity, row/channel tails folded in as
//     the out-of-range offset -- is computed ONCE in the prologue into VGPRs (XP x R*S of them);
//   * the K loop runs channel chunks outermost and the R*S taps fully unrolled innermost, so a step's loads are
//     `buffer_load_dwordx4 ... lds` with a pre-computed voffset and the chunk advance in the scalar soffset:
//     no VALU, no branches, no VGPR round trip (LDS-DMA zero-fills out-of-range lanes -- tools/probes);
//   * m -> (img, ho, wo) uses multiply-shift division.
// Requires R x S in {1x1, 3x3, 7x1, 3x1} and (Ci % BK == 0 or a single chunk); the rest stays on igemm_kernel.
template <int BM, int BN, int BK, int WGM, int WGN, bool OUT_F32, bool STATS, int NS>
struct Cfg2 {
  using C = Cfg<BM, BN, BK, WGM, WGN, OUT_F32, STATS, 2>;
  // NS = LDS stages.  A ring of 3-4 stages with counted vmcnt + raw s_barrier (DMA in flight across the barrier) was
  // measured at +-2 % over NS = 2 on every layer class, so only the double buffer is instantiated.
  static constexpr int WROWS = BN;
  static constexpr int STAGE = (BM + WROWS) * C::ROWB;
  static constexpr int LDS = cmax(cmax(NS * STAGE, C::EPI), C::RED);
  static_assert(LDS <= 160 * 1024, "lds");
};

// SCHED 1: the LDS-DMA pieces of step k+1 are issued IN BETWEEN the MFMA groups of step k instead of all at once right
// after the barrier (the 3x3 layers; the 1x1 layers issue them all at the head of the step).  Also tried (round 3, same box, conv_bench): pinning
// the fragment reads of group kk+1 at the head of group kk with sched_barrier(0) -- 2.62 vs 2.55 ms over the 12 heaviest
// shapes, slower than the compiler's own interleave, dropped.  Every wave comes out of the barrier at the same moment, and one piece
// costs its wave 60-180 issue cycles (v_readfirstlane + M0 write + the buffer_load itself): with 8 pieces up front both waves
// of a SIMD sit in their issue phase together and the matrix pipe idles for that long at the head of every step.
template <int BM, int BN, int BK, int WGM, int WGN, bool OUT_F32, bool STATS, int R, int S, int NS, bool PFA = false, bool POOL = false,
          int SCHED = 0, bool BNRED = false>
__global__ __launch_bounds__(64 * WGM * WGN) void igemm2_kernel(IGemmArgs p) {
  using C = Cfg<BM, BN, BK, WGM, WGN, OUT_F32, STATS, 2>;
  using C2 = Cfg2<BM, BN, BK, WGM, WGN, OUT_F32, STATS, NS>;
  constexpr int CPR = C::CPR, RPP = C::RPP, XP = C::XP, WP = C::WP, ROWB = C::ROWB, STAGE = C2::STAGE;
  constexpr int TM = C::TM, TN = C::TN, WTM = C::WTM, WTN = C::WTN;
  constexpr int NTAP = R * S;
  static_assert(RPP % 16 == 0, "the source-side swizzle must not depend on the pass");
  typedef __attribute__((address_space(3))) void* lptr_t;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  int logical;
  {
    const int nb = p.n_blocks, q = nb >> 3, r = nb & 7;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m0 = (int)fd_div((unsigned)logical, p.fd_ntn);
  const int tile_n = logical - tile_m0 * p.n_tiles_n;
  const int tile_m = tile_m0 + p.m_tile0;

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);

  const int chunk = tid % CPR;
  const int r0 = tid / CPR;
  const int csw = (chunk ^ swz<BK>(r0)) << 3;   // channel (element) offset of the 16-byte piece this lane fetches
  const bool c_ok = csw < p.Ci;                  // only a single-chunk layer can have a channel tail

  // ---- prologue: per-(row, tap) byte offsets of the activation tile, per-row offsets of the filter tile ----
  unsigned vx[XP][NTAP];
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int m = tile_m * BM + r0 + j * RPP;
    const bool live = c_ok && m < p.M;
    const unsigned img = fd_div((unsigned)m, p.fd_howo);
    const unsigned rem = (unsigned)m - img * (unsigned)p.HoWo;
    const unsigned ho = fd_div(rem, p.fd_wo);
    const unsigned wo = rem - ho * (unsigned)p.Wo;
    const int bh = (int)ho * p.so - p.pad, bw = (int)wo * p.so - p.pad_w;
    const unsigned base = img * (unsigned)p.x_img_pitch + (unsigned)csw;
    unsigned roff[R], coff[S];
    bool rok[R], cok[S];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int nh = bh + p.tsign * r;
      bool ok = live;
      if (p.sd == 2) {
        ok = ok && ((nh & 1) == 0);
        nh >>= 1;
      }
      rok[r] = ok && ((unsigned)nh < (unsigned)p.Hi);
      roff[r] = base + (unsigned)nh * (unsigned)p.x_row_pitch;
    }
#pragma unroll
    for (int q = 0; q < S; ++q) {
      int nw = bw + p.tsign * q;
      bool ok = true;
      if (p.sd == 2) {
        ok = ((nw & 1) == 0);
        nw >>= 1;
      }
      cok[q] = ok && ((unsigned)nw < (unsigned)p.Wi);
      coff[q] = (unsigned)nw * (unsigned)p.x_pix_pitch;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < S; ++q) vx[j][r * S + q] = (rok[r] && cok[q]) ? (roff[r] + coff[q]) * 2u : ASM_OOB;
  }
  unsigned vw[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const int row = r0 + j * RPP;
    const int n = tile_n * BN + row;
    vw[j] = (c_ok && row < BN && n < p.Co) ? ((unsigned)n * (unsigned)p.w_row_pitch + (unsigned)csw) * 2u : ASM_OOB;
  }

  const int wrow0 = wave * (64 / CPR);
  // pieces [lo, hi) of a step's XP + WP LDS-DMA pieces (activation rows first)
  auto issue_part = [&](int stage, const int t, unsigned xso, unsigned wso, const int lo, const int hi) {
    unsigned char* xs = smem + stage * STAGE;
    unsigned char* ws = xs + BM * ROWB;
#pragma unroll
    for (int j = 0; j < XP; ++j)
      if (j >= lo && j < hi)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xs + (j * RPP + wrow0) * ROWB), 16,
                                                 (int)vx[j][t], (int)xso, 0, 0);
#pragma unroll
    for (int j = 0; j < WP; ++j)
      if (XP + j >= lo && XP + j < hi && j * RPP + wrow0 < BN)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(ws + (j * RPP + wrow0) * ROWB), 16,
                                                 (int)vw[j], (int)wso, 0, 0);
  };
  auto issue = [&](int stage, const int t, unsigned xso, unsigned wso) {
    unsigned char* xs = smem + stage * STAGE;
    unsigned char* ws = xs + BM * ROWB;
#pragma unroll
    for (int j = 0; j < XP; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xs + (j * RPP + wrow0) * ROWB), 16,
                                               (int)vx[j][t], (int)xso, 0, 0);
#pragma unroll
    for (int j = 0; j < WP; ++j)
      if (j * RPP + wrow0 < BN)   // wave-uniform: rows past the filter tile are never written nor read
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(ws + (j * RPP + wrow0) * ROWB), 16,
                                                 (int)vw[j], (int)wso, 0, 0);
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  const int l31 = lane & 31, lhi = lane >> 5;
  // fragment byte offsets inside a stage (a = b = 0); +32 rows keeps the swizzle, so the other tiles are immediates
  unsigned fwo[BK / 16], fxo[BK / 16];
#pragma unroll
  for (int kk = 0; kk < BK / 16; ++kk) {
    const int ch = kk * 2 + lhi;
    const int rw_ = wn * WTN + l31, rx_ = wm * WTM + l31;
    fwo[kk] = BM * ROWB + rw_ * ROWB + ((ch ^ swz<BK>(rw_)) << 4);
    fxo[kk] = rx_ * ROWB + ((ch ^ swz<BK>(rx_)) << 4);
  }
  // Fragments are double-buffered in registers: the ds_reads of k-substep kk+1 are issued BEFORE the MFMAs of kk, and
  // the first fragments of the next stage right after the barrier that publishes it, before the last MFMA group of
  // the current step -- so LDS latency runs under MFMA execution instead of in front of it (the compiler emitted
  // read -> wait -> 4 MFMA per substep: ~45 % of every wave's loop time was spent waiting on lgkmcnt).
  constexpr int KK = BK / 16;
  bf16x8 fwb[2][TN], fxb[2][TM];
  auto load_frags = [&](int stage, const int kk, const int buf) {
    const unsigned char* sb = smem + stage * STAGE;
#pragma unroll
    for (int a = 0; a < TN; ++a) fwb[buf][a] = *reinterpret_cast<const bf16x8*>(sb + fwo[kk] + a * 32 * ROWB);
#pragma unroll
    for (int b = 0; b < TM; ++b) fxb[buf][b] = *reinterpret_cast<const bf16x8*>(sb + fxo[kk] + b * 32 * ROWB);
  };
  auto mma = [&](const int buf) {
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwb[buf][a], fxb[buf][b], acc[a][b], 0, 0, 0);
  };
  const unsigned tapw = (unsigned)p.Ci * 2u;   // bytes between consecutive taps of a filter row
  {
    // ---- DMA of step k+1 runs under the MFMAs of step k; one barrier per step ----
    issue(0, 0, 0u, (unsigned)p.wt0 * tapw);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_frags(0, 0, 0);
    int cur = 0;
    unsigned kcb = 0;                            // byte offset of the current channel chunk
#pragma unroll 1
    for (int kc = 0; kc < p.kchunks; ++kc) {
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const bool more = (t + 1 < NTAP) || (kc + 1 < p.kchunks);
        if constexpr (SCHED == 1 && KK >= 2) {
          constexpr int NP = XP + WP, G = KK - 1;            // pieces, MFMA groups in front of the barrier
          constexpr int PPG = (NP + G - 1) / G;
          const int tn = (t + 1 < NTAP) ? t + 1 : 0;         // compile-time after unrolling
          const unsigned xso = (t + 1 < NTAP) ? kcb : kcb + BK * 2;
          const unsigned wso = xso + (unsigned)(p.wt0 + (tn / S) * p.wtr + (tn % S) * p.wts) * tapw;
#pragma unroll
          for (int kk = 0; kk + 1 < KK; ++kk) {
            if (more) issue_part(cur ^ 1, tn, xso, wso, kk * PPG, (kk + 1) * PPG < NP ? (kk + 1) * PPG : NP);
            load_frags(cur, kk + 1, (kk + 1) & 1);
            mma(kk & 1);
          }
        } else {
        if (t + 1 < NTAP) {
          issue(cur ^ 1, t + 1, kcb, kcb + (unsigned)(p.wt0 + ((t + 1) / S) * p.wtr + ((t + 1) % S) * p.wts) * tapw);
        } else if (kc + 1 < p.kchunks) {
          issue(cur ^ 1, 0, kcb + BK * 2, kcb + BK * 2 + (unsigned)p.wt0 * tapw);
        }
#pragma unroll
        for (int kk = 0; kk + 1 < KK; ++kk) {
          load_frags(cur, kk + 1, (kk + 1) & 1);
          mma(kk & 1);
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // next tile visible; everyone's reads of this stage are in registers
        if (more) load_frags(cur ^ 1, 0, 0);
        mma((KK - 1) & 1);
        cur ^= 1;
      }
      kcb += BK * 2;
    }
  }

  igemm_epilogue<C, BM, BN, WTM, WTN, TM, TN, OUT_F32, STATS, PFA, POOL, false, BNRED>(p, acc, smem, tile_m, tile_n, tid, wm, wn, l31, lhi);
}

// ------------------------------------------------------------------------------------------------------------
// igemm3_kernel: 3x3 / stride 1 / pad 1 (fprop and its input gradient) with the ACTIVATION rows resident across the taps.
//
// Over the flattened pixel index m = (img, h, w) a 3x3 stride-1 convolution reads, for tap (r, s), the pixel
// m + (r - 1) * W + (s - 1): the nine taps are nine LINEAR row shifts of one [M][C] matrix, and what zero padding adds is
// a per-(pixel, tap) predicate (row / column of the tap outside the image), not a different address pattern.  igemm2
// stages a fresh 128-row activation tile for every tap (9 x 16 KB per 64-channel chunk, nearly all of it the same pixels
// again, through L2 -> LDS); here the 128 + 2 W + 2 rows the nine taps touch are LDS-DMA'd ONCE per 64-channel chunk
// (double-buffered over chunks: the next chunk's rows arrive one 32-row piece per tap under the current chunk's MFMAs),
// only the filter tile is streamed per tap, and a tap's activation fragment is a ds_read_b128 at a shifted row.  An
// out-of-image tap redirects the lane's fragment address to an all-zero row (one v_cndmask per tap and 32-row tile; the
// predicate bits are computed once per workgroup).  LDS-DMA traffic per chunk: 24 + 9 x 16 KB instead of 9 x 32 KB.
// Measured on the standalone probe (tools/probes/conv_probe.hip, MI355X, same box, plain epilogue): 14x14x512 -> 1024
// 464 -> 396 us, 14x14x256 -> 512 126 -> 115, 28x28x128 -> 256 138 -> 122, 7x7x256 -> 512 35.6 -> 32.7 us against the
// best gather-form tile of each layer; with the filter DMA as the only per-step traffic the loop runs within 4 % of its
// own MFMA + LDS-read time.  80 KB of LDS: two workgroups per CU, so one's epilogue runs under the other's main loop.
// Requires Ci % 64 == 0, at least two chunks (a single chunk has nothing to prefetch under), W <= 30, dense NHWC input.
// halo rows per buffer: 128 + 2 W + 2 <= H3_ROWS - 1 (rows past the live ones are zero-filled by the DMA: the last one is the
// all-zero row an out-of-image tap reads); 192 for W <= 30, 256 for the single-chunk form up to W = 62

// HR / NHB (round 5): halo rows per buffer and number of halo buffers.  A layer with ONE 64-channel chunk (Ci = 64: the SK
// convolutions of the 56 x 56 and 28 x 28 stages) has no next chunk to stage under the current one, so one buffer does, and
// with it 256 rows (W <= 62) fit 64 KB together with the two filter stages: two workgroups per CU like igemm2's tile, 178 +
// 9 x 128 instead of 9 x 256 staged rows per workgroup.
template <int BN, bool STATS, bool PFA, int H3_ROWS = 192, int NHB = 2, bool BNRED = false>
__global__ __launch_bounds__(256) void igemm3_kernel(IGemmArgs p) {
  constexpr int BM = 128, BK = 64, WGM = 2, WGN = 2;
  using C = Cfg<BM, BN, BK, WGM, WGN, false, STATS, 2>;
  constexpr int RPP = C::RPP, ROWB = C::ROWB, WP = C::WP;
  constexpr int TM = C::TM, TN = C::TN, WTM = C::WTM, WTN = C::WTN;
  constexpr int HP = H3_ROWS / RPP;                 // halo pieces per wave and chunk
  constexpr int HALO = H3_ROWS * ROWB, WST = BN * ROWB;
  constexpr int KK = BK / 16, NTAP = 9;
  constexpr unsigned ZERO_ROW = (unsigned)(H3_ROWS - 1) * ROWB;
  static_assert(RPP == 32 && H3_ROWS % RPP == 0 && (NHB == 1 || HP <= NTAP - 1), "halo pieces are issued one per tap");
  typedef __attribute__((address_space(3))) void* lptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // halo[2] | filter ring[2]; reused by the epilogue
  unsigned char* const wring = smem + NHB * HALO;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  int logical;
  {
    const int nb = p.n_blocks, q = nb >> 3, r = nb & 7;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m0 = (int)fd_div((unsigned)logical, p.fd_ntn);
  const int tile_n = logical - tile_m0 * p.n_tiles_n;
  const int tile_m = tile_m0 + p.m_tile0;

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);
  const int chunk = tid & 7, r0 = tid >> 3;
  const int csw = (chunk ^ swz<BK>(r0)) << 3;      // the source-side swizzle (RPP % 16 == 0: the same for every pass)
  const int Wd = p.Wi;
  const int hrows = BM + 2 * Wd + 2;

  // ---- prologue: halo / filter source offsets, per-(pixel, tap) predicates ----
  unsigned vh[HP];
#pragma unroll
  for (int j = 0; j < HP; ++j) {
    const int hr = r0 + j * RPP;
    const int g = tile_m * BM - (Wd + 1) + hr;      // pixel of halo row hr
    vh[j] = (hr < hrows && g >= 0 && g < p.M) ? ((unsigned)g * (unsigned)p.Ci + (unsigned)csw) * 2u : ASM_OOB;
  }
  unsigned vw[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const int n = tile_n * BN + r0 + j * RPP;
    vw[j] = n < p.Co ? ((unsigned)n * (unsigned)p.w_row_pitch + (unsigned)csw) * 2u : ASM_OOB;
  }
  const int l31 = lane & 31, lhi = lane >> 5;
  // loop tap t = (r, s) reads the pixel (h + ts * (r - 1), w + ts * (s - 1)), ts = +1 (fprop) / -1 (input gradient)
  unsigned vmask[TM];
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int m = tile_m * BM + wm * WTM + b * 32 + l31;
    const unsigned mm = m < p.M ? (unsigned)m : 0u;
    const unsigned img = fd_div(mm, p.fd_howo);
    const unsigned rem = mm - img * (unsigned)p.HoWo;
    const int h = (int)fd_div(rem, p.fd_wo);
    const int w = (int)rem - h * p.Wo;
    unsigned mk = 0;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      const int dr = p.tsign * (t / 3 - 1), ds = p.tsign * (t % 3 - 1);
      const bool ok = m < p.M && (unsigned)(h + dr) < (unsigned)p.Hi && (unsigned)(w + ds) < (unsigned)p.Wi;
      mk |= (ok ? 1u : 0u) << t;
    }
    vmask[b] = mk;
  }
  const int wrow0 = wave * 8;
  const unsigned tapw = (unsigned)p.Ci * 2u;

  auto issue_halo = [&](int hb, unsigned xso, const int j) {
    unsigned char* hs = smem + hb * HALO;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(hs + (j * RPP + wrow0) * ROWB), 16, (int)vh[j], (int)xso, 0, 0);
  };
  auto issue_w = [&](int stage, unsigned wso, const int lo, const int hi) {
    unsigned char* ws = wring + stage * WST;
#pragma unroll
    for (int j = 0; j < WP; ++j)
      if (j >= lo && j < hi)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(ws + (j * RPP + wrow0) * ROWB), 16, (int)vw[j], (int)wso, 0, 0);
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  unsigned fwo[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int rw_ = wn * WTN + l31;
    fwo[kk] = rw_ * ROWB + (((kk * 2 + lhi) ^ swz<BK>(rw_)) << 4);
  }
  const int xrow0 = wm * WTM + l31;     // halo row of this lane's first pixel at the shift (-1, -1)
  // byte offsets (inside a halo buffer) of this lane's activation fragments for loop tap t, k-substep 0; the swizzle is
  // keyed on the halo row (+32 rows keep it), an out-of-image tap reads the zero row
  unsigned xbv[TM];
  auto set_tap = [&](const int t) {
    const int dr = p.tsign * (t / 3 - 1), ds = p.tsign * (t % 3 - 1);
    const int row = xrow0 + (dr + 1) * Wd + (ds + 1);
    const unsigned xb = (unsigned)row * ROWB + ((unsigned)(lhi ^ swz<BK>(row)) << 4);
#pragma unroll
    for (int b = 0; b < TM; ++b) xbv[b] = ((vmask[b] >> t) & 1u) ? xb + (unsigned)(b * 32 * ROWB) : ZERO_ROW;
  };
  bf16x8 fwb[2][TN], fxb[2][TM];
  auto load_frags = [&](int hb, int stage, const int kk, const int buf) {
    const unsigned char* ws = wring + stage * WST;
    const unsigned char* hs = smem + hb * HALO;
#pragma unroll
    for (int a = 0; a < TN; ++a) fwb[buf][a] = *reinterpret_cast<const bf16x8*>(ws + fwo[kk] + a * 32 * ROWB);
#pragma unroll
    for (int b = 0; b < TM; ++b) fxb[buf][b] = *reinterpret_cast<const bf16x8*>(hs + (xbv[b] ^ (unsigned)(kk << 5)));
  };
  auto mma = [&](const int buf) {
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwb[buf][a], fxb[buf][b], acc[a][b], 0, 0, 0);
  };
  auto tap_off = [&](const int t) -> unsigned { return (unsigned)(p.wt0 + (t / 3) * p.wtr + (t % 3) * p.wts) * tapw; };

  {
    // ---- chunk 0's rows + the first filter tile; then one barrier per (chunk, tap) step.  A step opens with the barrier
    // that publishes its filter tile, issues the NEXT step's DMA at once (a whole step of MFMAs to land under: with only
    // 4 + 1 pieces per wave, spreading them over the step as igemm2 does leaves the last ones ~200 cycles before the wait)
    // and runs the previous step's last MFMA group under its own first fragment reads ----
#pragma unroll
    for (int j = 0; j < HP; ++j) issue_halo(0, 0u, j);
    issue_w(0, tap_off(0), 0, WP);
#pragma unroll
    for (int a = 0; a < TN; ++a) fwb[(KK - 1) & 1][a] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};   // the first "previous group" adds 0
#pragma unroll
    for (int b = 0; b < TM; ++b) fxb[(KK - 1) & 1][b] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    int cur = 0, hb = 0;
    unsigned kcb = 0;
#pragma unroll 1
    for (int kc = 0; kc < p.kchunks; ++kc) {
      const bool has_next = kc + 1 < p.kchunks;
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // this step's filter tile (at tap 0: this chunk's rows) visible; the
                                                 // stage / rows the DMA below overwrites have been read by every wave
        set_tap(t);
        load_frags(hb, cur, 0, 0);
        mma((KK - 1) & 1);
        const bool more = (t + 1 < NTAP) || has_next;
        const int tn = (t + 1 < NTAP) ? t + 1 : 0;           // compile time after unrolling
        if constexpr (NHB == 2) {
          if (has_next && t < HP) issue_halo(hb ^ 1, kcb + BK * 2, t);
        }
        if (more) issue_w(cur ^ 1, (t + 1 < NTAP ? kcb : kcb + BK * 2) + tap_off(tn), 0, WP);
#pragma unroll
        for (int kk = 0; kk + 1 < KK; ++kk) {
          load_frags(hb, cur, kk + 1, (kk + 1) & 1);
          mma(kk & 1);
        }
        cur ^= 1;
      }
      kcb += BK * 2;
      if constexpr (NHB == 2) hb ^= 1;
    }
    mma((KK - 1) & 1);
  }
  __syncthreads();   // every wave's fragment reads are done: the epilogue reuses the region
  igemm_epilogue<C, BM, BN, WTM, WTN, TM, TN, false, STATS, PFA, false, false, BNRED, 8>(p, acc, smem, tile_m, tile_n, tid, wm, wn, l31, lhi);
}

template <int BN, bool STATS, bool PFA, int HR = 192, int NHB = 2, bool BNRED = false>
int launch3_one(const IGemmArgs& a, hipStream_t st) {
  using C = Cfg<128, BN, 64, 2, 2, false, STATS, 2>;
  constexpr int LDS = cmax(cmax(NHB * HR * 128 + 2 * BN * 128, C::EPI), C::RED);
  static_assert(LDS <= 80 * 1024, "two workgroups per CU");
  auto kern = igemm3_kernel<BN, STATS, PFA, HR, NHB, BNRED>;
  static bool attr_done[ASM_MAX_DEVICES] = {};
  if (hipError_t e = asm_ensure_dyn_lds(kern, LDS, attr_done); e != hipSuccess)
    ASM_FAIL(ASM_EHIP, "igemm3_kernel: dynamic LDS opt-in: %s", hipGetErrorString(e));
  ASM_LAUNCH(kern, dim3(a.n_blocks), dim3(256), LDS, st, a);
  asm_last_conv_kernel = 3;
  ASM_CHECK_LAUNCH("igemm3_kernel");
  return ASM_OK;
}

// is the layer one igemm3_kernel covers?
bool igemm3_covers(const IGemmArgs& a, bool out_f32) {
  if (out_f32 || a.R != 3 || a.S != 3 || a.so != 1 || a.sd != 1 || a.y_strided || a.pool_dy) return false;
  if (!((a.tsign > 0 && a.pad == 1 && a.pad_w == 1) || (a.tsign < 0 && a.pad == -1 && a.pad_w == -1))) return false;
  if (a.wt0 != 0 || a.wtr != 3 || a.wts != 1) return false;
  const bool one_chunk = a.Ci == 64;      // asm_tuning.igemm3 >= 2: also the single-chunk layers (one halo buffer)
  if (one_chunk && asm_tune().igemm3 == 1) return false;
  if (a.Ci % 64 || a.Co <= 64 || a.Wi > ((one_chunk ? 256 : 192) - 1 - 128 - 2) / 2) return false;
  if (a.HoWo != a.Hi * a.Wi || a.Wo != a.Wi || a.M % a.HoWo) return false;
  if (a.x_pix_pitch != a.Ci || a.x_row_pitch != a.Wi * a.Ci || a.x_img_pitch != a.Hi * a.Wi * a.Ci) return false;
  return true;
}

// returns 1 when the layer is not one igemm3_kernel covers
int try_igemm3(IGemmArgs& a, bool out_f32, bool stats, hipStream_t st) {
  if (!igemm3_covers(a, out_f32)) return 1;
  const bool one_chunk = a.Ci == 64;
  a.n_tiles_n = cdiv(a.Co, 128);
  a.n_blocks = (cdiv(a.M, 128) - a.m_tile0) * a.n_tiles_n;
  a.kchunks = a.Ci / 64;
  a.fd_ntn = make_fastdiv((unsigned)a.n_tiles_n);
  const int pfa_env = asm_tune().igemm_pfa;
  const bool pfa = a.addend != nullptr && (pfa_env >= 0 ? pfa_env != 0 : a.n_blocks <= 1024);
  const bool bnred = stats && a.red_y;      // an input gradient that also reduces the batch-norm backward sums of its output
  if (one_chunk) {
    if (a.Wi <= 30) {
      if (bnred) return launch3_one<128, true, false, 192, 1, true>(a, st);
      if (stats) return launch3_one<128, true, false, 192, 1>(a, st);
      if (pfa) return launch3_one<128, false, true, 192, 1>(a, st);
      return launch3_one<128, false, false, 192, 1>(a, st);
    }
    if (bnred) return launch3_one<128, true, false, 256, 1, true>(a, st);
    if (stats) return launch3_one<128, true, false, 256, 1>(a, st);
    if (pfa) return launch3_one<128, false, true, 256, 1>(a, st);
    return launch3_one<128, false, false, 256, 1>(a, st);
  }
  if (bnred) return launch3_one<128, true, false, 192, 2, true>(a, st);
  if (stats) return launch3_one<128, true, false>(a, st);
  if (pfa) return launch3_one<128, false, true>(a, st);
  return launch3_one<128, false, false>(a, st);
}

// ------------------------------------------------------------------------------------------------------------
// conv_halo_kernel: stride-1 3x3 convolution (and its input gradient) for the NARROW high-resolution layers
// (Ci in {32, 64}, 112 x 112 maps: the BigLittle module-0 / ResNet-D stem convolutions).
// The gather-GEMM stages one activation tile PER TAP, so the 9 taps of a 3x3 read (almost) the same pixels 9 times
// through L2 -> LDS: these layers need ~0.6 GB of HBM traffic but ~5 GB of L2 -> LDS traffic and ran at 2.0-3.0 TB/s
// of HBM-equivalent rate (L2-bound, 14-20 % of the MFMA peak).  Here a workgroup owns an 8 x 16 pixel patch of one
// image: the 10 x 18 halo of the patch is staged in LDS ONCE (rows padded by 16 bytes: conflict-free ds_read_b128 at
// any tap offset), the whole 9-tap filter slice for the workgroup's output channels is LDS-DMA'd once, and the tap loop
// runs entirely out of LDS with compile-time tap offsets and NO barriers.  Epilogue (bf16 pack, coalesced stores, fused
// BN statistics / gradient fan-in addend) is the shared one.
// Requires H % 8 == 0, W % 16 == 0, stride 1, pad 1 (so tiles == M / 128 and the statistics partials keep their layout).
template <int KO, int CI>
struct HaloCfg {
  static constexpr int RB = CI * 2 + 16;                    // padded halo row (bytes)
  static constexpr int HALO = 10 * 18 * RB;
  static constexpr int FILT = 9 * KO * CI * 2;              // the whole 3x3 filter slice of the workgroup's KO rows
  static constexpr int EPI = 128 * (KO * 2 + 16);           // Cfg::EPI
  static constexpr int RED = (256 / (KO / 8)) * KO * 2 * 4; // Cfg::RED
  static constexpr int WORK = cmax(cmax(HALO, EPI), RED);   // halo, then output tile / statistics scratch
  static constexpr int LDS = FILT + WORK;
  static constexpr int PER_CU = (160 * 1024) / LDS < 4 ? (160 * 1024) / LDS : 4;
};

// PERSISTENT and weight-stationary: a workgroup loads its filter slice ONCE and then walks a contiguous run of patches
// (vertical / horizontal neighbours back to back: shared halo rows come from this XCD's L2).  The halo of patch i+1 is
// prefetched into registers while patch i is multiplied and stored, so a workgroup hides its own HBM latency.
template <int KO, int CI, int WGM, int WGN, bool STATS>
__global__ __launch_bounds__(256) void conv_halo_kernel(IGemmArgs p) {
  using C = Cfg<128, KO, CI, WGM, WGN, false, STATS, 2>;
  using H = HaloCfg<KO, CI>;
  constexpr int TM = C::TM, TN = C::TN, WTM = C::WTM, WTN = C::WTN;
  constexpr int HW_ = 18;                           // halo width in pixels (height 10)
  constexpr int RB = H::RB;
  constexpr int CPR = CI / 8;                       // 16-byte chunks per pixel / per filter row
  constexpr int WROWB = CI * 2;                     // filter tile row bytes (XOR-swizzled, as in igemm2)
  constexpr int WTAP = KO * WROWB;                  // one tap's filter tile
  constexpr int KK = CI / 16;
  constexpr int NV = 10 * HW_ * CPR;                // 16-byte vectors of one halo
  constexpr int HP = (NV + 255) / 256;              // halo vectors per thread
  static_assert(H::EPI == C::EPI && H::RED >= C::RED, "epilogue scratch size");
  typedef __attribute__((address_space(3))) void* lptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ws = smem;                         // filter (resident)
  unsigned char* xs = smem + H::FILT;               // halo of the current patch; reused by the epilogue

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, lhi = lane >> 5;

  // this workgroup's run of (row-tile) patches; block b runs on XCD b % 8, so give XCD x the x-th eighth of the patches
  const int n_m = p.M >> 7;
  int t_begin, t_end;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective remap
    const int per = n_m / nb, extra = n_m - per * nb;
    t_begin = logical * per + (logical < extra ? logical : extra);
    t_end = t_begin + per + (logical < extra ? 1 : 0);
  }
  const int tiles_x = p.Wi >> 4, tpi = tiles_x * (p.Hi >> 3);

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);

  // ---- filter: all 9 taps of the KO rows by LDS-DMA (lane-linear destination, swizzle on the source), once ----
  {
    const int chunk = tid % CPR, r0 = tid / CPR;
    constexpr int RPP = 256 / CPR;
    const int csw = (chunk ^ swz<CI>(r0)) << 3;
    const int wrow0 = wave * (64 / CPR);
#pragma unroll
    for (int j = 0; j < (KO + RPP - 1) / RPP; ++j) {
      const int row = r0 + j * RPP;
      const unsigned vw = (row < KO && row < p.Co) ? ((unsigned)row * (unsigned)p.w_row_pitch + (unsigned)csw) * 2u : ASM_OOB;
      if (j * RPP + wrow0 < KO) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(ws + t * WTAP + (j * RPP + wrow0) * WROWB), 16, (int)vw,
                                                   (int)((unsigned)t * (unsigned)p.Ci * 2u), 0, 0);
      }
    }
  }

  // halo vector i of a thread: pixel hp = i / CPR of the 10 x 18 halo, chunk ck
  auto load_halo = [&](int tile, u32x4 (&hv)[HP]) {
    const int img = tile / tpi, trem = tile - img * tpi;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int y0 = ty * 8, x0 = tx * 16;
    const unsigned img_off = (unsigned)img * (unsigned)p.x_img_pitch;
#pragma unroll
    for (int k = 0; k < HP; ++k) {
      const int i = k * 256 + tid;
      const int hp = i / CPR, ck = i - hp * CPR;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = (i < NV) && ((unsigned)gy < (unsigned)p.Hi) && ((unsigned)gx < (unsigned)p.Wi);
      const unsigned off = (img_off + (unsigned)gy * (unsigned)p.x_row_pitch + (unsigned)gx * (unsigned)p.x_pix_pitch +
                            (unsigned)ck * 8u) * 2u;
      hv[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : ASM_OOB, 0, 0);
    }
  };
  auto store_halo = [&](const u32x4 (&hv)[HP]) {
#pragma unroll
    for (int k = 0; k < HP; ++k) {
      const int i = k * 256 + tid;
      const int hp = i / CPR, ck = i - hp * CPR;
      if (i < NV) *reinterpret_cast<u32x4*>(xs + hp * RB + ck * 16) = hv[k];
    }
  };

  // per-lane fragment bases: activation pixel (py, px) of row m_local -> halo index py * 18 + px (top-left of its window)
  unsigned fxo[TM], fwo[TN];
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int row = wm * WTM + b * 32 + l31;
    fxo[b] = (unsigned)(((row >> 4) * HW_ + (row & 15)) * RB + lhi * 16);
  }
#pragma unroll
  for (int a = 0; a < TN; ++a) fwo[a] = (unsigned)((wn * WTN + a * 32 + l31) * WROWB);
  const int wsw = swz<CI>(wn * WTN + l31);   // rows 32 apart share the swizzle term

  u32x4 hv[HP];
  if (t_begin < t_end) load_halo(t_begin, hv);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the filter DMA (and the first halo) have landed

#pragma unroll 1
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();                 // the previous patch's epilogue is done with the scratch region (1st trip: filter visible)
    store_halo(hv);
    __syncthreads();
    if (tile + 1 < t_end) load_halo(tile + 1, hv);   // in flight under the MFMAs and the stores of this patch

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    // tap loop: fprop (tsign > 0) reads halo (py + r, px + s) with filter tap (r, s); the input gradient (tsign < 0)
    // reads halo (py + 2 - r, px + 2 - s) with the same filter tap (dx(h, w) = sum dy(h + 1 - r, w + 1 - s) . w(r, s))
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int r = t / 3, q = t - r * 3;
      const int hoff = (p.tsign > 0 ? (r * HW_ + q) : ((2 - r) * HW_ + (2 - q))) * RB;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        bf16x8 fw[TN], fx[TM];
#pragma unroll
        for (int a = 0; a < TN; ++a)
          fw[a] = *reinterpret_cast<const bf16x8*>(ws + t * WTAP + fwo[a] + (((kk * 2 + lhi) ^ wsw) << 4));
#pragma unroll
        for (int b = 0; b < TM; ++b) fx[b] = *reinterpret_cast<const bf16x8*>(xs + fxo[b] + hoff + kk * 32);
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[a], fx[b], acc[a][b], 0, 0, 0);
      }
    }
    __syncthreads();   // every wave is done reading the halo: the epilogue reuses the region for the output tile
    const int img = tile / tpi, trem = tile - img * tpi;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int patch_base = (img * p.Hi + ty * 8) * p.Wi + tx * 16;
    igemm_epilogue<C, 128, KO, WTM, WTN, TM, TN, false, STATS>(p, acc, xs, tile, 0, tid, wm, wn, l31, lhi, patch_base);
  }
}

template <int KO, int CI, int WGM, int WGN>
int launch_halo(IGemmArgs& a, bool stats, hipStream_t st) {
  using H = HaloCfg<KO, CI>;
  a.n_tiles_n = 1;
  const int n_m = a.M / 128;
  a.n_blocks = n_m < H::PER_CU * 256 ? n_m : H::PER_CU * 256;
  static bool attr_done[2][ASM_MAX_DEVICES] = {};
  if (stats) {
    auto kern = conv_halo_kernel<KO, CI, WGM, WGN, true>;
    if (hipError_t e = asm_ensure_dyn_lds(kern, H::LDS, attr_done[1]); e != hipSuccess)
      ASM_FAIL(ASM_EHIP, "conv_halo_kernel: dynamic LDS opt-in: %s", hipGetErrorString(e));
    ASM_LAUNCH(kern, dim3(a.n_blocks), dim3(256), H::LDS, st, a);
  } else {
    auto kern = conv_halo_kernel<KO, CI, WGM, WGN, false>;
    if (hipError_t e = asm_ensure_dyn_lds(kern, H::LDS, attr_done[0]); e != hipSuccess)
      ASM_FAIL(ASM_EHIP, "conv_halo_kernel: dynamic LDS opt-in: %s", hipGetErrorString(e));
    ASM_LAUNCH(kern, dim3(a.n_blocks), dim3(256), H::LDS, st, a);
  }
  asm_last_conv_kernel = 4;
  ASM_CHECK_LAUNCH("conv_halo_kernel");
  return ASM_OK;
}

// is the layer one the halo kernel covers?
bool halo_covers(const IGemmArgs& a, bool out_f32) {
  if (out_f32 || a.R != 3 || a.S != 3 || a.so != 1 || a.sd != 1 || a.y_strided || a.bn_scale) return false;
  if (a.Hi % 8 || a.Wi % 16 || a.M != (a.HoWo / a.Wo) * a.Wo * (a.M / a.HoWo) || a.HoWo != a.Hi * a.Wi) return false;
  if (a.x_pix_pitch != a.Ci || a.x_row_pitch != a.Wi * a.Ci || a.x_img_pitch != a.Hi * a.Wi * a.Ci) return false;
  if (!((a.tsign > 0 && a.pad == 1 && a.pad_w == 1) || (a.tsign < 0 && a.pad == -1 && a.pad_w == -1))) return false;
  if (a.wt0 != 0 || a.wtr != 3 || a.wts != 1) return false;
  return (a.Ci == 32 || a.Ci == 64) && (a.Co == 32 || a.Co == 64);
}

// returns 1 when the layer is not one the halo kernel covers
int try_halo(IGemmArgs& a, bool out_f32, bool stats, hipStream_t st) {
  if (!halo_covers(a, out_f32)) return 1;
  if (a.Ci == 64 && a.Co == 32) return launch_halo<32, 64, 4, 1>(a, stats, st);
  if (a.Ci == 32 && a.Co == 32) return launch_halo<32, 32, 4, 1>(a, stats, st);
  if (a.Ci == 32 && a.Co == 64) return launch_halo<64, 32, 2, 2>(a, stats, st);
  if (a.Ci == 64 && a.Co == 64) return launch_halo<64, 64, 2, 2>(a, stats, st);
  return 1;
}

template <int BM, int BN, int BK, int WGM, int WGN, bool OUT_F32, bool STATS, int R, int S, int NS = 2, bool PFA = false, bool POOL = false,
          int SCHED = 0, bool BNRED = false>
int launch2_one(const IGemmArgs& a, hipStream_t st) {
  using C = Cfg2<BM, BN, BK, WGM, WGN, OUT_F32, STATS, NS>;
  constexpr int NTHR = 64 * WGM * WGN;
  // measured (tools/conv_bench.py, same box, round 3): spreading the DMA issue is +2..4 % on the 3x3 layers (7 of 8 shapes,
  // fprop and dgrad) and -3..6 % on the deep 1x1 layers (their steps are short: the last pieces land too late), hence the
  // per-layer choice
  if constexpr (SCHED == 0 && BK == 64 && BN >= 128 && !OUT_F32 && !POOL && ((R == 3 && S == 3) || (R == 1 && S == 1))) {
    if (R == 3) return launch2_one<BM, BN, BK, WGM, WGN, OUT_F32, STATS, R, S, NS, PFA, POOL, 1>(a, st);
  }
  auto kern = igemm2_kernel<BM, BN, BK, WGM, WGN, OUT_F32, STATS, R, S, NS, PFA, POOL, SCHED, BNRED>;
  static bool attr_done[ASM_MAX_DEVICES] = {};
  if (hipError_t e = asm_ensure_dyn_lds(kern, C::LDS, attr_done); e != hipSuccess)
    ASM_FAIL(ASM_EHIP, "igemm2_kernel: dynamic LDS opt-in: %s", hipGetErrorString(e));
  ASM_LAUNCH(kern, dim3(a.n_blocks), dim3(NTHR), C::LDS, st, a);
  asm_last_conv_kernel = 2;
  ASM_CHECK_LAUNCH("igemm2_kernel");
  return ASM_OK;
}

// returns 1 if this (tile, tap shape) has no igemm2 instantiation
template <int BM, int BN, int BK, int WGM, int WGN>
int launch2_cfg(IGemmArgs& a, bool out_f32, bool stats, hipStream_t st) {
  a.n_tiles_n = cdiv(a.Co, BN);
  a.n_blocks = (cdiv(a.M, BM) - a.m_tile0) * a.n_tiles_n;     // (m_tile0 != 0 only on the way to a 128-row tile)
  a.kchunks = cdiv(a.Ci, BK);
  a.fd_ntn = make_fastdiv((unsigned)a.n_tiles_n);
  if (a.Ci % BK != 0 && a.kchunks != 1) return 1;
  // addend-prefetching epilogue (see igemm_epilogue): where a launch is one round of few workgroups, or one 256-row
  // workgroup per CU anyway; ASM_IGEMM_PFA=0 / 1 forces it off / on (tests, A/B)
  const int pfa_env = asm_tune().igemm_pfa;
  const bool pfa = a.addend != nullptr && !a.y_strided &&
                   (pfa_env >= 0 ? pfa_env != 0 : (BM == 256 || a.n_blocks <= 1024));
  if (a.R == 1 && a.S == 1) {
    if (out_f32) return launch2_one<BM, BN, BK, WGM, WGN, true, false, 1, 1>(a, st);
    if (stats && a.red_y) return launch2_one<BM, BN, BK, WGM, WGN, false, true, 1, 1, 2, false, false, 0, true>(a, st);
    if (stats) return launch2_one<BM, BN, BK, WGM, WGN, false, true, 1, 1>(a, st);
    if (a.pool_dy) return launch2_one<BM, BN, BK, WGM, WGN, false, false, 1, 1, 2, false, true>(a, st);
    if (pfa) return launch2_one<BM, BN, BK, WGM, WGN, false, false, 1, 1, 2, true>(a, st);
    return launch2_one<BM, BN, BK, WGM, WGN, false, false, 1, 1>(a, st);
  }
  if (out_f32) return 1;
  if (a.R == 3 && a.S == 3) {
    if (stats) return launch2_one<BM, BN, BK, WGM, WGN, false, true, 3, 3>(a, st);
    if (pfa) return launch2_one<BM, BN, BK, WGM, WGN, false, false, 3, 3, 2, true>(a, st);
    return launch2_one<BM, BN, BK, WGM, WGN, false, false, 3, 3>(a, st);
  }
  if (!stats) {   // parity classes of a stride-2 3x3 input gradient (asm_conv2d_dgrad)
    if (a.R == 1 && a.S == 2) return launch2_one<BM, BN, BK, WGM, WGN, false, false, 1, 2>(a, st);
    if (a.R == 2 && a.S == 1) return launch2_one<BM, BN, BK, WGM, WGN, false, false, 2, 1>(a, st);
    if (a.R == 2 && a.S == 2) return launch2_one<BM, BN, BK, WGM, WGN, false, false, 2, 2>(a, st);
  }
  if constexpr (BK == 32 && BM == 128 && BN <= 64) {   // the two stems (R = k, S = 1 over the 4-channel halo buffer)
    if (a.R == 7 && a.S == 1) {
      if (stats) return launch2_one<BM, BN, BK, WGM, WGN, false, true, 7, 1>(a, st);
      return launch2_one<BM, BN, BK, WGM, WGN, false, false, 7, 1>(a, st);
    }
    if (a.R == 3 && a.S == 1) {
      if (stats) return launch2_one<BM, BN, BK, WGM, WGN, false, true, 3, 1>(a, st);
      return launch2_one<BM, BN, BK, WGM, WGN, false, false, 3, 1>(a, st);
    }
  }
  return 1;
}

template <int BM, int BN, int BK, int WGM, int WGN, bool OUT_F32, bool STATS, int MODE, bool BNRED = false>
int launch_one(const IGemmArgs& a, hipStream_t st) {
  using C = Cfg<BM, BN, BK, WGM, WGN, OUT_F32, STATS, MODE>;
  auto kern = igemm_kernel<BM, BN, BK, WGM, WGN, OUT_F32, STATS, MODE, BNRED>;
  static bool attr_done[ASM_MAX_DEVICES] = {};
  if (hipError_t e = asm_ensure_dyn_lds(kern, C::LDS, attr_done); e != hipSuccess)
    ASM_FAIL(ASM_EHIP, "igemm_kernel: dynamic LDS opt-in: %s", hipGetErrorString(e));
  ASM_LAUNCH(kern, dim3(a.n_blocks), dim3(C::NT), C::LDS, st, a);
  asm_last_conv_kernel = 0;
  ASM_CHECK_LAUNCH("igemm_kernel");
  return ASM_OK;
}

template <int BM, int BN, int BK, int WGM, int WGN, int MODE>
int launch_mode(IGemmArgs& a, bool out_f32, bool stats, hipStream_t st) {
  a.n_tiles_n = cdiv(a.Co, BN);
  a.n_blocks = cdiv(a.M, BM) * a.n_tiles_n;
  a.kchunks = cdiv(a.Ci, BK);
  if (out_f32) return launch_one<BM, BN, BK, WGM, WGN, true, false, MODE>(a, st);
  if (stats && a.red_y) return launch_one<BM, BN, BK, WGM, WGN, false, true, MODE, true>(a, st);
  if (stats) return launch_one<BM, BN, BK, WGM, WGN, false, true, MODE>(a, st);
  return launch_one<BM, BN, BK, WGM, WGN, false, false, MODE>(a, st);
}

// ASM_IGEMM_MODE != 0 / ASM_IGEMM_TILE (1 = 128-row tiles, 3 = 256x256) force a choice (tests, tuning).
template <int BM, int BN, int BK, int WGM, int WGN>
int launch_cfg(IGemmArgs& a, bool out_f32, bool stats, hipStream_t st) {
  return launch_mode<BM, BN, BK, WGM, WGN, 2>(a, out_f32, stats, st);
}

// which of the specialised 3x3 kernels launch() would try for a layer (the knobs included)
struct Sel3 {
  bool bigv, to_igemm3, try8, try3;
};
Sel3 select_3x3(const IGemmArgs& a) {
  const int ftile = asm_tune().igemm_tile, h3 = asm_tune().igemm3, h8 = asm_tune().igemm8;
  const bool heavy = a.Ci % 64 == 0 && (long long)a.R * a.S * a.Ci >= 512;
  const long long b256v = (long long)cdiv(a.M, 256) * cdiv(a.Co, 256);
  Sel3 r;
  r.bigv = heavy && a.Co >= 256 && b256v >= 192;
  if (ftile == 1) r.bigv = false;
  if (ftile == 3 && a.Ci % 64 == 0) r.bigv = true;
  // (the short-reduction layers on >= 768 tiles stay on igemm3: 28x28x128 -> 256 forward 127 us there, 131 us on igemm8)
  r.to_igemm3 = h3 && b256v >= 768 && a.Ci <= 128;
  r.try8 = ftile == 0 && ((h8 == 1 && r.bigv && !r.to_igemm3) || h8 == 2);
  r.try3 = ftile == 0 && h3 && (h3 == 2 || !r.bigv || r.to_igemm3);
  return r;
}

// An input gradient that also reduces the batch-norm backward sums of its output (IGemmArgs::red_y) needs the BNRED instantiation
// of whatever kernel runs it: for the 3x3 layers those exist for igemm8 and igemm3.  Would launch() end up on one of them?
bool bnred_3x3_supported(const IGemmArgs& a) {
  if (asm_tune().igemm_mode != 0 || asm_tune().igemm_tile != 0) return false;
  if (halo_covers(a, false)) return false;       // launch() would send it to conv_halo_kernel: no such instantiation there
  const Sel3 s3 = select_3x3(a);
  return (s3.try8 && asm_igemm8_covers(a, false)) || (s3.try3 && igemm3_covers(a, false));
}

int launch(IGemmArgs& a, bool out_f32, bool stats, hipStream_t st, bool igemm2_only = false) {
  // Measured on MI355X (tools/conv_bench.py, Assemble-ResNet-50 shapes, batch 256):
  //  * HBM-bound layers (1x1, and everything at 112x112): LDS-DMA staging + the smallest footprint wins
  //    (3-8 workgroups per CU hide the load round trip of the very short K loops);
  //  * MFMA-bound layers (K = R*S*C >= 512): 256x256 / 8 waves wins once there are >= 192 such tiles (C>=256 outputs).
  // 1x1 layers stage 32 channels per step (smallest footprint: the bandwidth-bound ones want many workgroups per CU) --
  // except deep reductions on few tiles (7x7 / 14x14 / 28x28 maps), where each of the K / 32 steps is an exposed DMA
  // round trip behind a barrier: up to 4000 tiles of 128 x 128 they take 64-channel steps (tools/conv_bench.py, all 1x1
  // shapes of the network: fprop 2.36 -> 2.31 ms, input gradients 2.11 -> 2.03 ms per step).  (3x3 layers on 128-row tiles
  // with 32-channel steps -- half the LDS, four workgroups per CU -- measured slower and were removed in round 6.)
  const long long t128_all = (long long)cdiv(a.M, 128) * cdiv(a.Co, 128);
  constexpr long long bk64_tiles = 4000;
  const bool bk64 = a.Ci % 64 == 0 && (a.R * a.S > 1 || (a.Ci >= 256 && t128_all <= bk64_tiles));
  const bool heavy = a.Ci % 64 == 0 && (long long)a.R * a.S * a.Ci >= 512;
  const int fmode = asm_tune().igemm_mode, ftile = asm_tune().igemm_tile;
  a.fd_howo = make_fastdiv((unsigned)a.HoWo);
  a.fd_wo = make_fastdiv((unsigned)a.Wo);
  const bool v2 = true;     // (asm_tuning.igemm_mode != 0 is what sends a layer to the general kernel)
  if (a.pool_dy && !(v2 && fmode == 0 && !out_f32 && !stats && a.R == 1 && a.S == 1 && !a.y_strided))
    ASM_FAIL(ASM_ENOTSUP, "conv dgrad_pooled: only the 1x1 stride-1 igemm2 path folds an average-pool backward in");
  // only igemm8 and igemm3 have that instantiation for a 3x3 (in the bandwidth-bound halo kernel of the 112-wide maps the y reads
  // cost 0.22 ms per step and saved 0.09 of reduce passes: measured, removed)
  const bool bnred3 = stats && a.red_y && a.R * a.S > 1;
  if (v2 && fmode == 0 && ftile == 0 && !bnred3) {
    const int rc = try_halo(a, out_f32, stats, st);
    if (rc != 1) return rc;
  }
  if (v2 && fmode == 0 && ftile == 0 && a.R == 1 && a.S == 1 && asm_tune().gemm1 != 0) {
    const int rc = asm_gemm1_try(a, out_f32, stats, st);   // ring-pipelined GEMM form of the 1x1 layers (conv_gemm1.hip)
    if (rc != 1) return rc;
  }
  if (v2 && fmode == 0) {
    int rc;
    const Sel3 s3 = select_3x3(a);
    const bool bigv = s3.bigv;
    // igemm3_kernel (128 x 128 tiles, activation rows resident across the taps, two workgroups per CU) against igemm2
    // (tools/conv_bench.py --iters 50, same box, steady state): it wins wherever igemm2 would run 128-row tiles
    // (28x28x128 -> 256 input gradient 124 -> 112 us, 7x7x256 -> 512 37 -> 35 / 48.6 -> 40, 7x7x512 -> 1024 input gradient
    // 114 -> 108, 14x14x128 -> 256 input gradient 35.8 -> 33.2) and on the 784-tile short-reduction forward layer
    // (28x28x128 -> 256: 131 -> 117.5); against the 256 x 256 tile it loses 3 - 15 % where that tile fills the chip
    // (its 8-wave loop reads 0.75 instead of 1 LDS fragment per MFMA and has half the barriers), so those stay.
    // asm_tuning.igemm3 = 2 forces it wherever the shape allows (tests).
    // (asm_tuning.igemm3 = 4 -- also the deep 14- / 7-wide layers of the 256 x 256 tile on igemm3 -- measured 0.2 ms slower in the
    // step in rounds 5 and 6 and was removed.)
    // igemm8_kernel: the layers of the 256 x 256 tile on the wave-staggered multi-phase loop (bit-identical results)
    const int h8 = asm_tune().igemm8;
    if (s3.try8) {
      // The ragged last round.  One 128 KB workgroup per CU: n tiles take ceil(n / CUs) rounds, and 784 tiles on 256 CUs
      // (14x14x512 -> 1024 and 28x28x128 -> 256 at batch 256) spend a whole round on their last 16.  When the tail is short,
      // the row tiles of the full rounds go to igemm8 and the remaining rows to the 128 x 128 kernels (igemm3 / igemm2: four
      // times the workgroups, two per CU, the same accumulation order -- the tensor stays bit-identical): 3 rounds + one
      // small round (~0.3 of a big one when there is at most one small tile per CU, ~0.5 per round of two) instead of 4.
      // Measured (14x14x512 -> 1024 forward, same box): 490.2 us igemm2, 455.5 us split (three igemm8 rounds ~355 us + ~100 us
      // for the 64 small tiles, which run one per CU and are latency-bound: 72 lock-step steps).
      const int cus = asm_num_cus();
      const int nt8 = cdiv(a.Co, 256), mt8 = cdiv(a.M, 256);
      const long long n8 = (long long)nt8 * mt8;
      const int m_full = (int)((n8 / cus) * cus / nt8);                 // row tiles of the full rounds
      const long long rem = n8 - (long long)m_full * nt8;              // 256 x 256 tiles left over
      const double small = 4 * rem <= cus ? 0.3 : 0.5 * (double)((4 * rem + 2 * cus - 1) / (2 * cus));
      const double split_cost = (double)((long long)m_full * nt8 / cus) + small, whole_cost = (double)((n8 + cus - 1) / cus);
      if (h8 == 1 && m_full > 0 && rem > 0 && split_cost < whole_cost - 0.15 && (!bnred3 || igemm3_covers(a, out_f32))) {
        IGemmArgs head = a;
        head.M = m_full * 256;            // rows of the full rounds (the gather itself is bounded by the tensor, not by M)
        rc = asm_igemm8_try(head, out_f32, stats, st);
        if (rc == ASM_OK) {
          IGemmArgs tail = a;
          tail.m_tile0 = m_full * 2;      // in 128-row tiles
          rc = try_igemm3(tail, out_f32, stats, st);
          if (rc == 1) rc = launch2_cfg<128, 128, 64, 2, 2>(tail, out_f32, stats, st);
          if (rc == 1) ASM_FAIL(ASM_EINVAL, "conv: no 128-row kernel for the last rows of an igemm8 layer");
          return rc;
        }
        if (rc != 1) return rc;
      } else {
        rc = asm_igemm8_try(a, out_f32, stats, st);
        if (rc != 1) return rc;
      }
    }
    if (s3.try3) {
      rc = try_igemm3(a, out_f32, stats, st);
      if (rc != 1) return rc;
    }
    if (bnred3) ASM_FAIL(ASM_ENOTSUP, "conv dgrad_bnred: no kernel with the batch-norm sums for this 3x3 layer");
    if (a.Co <= 32) rc = bk64 ? launch2_cfg<128, 32, 64, 4, 1>(a, out_f32, stats, st) : launch2_cfg<128, 32, 32, 4, 1>(a, out_f32, stats, st);
    else if (a.Co <= 64) rc = bk64 ? launch2_cfg<128, 64, 64, 2, 2>(a, out_f32, stats, st) : launch2_cfg<128, 64, 32, 2, 2>(a, out_f32, stats, st);
    else if (bigv) rc = launch2_cfg<256, 256, 64, 4, 2>(a, out_f32, stats, st);
    else rc = bk64 ? launch2_cfg<128, 128, 64, 2, 2>(a, out_f32, stats, st) : launch2_cfg<128, 128, 32, 2, 2>(a, out_f32, stats, st);
    if (rc != 1 || igemm2_only) return rc;
    if (a.pool_dy) ASM_FAIL(ASM_ENOTSUP, "conv dgrad_pooled: no igemm2 instantiation for this shape");
  }
  if (igemm2_only) return 1;
  if (a.Co <= 32) return bk64 ? launch_cfg<128, 32, 64, 4, 1>(a, out_f32, stats, st)
                              : launch_cfg<128, 32, 32, 4, 1>(a, out_f32, stats, st);
  if (a.Co <= 64) return bk64 ? launch_cfg<128, 64, 64, 2, 2>(a, out_f32, stats, st)
                              : launch_cfg<128, 64, 32, 2, 2>(a, out_f32, stats, st);
  const long long b256 = (long long)cdiv(a.M, 256) * cdiv(a.Co, 256);
  bool big = heavy && a.Co >= 256 && b256 >= 192;
  if (ftile == 1) big = false;
  if (ftile == 3 && a.Ci % 64 == 0) big = true;
  if (big) return launch_cfg<256, 256, 64, 4, 2>(a, out_f32, stats, st);
  return bk64 ? launch_cfg<128, 128, 64, 2, 2>(a, out_f32, stats, st)
              : launch_cfg<128, 128, 32, 2, 2>(a, out_f32, stats, st);
}

int check_desc(const asm_conv_desc* d) {
  ASM_REQUIRE(d != nullptr, "conv: null descriptor");
  ASM_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0,
              "conv: non-positive dimension");
  ASM_REQUIRE(d->C % 8 == 0, "conv: C=%d must be a multiple of 8", d->C);
  ASM_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride %d not supported", d->stride);
  ASM_REQUIRE(d->Ho > 0 && d->Wo > 0, "conv: bad output size");
  ASM_REQUIRE(d->pad >= 0 && d->pad < 64, "conv: bad pad");
  return ASM_OK;
}

}  // namespace

static inline int64_t img_pitch(const asm_conv_desc* d) {
  return d->x_img_pitch ? d->x_img_pitch : (int64_t)d->H * d->W * d->C;
}
static inline int row_pitch(const asm_conv_desc* d) { return d->x_row_pitch ? d->x_row_pitch : d->W * d->C; }
static inline int pix_pitch(const asm_conv_desc* d) { return d->x_pix_pitch ? d->x_pix_pitch : d->C; }

extern "C" int asm_debug_last_conv_kernel(void) { return asm_last_conv_kernel; }

extern "C" int asm_conv2d_stats_blocks(const asm_conv_desc* d) {
  if (!d) return ASM_EINVAL;
  return cdiv(d->N * d->Ho * d->Wo, STATS_BM);
}

static int fprop_impl(const asm_conv_desc* d, const void* x, const void* w, void* y, float* stats_partial,
                      const float* bn_scale, const float* bn_shift, const void* residual, int relu, void* stream) {
  if (int e = check_desc(d)) return e;
  ASM_REQUIRE(x && w && y, "conv fprop: null pointer");
  const int64_t xelems = (int64_t)d->N * img_pitch(d);
  ASM_REQUIRE(xelems * 2 < (int64_t)ASM_OOB, "conv fprop: input larger than 2 GiB");
  const int ldy = d->ldy ? d->ldy : d->K;
  ASM_REQUIRE(ldy % (d->out_f32 ? 4 : 8) == 0 && ldy >= d->K, "conv fprop: bad ldy %d", ldy);
  ASM_REQUIRE(!(stats_partial && d->out_f32), "conv fprop: fused statistics need bf16 output");
  IGemmArgs a;
  a.x = x; a.w = w; a.y = y; a.addend = residual; a.addend_mask = nullptr; a.stats = stats_partial;
  a.x_bytes = (unsigned)(xelems * 2);
  a.w_bytes = (unsigned)((int64_t)d->K * d->R * d->S * d->C * 2);
  a.M = d->N * d->Ho * d->Wo;
  a.Hi = d->H; a.Wi = d->W; a.Ci = d->C;
  a.Wo = d->Wo; a.HoWo = d->Ho * d->Wo;
  a.Co = d->K; a.ldy = ldy;
  a.R = d->R; a.S = d->S;
  a.so = d->stride; a.sd = 1; a.tsign = 1; a.pad = d->pad;
  a.pad_w = a.pad; a.wt0 = 0; a.wtr = a.S; a.wts = 1; a.y_strided = 0;
  a.y_base = a.y_img_pitch = a.y_row_pitch = a.y_pix_pitch = 0;
  a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.bn_relu = relu;
  a.pool_dy = nullptr; a.pool_k = a.pool_stride = a.pool_pad = a.pool_Hp = a.pool_Wp = a.pool_cv = a.pool_H = 0;
  a.red_y = nullptr; a.red_mask = nullptr;
  a.x_img_pitch = (int)img_pitch(d); a.x_row_pitch = row_pitch(d); a.x_pix_pitch = pix_pitch(d);
  a.w_row_pitch = d->R * d->S * d->C;
  a.m_tile0 = 0;
  return launch(a, d->out_f32 != 0, stats_partial != nullptr, (hipStream_t)stream);
}

extern "C" int asm_conv2d_fprop(const asm_conv_desc* d, const void* x, const void* w, void* y,
                                float* stats_partial, void* stream) {
  return fprop_impl(d, x, w, y, stats_partial, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int asm_conv2d_fprop_bn(const asm_conv_desc* d, const void* x, const void* w, void* y, const float* scale,
                                   const float* shift, const void* residual, int relu, void* stream) {
  ASM_REQUIRE(d && scale && shift, "conv fprop_bn: null pointer");
  ASM_REQUIRE(!d->out_f32 && d->K % 8 == 0, "conv fprop_bn: needs bf16 output and K %% 8 == 0 (K=%d)", d->K);
  ASM_REQUIRE(d->ldy == 0 || d->ldy == d->K, "conv fprop_bn: padded output rows not supported");
  return fprop_impl(d, x, w, y, nullptr, scale, shift, residual, relu ? 1 : 0, stream);
}

struct PoolAdd {
  const void* dy;
  int k, stride, pad, Hp, Wp, cv;
};
struct BnRed {      // asm_conv2d_dgrad_bnred: batch-norm backward sums of the gradient this launch writes
  const void* y;
  const uint8_t* mask;
  float* partial;
};
static int dgrad_impl(const asm_conv_desc* d, const void* dy, const void* wt, const void* addend,
                      const uint8_t* addend_mask, void* dx, void* stream, const PoolAdd* pool = nullptr,
                      const BnRed* red = nullptr);
// conv_dgrad_s2.hip: the one-launch 3x3 / stride-2 input gradient (returns 1 when the layer is not one it covers)
int asm_dgrad_s2_try(const asm_conv_desc* d, const void* dy, const void* wt, const void* addend, const uint8_t* addend_mask,
                     void* dx, void* stream);

extern "C" int asm_conv2d_dgrad(const asm_conv_desc* d, const void* dy, const void* wt, const void* addend,
                                void* dx, void* stream) {
  return dgrad_impl(d, dy, wt, addend, nullptr, dx, stream);
}

extern "C" int asm_conv2d_dgrad_masked(const asm_conv_desc* d, const void* dy, const void* wt, const void* addend,
                                       const uint8_t* addend_mask, void* dx, void* stream) {
  ASM_REQUIRE(addend && addend_mask, "conv dgrad_masked: needs the addend and its mask");
  ASM_REQUIRE(d && d->C % 8 == 0, "conv dgrad_masked: C must be a multiple of 8");
  return dgrad_impl(d, dy, wt, addend, addend_mask, dx, stream);
}

extern "C" int asm_conv2d_dgrad_pooled(const asm_conv_desc* d, const void* dy, const void* wt, const void* addend,
                                       const uint8_t* addend_mask, const void* pool_dy, int pool_k, int pool_stride,
                                       int pool_pad, int pool_Ho, int pool_Wo, int count_valid, void* dx, void* stream) {
  ASM_REQUIRE(d && pool_dy, "conv dgrad_pooled: null pointer");
  ASM_REQUIRE(d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->C % 8 == 0,
              "conv dgrad_pooled: needs a 1x1 stride-1 convolution with C %% 8 == 0");
  ASM_REQUIRE(pool_k >= 1 && (pool_stride == 1 || pool_stride == 2) && pool_pad >= 0 && pool_Ho > 0 && pool_Wo > 0,
              "conv dgrad_pooled: bad pooling geometry");
  if (pool_k > 2 * pool_stride || pool_k < pool_stride)
    ASM_FAIL(ASM_ENOTSUP, "conv dgrad_pooled: window %d / stride %d puts a pixel in more than two windows per dimension",
             pool_k, pool_stride);
  ASM_REQUIRE(!addend_mask || addend, "conv dgrad_pooled: a mask needs its addend");
  const PoolAdd pa = {pool_dy, pool_k, pool_stride, pool_pad, pool_Ho, pool_Wo, count_valid ? 1 : 0};
  return dgrad_impl(d, dy, wt, addend, addend_mask, dx, stream, &pa);
}

// does asm_conv2d_dgrad_bnred cover this layer?  1x1 / stride 1 / no padding (every kernel of that class has the instantiation),
// and the 3x3 / stride 1 / pad 1 layers that launch() sends to igemm8_kernel or igemm3_kernel
static bool dgrad_bnred_covers(const asm_conv_desc* d) {
  if (!d || d->C % 8 || d->K % 8 || d->stride != 1 || d->x_img_pitch || d->x_row_pitch || d->x_pix_pitch || d->out_f32) return false;
  if (d->R == 1 && d->S == 1 && d->pad == 0) return true;
  if (d->R != 3 || d->S != 3 || d->pad != 1 || d->Ho != d->H || d->Wo != d->W) return false;
  IGemmArgs a;            // the geometry of dgrad_impl's generic form, as far as the selection reads it
  a.M = d->N * d->H * d->W;
  a.Hi = d->Ho; a.Wi = d->Wo; a.Ci = d->K;
  a.Wo = d->W; a.HoWo = d->H * d->W;
  a.Co = d->C; a.ldy = d->C;
  a.R = 3; a.S = 3;
  a.x_bytes = (unsigned)((int64_t)d->N * d->Ho * d->Wo * d->K * 2);
  a.x_img_pitch = d->Ho * d->Wo * d->K; a.x_row_pitch = d->Wo * d->K; a.x_pix_pitch = d->K;
  a.wt0 = 0; a.wtr = 3; a.wts = 1; a.y_strided = 0; a.m_tile0 = 0;
  a.pool_dy = nullptr; a.bn_scale = nullptr;
  a.so = 1; a.sd = 1; a.tsign = -1; a.pad = -1; a.pad_w = -1;
  return bnred_3x3_supported(a);
}

// partial rows of asm_conv2d_dgrad_bnred for this layer; 0: the layer is not one it covers
extern "C" int asm_conv2d_dgrad_bnred_blocks(const asm_conv_desc* d) {
  if (!d) return ASM_EINVAL;
  if (!dgrad_bnred_covers(d)) return 0;
  return cdiv(d->N * d->H * d->W, STATS_BM);
}

extern "C" int asm_conv2d_dgrad_bnred(const asm_conv_desc* d, const void* dy, const void* wt, const void* addend,
                                      const uint8_t* addend_mask, const void* bn_y, const uint8_t* bn_relu_mask,
                                      float* partial, void* dx, void* stream) {
  ASM_REQUIRE(d && bn_y && partial, "conv dgrad_bnred: null pointer");
  ASM_REQUIRE(!addend_mask || addend, "conv dgrad_bnred: a mask needs its addend");
  if (!dgrad_bnred_covers(d))
    ASM_FAIL(ASM_ENOTSUP, "conv dgrad_bnred: 1x1 stride-1 layers, and the 3x3 stride-1 layers of the igemm8 / igemm3 kernels, with "
             "C %% 8 == 0 only (%dx%d, stride %d, C %d)", d->R, d->S, d->stride, d->C);
  const BnRed r = {bn_y, bn_relu_mask, partial};
  return dgrad_impl(d, dy, wt, addend, addend_mask, dx, stream, nullptr, &r);
}

static int dgrad_impl(const asm_conv_desc* d, const void* dy, const void* wt, const void* addend,
                      const uint8_t* addend_mask, void* dx, void* stream, const PoolAdd* pool, const BnRed* red) {
  if (int e = check_desc(d)) return e;
  ASM_REQUIRE(dy && wt && dx, "conv dgrad: null pointer");
  ASM_REQUIRE(d->K % 8 == 0, "conv dgrad: K=%d must be a multiple of 8 (pad dy)", d->K);
  ASM_REQUIRE(d->x_img_pitch == 0 && d->x_row_pitch == 0 && d->x_pix_pitch == 0 && !d->out_f32,
              "conv dgrad: custom pitches / f32 output not supported");
  const int64_t dyelems = (int64_t)d->N * d->Ho * d->Wo * d->K;
  ASM_REQUIRE(dyelems * 2 < (int64_t)ASM_OOB, "conv dgrad: dy larger than 2 GiB");
  IGemmArgs a;
  a.x = dy; a.w = wt; a.y = dx; a.addend = addend; a.addend_mask = addend_mask;
  a.stats = red ? red->partial : nullptr;
  a.red_y = red ? red->y : nullptr; a.red_mask = red ? red->mask : nullptr;
  a.x_bytes = (unsigned)(dyelems * 2);
  a.w_bytes = (unsigned)((int64_t)d->K * d->R * d->S * d->C * 2);
  a.M = d->N * d->H * d->W;
  a.Hi = d->Ho; a.Wi = d->Wo; a.Ci = d->K;
  a.Wo = d->W; a.HoWo = d->H * d->W;
  a.Co = d->C; a.ldy = d->C;
  a.R = d->R; a.S = d->S;
  a.x_img_pitch = d->Ho * d->Wo * d->K; a.x_row_pitch = d->Wo * d->K; a.x_pix_pitch = d->K;
  a.w_row_pitch = d->R * d->S * d->K;
  a.wt0 = 0; a.wtr = a.S; a.wts = 1; a.y_strided = 0; a.m_tile0 = 0;
  a.y_base = a.y_img_pitch = a.y_row_pitch = a.y_pix_pitch = 0;
  a.bn_scale = a.bn_shift = nullptr; a.bn_relu = 0;
  a.pool_dy = nullptr; a.pool_k = a.pool_stride = a.pool_pad = a.pool_Hp = a.pool_Wp = a.pool_cv = a.pool_H = 0;
  if (pool) {
    a.pool_dy = pool->dy; a.pool_k = pool->k; a.pool_stride = pool->stride; a.pool_pad = pool->pad;
    a.pool_Hp = pool->Hp; a.pool_Wp = pool->Wp; a.pool_cv = pool->cv; a.pool_H = d->H;
  }
  // Stride-2 3x3: three quarters of the (pixel, tap) pairs of the generic gather are parity misses (multiplied as
  // zeros).  Split dx into its four (h % 2, w % 2) classes instead: within a class every pixel uses the same
  // 1 / 2 / 2 / 4 taps, so each class is a dense stride-1 gather over dy with a 1x1 / 1x2 / 2x1 / 2x2 sub-filter
  //   dx(2hh+ph, 2ww+pw) = sum_{i,j} dy(hh + dh0 - i, ww + dw0 - j) . w(r0 + 2i, s0 + 2j),
  //   r0 = (ph + pad) & 1, dh0 = (ph + pad - r0) / 2   (same for columns)
  // written through the strided-output epilogue: 9/4 instead of 9 tap passes.
  const int split_ok = asm_tune().dgrad_parity;
  const bool k3 = d->R == 3 && d->S == 3, k1 = d->R == 1 && d->S == 1 && d->pad == 0;
  if (k3 && d->stride == 2 && !pool && asm_tune().igemm_mode == 0) {   // all four parity classes in one launch
    const int rc = asm_dgrad_s2_try(d, dy, wt, addend, addend_mask, dx, stream);
    if (rc != 1) return rc;
  }
  if (split_ok && d->stride == 2 && (k3 || k1) && asm_tune().igemm_mode == 0) {
    // a 1x1 / 2 projection touches only the (even, even) class: the other three are zero (or just the addend)
    bool launched = false;
    for (int cls = 0; cls < 4; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      const int Hc = (d->H - ph + 1) / 2, Wc = (d->W - pw + 1) / 2;
      const int r0 = (ph + d->pad) & 1, s0 = (pw + d->pad) & 1;
      const int Rc = (d->R - r0 + 1) / 2, Sc = (d->S - s0 + 1) / 2;   // taps r0, r0 + 2, ... (< R)
      if (Hc <= 0 || Wc <= 0 || Rc <= 0 || Sc <= 0) continue;
      IGemmArgs c = a;
      c.R = Rc; c.S = Sc;
      c.M = d->N * Hc * Wc;
      c.Wo = Wc; c.HoWo = Hc * Wc;
      c.so = 1; c.sd = 1; c.tsign = -1;
      c.pad = -((ph + d->pad - r0) / 2); c.pad_w = -((pw + d->pad - s0) / 2);
      c.wt0 = r0 * d->S + s0; c.wtr = 2 * d->S; c.wts = 2;
      c.y_strided = 1;
      c.y_base = (ph * d->W + pw) * d->C;
      c.y_img_pitch = d->H * d->W * d->C; c.y_row_pitch = 2 * d->W * d->C; c.y_pix_pitch = 2 * d->C;
      if (!launched && k1 && addend_mask)
        ASM_FAIL(ASM_ENOTSUP, "conv dgrad_masked: a masked addend is not supported for the 1x1 stride-2 input gradient");
      if (!launched && k1) {   // fill the untouched classes before the one launch that overwrites its own pixels
        const size_t bytes = (size_t)a.M * d->C * 2;
        hipError_t e = (addend && addend != dx) ? asm_fill_async(dx, addend, 0, bytes, (hipStream_t)stream)
                     : addend ? hipSuccess
                              : asm_fill_async(dx, nullptr, 0, bytes, (hipStream_t)stream);
        if (e != hipSuccess) ASM_FAIL(ASM_EHIP, "conv dgrad: fill: %s", hipGetErrorString(e));
      }
      const int rc = launch(c, false, false, (hipStream_t)stream, /*igemm2_only=*/true);
      if (rc == 1) {           // no igemm2 instantiation for this shape: nothing was launched for this class
        if (launched) ASM_FAIL(ASM_ENOTSUP, "conv dgrad: parity classes launched inconsistently");
        break;                 // (a stray fill above is overwritten by the generic kernel below)
      }
      if (rc != ASM_OK) return rc;
      launched = true;
    }
    if (launched) return ASM_OK;
  }
  // generic form: p = (h + pad - r) / stride  when divisible
  a.so = 1; a.sd = d->stride; a.tsign = -1; a.pad = -d->pad; a.pad_w = a.pad;
  return launch(a, false, red != nullptr, (hipStream_t)stream);
}
