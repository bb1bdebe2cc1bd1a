// Probe (round 4): what bounds the MFMA main loop of the 3x3 implicit-GEMM kernels?
//
// A 3x3 / stride-1 / pad-1 convolution over NHWC, written over the flattened pixel index m, is
//     Y[m][n] = sum_{t=(r,s)} sum_c X[m + r*W + s][c] * Wt[n][t][c]        (X front-padded by W+1 pixel rows)
// -- the nine taps are nine LINEAR row shifts of the same activation matrix (border pixels are a per-(pixel, tap)
// mask on top, not part of the probe).  Two ways to feed the matrix pipe:
//   gather form (what igemm2_kernel does): every K-step (tap, 64-channel chunk) LDS-DMAs a fresh BM-row activation
//     tile and a BN-row filter tile;  (BM + BN) * 128 B per step;
//   halo form: per 64-channel chunk the BM + 2W + 2 rows the nine taps touch are staged ONCE, the taps read them at
//     shifted row offsets, only the filter tile is streamed per step;  (BM + 2W + 2) * 128 + 9 * BN * 128 B per chunk.
// Variants: ring depth NS (counted vmcnt + raw s_barrier), parts of the DMA switched off (timing only: how much of
// the loop time is DMA volume), MFMA switched off (DMA rate alone).
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/conv_probe.hip -o tools/probes/bin/conv_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <type_traits>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lptr_t;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }   // 128-byte rows, 8 chunks

struct Args {
  const void* x;   // [(M + 2W + 2)][C] bf16
  const void* w;   // [N][9][C] bf16
  void* y;         // [M][N] bf16
  unsigned x_bytes, w_bytes;
  int M, N, C, W, kchunks, n_tiles_n, n_blocks;
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// n is a compile-time constant after unrolling, but not a constant expression: the switch folds to one s_waitcnt
__device__ __forceinline__ void wait_vm_dyn(const int n) {
  switch (n) {
    case 0: wait_vm<0>(); break;
    case 1: wait_vm<1>(); break;
    case 2: wait_vm<2>(); break;
    case 3: wait_vm<3>(); break;
    case 4: wait_vm<4>(); break;
    case 5: wait_vm<5>(); break;
    case 6: wait_vm<6>(); break;
    case 7: wait_vm<7>(); break;
    case 8: wait_vm<8>(); break;
    case 9: wait_vm<9>(); break;
    case 10: wait_vm<10>(); break;
    case 11: wait_vm<11>(); break;
    case 12: wait_vm<12>(); break;
    default: wait_vm<0>(); break;
  }
}

// DMASK bit 0: skip the activation DMA after the first step, bit 1: skip the filter DMA after the first step,
// bit 2: no MFMA / no fragment reads (DMA + barriers only)
template <int BM, int BN, int WGM, int WGN, int NS, int DMASK>
__global__ __launch_bounds__(64 * WGM * WGN) void gather_kernel(Args p) {
  constexpr int NT = 64 * WGM * WGN, RPP = NT / 8, XP = BM / RPP, WP = BN / RPP, ROWB = 128;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 32, TN = WTN / 32, KK = 4, NTAP = 9;
  constexpr int P = XP + WP;
  static_assert(WP >= 1 && XP >= 1 && NS >= 2 && NS - 1 <= NTAP, "cfg");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  int logical;
  {
    const int nb = p.n_blocks, q = nb >> 3, r = nb & 7;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = logical / p.n_tiles_n, tile_n = logical - tile_m * p.n_tiles_n;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);
  const int chunk = tid & 7, r0 = tid >> 3;
  const int csw = (chunk ^ swz(r0)) << 3;
  unsigned vx[XP][NTAP];
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int m = tile_m * BM + r0 + j * RPP;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) vx[j][t] = ((unsigned)(m + (t / 3) * p.W + (t % 3)) * (unsigned)p.C + (unsigned)csw) * 2u;
  }
  unsigned vw[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) vw[j] = ((unsigned)(tile_n * BN + r0 + j * RPP) * (unsigned)(9 * p.C) + (unsigned)csw) * 2u;
  const int wrow0 = wave * 8;
  const unsigned tapw = (unsigned)p.C * 2u;

  auto issue_part = [&](int stage, const int t, unsigned xso, unsigned wso, const int lo, const int hi, bool first) {
    unsigned char* xs = smem + stage * STAGE;
    unsigned char* ws = xs + BM * ROWB;
#pragma unroll
    for (int j = 0; j < XP; ++j)
      if (j >= lo && j < hi && (first || !(DMASK & 1)))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xs + (j * RPP + wrow0) * ROWB), 16, (int)vx[j][t], (int)xso, 0, 0);
#pragma unroll
    for (int j = 0; j < WP; ++j)
      if (XP + j >= lo && XP + j < hi && (first || !(DMASK & 2)))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(ws + (j * RPP + wrow0) * ROWB), 16, (int)vw[j], (int)wso, 0, 0);
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  const int l31 = lane & 31, lhi = lane >> 5;
  unsigned fwo[KK], fxo[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int ch = kk * 2 + lhi;
    const int rw_ = wn * WTN + l31, rx_ = wm * WTM + l31;
    fwo[kk] = BM * ROWB + rw_ * ROWB + ((ch ^ swz(rw_)) << 4);
    fxo[kk] = rx_ * ROWB + ((ch ^ swz(rx_)) << 4);
  }
  bf16x8 fwb[2][TN], fxb[2][TM];
  auto load_frags = [&](int stage, const int kk, const int buf) {
    if (DMASK & 4) return;
    const unsigned char* sb = smem + stage * STAGE;
#pragma unroll
    for (int a = 0; a < TN; ++a) fwb[buf][a] = *reinterpret_cast<const bf16x8*>(sb + fwo[kk] + a * 32 * ROWB);
#pragma unroll
    for (int b = 0; b < TM; ++b) fxb[buf][b] = *reinterpret_cast<const bf16x8*>(sb + fxo[kk] + b * 32 * ROWB);
  };
  auto mma = [&](const int buf) {
    if (DMASK & 4) return;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwb[buf][a], fxb[buf][b], acc[a][b], 0, 0, 0);
  };

  // prologue: steps 0 .. NS-2 in flight
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue_part(s, s, 0u, (unsigned)s * tapw, 0, P, true);
  int cur = 0;
  unsigned kcb = 0;
  bool started = false;
#pragma unroll 1
  for (int kc = 0; kc < p.kchunks; ++kc) {
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      // step k = kc * 9 + t is computed from stage cur; its DMA was issued NS-1 steps ago
      wait_vm<P*(NS - 2)>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      load_frags(cur, 0, 0);
      if (started) mma((KK - 1) & 1);           // last MFMA group of the previous step, under this step's first LDS reads
      started = true;
      // DMA of step k + NS - 1 into the stage the previous step has just released
      constexpr int G = KK - 1, PPG = (P + G - 1) / G;
      const int tn = (t + NS - 1) % NTAP;       // compile time
      const int carry = (t + NS - 1) / NTAP;
      const bool more = carry == 0 || kc + 1 < p.kchunks;
      const unsigned xso = kcb + (carry ? 128u : 0u);
      const unsigned wso = xso + (unsigned)tn * tapw;
      int nst = cur + NS - 1;
      if (nst >= NS) nst -= NS;
#pragma unroll
      for (int kk = 0; kk + 1 < KK; ++kk) {
        // the vmcnt immediates assume every step issues P pieces: the tail steps issue theirs from a clamped (valid) source
        issue_part(nst, tn, more ? xso : 0u, more ? wso : 0u, kk * PPG, (kk + 1) * PPG < P ? (kk + 1) * PPG : P, false);
        load_frags(cur, kk + 1, (kk + 1) & 1);
        mma(kk & 1);
      }
      cur = cur + 1 == NS ? 0 : cur + 1;
    }
    kcb += 128u;
  }
  mma((KK - 1) & 1);
  wait_vm<0>();

  // epilogue (probe: straight from the accumulators, 8 bytes per lane and quad)
  bf16_t* y = reinterpret_cast<bf16_t*>(p.y);
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int m = tile_m * BM + wm * WTM + b * 32 + l31;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = tile_n * BN + wn * WTN + a * 32 + 8 * g + 4 * lhi;
        u32x2 v;
        v.x = pack2bf(acc[a][b][4 * g], acc[a][b][4 * g + 1]);
        v.y = pack2bf(acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]);
        if (m < p.M) *reinterpret_cast<u32x2*>(y + (size_t)m * p.N + n) = v;
      }
    }
}

// ---- halo form ------------------------------------------------------------------------------------------------------
// LDS: halo[2][HR][128 B] (double-buffered over channel chunks), filter ring[NS][BN][128 B].
// HRMAX: compile-time bound of the halo rows (BM + 2W + 2 rounded up to the pass height).
template <int BM, int BN, int WGM, int WGN, int NS, int HRMAX, int DMASK, int NHB = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void halo_kernel(Args p) {
  constexpr int NT = 64 * WGM * WGN, RPP = NT / 8, WP = BN / RPP, ROWB = 128;
  constexpr int HP = HRMAX / RPP;                 // halo passes (pieces per wave and chunk)
  constexpr int HALO = HRMAX * ROWB, WST = BN * ROWB;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 32, TN = WTN / 32, KK = 4, NTAP = 9;
  // halo pieces of chunk kc+1 are issued during taps 0 .. HT-1 of chunk kc, HQ per tap: all of them are older than the
  // filter tile of the next chunk's first step, whatever NS
  constexpr int HT = NTAP - (NS - 1);
  constexpr int HQ = (HP + HT - 1) / HT;
  static_assert(WP >= 1 && HRMAX % RPP == 0 && NS >= 2 && HT >= 1, "cfg");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const wring = smem + NHB * HALO;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  int logical;
  {
    const int nb = p.n_blocks, q = nb >> 3, r = nb & 7;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = logical / p.n_tiles_n, tile_n = logical - tile_m * p.n_tiles_n;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);
  const int chunk = tid & 7, r0 = tid >> 3;
  const int csw = (chunk ^ swz(r0)) << 3;
  const int hrows = BM + 2 * p.W + 2;              // live halo rows (<= HRMAX)
  unsigned vh[HP];
#pragma unroll
  for (int j = 0; j < HP; ++j) {
    const int hr = r0 + j * RPP;
    vh[j] = hr < hrows ? ((unsigned)(tile_m * BM + hr) * (unsigned)p.C + (unsigned)csw) * 2u : 0x80000000u;
  }
  unsigned vw[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) vw[j] = ((unsigned)(tile_n * BN + r0 + j * RPP) * (unsigned)(9 * p.C) + (unsigned)csw) * 2u;
  const int wrow0 = wave * 8;
  const unsigned tapw = (unsigned)p.C * 2u;

  auto issue_halo = [&](int hb, unsigned xso, const int lo, const int hi, bool first) {
    unsigned char* hs = smem + hb * HALO;
#pragma unroll
    for (int j = 0; j < HP; ++j)
      if (j >= lo && j < hi && (first || !(DMASK & 1)))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(hs + (j * RPP + wrow0) * ROWB), 16, (int)vh[j], (int)xso, 0, 0);
  };
  auto issue_w = [&](int stage, unsigned wso, bool first) {
    unsigned char* ws = wring + stage * WST;
#pragma unroll
    for (int j = 0; j < WP; ++j)
      if (first || !(DMASK & 2))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(ws + (j * RPP + wrow0) * ROWB), 16, (int)vw[j], (int)wso, 0, 0);
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  const int l31 = lane & 31, lhi = lane >> 5;
  unsigned fwo[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int rw_ = wn * WTN + l31;
    fwo[kk] = rw_ * ROWB + (((kk * 2 + lhi) ^ swz(rw_)) << 4);
  }
  const int xrow0 = wm * WTM + l31;     // halo row of this lane's pixel at tap (0, 0)
  bf16x8 fwb[2][TN], fxb[2][TM];
  // activation fragment of tap t: halo row xrow0 + (t/3)*W + t%3 (+ 32 b), swizzle keyed on the halo row
  auto xbase = [&](const int t) -> unsigned {
    const int row = xrow0 + (t / 3) * p.W + (t % 3);
    return (unsigned)row * ROWB + ((unsigned)(lhi ^ swz(row)) << 4);
  };
  auto load_frags = [&](int hb, int stage, unsigned xb, const int kk, const int buf) {
    if (DMASK & 4) return;
    const unsigned char* ws = wring + stage * WST;
    const unsigned char* hs = smem + hb * HALO;
#pragma unroll
    for (int a = 0; a < TN; ++a) fwb[buf][a] = *reinterpret_cast<const bf16x8*>(ws + fwo[kk] + a * 32 * ROWB);
#pragma unroll
    for (int b = 0; b < TM; ++b) fxb[buf][b] = *reinterpret_cast<const bf16x8*>(hs + (xb ^ (unsigned)(kk << 5)) + b * 32 * ROWB);
  };
  auto mma = [&](const int buf) {
    if (DMASK & 4) return;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwb[buf][a], fxb[buf][b], acc[a][b], 0, 0, 0);
  };

  // prologue: halo of chunk 0, then filter tiles of steps 0 .. NS-2
  issue_halo(0, 0u, 0, HP, true);
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue_w(s, (unsigned)s * tapw, true);
  int cur = 0, hb = 0;
  unsigned kcb = 0;
  bool started = false;

  auto chunk_body = [&](auto HN) {
    constexpr bool has_next = decltype(HN)::value;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      // pieces issued after this step's filter tile: the (halo part + filter tile) of the NS-2 steps before this one.
      // Window steps t-(NS-2) .. t-1; those with a negative index belong to the previous chunk (taps >= HT: no halo part).
      int nwin = 0;
#pragma unroll
      for (int j = 1; j <= NS - 2; ++j) {
        const int tj = t - j;
        int hq = 0;
        if (NHB == 2 && tj >= 0 && has_next && tj < HT) {
          const int lo = tj * HQ, hi = (tj + 1) * HQ < HP ? (tj + 1) * HQ : HP;
          hq = hi > lo ? hi - lo : 0;
        }
        nwin += hq + WP;
      }
      wait_vm_dyn(nwin);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned xb = xbase(t);
      load_frags(hb, cur, xb, 0, 0);
      if (started) mma((KK - 1) & 1);
      started = true;
      // this step's issues: halo part of the next chunk first, then the filter tile of step k + NS - 1
      if (NHB == 2 && has_next && t < HT) issue_halo(hb ^ 1, kcb + 128u, t * HQ, (t + 1) * HQ < HP ? (t + 1) * HQ : HP, false);
      {
        const int tn = (t + NS - 1) % NTAP;
        const int carry = (t + NS - 1) / NTAP;
        const bool more = carry == 0 || has_next;
        int nst = cur + NS - 1;
        if (nst >= NS) nst -= NS;
        issue_w(nst, more ? kcb + (carry ? 128u : 0u) + (unsigned)tn * tapw : 0u, false);
      }
#pragma unroll
      for (int kk = 0; kk + 1 < KK; ++kk) {
        load_frags(hb, cur, xb, kk + 1, (kk + 1) & 1);
        mma(kk & 1);
      }
      cur = cur + 1 == NS ? 0 : cur + 1;
    }
    kcb += 128u;
    if (NHB == 2) hb ^= 1;
  };
#pragma unroll 1
  for (int kc = 0; kc + 1 < p.kchunks; ++kc) chunk_body(std::true_type{});
  chunk_body(std::false_type{});
  mma((KK - 1) & 1);
  wait_vm<0>();

  bf16_t* y = reinterpret_cast<bf16_t*>(p.y);
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      const int m = tile_m * BM + wm * WTM + b * 32 + l31;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = tile_n * BN + wn * WTN + a * 32 + 8 * g + 4 * lhi;
        u32x2 v;
        v.x = pack2bf(acc[a][b][4 * g], acc[a][b][4 * g + 1]);
        v.y = pack2bf(acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]);
        if (m < p.M) *reinterpret_cast<u32x2*>(y + (size_t)m * p.N + n) = v;
      }
    }
}

// ---- naive reference (small problems) ---------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((unsigned)b) << 16); }
__global__ void ref_kernel(const bf16_t* x, const bf16_t* w, float* y, int M, int N, int C, int W) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), m = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (m >= M || n >= N) return;
  float acc = 0.f;
  for (int t = 0; t < 9; ++t) {
    const bf16_t* xr = x + (size_t)(m + (t / 3) * W + (t % 3)) * C;
    const bf16_t* wr = w + ((size_t)n * 9 + t) * C;
    for (int c = 0; c < C; ++c) acc += bf2f(xr[c]) * bf2f(wr[c]);
  }
  y[(size_t)m * N + n] = acc;
}

// ---- host -------------------------------------------------------------------------------------------------------------
struct Problem {
  const char* name;
  int M, N, C, W;
};
struct Buffers {
  bf16_t *x, *w, *y;
  size_t xe, we, ye;
};
static bf16_t h_f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
static float h_bf2f(bf16_t b) {
  unsigned u = (unsigned)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

typedef void (*kern_t)(Args);
struct Variant {
  std::string name;
  kern_t kern;
  int BM, BN, NT, lds, hrmax;
};

template <int BM, int BN, int WGM, int WGN, int NS, int DMASK>
Variant mk_gather(const char* nm) {
  Variant v;
  v.name = nm;
  v.kern = gather_kernel<BM, BN, WGM, WGN, NS, DMASK>;
  v.BM = BM; v.BN = BN; v.NT = 64 * WGM * WGN;
  v.lds = NS * (BM + BN) * 128;
  v.hrmax = 0;
  return v;
}
template <int BM, int BN, int WGM, int WGN, int NS, int HRMAX, int DMASK, int NHB = 2>
Variant mk_halo(const char* nm) {
  Variant v;
  v.name = nm;
  v.kern = halo_kernel<BM, BN, WGM, WGN, NS, HRMAX, DMASK, NHB>;
  v.BM = BM; v.BN = BN; v.NT = 64 * WGM * WGN;
  v.lds = NHB * HRMAX * 128 + NS * BN * 128;
  v.hrmax = HRMAX;
  return v;
}

static float run(const Variant& v, const Problem& pb, const Buffers& b, int iters, bool check, const float* yref, double* err) {
  Args a;
  a.x = b.x; a.w = b.w; a.y = b.y;
  a.x_bytes = (unsigned)(b.xe * 2); a.w_bytes = (unsigned)(b.we * 2);
  a.M = pb.M; a.N = pb.N; a.C = pb.C; a.W = pb.W; a.kchunks = pb.C / 64;
  a.n_tiles_n = pb.N / v.BN;
  a.n_blocks = (pb.M / v.BM) * a.n_tiles_n;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.kern), hipFuncAttributeMaxDynamicSharedMemorySize, v.lds));
  hipLaunchKernelGGL(v.kern, dim3(a.n_blocks), dim3(v.NT), v.lds, 0, a);
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  if (check) {
    std::vector<bf16_t> hy(b.ye);
    CK(hipMemcpy(hy.data(), b.y, b.ye * 2, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (size_t i = 0; i < b.ye; ++i) {
      const double d = (double)h_bf2f(hy[i]) - (double)yref[i];
      num += d * d;
      den += (double)yref[i] * yref[i];
    }
    *err = sqrt(num / (den + 1e-30));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(v.kern, dim3(a.n_blocks), dim3(v.NT), v.lds, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms / iters;
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  const bool gauss = argc > 2 && !strcmp(argv[2], "gauss");   // normal operands (sum of 12 uniforms) instead of uniform ones
  std::vector<Problem> probs = {
      {"check  M4096 N256 C128 W14", 4096, 256, 128, 14},
      {"14x14  M50176 N1024 C512 W14", 50176, 1024, 512, 14},
      {"56x56  M802816 N128 C64 W56", 802816, 128, 64, 56},
      {"7x7    M12544 N512 C256 W7", 12544, 512, 256, 7},
      {"28x28  M200704 N256 C128 W28", 200704, 256, 128, 28},
      {"14x14b M50176 N512 C256 W14", 50176, 512, 256, 14},
      {"56x56d M802816 N64 C128 W56", 802816, 64, 128, 56},
      {"28x28s M200704 N128 C64 W28", 200704, 128, 64, 28},
  };
  std::vector<Variant> vars;
  // round 3 of the probe: single-chunk layers (C = 64: one halo buffer, nothing to double-buffer) at two workgroups per CU
  vars.push_back(mk_gather<128, 128, 2, 2, 2, 0>("g128 ns2"));
  vars.push_back(mk_gather<128, 64, 2, 2, 2, 0>("g128x64 ns2"));
  vars.push_back(mk_halo<128, 128, 2, 2, 2, 256, 0, 1>("h128 ns2 hr256 1buf"));     // 64 KB: two per CU
  vars.push_back(mk_halo<128, 128, 2, 2, 3, 256, 0, 1>("h128 ns3 hr256 1buf"));     // 80 KB: two per CU
  vars.push_back(mk_halo<128, 64, 2, 2, 2, 256, 0, 1>("h128x64 ns2 hr256 1buf"));   // 48 KB: three per CU
  vars.push_back(mk_halo<128, 64, 2, 2, 3, 256, 0, 1>("h128x64 ns3 hr256 1buf"));
  vars.push_back(mk_halo<128, 64, 2, 2, 2, 256, 0, 2>("h128x64 ns2 hr256 2buf"));   // 80 KB
  vars.push_back(mk_halo<128, 128, 2, 2, 2, 192, 0>("h128 ns2 hr192"));

  int dev = 0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  printf("device: %s, %d CUs, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);

  for (size_t pi = 0; pi < probs.size(); ++pi) {
    const Problem& pb = probs[pi];
    if (quick && pi > 1) break;
    Buffers b;
    b.xe = (size_t)(pb.M + 2 * pb.W + 2 + 512) * pb.C;   // + slack rows: halo passes read up to HRMAX rows
    b.we = (size_t)pb.N * 9 * pb.C;
    b.ye = (size_t)pb.M * pb.N;
    std::vector<bf16_t> hx(b.xe), hw(b.we);
    unsigned s = 12345u + (unsigned)pi;
    auto rnd = [&]() {
      s = s * 1664525u + 1013904223u;
      return ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
    };
    auto gs = [&]() {
      if (!gauss) return rnd();
      float a = 0.f;
      for (int i = 0; i < 12; ++i) a += rnd();
      return a * 0.5f;
    };
    for (auto& v : hx) v = h_f2bf(gs());
    const float wsc = 1.0f / sqrtf(9.0f * pb.C);
    for (auto& v : hw) v = h_f2bf(gs() * wsc);
    CK(hipMalloc(&b.x, b.xe * 2));
    CK(hipMalloc(&b.w, b.we * 2));
    CK(hipMalloc(&b.y, b.ye * 2));
    CK(hipMemcpy(b.x, hx.data(), b.xe * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(b.w, hw.data(), b.we * 2, hipMemcpyHostToDevice));
    const bool check = pi == 0;
    std::vector<float> yref;
    if (check) {
      float* dref;
      CK(hipMalloc(&dref, b.ye * 4));
      hipLaunchKernelGGL(ref_kernel, dim3(pb.N / 64, pb.M / 4), dim3(256), 0, 0, b.x, b.w, dref, pb.M, pb.N, pb.C, pb.W);
      CK(hipDeviceSynchronize());
      yref.resize(b.ye);
      CK(hipMemcpy(yref.data(), dref, b.ye * 4, hipMemcpyDeviceToHost));
      CK(hipFree(dref));
    }
    const double gflop = 2.0 * pb.M * pb.N * 9.0 * pb.C / 1e9;
    printf("\n== %s  (%.1f GFLOP) ==\n", pb.name, gflop);
    const int long_iters = argc > 3 ? atoi(argv[3]) : 5;
    const int rounds = check ? 1 : 3, iters = check ? 2 : long_iters;
    std::vector<float> best(vars.size(), 1e30f);
    std::vector<double> errs(vars.size(), -1.0);
    for (int r = 0; r < rounds; ++r)
      for (size_t vi = 0; vi < vars.size(); ++vi) {
        const Variant& v = vars[vi];
        if (pb.M % v.BM || pb.N % v.BN) continue;
        if (v.hrmax && v.BM + 2 * pb.W + 2 > v.hrmax) continue;
        CK(hipMemset(b.y, 0, b.ye * 2));
        double err = -1;
        const float ms = run(v, pb, b, iters, check && r == 0, yref.data(), &err);
        if (check && r == 0) errs[vi] = err;
        if (ms < best[vi]) best[vi] = ms;
      }
    for (size_t vi = 0; vi < vars.size(); ++vi) {
      if (best[vi] > 1e29f) continue;
      printf("  %-20s %9.1f us  %7.0f TF/s  lds %6d", vars[vi].name.c_str(), best[vi] * 1e3, gflop / best[vi], vars[vi].lds);
      if (errs[vi] >= 0) printf("  rel-l2 %.2e", errs[vi]);
      printf("\n");
    }
    CK(hipFree(b.x));
    CK(hipFree(b.w));
    CK(hipFree(b.y));
  }
  return 0;
}
