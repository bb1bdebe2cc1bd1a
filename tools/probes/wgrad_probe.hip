// Probe (end of round 4, for round 5): what holds the 3x3 weight gradient at 41 % MFMA busy, and a form that does better.
//
// Problem: the workload's heaviest layer, 14 x 14 x 512 -> 1024 at batch 256 (474 GFLOP; library kernel: 511 us),
//     dW[n][t][c] = sum_m dY[m][n] * X[m + r W + s][c] * valid(m, t)       t = (r, s), X front-padded by W + 1 pixel rows
// i.e. a GEMM [N x M] . [M x 9C] whose right operand is nine linear row shifts of one matrix, the convolution's zero padding
// a per-(pixel, tap) predicate inside the reduction.  Every variant writes fp32 slabs [split][N][9C] like the library kernel
// and is checked against an fp64 reference on 96 sampled outputs.
//
//   wgrad_dma_kernel    the library's 256 x 256 / 8-wave tile, operands staged by LDS-DMA (swizzle on the source side) through
//                       a ring of NS stages of WPX pixels: 557 us (2 x 64), 626 (4 x 32), 814 (8 x 16) -- no better than
//                       register staging; a deeper prefetch is not what is missing
//   wgrad_dma4_kernel   the same tile on four waves of 128 x 128 (half the fragment bytes per MFMA): 577 us -- nor LDS reads
//   wgrad_halo9_kernel  resident rows: 128 (n) x [9 taps x 64 channels] per workgroup, the x rows of a step staged ONCE for
//                       all nine taps (28 KB per step instead of 64), one accumulator tile per tap: 428 us unmasked;
//                       MASK: the padding as four factor masks (row ok for r = 0 / 2, column ok for s = 0 / 2) built per step
//                       into LDS and ANDed onto the dy fragment: 469 - 484 us
//   wgrad_halo9b_kernel the same tile with the waves cut 2 (n) x 2 (channels) x 2 (taps 0-4 / 5-8): 414 us unmasked,
//                       463 us with the padding = 1 024 TFLOP/s, 9.5 % under the library kernel
// (borders are left out of the first two, as in conv_probe.hip; times on one MI355X, 4 pixel splits = one round of 256
// workgroups for the resident-row forms, 7 splits for the 256 x 256 tiles)
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/wgrad_probe.hip -o tools/probes/bin/wgrad_probe
// run:   tools/probes/bin/wgrad_probe [splits] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <type_traits>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
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

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ bf16x4 ds_read_tr(const unsigned char* p) {
  typedef __attribute__((ext_vector_type(4))) short s4;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(const_cast<unsigned char*>(p)));
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Args {
  const void* x;    // [(M + 2W + 2)][C] bf16
  const void* dy;   // [M][N] bf16
  float* out;       // [splits][N][9C] fp32
  unsigned x_bytes, dy_bytes;
  int M, N, C, W, cols, tiles_n, tiles_c, splits, m_per_split, H, masked;
};

constexpr int ROWB = 512;                 // 256 bf16 per tile row
constexpr unsigned OOB = 0x7fffff00u;     // beyond any buffer: the DMA writes zeros

__device__ __forceinline__ int tr_swz(int row) { return (row & 3) << 2; }   // conv_wgrad.hip tr_swz<512>

// WPX pixels per stage, NS stages, loads of step k + NS - 1 issued while step k is multiplied
template <int WPX, int NS>
__global__ __launch_bounds__(512) void wgrad_dma_kernel(Args p) {
  constexpr int YT = WPX * ROWB, STAGE = 2 * YT;
  constexpr int PASSES = WPX / 16;          // 8 waves x 2 rows per DMA instruction
  constexpr int PER_STEP = 2 * PASSES;      // DMA instructions per lane and step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wc = wave & 1;
  const int nbase = wn * 64, cbase = wc * 128;

  int bid;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7;
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles = p.tiles_n * p.tiles_c;
  const int split = bid / tiles;
  bid -= split * tiles;
  const int tile_n = bid / p.tiles_c, tile_c = bid - tile_n * p.tiles_c;

  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);

  // DMA geometry: in pass j this lane fills row (16 j + 2 wave + (lane >> 5)), 16-byte slot (lane & 31) of the tile; the slot
  // holds source chunk slot ^ swz(row), and row & 3 does not depend on j
  const int rloc = 2 * wave + (lane >> 5);
  const int slot = lane & 31;
  const int csrc = slot ^ tr_swz(rloc);
  const int n0 = tile_n * 256 + csrc * 8;
  const int j0 = tile_c * 256 + csrc * 8;
  const bool col_ok = j0 < p.cols;
  const int tap = col_ok ? j0 / p.C : 0;
  const int tap_c = j0 - tap * p.C;
  const int shift = (tap / 3) * p.W + (tap % 3);
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int steps = (m_end - m_begin + WPX - 1) / WPX;

  auto issue = [&](int step) {
    unsigned char* ys = smem + (step % NS) * STAGE;
    unsigned char* xs = ys + YT;
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
      const int m = m_begin + step * WPX + 16 * j + rloc;
      const bool ok = m < m_end && step < steps;
      const unsigned offy = ((unsigned)m * (unsigned)p.N + (unsigned)n0) * 2u;
      const unsigned offx = ((unsigned)(m + shift) * (unsigned)p.C + (unsigned)tap_c) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lptr_t)(ys + (16 * j + 2 * wave) * ROWB), 16, (int)(ok ? offy : OOB), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xs + (16 * j + 2 * wave) * ROWB), 16,
                                               (int)((ok && col_ok) ? offx : OOB), 0, 0, 0);
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  const int t16 = lane & 15, g = lane >> 4;
  const int colsel = (g & 1) * 16, pgrp = (g >> 1) * 8, trow = t16 >> 2, tcol = (t16 & 3) * 4;

  auto compute = [&](int step) {
    const unsigned char* ys = smem + (step % NS) * STAGE;
    const unsigned char* xs = ys + YT;
#pragma unroll
    for (int kk = 0; kk < WPX / 16; ++kk) {
      const int row = kk * 16 + pgrp + trow, row2 = row + 4;
      bf16x8 fy[2], fx[4];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int cy = nbase + a * 32 + colsel + tcol;
        const int o = ((cy & 4) << 1);
        const bf16x4 y0 = ds_read_tr(ys + row * ROWB + ((((cy >> 3) ^ tr_swz(row)) << 4) | o));
        const bf16x4 y1 = ds_read_tr(ys + row2 * ROWB + ((((cy >> 3) ^ tr_swz(row2)) << 4) | o));
        fy[a] = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int cx = cbase + b * 32 + colsel + tcol;
        const int o = ((cx & 4) << 1);
        const bf16x4 x0 = ds_read_tr(xs + row * ROWB + ((((cx >> 3) ^ tr_swz(row)) << 4) | o));
        const bf16x4 x1 = ds_read_tr(xs + row2 * ROWB + ((((cx >> 3) ^ tr_swz(row2)) << 4) | o));
        fx[b] = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[a], fx[b], acc[a][b], 0, 0, 0);
    }
  };

  // prologue: steps 0 .. NS - 2 in flight
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);
#pragma unroll 1
  for (int step = 0; step < steps; ++step) {
    wait_vm<(NS - 2) * PER_STEP>();      // this lane's part of step `step` has landed ...
    __syncthreads();                     // ... so has everybody's; and everybody is through with step - 1's stage
    issue(step + NS - 1);                // refill the stage step - 1 used
    compute(step);
  }

  float* out = p.out + (size_t)split * p.N * p.cols;
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = tile_c * 256 + cbase + b * 32 + l31;
      if (col < p.cols) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = tile_n * 256 + nbase + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (n < p.N) out[(size_t)n * p.cols + col] = acc[a][b][r];
        }
      }
    }
}

// The same tile on FOUR waves, each 128 (n) x 128 (columns): 256 accumulator registers per lane (one wave per SIMD may hold
// 512), half the fragment bytes per MFMA of the 64 x 128 form; fragments of kk + 1 are read while kk is multiplied.
template <int WPX, int NS>
__global__ __launch_bounds__(256) void wgrad_dma4_kernel(Args p) {
  constexpr int YT = WPX * ROWB, STAGE = 2 * YT;
  constexpr int PASSES = WPX / 8;           // 4 waves x 2 rows per DMA instruction
  constexpr int PER_STEP = 2 * PASSES;
  constexpr int KK = WPX / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nbase = (wave >> 1) * 128, cbase = (wave & 1) * 128;
  int bid;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7;
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles = p.tiles_n * p.tiles_c;
  const int split = bid / tiles;
  bid -= split * tiles;
  const int tile_n = bid / p.tiles_c, tile_c = bid - tile_n * p.tiles_c;
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
  const int rloc = 2 * wave + (lane >> 5);
  const int slot = lane & 31;
  const int csrc = slot ^ tr_swz(rloc);
  const int n0 = tile_n * 256 + csrc * 8;
  const int j0 = tile_c * 256 + csrc * 8;
  const bool col_ok = j0 < p.cols;
  const int tap = col_ok ? j0 / p.C : 0;
  const int tap_c = j0 - tap * p.C;
  const int shift = (tap / 3) * p.W + (tap % 3);
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int steps = (m_end - m_begin + WPX - 1) / WPX;

  auto issue = [&](int step) {
    unsigned char* ys = smem + (step % NS) * STAGE;
    unsigned char* xs = ys + YT;
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
      const int m = m_begin + step * WPX + 8 * j + rloc;
      const bool ok = m < m_end && step < steps;
      const unsigned offy = ((unsigned)m * (unsigned)p.N + (unsigned)n0) * 2u;
      const unsigned offx = ((unsigned)(m + shift) * (unsigned)p.C + (unsigned)tap_c) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lptr_t)(ys + (8 * j + 2 * wave) * ROWB), 16, (int)(ok ? offy : OOB), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xs + (8 * j + 2 * wave) * ROWB), 16,
                                               (int)((ok && col_ok) ? offx : OOB), 0, 0, 0);
    }
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  const int t16 = lane & 15, g = lane >> 4;
  const int colsel = (g & 1) * 16, pgrp = (g >> 1) * 8, trow = t16 >> 2, tcol = (t16 & 3) * 4;

  bf16x8 fy[2][4], fx[2][4];
  auto frags = [&](int step, const int kk, const int buf) {
    const unsigned char* ys = smem + (step % NS) * STAGE;
    const unsigned char* xs = ys + YT;
    const int row = kk * 16 + pgrp + trow, row2 = row + 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int cy = nbase + a * 32 + colsel + tcol;
      const int o = ((cy & 4) << 1);
      const bf16x4 y0 = ds_read_tr(ys + row * ROWB + ((((cy >> 3) ^ tr_swz(row)) << 4) | o));
      const bf16x4 y1 = ds_read_tr(ys + row2 * ROWB + ((((cy >> 3) ^ tr_swz(row2)) << 4) | o));
      fy[buf][a] = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int cx = cbase + b * 32 + colsel + tcol;
      const int o = ((cx & 4) << 1);
      const bf16x4 x0 = ds_read_tr(xs + row * ROWB + ((((cx >> 3) ^ tr_swz(row)) << 4) | o));
      const bf16x4 x1 = ds_read_tr(xs + row2 * ROWB + ((((cx >> 3) ^ tr_swz(row2)) << 4) | o));
      fx[buf][b] = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto mma = [&](const int buf) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[buf][a], fx[buf][b], acc[a][b], 0, 0, 0);
  };

#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);
#pragma unroll 1
  for (int step = 0; step < steps; ++step) {
    wait_vm<(NS - 2) * PER_STEP>();
    __syncthreads();
    issue(step + NS - 1);
    frags(step, 0, 0);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (kk + 1 < KK) frags(step, kk + 1, (kk + 1) & 1);
      mma(kk & 1);
    }
  }

  float* out = p.out + (size_t)split * p.N * p.cols;
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = tile_c * 256 + cbase + b * 32 + l31;
      if (col < p.cols) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = tile_n * 256 + nbase + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (n < p.N) out[(size_t)n * p.cols + col] = acc[a][b][r];
        }
      }
    }
}

// Resident-row form: a workgroup owns 128 (n) x [9 taps x 64 channels] over a pixel range.  Per 64-pixel step it stages the
// dy rows (64 x 128 ch) and ONCE the 64 + 2W + 2 x rows the nine taps touch (64 ch); tap t's fragment is read at row offset
// r W + s.  28 KB staged per 9.4 MFLOP step instead of 64 KB per 8.4: 2.6 x less volume through L2 -> LDS.  Waves: 4 (n: 32
// each) x 2 (channel tile of 32): 9 accumulator tiles per wave (one per tap).
// WC: the map width as a compile-time constant (0 = p.W at run time); ROLL: the four 16-pixel sub-steps as a rolled loop
// MASK: the zero padding of the convolution: a per-(pixel, tap) validity mask, built per step into LDS (one bf16-wide word per
// pixel and tap) and ANDed onto the dy fragment of that tap (8 consecutive pixels per lane = one 16-byte broadcast read)
template <int NS, int WC = 0, bool ROLL = false, bool MASK = false>
__global__ __launch_bounds__(512) void wgrad_halo9_kernel(Args p) {
  const int Wd = WC ? WC : p.W;
  constexpr int WPX = 64, XROWS = 96, YRB = 256, XRB = 128;
  constexpr int YT = WPX * YRB, XT = XROWS * XRB, MT = MASK ? 4 * WPX * 2 : 0, STAGE = YT + XT + MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wc = wave & 1;
  int bid;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7;
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_n = p.N / 128, cchunks = p.C / 64;
  const int tiles = tiles_n * cchunks;
  const int split = bid / tiles;
  bid -= split * tiles;
  const int tile_n = bid / cchunks, cchunk = bid - tile_n * cchunks;
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int steps = (m_end - m_begin + WPX - 1) / WPX;
  const int xrows_total = p.M + 2 * Wd + 2;

  // dy: 256-byte rows, 4 rows per DMA instruction, 16 instructions per step: wave w issues rows 4 (w + 8 j) .., j = 0, 1
  const int yr = lane >> 4, yslot = lane & 15;
  // x: 128-byte rows, 8 rows per instruction, 12 instructions: wave w issues rows 8 (w + 8 j) .., j = 0 (, 1 for w < 4)
  const int xr = lane >> 3, xslot = lane & 7;
  auto issue = [&](int step) {
    unsigned char* ys = smem + (step % NS) * STAGE;
    unsigned char* xs = ys + YT;
    const int m0 = m_begin + step * WPX;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 4 * (wave + 8 * j) + yr;
      const int m = m0 + row;
      const int cs = yslot ^ ((row & 3) << 2);
      const bool ok = m < m_end && step < steps;
      const unsigned off = ((unsigned)m * (unsigned)p.N + (unsigned)(tile_n * 128 + cs * 8)) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lptr_t)(ys + 4 * (wave + 8 * j) * YRB), 16, (int)(ok ? off : OOB), 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (wave + 8 * j < XROWS / 8) {
        const int row = 8 * (wave + 8 * j) + xr;
        const int xm = m0 + row;                                  // row of the front-padded x
        const int cs = xslot ^ (((row >> 1) & 1) << 2);
        const bool ok = xm < xrows_total && step < steps;
        const unsigned off = ((unsigned)xm * (unsigned)p.C + (unsigned)(cchunk * 64 + cs * 8)) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xs + 8 * (wave + 8 * j) * XRB), 16, (int)(ok ? off : OOB), 0, 0, 0);
      }
    }
    if (MASK) {      // valid(m, (r, s)) = rowok_r(h) & colok_s(w): four masks per pixel -- r = 0, r = 2, s = 0, s = 2 (H, W constants)
      unsigned short* ms = reinterpret_cast<unsigned short*>(xs + XT);
      if (tid < 4 * WPX) {
        constexpr int H_ = WC ? WC : 1;          // square maps in this probe
        const int which = tid / WPX, px = tid - which * WPX;
        const int m = m0 + px;
        const int q = m / H_, w = m - q * H_, h = q % H_;
        const bool v = which == 0 ? h >= 1 : which == 1 ? h <= H_ - 2 : which == 2 ? w >= 1 : w <= H_ - 2;
        ms[tid] = v ? 0xffffu : 0u;
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const int t16 = lane & 15, g = lane >> 4;
  const int colsel = (g & 1) * 16, pgrp = (g >> 1) * 8, trow = t16 >> 2, tcol = (t16 & 3) * 4;
  const int cy = wn * 32 + colsel + tcol, oy = ((cy & 4) << 1);
  const int cx = wc * 32 + colsel + tcol, ox = ((cx & 4) << 1);

  // x fragment addresses with W known at compile time: (row + sh) * 128 + (c ^ (((row + sh) & 2) << 5)); row & 3 == trow, so the
  // XOR term takes one of four lane values selected by sh & 3 and everything else is an immediate offset of the read
  const int cxb = ((cx >> 3) << 4) | ox;
  int xl[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) xl[q] = (pgrp + trow) * XRB + (cxb ^ (((trow + q) & 2) << 5));
  const int yl = (pgrp + trow) * YRB + ((((cy >> 3) ^ (trow << 2)) << 4) | oy);      // (row & 3) == (row2 & 3) == trow

  auto compute = [&](int step) {
    const unsigned char* ys = smem + (step % NS) * STAGE;
    const unsigned char* xs = ys + YT;
#pragma unroll(ROLL ? 1 : 4)
    for (int kk = 0; kk < WPX / 16; ++kk) {
      bf16x8 fy;
      if (WC) {
        const bf16x4 y0 = ds_read_tr(ys + yl + kk * 16 * YRB);
        const bf16x4 y1 = ds_read_tr(ys + yl + kk * 16 * YRB + 4 * YRB);
        fy = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
      } else {
        const int row = kk * 16 + pgrp + trow, row2 = row + 4;
        const bf16x4 y0 = ds_read_tr(ys + row * YRB + ((((cy >> 3) ^ ((row & 3) << 2)) << 4) | oy));
        const bf16x4 y1 = ds_read_tr(ys + row2 * YRB + ((((cy >> 3) ^ ((row2 & 3) << 2)) << 4) | oy));
        fy = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      u32x4 fyr0, fyr2, mc0, mc2;
      if (MASK) {
        const unsigned char* mb = xs + XT + (pgrp + kk * 16) * 2;
        fyr0 = __builtin_bit_cast(u32x4, fy) & *reinterpret_cast<const u32x4*>(mb);
        fyr2 = __builtin_bit_cast(u32x4, fy) & *reinterpret_cast<const u32x4*>(mb + WPX * 2);
        mc0 = *reinterpret_cast<const u32x4*>(mb + 2 * WPX * 2);
        mc2 = *reinterpret_cast<const u32x4*>(mb + 3 * WPX * 2);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int sh = (t / 3) * Wd + (t % 3);
        bf16x4 x0, x1;
        if (WC) {
          constexpr int W_ = WC ? WC : 1;
          const int shc = (t / 3) * W_ + (t % 3);
          x0 = ds_read_tr(xs + xl[shc & 3] + (kk * 16 + shc) * XRB);
          x1 = ds_read_tr(xs + xl[shc & 3] + (kk * 16 + shc + 4) * XRB);
        } else {
          const int row = kk * 16 + pgrp + trow;
          const int ra = row + sh, rb = row + 4 + sh;
          x0 = ds_read_tr(xs + ra * XRB + ((((cx >> 3) ^ (((ra >> 1) & 1) << 2)) << 4) | ox));
          x1 = ds_read_tr(xs + rb * XRB + ((((cx >> 3) ^ (((rb >> 1) & 1) << 2)) << 4) | ox));
        }
        const bf16x8 fx = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
        bf16x8 fyt = fy;
        if (MASK) {
          u32x4 v = t / 3 == 0 ? fyr0 : t / 3 == 2 ? fyr2 : __builtin_bit_cast(u32x4, fy);
          if (t % 3 == 0) v &= mc0;
          if (t % 3 == 2) v &= mc2;
          fyt = __builtin_bit_cast(bf16x8, v);
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fyt, fx, acc[t], 0, 0, 0);
      }
    }
  };

#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);
#pragma unroll 1
  for (int step = 0; step < steps; ++step) {
    if (wave < 4) wait_vm<(NS - 2) * 4>();      // waves 0 - 3 issue 4 DMA instructions per step, waves 4 - 7 three
    else wait_vm<(NS - 2) * 3>();
    __syncthreads();
    issue(step + NS - 1);
    compute(step);
  }

  float* out = p.out + (size_t)split * p.N * p.cols;
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int col = t * p.C + cchunk * 64 + wc * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = tile_n * 128 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      out[(size_t)n * p.cols + col] = acc[t][r];
    }
  }
}

// The same workgroup tile (128 n x 9 taps x 64 channels) with the waves cut differently: 2 (n: 64 each) x 2 (channel tile of
// 32) x 2 (taps 0 - 4 / 5 - 8): 10 or 8 accumulator tiles per wave, and per 16-pixel sub-step 2 dy + 5 (4) x fragments for 10 (8)
// MFMAs instead of 1 + 9 for 9 -- 35 % fewer LDS fragment bytes per MFMA.  Waves w and w + 4 (the two tap groups) share a SIMD.
template <int NS, int WC, bool MASK>
__global__ __launch_bounds__(512) void wgrad_halo9b_kernel(Args p) {
  constexpr int WPX = 64, XROWS = 96, YRB = 256, XRB = 128;
  constexpr int YT = WPX * YRB, XT = XROWS * XRB, MT = MASK ? 4 * WPX * 2 : 0, STAGE = YT + XT + MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tg = wave >> 2, ng = (wave >> 1) & 1, cg = wave & 1;
  int bid;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7;
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_n = p.N / 128, cchunks = p.C / 64;
  const int tiles = tiles_n * cchunks;
  const int split = bid / tiles;
  bid -= split * tiles;
  const int tile_n = bid / cchunks, cchunk = bid - tile_n * cchunks;
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int steps = (m_end - m_begin + WPX - 1) / WPX;
  const int xrows_total = p.M + 2 * WC + 2;
  const int yr = lane >> 4, yslot = lane & 15;
  const int xr = lane >> 3, xslot = lane & 7;
  auto issue = [&](int step) {
    unsigned char* ys = smem + (step % NS) * STAGE;
    unsigned char* xs = ys + YT;
    const int m0 = m_begin + step * WPX;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 4 * (wave + 8 * j) + yr;
      const int m = m0 + row;
      const int cs = yslot ^ ((row & 3) << 2);
      const bool ok = m < m_end && step < steps;
      const unsigned off = ((unsigned)m * (unsigned)p.N + (unsigned)(tile_n * 128 + cs * 8)) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lptr_t)(ys + 4 * (wave + 8 * j) * YRB), 16, (int)(ok ? off : OOB), 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (wave + 8 * j < XROWS / 8) {
        const int row = 8 * (wave + 8 * j) + xr;
        const int xm = m0 + row;
        const int cs = xslot ^ (((row >> 1) & 1) << 2);
        const bool ok = xm < xrows_total && step < steps;
        const unsigned off = ((unsigned)xm * (unsigned)p.C + (unsigned)(cchunk * 64 + cs * 8)) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xs + 8 * (wave + 8 * j) * XRB), 16, (int)(ok ? off : OOB), 0, 0, 0);
      }
    }
    if (MASK) {
      unsigned short* ms = reinterpret_cast<unsigned short*>(xs + XT);
      if (tid < 4 * WPX) {
        const int which = tid / WPX, px = tid - which * WPX;
        const int m = m0 + px;
        const int q = m / WC, w = m - q * WC, h = q % WC;
        const bool v = which == 0 ? h >= 1 : which == 1 ? h <= WC - 2 : which == 2 ? w >= 1 : w <= WC - 2;
        ms[tid] = v ? 0xffffu : 0u;
      }
    }
  };

  f32x16 acc[2][5];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][t][i] = 0.f;
  const int t16 = lane & 15, g = lane >> 4;
  const int colsel = (g & 1) * 16, pgrp = (g >> 1) * 8, trow = t16 >> 2, tcol = (t16 & 3) * 4;
  const int cx = cg * 32 + colsel + tcol, ox = ((cx & 4) << 1);
  const int cxb = ((cx >> 3) << 4) | ox;
  int xl[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) xl[q] = (pgrp + trow) * XRB + (cxb ^ (((trow + q) & 2) << 5));
  int yl[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int cy = ng * 64 + a * 32 + colsel + tcol;
    yl[a] = (pgrp + trow) * YRB + ((((cy >> 3) ^ (trow << 2)) << 4) | ((cy & 4) << 1));
  }

  auto taps = [&](const unsigned char* ys, const unsigned char* xs, auto T0, auto NT) {
    constexpr int t0 = decltype(T0)::value, nt = decltype(NT)::value;
#pragma unroll
    for (int kk = 0; kk < WPX / 16; ++kk) {
      u32x4 fy[2], fyr0[2], fyr2[2], mc0, mc2;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const bf16x4 y0 = ds_read_tr(ys + yl[a] + kk * 16 * YRB);
        const bf16x4 y1 = ds_read_tr(ys + yl[a] + kk * 16 * YRB + 4 * YRB);
        fy[a] = __builtin_bit_cast(u32x4, __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
      if (MASK) {
        const unsigned char* mb = xs + XT + (pgrp + kk * 16) * 2;
        const u32x4 r0 = *reinterpret_cast<const u32x4*>(mb), r2 = *reinterpret_cast<const u32x4*>(mb + WPX * 2);
        mc0 = *reinterpret_cast<const u32x4*>(mb + 2 * WPX * 2);
        mc2 = *reinterpret_cast<const u32x4*>(mb + 3 * WPX * 2);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          fyr0[a] = fy[a] & r0;
          fyr2[a] = fy[a] & r2;
        }
      }
#pragma unroll
      for (int i = 0; i < nt; ++i) {
        constexpr int W_ = WC;
        const int t = t0 + i;
        const int shc = (t / 3) * W_ + (t % 3);
        const bf16x4 x0 = ds_read_tr(xs + xl[shc & 3] + (kk * 16 + shc) * XRB);
        const bf16x4 x1 = ds_read_tr(xs + xl[shc & 3] + (kk * 16 + shc + 4) * XRB);
        const bf16x8 fx = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          u32x4 v = fy[a];
          if (MASK) {
            v = t / 3 == 0 ? fyr0[a] : t / 3 == 2 ? fyr2[a] : fy[a];
            if (t % 3 == 0) v &= mc0;
            if (t % 3 == 2) v &= mc2;
          }
          acc[a][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v), fx, acc[a][i], 0, 0, 0);
        }
      }
    }
  };

#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);
#pragma unroll 1
  for (int step = 0; step < steps; ++step) {
    if (wave < 4) wait_vm<(NS - 2) * 4>();
    else wait_vm<(NS - 2) * 3>();
    __syncthreads();
    issue(step + NS - 1);
    const unsigned char* ys = smem + (step % NS) * STAGE;
    if (tg == 0) taps(ys, ys + YT, std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
    else taps(ys, ys + YT, std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{});
  }

  float* out = p.out + (size_t)split * p.N * p.cols;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int nt = tg ? 4 : 5, t0 = tg ? 5 : 0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i < nt) {
        const int col = (t0 + i) * p.C + cchunk * 64 + cg * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = tile_n * 128 + ng * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          out[(size_t)n * p.cols + col] = acc[a][i][r];
        }
      }
    }
}

// reference for sampled (n, col) pairs: fp64 sum over all pixels
__global__ void ref_kernel(const bf16_t* x, const bf16_t* dy, const int* samples, int ns, int M, int N, int C, int W, int H,
                           double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const int n = samples[2 * i], col = samples[2 * i + 1];
  const int tap = col / C, c = col - tap * C;
  const int shift = (tap / 3) * W + (tap % 3);
  double s = 0.0;
  for (int m = 0; m < M; ++m) {
    if (H) {
      const int q = m / W, w = m - q * W, h = q % H;
      if ((unsigned)(h + tap / 3 - 1) >= (unsigned)H || (unsigned)(w + tap % 3 - 1) >= (unsigned)W) continue;
    }
    const float a = __uint_as_float(((unsigned)dy[(size_t)m * N + n]) << 16);
    const float b = __uint_as_float(((unsigned)x[(size_t)(m + shift) * C + c]) << 16);
    s += (double)a * (double)b;
  }
  out[i] = s;
}

__global__ void fill_kernel(bf16_t* p, size_t n, unsigned seed) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u + seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  const float f = ((float)(h & 0xffff) / 65536.0f - 0.5f);      // uniform in [-0.5, 0.5)
  __bf16 b = (__bf16)f;
  p[i] = __builtin_bit_cast(unsigned short, b);
}

static void run(const char* name, void (*kernel)(Args), int threads, int lds, Args a, int iters, const int* d_samples, int ns,
                const double* h_ref, int grid_tiles = 0) {
  CK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipMemset(a.out, 0xff, (size_t)a.splits * a.N * a.cols * sizeof(float)));
  const int grid = (grid_tiles ? grid_tiles : a.tiles_n * a.tiles_c) * a.splits;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, 0, a);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // check the samples: sum of the slabs against the fp64 reference
  std::vector<int> hs(2 * ns);
  CK(hipMemcpy(hs.data(), d_samples, sizeof(int) * 2 * ns, hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (int i = 0; i < ns; ++i) {
    double s = 0.0;
    for (int k = 0; k < a.splits; ++k) {
      float v;
      CK(hipMemcpy(&v, a.out + ((size_t)k * a.N + hs[2 * i]) * a.cols + hs[2 * i + 1], sizeof(float), hipMemcpyDeviceToHost));
      s += v;
    }
    const double err = fabs(s - h_ref[i]) / (fabs(h_ref[i]) + 1.0);
    if (err > worst) worst = err;
  }
  const double us = 1000.0 * ms / iters;
  const double flops = 2.0 * a.M * (double)a.N * a.cols;
  printf("%-28s grid %4d  LDS %3d KB  %8.1f us  %7.1f TFLOP/s  worst rel err %.2e %s\n", name, grid, lds >> 10, us,
         flops / us * 1e-6, worst, worst < 2e-3 ? "ok" : "WRONG");
}

int main(int argc, char** argv) {
  const int splits = argc > 1 ? atoi(argv[1]) : 7;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  const int NB = 256, H = 14, W = 14, C = 512, N = 1024;
  const int M = NB * H * W;
  Args a = {};
  a.M = M; a.N = N; a.C = C; a.W = W; a.cols = 9 * C;
  a.tiles_n = N / 256; a.tiles_c = (a.cols + 255) / 256; a.splits = splits;
  a.m_per_split = ((M + splits - 1) / splits + 63) / 64 * 64;
  const size_t xn = (size_t)(M + 2 * W + 2) * C, yn = (size_t)M * N;
  bf16_t *x, *dy;
  CK(hipMalloc(&x, xn * 2));
  CK(hipMalloc(&dy, yn * 2));
  CK(hipMalloc(&a.out, (size_t)splits * N * a.cols * sizeof(float)));
  a.x = x; a.dy = dy; a.x_bytes = (unsigned)(xn * 2); a.dy_bytes = (unsigned)(yn * 2);
  fill_kernel<<<(unsigned)((xn + 255) / 256), 256>>>(x, xn, 1u);
  fill_kernel<<<(unsigned)((yn + 255) / 256), 256>>>(dy, yn, 7u);
  const int ns = 96;
  std::vector<int> hs(2 * ns);
  for (int i = 0; i < ns; ++i) {
    hs[2 * i] = (i * 131 + 17) % N;
    hs[2 * i + 1] = (i * 977 + (i % 9) * C + 5) % a.cols;
  }
  int* d_samples;
  double* d_ref;
  CK(hipMalloc(&d_samples, sizeof(int) * 2 * ns));
  CK(hipMalloc(&d_ref, sizeof(double) * ns));
  CK(hipMemcpy(d_samples, hs.data(), sizeof(int) * 2 * ns, hipMemcpyHostToDevice));
  ref_kernel<<<(ns + 63) / 64, 64>>>(x, dy, d_samples, ns, M, N, C, W, 0, d_ref);
  std::vector<double> href(ns);
  CK(hipMemcpy(href.data(), d_ref, sizeof(double) * ns, hipMemcpyDeviceToHost));
  printf("wgrad 3x3 N%d %dx%dx%d -> %d, 256 x 256 tiles, %d x %d tiles x %d splits, %d pixels per split\n", NB, H, W, C, N,
         a.tiles_n, a.tiles_c, splits, a.m_per_split);
#define RUN(name, K, T, WPX, NS) run(name, K<WPX, NS>, T, NS * 2 * WPX * ROWB, a, iters, d_samples, ns, href.data())
  RUN("8 waves, 2 x 64-pixel stages", wgrad_dma_kernel, 512, 64, 2);
  RUN("4 waves, 2 x 64-pixel stages", wgrad_dma4_kernel, 256, 64, 2);
  run("resident rows 128 x 9 x 64, 3 st", wgrad_halo9_kernel<3>, 512, 3 * (64 * 256 + 96 * 128), a, iters, d_samples, ns, href.data(),
      (N / 128) * (C / 64));
  run("resident rows, 3 st, W const", wgrad_halo9_kernel<3, 14>, 512, 3 * (64 * 256 + 96 * 128), a, iters, d_samples, ns, href.data(),
      (N / 128) * (C / 64));
  run("resident rows, 2 st, W const", wgrad_halo9_kernel<2, 14>, 512, 2 * (64 * 256 + 96 * 128), a, iters, d_samples, ns, href.data(),
      (N / 128) * (C / 64));
  run("resident rows, 4 st, W const", wgrad_halo9_kernel<4, 14>, 512, 4 * (64 * 256 + 96 * 128), a, iters, d_samples, ns, href.data(),
      (N / 128) * (C / 64));
  run("resident rows, 3 st, rolled kk", wgrad_halo9_kernel<3, 0, true>, 512, 3 * (64 * 256 + 96 * 128), a, iters, d_samples, ns, href.data(),
      (N / 128) * (C / 64));
  run("resident rows, 3 st, W const, rolled", wgrad_halo9_kernel<3, 14, true>, 512, 3 * (64 * 256 + 96 * 128), a, iters, d_samples, ns, href.data(),
      (N / 128) * (C / 64));
  run("resident rows, 4 st, W const, rolled", wgrad_halo9_kernel<4, 14, true>, 512, 4 * (64 * 256 + 96 * 128), a, iters, d_samples, ns, href.data(),
      (N / 128) * (C / 64));
  run("resident rows, 2 st, W const, rolled", wgrad_halo9_kernel<2, 14, true>, 512, 2 * (64 * 256 + 96 * 128), a, iters, d_samples, ns, href.data(),
      (N / 128) * (C / 64));
  run("tap-split waves, 2 st", wgrad_halo9b_kernel<2, 14, false>, 512, 2 * (64 * 256 + 96 * 128), a, iters, d_samples, ns, href.data(),
      (N / 128) * (C / 64));
  run("tap-split waves, 3 st", wgrad_halo9b_kernel<3, 14, false>, 512, 3 * (64 * 256 + 96 * 128), a, iters, d_samples, ns, href.data(),
      (N / 128) * (C / 64));
  // the real convolution: zero padding as per-(pixel, tap) masks
  a.H = H;
  ref_kernel<<<(ns + 63) / 64, 64>>>(x, dy, d_samples, ns, M, N, C, W, H, d_ref);
  CK(hipMemcpy(href.data(), d_ref, sizeof(double) * ns, hipMemcpyDeviceToHost));
  run("resident rows, 3 st, W const, MASKED", wgrad_halo9_kernel<3, 14, false, true>, 512, 3 * (64 * 256 + 96 * 128 + 4 * 64 * 2), a, iters,
      d_samples, ns, href.data(), (N / 128) * (C / 64));
  run("resident rows, 2 st, W const, MASKED", wgrad_halo9_kernel<2, 14, false, true>, 512, 2 * (64 * 256 + 96 * 128 + 4 * 64 * 2), a, iters,
      d_samples, ns, href.data(), (N / 128) * (C / 64));
  run("resident rows, 4 st, W const, MASKED", wgrad_halo9_kernel<4, 14, false, true>, 512, 4 * (64 * 256 + 96 * 128 + 4 * 64 * 2), a, iters,
      d_samples, ns, href.data(), (N / 128) * (C / 64));
  run("resident rows, 2 st, W const, rolled, MASKED", wgrad_halo9_kernel<2, 14, true, true>, 512, 2 * (64 * 256 + 96 * 128 + 4 * 64 * 2), a,
      iters, d_samples, ns, href.data(), (N / 128) * (C / 64));
  run("tap-split waves, 2 st, MASKED", wgrad_halo9b_kernel<2, 14, true>, 512, 2 * (64 * 256 + 96 * 128 + 4 * 64 * 2), a, iters, d_samples, ns,
      href.data(), (N / 128) * (C / 64));
  run("tap-split waves, 3 st, MASKED", wgrad_halo9b_kernel<3, 14, true>, 512, 3 * (64 * 256 + 96 * 128 + 4 * 64 * 2), a, iters, d_samples, ns,
      href.data(), (N / 128) * (C / 64));
  return 0;
}
