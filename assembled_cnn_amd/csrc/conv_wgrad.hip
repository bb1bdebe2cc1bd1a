// MFMA weight-gradient kernel for gfx950 (split-K over output pixels, deterministic slab reduce).
//
//   dW[n][t][c] = sum_m  dY[m][n] * X[gather(m, t)][c]          m = (img, ho, wo)
//
// GEMM view: rows = output channels n (128 / block), columns j = (tap t, channel c) flattened
// (128 / block), reduction = pixels (64 per step).  Both operands arrive "reduction-major"
// ([pixel][channel], channel contiguous) which is the transpose of what an MFMA fragment needs
// (8 consecutive reduction elements per lane), so the fragments are read from LDS with gfx950's
// transposing ds_read_b64_tr_b16: every 16-lane group reads a 4(pixel) x 16(channel) block and
// receives one channel column per lane.  LDS rows are XOR-swizzled in 16-byte chunks by
// (pixel & 3) so the four pixel rows a transposing read touches sit on disjoint banks.
//
// Each block accumulates a 128 x 128 fp32 tile over its pixel range and writes it to a slab
// [split][Co][cols]; asm_wgrad_reduce sums the slabs in a fixed order (no atomics ->
// bit-reproducible).  With one split the tile goes straight to dW.
#include "common.h"
#include "../../include/asm_hip_debug.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace {

struct WgradArgs {
  const void* dy;
  const void* x;
  float* out;  // slab base (or dW when splits == 1)
  unsigned dy_bytes, x_bytes;
  int M;
  int Hi, Wi, Ci;
  int Co, ldy;
  int R, S, so, pad;
  int x_img_pitch, x_row_pitch, x_pix_pitch;
  int cols;  // R*S*Ci
  int tiles_n, tiles_c, splits;
  int m_per_split;  // multiple of 64
  FastDiv fd_howo, fd_wo;
  int HoWo, Wo;
};

#ifndef WG_WPX
#define WG_WPX 64
#endif
constexpr int WPX = WG_WPX;   // pixels per step

template <int N>
struct IC8 {
  static constexpr int value = N;
};

__device__ __forceinline__ bf16x4 ds_read_tr(const unsigned char* p) {
  typedef __attribute__((ext_vector_type(4))) short s4;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s4*)(const_cast<unsigned char*>(p)));
}

// 16-byte-chunk XOR swizzle for an LDS tile whose rows are RB bytes: the 4 pixel rows x 64 bytes one
// transposing read touches must cover all 64 banks exactly once.
template <int RB>
__device__ __forceinline__ int tr_swz(int row) {
  return RB >= 256 ? ((row & 3) << 2) : (RB == 128 ? (((row >> 1) & 1) << 2) : 0);
}

// BNW = output-channel (dy) tile width: 128 / 64 / 32, so narrow layers (K = 32 / 64 at 112x112 and 56x56,
// where the pixel count is largest) neither waste MFMAs on zero rows nor LDS on empty tiles.  BCW = (tap, channel)
// column tile: 128.  (The 256 x 256 tile of the layers whose dW is at least that large is wgrad8_kernel below.)
// LIN: 1x1, stride 1, no padding over a densely packed x: output pixel m IS input pixel m and column j IS channel j, so
// the staging loads need no (image, row, column) decode at all (two multiply-shift divisions, ~25 VALU per row and step in
// the general form; most weight-gradient launches of a bottleneck network are such 1x1 layers).
// NS = 2 (LIN, 128-column tiles): the two tiles of a step arrive by LDS-DMA into the other of two stages, requested one step
// ahead (csrc/conv_gemm1.hip has the forward / input-gradient twin): no staging registers, no ds_write.  The
// XOR swizzle of the transposing reads is applied on the SOURCE side (a lane's LDS destination is base + 16 * lane).  Same
// steps in the same order as the register-staged loop: bit-identical sums.
template <int BNW, int BCW, bool LIN = false, int NS = 0>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs p) {
  static_assert(BCW == 128 && BNW <= 128, "the 256 x 256 tile is wgrad8_kernel");
  constexpr int NTHR = 256;
  constexpr int WROWB = BCW * 2;            // x tile row bytes
  constexpr int WTILE = WPX * WROWB;
  constexpr int YROWB = BNW * 2;            // dy tile row bytes
  constexpr int YTILE = WPX * YROWB;
  constexpr int STAGE = YTILE + WTILE;
  constexpr int CX = BCW / 8;               // 16-byte chunks per x row
  constexpr int XRP = NTHR / CX;            // x rows staged per pass (16)
  constexpr int XP = WPX / XRP;             // x passes (4)
  constexpr int CY = BNW / 8;               // 16-byte chunks per dy row
  constexpr int YRP = NTHR / CY;            // dy rows staged per pass
  constexpr int YP = WPX / YRP > 0 ? WPX / YRP : 1;             // dy passes (4 / 2 / 1)
  constexpr int NT = BNW >= 64 ? 2 : 1;     // 32-row dy tiles per wave
  constexpr int CT = BNW == 128 ? 2 : 1;    // 32-col x tiles per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * STAGE

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = BNW == 128 ? (wave >> 1) : 0;
  const int wc = BNW == 128 ? (wave & 1) : wave;
  const int nbase = wn * 64;                                        // wave's first dy channel within the tile
  const int cbase = wc * (BNW == 128 ? 64 : 32);                    // wave's first column within the tile

  // block -> (split, tile_n, tile_c): tiles of one split adjacent (they re-read the same pixels)
  // XCD-aware bijective remap (block b runs on XCD b % 8, each XCD has its own L2): give every XCD a
  // contiguous range of logical blocks, i.e. whole pixel ranges (splits) / whole dy row-tiles, so the tiles
  // that share operands hit the same L2 instead of every XCD re-fetching every operand from HBM.
  int bid;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7;
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles = p.tiles_n * p.tiles_c;
  const int split = bid / tiles;
  bid -= split * tiles;
  const int tile_n = bid / p.tiles_c;
  const int tile_c = bid - tile_n * p.tiles_c;

  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);

  const int chunk = tid % CX;  // x tile: 16-byte chunk (8 columns) within the row
  const int prow = tid / CX;   // x tile: pixel row 0..15 (+16 per pass)
  const int ychunk = tid % CY;
  const int yrow = tid / CY;

  // this thread's fixed column group of the x tile: tap + channel
  const int j0 = tile_c * BCW + chunk * 8;
  const bool col_ok = j0 < p.cols;
  int tap_r = 0, tap_s = 0, tap_c = 0;
  if (col_ok) {
    const int t = j0 / p.Ci;
    tap_c = j0 - t * p.Ci;
    tap_r = t / p.S;
    tap_s = t - tap_r * p.S;
  }
  const int n0 = tile_n * BNW + ychunk * 8;
  const bool n_ok = n0 < p.Co;

  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int steps = (m_end - m_begin + WPX - 1) / WPX;

  // two register tile sets: 2-deep global prefetch (see conv_igemm.hip)
  u32x4 ya[YP], xa[XP], yb[YP], xb[XP];
  auto load_tile = [&](int step, u32x4(&ry)[YP], u32x4(&rxv)[XP]) {
#pragma unroll
    for (int j = 0; j < YP; ++j) {
      const int m = m_begin + step * WPX + yrow + YRP * j;
      const unsigned offy = ((unsigned)m * (unsigned)p.ldy + (unsigned)n0) * 2u;
      ry[j] = __builtin_amdgcn_raw_buffer_load_b128(rdy, (m < m_end && n_ok) ? offy : ASM_OOB, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int m = m_begin + step * WPX + prow + XRP * j;
      const bool mok = m < m_end;
      if constexpr (LIN) {
        const unsigned offl = ((unsigned)m * (unsigned)p.Ci + (unsigned)j0) * 2u;
        rxv[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (mok && col_ok) ? offl : ASM_OOB, 0, 0);
        continue;
      }
      const unsigned um = mok ? (unsigned)m : 0u;
      const unsigned img = fd_div(um, p.fd_howo);
      const unsigned rem = um - img * (unsigned)p.HoWo;
      const unsigned ho = fd_div(rem, p.fd_wo);
      const unsigned wo = rem - ho * (unsigned)p.Wo;
      const int ih = (int)ho * p.so + tap_r - p.pad;
      const int iw = (int)wo * p.so + tap_s - p.pad;
      const bool ok = mok && col_ok && ((unsigned)ih < (unsigned)p.Hi) && ((unsigned)iw < (unsigned)p.Wi);
      const unsigned offx = (img * (unsigned)p.x_img_pitch + (unsigned)ih * (unsigned)p.x_row_pitch +
                             (unsigned)iw * (unsigned)p.x_pix_pitch + (unsigned)tap_c) * 2u;
      rxv[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? offx : ASM_OOB, 0, 0);
    }
  };
  auto store_tile = [&](int stage, const u32x4(&ry)[YP], const u32x4(&rxv)[XP]) {
    unsigned char* ys = smem + stage * STAGE;
    unsigned char* xs = ys + YTILE;
#pragma unroll
    for (int j = 0; j < YP; ++j) {
      const int row = yrow + YRP * j;
      *reinterpret_cast<u32x4*>(ys + row * YROWB + ((ychunk ^ tr_swz<YROWB>(row)) << 4)) = ry[j];
    }
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int row = prow + XRP * j;
      *reinterpret_cast<u32x4*>(xs + row * WROWB + ((chunk ^ tr_swz<WROWB>(row)) << 4)) = rxv[j];
    }
  };

  f32x16 acc[NT][CT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < CT; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // transposing-read lane geometry (32x32x16 fragment: channel = lane&31, reduction group = lane>>5)
  const int t16 = lane & 15;          // lane within its 16-lane group
  const int g = lane >> 4;            // group 0..3
  const int colsel = (g & 1) * 16;    // which 16 of the fragment's 32 channels
  const int pgrp = (g >> 1) * 8;      // reduction offset 0 / 8
  const int trow = t16 >> 2;          // pixel row within the 4-row block
  const int tcol = (t16 & 3) * 4;     // channel offset of this lane's 4-element source

  auto compute = [&](int stage) {
    const unsigned char* ys = smem + stage * STAGE;
    const unsigned char* xs = ys + YTILE;
#pragma unroll
    for (int kk = 0; kk < WPX / 16; ++kk) {
      const int row = kk * 16 + pgrp + trow;
      const int row2 = row + 4;
      bf16x8 fy[NT], fx[CT];
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const int cy = nbase + a * 32 + colsel + tcol;  // channel (element) index in the dy row
        const int o = ((cy & 4) << 1);
        const bf16x4 y0 = ds_read_tr(ys + row * YROWB + ((((cy >> 3) ^ tr_swz<YROWB>(row)) << 4) | o));
        const bf16x4 y1 = ds_read_tr(ys + row2 * YROWB + ((((cy >> 3) ^ tr_swz<YROWB>(row2)) << 4) | o));
        fy[a] = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int b = 0; b < CT; ++b) {
        const int cx = cbase + b * 32 + colsel + tcol;
        const int o = ((cx & 4) << 1);
        const bf16x4 x0 = ds_read_tr(xs + row * WROWB + ((((cx >> 3) ^ tr_swz<WROWB>(row)) << 4) | o));
        const bf16x4 x1 = ds_read_tr(xs + row2 * WROWB + ((((cx >> 3) ^ tr_swz<WROWB>(row2)) << 4) | o));
        fx[b] = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[a], fx[b], acc[a][b], 0, 0, 0);
    }
  };

  if constexpr (NS >= 2) {
    static_assert(LIN && NS == 2, "the ring form covers the linear-address layers, two stages");
    // NOTE (round 6): behind a compiler-visible LDS read (the ds_read_tr builtin) hipcc waits for EVERY outstanding LDS-DMA --
    // s_waitcnt vmcnt(0) before the first fragment read of a step, i.e. also for the tiles requested a moment earlier -- so
    // inside one workgroup this ring overlaps nothing (and the depths 3 / 4 that existed until round 6 were the same loop with
    // less occupancy: removed); the overlap comes from the second workgroup on the CU.  Hiding the reads in inline asm with hand-counted lgkmcnt (what wgrad8_kernel does)
    // was measured here too (tools/gemm1_sweep.py --wgrad, same box, every 1x1 layer at batch 256): 2751 against 2780 us per
    // step summed over the layers, single layers +-10 % either way -- within the noise, so the simpler form stays.
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int YPW = YROWB == 256 ? 4 : (YROWB == 128 ? 2 : 1);   // dy pieces (1 KiB) per wave and step
    constexpr int YRPP = 1024 / YROWB;                                // dy rows per piece: 4 / 8 / 16
    auto issue = [&](int stage, int step) {
      unsigned char* ys = smem + stage * STAGE;
      unsigned char* xs = ys + YTILE;
      const int m0 = m_begin + step * WPX;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int piece = wave + 4 * j;
        const int row = 4 * piece + (lane >> 4);
        const int cs = (lane & 15) ^ tr_swz<WROWB>(row);
        const int m = m0 + row, c = tile_c * BCW + cs * 8;
        const unsigned off = ((unsigned)m * (unsigned)p.Ci + (unsigned)c) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xs + 4 * piece * WROWB), 16,
                                                 (int)((m < m_end && c < p.cols) ? off : ASM_OOB), 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < YPW; ++j) {
        const int piece = wave + 4 * j;
        const int row = YRPP * piece + lane / CY;
        const int cs = (lane % CY) ^ tr_swz<YROWB>(row);
        const int m = m0 + row, n = tile_n * BNW + cs * 8;
        const unsigned off = ((unsigned)m * (unsigned)p.ldy + (unsigned)n) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lptr_t)(ys + YRPP * piece * YROWB), 16,
                                                 (int)((m < m_end && n < p.Co) ? off : ASM_OOB), 0, 0, 0);
      }
    };
    if (steps > 0) issue(0, 0);
    int cur = 0;
#pragma unroll 1
    for (int step = 0; step < steps; ++step) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // this step's tiles visible; step - 1's stage released
      if (step + 1 < steps) issue(cur ^ 1, step + 1);
      compute(cur);
      cur ^= 1;
    }
  } else {
  // steps beyond the range load nothing (m >= m_end -> zeros), so the pair loop needs no tail branch
  load_tile(0, ya, xa);
  store_tile(0, ya, xa);
  load_tile(1, ya, xa);
  __syncthreads();
#pragma unroll 1
  for (int step = 0; step < steps; step += 2) {
    load_tile(step + 2, yb, xb);
    compute(0);
    store_tile(1, ya, xa);
    __syncthreads();
    load_tile(step + 3, ya, xa);
    compute(1);
    store_tile(0, yb, xb);
    __syncthreads();
  }
  }

  // D[i = n_local][j = col_local]: col = lane&31, n = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float* out = p.out + (size_t)split * p.Co * p.cols;
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < CT; ++b) {
      const int col = tile_c * BCW + cbase + b * 32 + l31;
      if (col < p.cols) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = tile_n * BNW + nbase + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (n < p.Co) out[(size_t)n * p.cols + col] = acc[a][b][r];
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------------------
// wgrad8_kernel: the 256 x 256 weight-gradient tile on a wave-staggered, multi-phase main loop (conv_igemm8.hip is the forward /
// input-gradient twin; same reasoning, different operand geometry).
//
// Same math and the same (pixel block of 16, MFMA) accumulation order as wgrad_kernel<256, 256> on the same plan -- bit-identical
// sums.  What differs is how the operands travel:
//   * a PHASE is one 16-pixel block of the reduction: 12 transposing fragment reads, 8 MFMAs 32x32x16 on the wave's 8 accumulators
//     (no MFMA depends on one nearer than 8 issue slots);
//   * LDS is a ring of eight such blocks per operand (dy: 16 rows x 512 B, x: 16 rows x 512 B; 2 x 64 KB), filled by LDS-DMA: in
//     phase g every wave requests its two rows of block g + 6 (one dy piece, one x piece of 1 KB) -- no staging registers, no
//     ds_write, no (image, row, column) decode for more than ONE pixel per lane and phase;
//   * the DMA never drains: vmcnt(10) in phase g retires block g + 1 (requested five phases earlier), the barriers are raw;
//   * waves w and w + 4 (one SIMD) run ONE BARRIER APART: in every barrier interval one of them multiplies (s_setprio 1) while the
//     other reads fragments and issues DMA.
//   Write-after-read: block g + 6 lands in the slot of block g - 2; its readers retired their reads in phase g - 2 (lgkmcnt(0)
//     behind that phase's first barrier) and the staggered group is one barrier late: at least three barriers in between.
//   Read-after-write: a wave's wait for block g + 1 sits before the first barrier of ITS phase g; the first read of that block
//     (phase g + 1 of the group that is ahead) is behind the second one.
// Blocks past the end of the pixel range are requested at the out-of-range offset (zero-fill; their products are zeros): the
// counted waits need no tail case.
template <bool LIN>
__global__ __launch_bounds__(512) void wgrad8_kernel(WgradArgs p) {
    constexpr int ROWB = 512, BLK = 16 * ROWB, RING = 8, XBASE = RING * BLK, DIST = 6, INFL = 2 * (DIST - 1);
  typedef __attribute__((address_space(3))) void* lptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 x 64 KB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                       // waves w and w + 4 share a SIMD: one of each group per SIMD
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
  const int tile_n = bid / p.tiles_c;
  const int tile_c = bid - tile_n * p.tiles_c;

  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);

  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int groups = (m_end - m_begin + 127) >> 7;          // eight phases = 128 pixels per trip of the loop

  // ---- what this lane requests in every phase: row rb of the block, 16-byte piece cp of the row (source-side swizzle) ----
  const int rb = 2 * wave + (lane >> 5), cp = lane & 31;
  const int lc = cp ^ tr_swz<ROWB>(rb);
  const int j0 = tile_c * 256 + lc * 8;
  const bool col_ok = j0 < p.cols;
  int tap_c = 0, dr = 0, ds = 0;
  if (!LIN && col_ok) {
    const int t = j0 / p.Ci;
    tap_c = j0 - t * p.Ci;
    const int tr = t / p.S;
    dr = tr - p.pad;
    ds = (t - tr * p.S) - p.pad;
  }
  const int n0 = tile_n * 256 + lc * 8;
  const bool n_ok = n0 < p.Co;
  unsigned char* const ydst = smem + 2 * wave * ROWB;
  unsigned char* const xdst = smem + XBASE + 2 * wave * ROWB;
  int mrow = m_begin + rb;                                   // the pixel of the next block to request

  // branch-free on purpose (bitwise tests, selects): a short-circuit here becomes exec-masked control flow around the requests
  auto request = [&](const int slot) {
    const int m = mrow;
    mrow += 16;
    const unsigned mok = (unsigned)(m < m_end);
    const unsigned offy = ((unsigned)m * (unsigned)p.ldy + (unsigned)n0) * 2u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lptr_t)(ydst + slot * BLK), 16, (int)((mok & (unsigned)n_ok) ? offy : ASM_OOB), 0, 0, 0);
    unsigned offx, ok = mok & (unsigned)col_ok;
    if constexpr (LIN) {
      offx = ((unsigned)m * (unsigned)p.Ci + (unsigned)j0) * 2u;
    } else {
      const unsigned um = (unsigned)min(m, p.M - 1);
      const unsigned img = fd_div(um, p.fd_howo);
      const unsigned rem = um - img * (unsigned)p.HoWo;
      const unsigned ho = fd_div(rem, p.fd_wo);
      const unsigned wo = rem - ho * (unsigned)p.Wo;
      const int ih = (int)ho * p.so + dr;
      const int iw = (int)wo * p.so + ds;
      ok &= (unsigned)((unsigned)ih < (unsigned)p.Hi) & (unsigned)((unsigned)iw < (unsigned)p.Wi);
      offx = (img * (unsigned)p.x_img_pitch + (unsigned)ih * (unsigned)p.x_row_pitch + (unsigned)iw * (unsigned)p.x_pix_pitch +
              (unsigned)tap_c) * 2u;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xdst + slot * BLK), 16, (int)(ok ? offx : ASM_OOB), 0, 0, 0);
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // transposing-read lane geometry (wgrad_kernel): the two reads of a fragment are rows `row` and `row + 4` of the block
  const int t16 = lane & 15, g4 = lane >> 4;
  const int colsel = (g4 & 1) * 16, tcol = (t16 & 3) * 4;
  const int row = (g4 >> 1) * 8 + (t16 >> 2);              // (row + 4) & 3 == row & 3: one swizzle term for both
  unsigned fyo[2], fxo[4];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int cy = nbase + a * 32 + colsel + tcol;
    fyo[a] = (unsigned)(row * ROWB + ((((cy >> 3) ^ tr_swz<ROWB>(row)) << 4) | ((cy & 4) << 1)));
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int cx = cbase + b * 32 + colsel + tcol;
    fxo[b] = (unsigned)(XBASE + row * ROWB + ((((cx >> 3) ^ tr_swz<ROWB>(row)) << 4) | ((cx & 4) << 1)));
  }

  // ---- pipeline fill: blocks 0 .. 5; block 0 has landed behind vmcnt(10) ----
#pragma unroll
  for (int s = 0; s < DIST; ++s) request(s);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL) : "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();      // the stagger: group 1 runs one barrier behind group 0

  // The fragment reads are inline asm: behind a compiler-visible LDS read hipcc drains every outstanding LDS-DMA (s_waitcnt
  // vmcnt(0) before the first ds_read of each phase -- the whole ring would be pointless).  Their completion is counted by the
  // lgkmcnt(0) statement below, which names every destination (cdna_hip_programming.md 5.7, form (ii)).
  typedef __attribute__((ext_vector_type(4))) short s4;
#define WG8_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
  auto phase = [&](auto sc) {
    constexpr int S = decltype(sc)::value;
    s4 y0[2], y1[2], x0[4], x1[4];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      WG8_TR(y0[a], fyo[a], S * BLK);
      WG8_TR(y1[a], fyo[a], S * BLK + 4 * ROWB);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      WG8_TR(x0[b], fxo[b], S * BLK);
      WG8_TR(x1[b], fxo[b], S * BLK + 4 * ROWB);
    }
    request((S + DIST) % RING);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(y0[0]), "+v"(y1[0]), "+v"(y0[1]), "+v"(y1[1]), "+v"(x0[0]), "+v"(x1[0]), "+v"(x0[1]), "+v"(x1[1]),
                   "+v"(x0[2]), "+v"(x1[2]), "+v"(x0[3]), "+v"(x1[3])
                 :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 fy[2], fx[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
      fy[a] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(y0[a], y1[a], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
    for (int b = 0; b < 4; ++b)
      fx[b] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(x0[b], x1[b], 0, 1, 2, 3, 4, 5, 6, 7));
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[a], fx[b], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]),
                      "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]));
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
#undef WG8_TR
#pragma unroll 1
  for (int gi = 0; gi < groups; ++gi) {
    phase(IC8<0>{}); phase(IC8<1>{}); phase(IC8<2>{}); phase(IC8<3>{});
    phase(IC8<4>{}); phase(IC8<5>{}); phase(IC8<6>{}); phase(IC8<7>{});
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();      // group 0 waits for group 1's last phase
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the zero-fill requests past the end

  float* out = p.out + (size_t)split * p.Co * p.cols;
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
          if (n < p.Co) out[(size_t)n * p.cols + col] = acc[a][b][r];
        }
      }
    }
}

// slab reduce: dw[i] = sum_k slab[k][i].  256 threads = 32 float4 columns x 8 split-lanes; every lane keeps 4
// independent loads in flight (the serial-k version was latency-bound), partial sums meet in LDS in a fixed
// order, so the result stays bit-reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           size_t n, int splits) {
  __shared__ f32x4 red[8][32];
  const int cx = threadIdx.x & 31, ky = threadIdx.x >> 5;
  const size_t i = ((size_t)blockIdx.x * 32 + cx) * 4;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (i < n) {  // n is a multiple of 8 (C % 8 == 0)
    int k = ky;
    for (; k + 24 < splits; k += 32) {
      s0 += *reinterpret_cast<const f32x4*>(slab + (size_t)k * n + i);
      s1 += *reinterpret_cast<const f32x4*>(slab + (size_t)(k + 8) * n + i);
      s2 += *reinterpret_cast<const f32x4*>(slab + (size_t)(k + 16) * n + i);
      s3 += *reinterpret_cast<const f32x4*>(slab + (size_t)(k + 24) * n + i);
    }
    for (; k < splits; k += 8) s0 += *reinterpret_cast<const f32x4*>(slab + (size_t)k * n + i);
  }
  red[ky][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ky == 0 && i < n) {
    f32x4 t = red[0][cx];
#pragma unroll
    for (int r = 1; r < 8; ++r) t += red[r][cx];
    *reinterpret_cast<f32x4*>(dw + i) = t;
  }
}


// ------------------------------------------------------------------------------------------------------------
// wgrad_halo_kernel: weight gradient of the NARROW 3x3 stride-1 layers (C in {32, 64}, K in {32, 64, 128}), persistent.
// The general kernel gathers one x tile PER TAP COLUMN GROUP, i.e. reads (nearly) the same pixels 9 times through L2 and spends
// more VALU on the (image, row, column) decode of its gather than MFMA cycles on these layers (28 % MFMA-busy; 56x56x64 -> 128 at
// batch 256 takes 300 us in the step, six times its bound).  Here a workgroup walks a contiguous run of pixel PATCHES of one
// image -- PH x PW pixels: 8 x 16 on the 112 x 112 maps, whole-row bands of 2 x 56 / 4 x 28 on the 56 x 56 / 28 x 28 ones -- stages
// the patch's dy rows and the (PH + 2) x (PW + 2) halo of x in LDS once, and accumulates ALL nine taps of dW[k][t][c] in
// registers from them (transposing ds_read_b64_tr_b16 fragments; a tap is a constant offset into the halo).  The reduction runs
// over the patch's pixels in groups of 16 (a group may straddle image rows: every lane addresses its own pixel).  The next patch
// is prefetched into registers under the MFMAs.  With K = 128 a workgroup owns one 64-channel half of dy (two workgroups per
// patch run, adjacent in the XCD-contiguous order: the second one's x halo comes from L2).  One fp32 slab [K][9 C] per patch
// run, summed by wgrad_reduce_kernel in fixed order (deterministic).
// NW waves per workgroup: 4 (two workgroups per CU), or 8 for the 64 x 64 channel form -- its 36 accumulator units are 9 per wave
// on four waves (144 accumulator + 144 other registers: one wave per SIMD, i.e. ONE four-wave workgroup per CU and nothing to
// overlap a patch's staging with); on eight waves 5 per wave, 2 waves per SIMD.
template <int KF, int CI, int PH, int PW, int NW = (KF == 64 && CI == 64 ? 8 : 4)>
struct WHalo {
  static constexpr int NTHR = 64 * NW;
  static constexpr int PPX = PH * PW, NKK = PPX / 16;            // pixels per patch (128 / 112), reduction groups
  static constexpr int HWD = PW + 2, NHP = (PH + 2) * HWD;       // halo width, halo pixels
  static constexpr int XRB = CI == 32 ? 64 : 192;    // row strides = 16 dwords mod 64: the 4 pixel rows of a transposing
  static constexpr int YRB = KF == 32 ? 64 : 192;    // read land on disjoint banks (C = 64: 128 data + 64 pad bytes)
  static constexpr int XS = NHP * XRB, YS = PPX * YRB;
  static constexpr int LDS = XS + YS;
  static constexpr int KT = KF / 32, CT = CI / 32;
  static constexpr int NU = KT * CT * 9;             // 32 x 32 accumulator units (tap, k-tile, c-tile)
  static constexpr int UPW = (NU + NW - 1) / NW;     // per wave
  static constexpr int NVX = NHP * (CI / 8), NVY = PPX * (KF / 8);
  static constexpr int HPX = (NVX + NTHR - 1) / NTHR, HPY = (NVY + NTHR - 1) / NTHR;
  static_assert(PPX % 16 == 0 && LDS <= 80 * 1024, "patch");
};

template <int KF, int CI, int PH, int PW>
__global__ __launch_bounds__((WHalo<KF, CI, PH, PW>::NTHR)) void wgrad_halo_kernel(WgradArgs p) {
  using W = WHalo<KF, CI, PH, PW>;
  constexpr int XRB = W::XRB, YRB = W::YRB, CPX = CI / 8, CPY = KF / 8, HWD = W::HWD, NTHR = W::NTHR, NW = NTHR / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xs = smem;
  unsigned char* ys = smem + W::XS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  const int khalves = p.tiles_n;                     // K / KF workgroups per patch run
  const int tiles_x = p.Wi / PW, tpi = tiles_x * (p.Hi / PH);
  const int n_m = (p.M / (p.Hi * p.Wi)) * tpi;       // patches
  int t_begin, t_end, logical, khalf;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int lg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // XCD-contiguous runs of patches
    logical = lg / khalves;
    khalf = lg - logical * khalves;
    const int runs = nb / khalves;
    const int per = n_m / runs, extra = n_m - per * runs;
    t_begin = logical * per + (logical < extra ? logical : extra);
    t_end = t_begin + per + (logical < extra ? 1 : 0);
  }
  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);

  u32x4 hx[W::HPX], hy[W::HPY];
  auto load_patch = [&](int tile) {
    const int img = tile / tpi, trem = tile - img * tpi;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int y0 = ty * PH, x0 = tx * PW;
#pragma unroll
    for (int k = 0; k < W::HPX; ++k) {
      const int i = k * NTHR + tid;
      const int hp = i / CPX, ck = i - hp * CPX;
      const int hyy = hp / HWD, hxx = hp - hyy * HWD;
      const int gy = y0 - 1 + hyy, gx = x0 - 1 + hxx;
      const bool ok = (i < W::NVX) && ((unsigned)gy < (unsigned)p.Hi) && ((unsigned)gx < (unsigned)p.Wi);
      const unsigned off = (((unsigned)img * (unsigned)p.Hi + (unsigned)gy) * (unsigned)p.Wi + (unsigned)gx) * (unsigned)p.Ci * 2u +
                           (unsigned)ck * 16u;
      hx[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : ASM_OOB, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < W::HPY; ++k) {
      const int i = k * NTHR + tid;
      const int row = i / CPY, ck = i - row * CPY;
      const int ry = row / PW, rxx = row - ry * PW;
      const unsigned m = ((unsigned)img * (unsigned)p.Hi + (unsigned)(y0 + ry)) * (unsigned)p.Wi + (unsigned)(x0 + rxx);
      hy[k] = __builtin_amdgcn_raw_buffer_load_b128(
          rdy, i < W::NVY ? (m * (unsigned)p.ldy + (unsigned)(khalf * KF) + (unsigned)ck * 8u) * 2u : ASM_OOB, 0, 0);
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int k = 0; k < W::HPX; ++k) {
      const int i = k * NTHR + tid;
      const int hp = i / CPX, ck = i - hp * CPX;
      if (i < W::NVX) *reinterpret_cast<u32x4*>(xs + hp * XRB + ck * 16) = hx[k];
    }
#pragma unroll
    for (int k = 0; k < W::HPY; ++k) {
      const int i = k * NTHR + tid;
      const int row = i / CPY, ck = i - row * CPY;
      if (i < W::NVY) *reinterpret_cast<u32x4*>(ys + row * YRB + ck * 16) = hy[k];
    }
  };

  // Which accumulator units (tap t, k-tile kt, c-tile ct) a wave owns.  General: unit wave + NW * i of the (t, kt, ct) order.
  // The eight-wave 64 x 64 form (36 units, two k-tiles): the two k-tiles of a (t, ct) PAIR multiply the same x fragment, so a wave
  // takes pairs -- units 0 - 3 = pairs 2 w and 2 w + 1, and waves 0 - 3 one more unit, half of pairs 16 / 17: 6 instead of 10 x
  // fragment reads per 16 pixels for the five-unit waves (the LDS pipe, not the MFMA pipe, bounds this kernel).
  constexpr bool PAIRED = NW == 8 && W::KT == 2 && W::NU == 36;
  auto unit_of = [&](const int i, int& t, int& kt, int& ct) -> bool {
    if constexpr (PAIRED) {
      const int pair = i < 4 ? 2 * wave + (i >> 1) : 16 + (wave >> 1);
      kt = i < 4 ? (i & 1) : (wave & 1);
      t = pair / W::CT;
      ct = pair - t * W::CT;
      return i < 4 || wave < 4;
    } else {
      const int u = wave + NW * i;
      t = u / (W::KT * W::CT);
      const int rem = u - t * (W::KT * W::CT);
      kt = rem / W::CT;
      ct = rem - kt * W::CT;
      return u < W::NU;
    }
  };

  f32x16 acc[W::UPW];
#pragma unroll
  for (int i = 0; i < W::UPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // transposing-read lane geometry (see wgrad_kernel): lane l of a 32x32x16 operand = channel l & 31, pixels (l >> 5) * 8 ..
  const int t16 = lane & 15, g = lane >> 4;
  const int chan = (g & 1) * 16 + (t16 & 3) * 4;    // first of this lane's 4 source channels within a 32-channel tile
  const int prow = (g >> 1) * 8 + (t16 >> 2);       // source pixel within the 16-pixel reduction group (second read: + 4)
  // halo row (tap (0, 0)) of this lane's two source pixels of every reduction group, in bytes
  unsigned hb0[W::NKK], hb1[W::NKK];
#pragma unroll
  for (int kk = 0; kk < W::NKK; ++kk) {
    const int p0 = kk * 16 + prow, p1 = p0 + 4;
    hb0[kk] = (unsigned)(((p0 / PW) * HWD + p0 % PW) * XRB);
    hb1[kk] = (unsigned)(((p1 / PW) * HWD + p1 % PW) * XRB);
  }

  if (t_begin < t_end) load_patch(t_begin);
#pragma unroll 1
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();                    // every wave is done with the previous patch
    store_patch();
    __syncthreads();
    if (tile + 1 < t_end) load_patch(tile + 1);
#pragma unroll
    for (int kk = 0; kk < W::NKK; ++kk) {    // 16 reduction pixels
      bf16x8 fy[W::KT];
#pragma unroll
      for (int kt = 0; kt < W::KT; ++kt) {
        const unsigned char* a = ys + (kk * 16 + prow) * YRB + (kt * 32 + chan) * 2;
        const bf16x4 y0 = ds_read_tr(a), y1 = ds_read_tr(a + 4 * YRB);
        fy[kt] = __builtin_shufflevector(y0, y1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      bf16x8 fx;
#pragma unroll
      for (int i = 0; i < W::UPW; ++i) {
        int t, kt, ct;
        if (unit_of(i, t, kt, ct)) {                // wave-uniform
          if (!(PAIRED && i < 4 && (i & 1))) {      // (the second unit of a pair multiplies the same x fragment)
            const int r = t / 3, q = t - r * 3;
            const unsigned tapo = (unsigned)((r * HWD + q) * XRB + (ct * 32 + chan) * 2);
            const bf16x4 x0 = ds_read_tr(xs + hb0[kk] + tapo), x1 = ds_read_tr(xs + hb1[kk] + tapo);
            fx = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
          }
          const bf16x8 fa = (W::KT == 1 || kt == 0) ? fy[0] : fy[W::KT - 1];   // static register indices only
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fx, acc[i], 0, 0, 0);
        }
      }
    }
  }
  // this workgroup's partial dW: rows [khalf * KF, + KF) of slab [logical][K][9 * CI]
  const int cols = 9 * CI;
  float* out = p.out + ((size_t)logical * p.Co + (size_t)khalf * KF) * cols;
  const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int i = 0; i < W::UPW; ++i) {
    int t, kt, ct;
    if (unit_of(i, t, kt, ct)) {
      const int col = t * CI + ct * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        out[(size_t)n * cols + col] = acc[i][r];
      }
    }
  }
}

// patch geometry of the halo form for a map: 8 x 16 patches where the map tiles into them, whole-row bands of 112 pixels on the
// 56- and 28-wide maps; 0 = none
inline int halo_patch_w(const asm_conv_desc* d) {
  if (d->H % 8 == 0 && d->W % 16 == 0) return 16;
  if (d->W == 56 && d->H % 2 == 0) return 56;
  if (d->W == 28 && d->H % 4 == 0) return 28;
  return 0;
}

// patch runs (= slabs) of the persistent halo form for this layer, or 0 if it does not apply; the launch has K / KF workgroups
// per run (KF = 64 for K = 128)
int wgrad_halo_blocks(const asm_conv_desc* d) {
  const int mode = asm_tune().wgrad_halo;     // 0 off, 1 on for the large maps, 2 whenever the shape allows (tests)
  if (!mode) return 0;
  if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->Ho != d->H || d->Wo != d->W) return 0;
  if (d->x_img_pitch || d->x_row_pitch || d->x_pix_pitch) return 0;
  const int pw = halo_patch_w(d);
  if (!pw) return 0;
  if (!((d->C == 32 || d->C == 64) && (d->K == 32 || d->K == 64 || d->K == 128))) return 0;
  const int ppx = pw == 16 ? 128 : 112;
  const int n_m = d->N * d->H * d->W / ppx;
  if (mode != 2 && n_m < 512) return 0;   // a slab per workgroup only pays on the large maps
  const int kh = d->K == 128 ? 2 : 1;
  const int wgs = (d->C == 64 && d->K >= 64) ? 256 : 512;    // the eight-wave form runs one workgroup per CU, the others two
  return n_m < wgs / kh ? n_m : wgs / kh;
}

struct Plan {
  int bnw, bcw, tiles_n, tiles_c, splits, m_per_split;
};

Plan make_plan(const asm_conv_desc* d) {
  Plan pl;
  const int M = d->N * d->Ho * d->Wo;
  const int cols = d->R * d->S * d->C;
  pl.bnw = (d->K <= 32 && WPX == 64) ? 32 : (d->K <= 64 ? 64 : 128);
  pl.bcw = 128;
  // 256 x 256 / 8 waves (wgrad8_kernel) when dW tiles exactly (no padded MFMAs) and there is enough of it; ASM_WGRAD_BIG=0/1
  // forces.  Round-6 same-box sweep of wgrad8 against the 128-wide kernels at batch 256 (tools/conv_bench.py --kinds wgrad,
  // ASM_WGRAD_BIG=1 against the default): 3x3 layers win from 30 GFLOP up (7x7x256->512 69 -> 61 us, 14x14x128->256 60 -> 56 us;
  // the >= 118 GFLOP ones 13 - 15 %); 1x1 layers from 53 GFLOP up, also with one column tile (28x28x256->512 92 -> 83 us), and lose
  // 10 - 20 % at 26 GFLOP and below.  Column padding of up to 1/8 (3x3 with 128 input channels: 1152 -> 1280) still nets a win.
  const int big_env = asm_tune().wgrad_big;
  const int cpad = cdiv(cols, 256) * 256;
  const double gflop = 2.0 * (double)M * d->K * cols * 1e-9;
  const bool taps = d->R * d->S > 1;
  bool big = d->K % 256 == 0 && (cpad - cols) * 8 <= cpad &&
             (taps ? (cols >= 512 && gflop >= 25.0) : (cols >= 256 && gflop >= 40.0));
  if (big_env == 0) big = false;
  if (big_env == 1 && d->K >= 256 && cols >= 256) big = true;
  if (big) pl.bnw = pl.bcw = 256;
  pl.tiles_n = cdiv(d->K, pl.bnw);
  pl.tiles_c = cdiv(cols, pl.bcw);
  const int tiles = pl.tiles_n * pl.tiles_c;
  const int msteps = cdiv(M, WPX);
  // Split count from a small cost model (microseconds): the MFMA/stream time scales with how evenly
  // tiles*splits blocks fill the resident slots (256 CUs x 1..4 workgroups), the slab path costs
  // (splits + 1) x |dW| of fp32 traffic.  With the XCD-contiguous block order every split's pixel range stays
  // on one XCD's L2, so extra splits do not add HBM reads of x / dy -- as long as the splits do not straddle XCDs:
  // Round-6 sweep (tools/wgrad_split_sweep.py, every layer at batch 256, same box): with several tiles per split a split
  // count that is a multiple of 8 gives every XCD whole splits, and one that is not costs 8 - 15 % (56x56x64 -> 128 3x3:
  // 96 splits 179 us, 102 splits 206; 28x28x256 -> 512 1x1: 96 splits 78 us, 126 splits 94).  The 256 x 256 tile (one 128 KB
  // workgroup per CU) pays ~12 us per round of the chip in pipeline fill and slab epilogue, and an under-full single round
  // costs it less than its share of idle CUs (the busy ones clock higher: 14x14x512 -> 1024 3x3 on 216 of 256 CUs in one round
  // 404 us, on 504 workgroups in two rounds 430).
  const size_t wbytes = (size_t)d->K * cols * sizeof(float);
  const double flops = 2.0 * (double)M * d->K * cols;
  const double io_bytes = 2.0 * ((double)M * d->C + (double)M * d->K);
  const bool big8 = pl.bnw == 256;
  const double work_us = fmax(flops / (big8 ? 9.0e8 : 6.0e8), io_bytes / 4.0e6);      // ~900 / ~600 TFLOP/s or ~4 TB/s
  const int slots = 256 * (pl.bnw == 256 ? 1 : (pl.bnw == 128 ? 2 : (pl.bnw == 64 ? 3 : 4))) * (64 / WPX);
  const int max_splits = msteps / 4 > 0 ? msteps / 4 : 1;            // at least 4 steps per block
  int splits = 1;
  double best = 1e30;
  for (int req = 1; req <= 256 && req <= max_splits; ++req) {
    const int sp = cdiv(msteps, cdiv(msteps, req));                   // the split count a request for `req` ends up with
    if (sp != req) continue;
    const double blocks = (double)tiles * sp;
    const double rounds = ceil(blocks / slots);
    const double fill = rounds / (blocks / slots);                    // >= 1: quantisation of the last round
    const double under = blocks >= slots ? 1.0                        // too few blocks: less overlap
                         : (big8 ? 1.0 + 0.35 * ((double)slots / blocks - 1.0) : (double)slots / blocks * 0.5 + 0.5);
    const double straddle = (tiles > 1 && sp > 8 && sp % 8 != 0) ? 1.08 : 1.0;
    const double slab = sp > 1 ? ((double)(sp + 1) * wbytes / 4.0e6 + 4.0) : 0.0;
    const double est = work_us * (blocks < slots ? under : fill) * straddle + slab + (big8 ? 12.0 * rounds : 0.0);
    if (est < best) {
      best = est;
      splits = sp;
    }
  }
  // ASM_WGRAD_SPLITS=n forces the pixel split (tests: slab path on small shapes; tuning)
  const int forced = asm_tune().wgrad_splits;
  if (forced > 0) splits = forced < msteps ? forced : msteps;
  int steps_per = cdiv(msteps, splits);
  pl.m_per_split = steps_per * WPX;
  pl.splits = cdiv(M, pl.m_per_split);
  return pl;
}

// ring depth of the LDS-DMA form for a linear-address 1x1 layer with 128-column tiles (asm_tuning.wgrad_ring: 0 never,
// n >= 1 the two-stage ring forced, -1 per layer from the same-box sweep, tools/gemm1_sweep.py --wgrad), or 0 for the register-staged loop
int wgrad_ring_depth(const asm_conv_desc* d, const Plan& pl) {
  const int mode = asm_tune().wgrad_ring;
  if (mode == 0 || WPX != 64) return 0;
  if (mode > 0) return 2;
  // Round-5 sweep (tools/gemm1_sweep.py --wgrad, every 1x1 shape of Assemble-ResNet-50 at batch 256, bit-identical sums): two
  // stages (64 KB, two workgroups per CU like the register-staged loop) win 3 - 10 % on the 28 x 28 and smaller maps and on
  // the 56 x 56 layers with >= 128 input channels, and lose 20 - 30 % on the narrow 56 x 56 ones (x rows of 64 / 128 bytes:
  // a quarter / half of every 256-byte DMA row is padding).  Deeper rings (three / four stages, one workgroup per CU) lost
  // everywhere and were removed in round 6 (see the note in wgrad_kernel: hipcc drains the ring before every fragment read).
  (void)pl;
  if ((long long)d->N * d->H * d->W >= 500000 && d->C <= 64) return 0;
  return 2;
}

}  // namespace

extern "C" size_t asm_conv2d_wgrad_workspace_bytes(const asm_conv_desc* d) {
  if (!d) return 0;
  if (const int hb = wgrad_halo_blocks(d)) return (size_t)hb * d->K * 9 * d->C * sizeof(float);
  Plan pl = make_plan(d);
  if (pl.splits <= 1) return 0;
  return (size_t)pl.splits * d->K * d->R * d->S * d->C * sizeof(float);
}

extern "C" int asm_conv2d_wgrad_plan(const asm_conv_desc* d, int32_t plan[6]) {
  ASM_REQUIRE(d && plan, "conv wgrad plan: null pointer");
  if (const int hb = wgrad_halo_blocks(d)) {      // persistent halo form: {K, -1, 1, 1, workgroups, pixels per workgroup}
    plan[0] = d->K; plan[1] = -1; plan[2] = 1; plan[3] = 1; plan[4] = hb;
    plan[5] = (d->N * d->H * d->W + hb - 1) / hb;
    return ASM_OK;
  }
  const Plan pl = make_plan(d);
  plan[0] = pl.bnw; plan[1] = pl.bcw; plan[2] = pl.tiles_n; plan[3] = pl.tiles_c; plan[4] = pl.splits;
  plan[5] = pl.m_per_split;
  return ASM_OK;
}

extern "C" int asm_conv2d_wgrad(const asm_conv_desc* d, const void* x, const void* dy, float* dw,
                                void* workspace, size_t workspace_bytes, void* stream) {
  ASM_REQUIRE(d && x && dy && dw, "conv wgrad: null pointer");
  ASM_REQUIRE(d->C % 8 == 0, "conv wgrad: C=%d must be a multiple of 8", d->C);
  ASM_REQUIRE(d->stride == 1 || d->stride == 2, "conv wgrad: stride %d not supported", d->stride);
  const int ldy = d->ldy ? d->ldy : d->K;
  ASM_REQUIRE(ldy % 8 == 0 && ldy >= d->K, "conv wgrad: dy row stride %d must be a multiple of 8", ldy);
  const int64_t img_p = d->x_img_pitch ? d->x_img_pitch : (int64_t)d->H * d->W * d->C;
  const int64_t xelems = (int64_t)d->N * img_p;
  const int64_t dyelems = (int64_t)d->N * d->Ho * d->Wo * ldy;
  ASM_REQUIRE(xelems * 2 < (int64_t)ASM_OOB && dyelems * 2 < (int64_t)ASM_OOB, "conv wgrad: tensor larger than 2 GiB");
  Plan pl = make_plan(d);
  const size_t need = asm_conv2d_wgrad_workspace_bytes(d);
  ASM_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), "conv wgrad: workspace too small (%zu < %zu)",
              workspace_bytes, need);
  if (const int hb = wgrad_halo_blocks(d)) {
    WgradArgs h;
    const int kf = d->K == 128 ? 64 : d->K, kh = d->K / kf, pw = halo_patch_w(d);
    h.dy = dy; h.x = x; h.out = reinterpret_cast<float*>(workspace);
    h.dy_bytes = (unsigned)(dyelems * 2); h.x_bytes = (unsigned)(xelems * 2);
    h.M = d->N * d->H * d->W; h.Hi = d->H; h.Wi = d->W; h.Ci = d->C; h.Co = d->K; h.ldy = ldy;
    h.R = 3; h.S = 3; h.so = 1; h.pad = 1;
    h.x_img_pitch = d->H * d->W * d->C; h.x_row_pitch = d->W * d->C; h.x_pix_pitch = d->C;
    h.cols = 9 * d->C; h.tiles_n = kh; h.tiles_c = 1; h.splits = hb; h.m_per_split = 0;
    h.HoWo = d->H * d->W; h.Wo = d->W;
    h.fd_howo = make_fastdiv((unsigned)h.HoWo); h.fd_wo = make_fastdiv((unsigned)h.Wo);
    hipStream_t hs = (hipStream_t)stream;
#define LAUNCH_WH(KF, CI, PH, PW)                                                                                      \
    do {                                                                                                               \
      constexpr int LDS_ = WHalo<KF, CI, PH, PW>::LDS;                                                                 \
      static bool done_[ASM_MAX_DEVICES] = {};                                                                         \
      if (hipError_t e = asm_ensure_dyn_lds(wgrad_halo_kernel<KF, CI, PH, PW>, LDS_, done_); e != hipSuccess)          \
        ASM_FAIL(ASM_EHIP, "wgrad_halo_kernel: dynamic LDS opt-in: %s", hipGetErrorString(e));                         \
      ASM_LAUNCH((wgrad_halo_kernel<KF, CI, PH, PW>), dim3(hb * kh), dim3(WHalo<KF, CI, PH, PW>::NTHR), LDS_, hs, h);                          \
    } while (0)
#define LAUNCH_WH_GEO(KF, CI)                                  \
    do {                                                       \
      if (pw == 16) LAUNCH_WH(KF, CI, 8, 16);                  \
      else if (pw == 56) LAUNCH_WH(KF, CI, 2, 56);             \
      else LAUNCH_WH(KF, CI, 4, 28);                           \
    } while (0)
    if (kf == 32 && d->C == 64) LAUNCH_WH_GEO(32, 64);
    else if (kf == 64 && d->C == 32) LAUNCH_WH_GEO(64, 32);
    else if (kf == 32 && d->C == 32) LAUNCH_WH_GEO(32, 32);
    else LAUNCH_WH_GEO(64, 64);
#undef LAUNCH_WH_GEO
#undef LAUNCH_WH
    ASM_CHECK_LAUNCH("wgrad_halo_kernel");
    const size_t n = (size_t)d->K * 9 * d->C;
    ASM_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)cdivz(n, 128)), dim3(256), 0, hs,
                       reinterpret_cast<const float*>(workspace), dw, n, hb);
    ASM_CHECK_LAUNCH("wgrad_reduce_kernel");
    return ASM_OK;
  }
  WgradArgs a;
  a.dy = dy; a.x = x;
  a.out = pl.splits > 1 ? reinterpret_cast<float*>(workspace) : dw;
  a.dy_bytes = (unsigned)(dyelems * 2);
  a.x_bytes = (unsigned)(xelems * 2);
  a.M = d->N * d->Ho * d->Wo;
  a.Hi = d->H; a.Wi = d->W; a.Ci = d->C;
  a.Co = d->K; a.ldy = ldy;
  a.R = d->R; a.S = d->S; a.so = d->stride; a.pad = d->pad;
  a.x_img_pitch = (int)img_p;
  a.x_row_pitch = d->x_row_pitch ? d->x_row_pitch : d->W * d->C;
  a.x_pix_pitch = d->x_pix_pitch ? d->x_pix_pitch : d->C;
  a.cols = d->R * d->S * d->C;
  a.tiles_n = pl.tiles_n; a.tiles_c = pl.tiles_c; a.splits = pl.splits; a.m_per_split = pl.m_per_split;
  a.HoWo = d->Ho * d->Wo; a.Wo = d->Wo;
  a.fd_howo = make_fastdiv((unsigned)a.HoWo);
  a.fd_wo = make_fastdiv((unsigned)a.Wo);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(pl.tiles_n * pl.tiles_c * pl.splits);
  const bool lin = d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->Ho == d->H && d->Wo == d->W &&
                   a.x_pix_pitch == d->C && a.x_row_pitch == d->W * d->C && a.x_img_pitch == d->H * d->W * d->C;
  const int ring = lin && pl.bcw != 256 ? wgrad_ring_depth(d, pl) : 0;
  if (ring >= 2) {
#define LAUNCH_RING(BNW_, NS_)                                                                                         \
    do {                                                                                                               \
      constexpr int LDS_ = NS_ * (WPX * BNW_ * 2 + WPX * 256);                                                         \
      static bool done_[ASM_MAX_DEVICES] = {};                                                                         \
      if (hipError_t e = asm_ensure_dyn_lds(wgrad_kernel<BNW_, 128, true, NS_>, LDS_, done_); e != hipSuccess)         \
        ASM_FAIL(ASM_EHIP, "wgrad_kernel (ring): dynamic LDS opt-in: %s", hipGetErrorString(e));                       \
      ASM_LAUNCH((wgrad_kernel<BNW_, 128, true, NS_>), grid, dim3(256), LDS_, st, a);                                  \
    } while (0)
#define LAUNCH_RING_NS(NS_)                                      \
    do {                                                         \
      if (pl.bnw == 128) LAUNCH_RING(128, NS_);                  \
      else if (pl.bnw == 64) LAUNCH_RING(64, NS_);               \
      else LAUNCH_RING(32, NS_);                                 \
    } while (0)
    LAUNCH_RING_NS(2);
#undef LAUNCH_RING_NS
#undef LAUNCH_RING
  } else if (lin && pl.bcw != 256) {
    if (pl.bnw == 128) ASM_LAUNCH((wgrad_kernel<128, 128, true>), grid, dim3(256), 2 * (WPX * 256 + WPX * 256), st, a);
    else if (pl.bnw == 64) ASM_LAUNCH((wgrad_kernel<64, 128, true>), grid, dim3(256), 2 * (WPX * 128 + WPX * 256), st, a);
    else ASM_LAUNCH((wgrad_kernel<32, 128, true>), grid, dim3(256), 2 * (WPX * 64 + WPX * 256), st, a);
  } else if (pl.bcw == 256) {
    constexpr int LDS = 2 * 8 * 16 * 512;              // 128 KiB: two rings of eight 16-pixel blocks
    if (lin) {
      static bool attr_done_l[ASM_MAX_DEVICES] = {};
      if (hipError_t e = asm_ensure_dyn_lds(wgrad8_kernel<true>, LDS, attr_done_l); e != hipSuccess)
        ASM_FAIL(ASM_EHIP, "wgrad8_kernel: dynamic LDS opt-in: %s", hipGetErrorString(e));
      ASM_LAUNCH((wgrad8_kernel<true>), grid, dim3(512), LDS, st, a);
    } else {
      static bool attr_done[ASM_MAX_DEVICES] = {};
      if (hipError_t e = asm_ensure_dyn_lds(wgrad8_kernel<false>, LDS, attr_done); e != hipSuccess)
        ASM_FAIL(ASM_EHIP, "wgrad8_kernel: dynamic LDS opt-in: %s", hipGetErrorString(e));
      ASM_LAUNCH((wgrad8_kernel<false>), grid, dim3(512), LDS, st, a);
    }
  } else if (pl.bnw == 128) {
    ASM_LAUNCH((wgrad_kernel<128, 128>), grid, dim3(256), 2 * (WPX * 256 + WPX * 256), st, a);
  } else if (pl.bnw == 64) {
    ASM_LAUNCH((wgrad_kernel<64, 128>), grid, dim3(256), 2 * (WPX * 128 + WPX * 256), st, a);
  } else {
    ASM_LAUNCH((wgrad_kernel<32, 128>), grid, dim3(256), 2 * (WPX * 64 + WPX * 256), st, a);
  }
  ASM_CHECK_LAUNCH("wgrad_kernel");
  if (pl.splits > 1) {
    const size_t n = (size_t)d->K * a.cols;
    ASM_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)cdivz(n, 128)), dim3(256), 0, st,
                       reinterpret_cast<const float*>(workspace), dw, n, pl.splits);
    ASM_CHECK_LAUNCH("wgrad_reduce_kernel");
  }
  return ASM_OK;
}
