// igemm8_kernel: the 3x3 / stride-1 / pad-1 convolution (forward and input gradient) of the wide layers on a
// wave-staggered, multi-phase MFMA main loop.
//
// Same math, same operands and the same (chunk, tap, k) accumulation order as igemm2_kernel<256, 256, 64> (conv_igemm.hip) -- the
// results are bit-identical to it (tests/test_gpu_conv.py) -- on a different main loop.  igemm2's loop is lock-step: every wave
// issues the LDS-DMA of step k+1, reads its fragments, multiplies, then ALL of them drain vmcnt(0) and meet at one barrier per
// step.  Here
//   * the 256 x 256 x 64 step (tap t of a 64-channel chunk) is cut into 4 phases, each {fragment reads, ONE half-tile of
//     LDS-DMA, barrier, 8 MFMAs 32x32x16, barrier}.  A wave (wm, wn) owns pixels {h * 128 + wm * 64 + [0, 64)} x channels
//     {a * 128 + wn * 32 + [0, 32)}, h, a in {0, 1}.  Phases 1 / 2: pixel half h = 0 against BOTH filter halves over k-substeps
//     {0, 1} / {2, 3}; phases 3 / 4: pixel half 1 (the filter fragments of all four substeps are still in registers).  A phase's
//     8 MFMAs go round FOUR accumulators, so a dependent MFMA sits four issue slots (128 cycles) behind its producer;
//   * the two waves of a SIMD (waves w and w + 4: wm = 0 / 1) run ONE BARRIER APART, so in every barrier interval one of them
//     multiplies while the other reads LDS and issues DMA, and s_setprio around the MFMA segment lets it win the arbitration;
//   * the DMA never drains: a half-tile is requested 2 - 4 phases before the wait that retires it; the waits are counted
//     (vmcnt(6) in phase 2, vmcnt(4) in phase 4: two or three half-tiles stay in flight across every barrier), barriers raw.
// LDS: two 64 KB step buffers, each [filter rows 0-127 | pixel rows 0-127 | filter rows 128-255 | pixel rows 128-255], 128-byte
// rows, 16-byte pieces XOR-swizzled as in igemm2 (source-side swizzle: the DMA destination is lane-linear).
// Staging, one half-tile per phase: filter half 1, pixel half 0, pixel half 1 of step T+1 in phases 1 - 3 of step T, filter half 0
// of step T+2 in phase 4.
//   Write-after-read: the filter halves and pixel half 0 are last read in phase 2, pixel half 1 in phase 4; every slot is
//     re-staged at least two phases after its tenant's last read (the reads are retired by the lgkmcnt(0) behind the reading
//     phase's first barrier, and the staggered group is one barrier late: two phases = at least three barriers).
//   Read-after-write: vmcnt(6) in phase 2 retires pixel half 1 of THIS step (read from phase 3 on), vmcnt(4) in phase 4 both
//     filter halves and pixel half 0 of step T+1 (read from its phase 1 on): always one phase -- two barriers, so one that every
//     wave of BOTH groups has passed after its own wait -- between the wait and the first read.
// Addressing: a tap is a wave-uniform shift of the pixel offset, so it rides in the scalar offset of the buffer load (the
// resource base is moved back by one row + one pixel so that every shift is non-negative); what is per lane is the zero
// padding: nine validity bits per staged row, tested with one v_and / v_cmp / v_cndmask per DMA piece (out-of-image taps, the
// row tail and the channel tail fetch the out-of-range offset: the DMA zero-fills).  8 VGPRs instead of igemm2's 36; 227 in all.
// The steps past the end of the reduction are staged as zero-fill (never read): the counted waits need no tail case.
//
// Measured (round 6, MI355X, tools/conv_bench.py --iters 20, normal data; same box, interleaved):
//   64 x 16x16x512 -> 1024 (256 tiles = ONE round of the chip):   igemm2 119.5 us (1294 TFLOP/s)   igemm8 117.5 us (1316)
//   192 x 16x16x512 -> 1024 (768 tiles = three rounds):            igemm2 353.8 us                  igemm8 344.9 us
//   256 x 14x14x512 -> 1024 (784 tiles: three rounds + 16 tiles):  igemm2 474.6 / 410.4 us          igemm8 449.9 / 394.2 us  (fprop / dgrad)
//   the quadrant schedule of the guide's 8-phase template (2 accumulators per phase): 119.0 / 350.0 / 463.5 us; this schedule
//   WITHOUT the stagger (all waves in lock step): 132.7 / 397.6 / 539.3 us.
// SQ counters on the one-round shape (rocprofv3 --pmc): SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs = 147.5 k of 207.2 k cycles = 71 % of
// every SIMD's cycles, at an effective 1.72 GHz (GRBM_GUI_ACTIVE / duration): the chip is at its power limit -- 71 % of
// 2.5 PFLOP/s x 1.72 / 2.4 is the 1.3 PFLOP/s measured.  What the stagger buys in cycles the clock partly takes back
// (MI355X_MICROARCH.md "DVFS give-back"); what is left on this layer is the ragged last round, see the split in conv_igemm.hip.
// Reference: what this computes is tf.layers.conv2d of conv2d_fixed_padding (nets/model_helper.py:67-78) for the SK unit's
// 3x3 convolution to 2F channels (nets/blocks.py:113-116) and its input gradient.
#include "common.h"
#include "igemm_common.h"

using namespace asm_igemm;

namespace {

template <int N>
struct IC {
  static constexpr int value = N;
};

template <bool STATS, bool PFA, bool BNRED = false>
__global__ __launch_bounds__(512) void igemm8_kernel(IGemmArgs p) {
  using C = Cfg<256, 256, 64, 2, 4, false, STATS, 2>;
  constexpr int ROWB = 128, HALF = 128 * ROWB, TILE = 4 * HALF;
  constexpr int OFF_W0 = 0, OFF_X0 = HALF, OFF_W1 = 2 * HALF, OFF_X1 = 3 * HALF;
  constexpr int NTAP = 9, KK = 4;
  typedef __attribute__((address_space(3))) void* lptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;      // waves w and w + 4 share a SIMD: one of each group per SIMD

  int logical;
  {
    const int nb = p.n_blocks, q = nb >> 3, r = nb & 7;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = (int)fd_div((unsigned)logical, p.fd_ntn);
  const int tile_n = logical - tile_m * p.n_tiles_n;

  // tap (r, s) reads pixel (ho + dh, wo + dw), dh = tsign * r - pad: a shift of dh * row_pitch + dw * pix_pitch elements
  const int shift = p.x_row_pitch + p.x_pix_pitch;     // |most negative shift|
  const __amdgpu_buffer_rsrc_t rx =
      make_rsrc(reinterpret_cast<const unsigned char*>(p.x) - (size_t)shift * 2u, p.x_bytes + (unsigned)shift * 2u);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);

  const int chunk = tid & 7, r0 = tid >> 3;
  const int csw = (chunk ^ swz<64>(r0)) << 3;      // source-side swizzle (staging passes are 64 rows apart: the same for all)

  // ---- prologue: per staged row the byte offset of its pixel and the nine tap-validity bits; filter row offsets ----
  unsigned vb[4], vmk[4], vw[4];                   // q = half * 2 + pass: tile row q * 64 + r0
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = tile_m * 256 + q * 64 + r0;
    const bool live = m < p.M;
    const unsigned mm = live ? (unsigned)m : 0u;
    const unsigned img = fd_div(mm, p.fd_howo);
    const unsigned rem = mm - img * (unsigned)p.HoWo;
    const int ho = (int)fd_div(rem, p.fd_wo);
    const int wo = (int)rem - ho * p.Wo;
    vb[q] = (img * (unsigned)p.x_img_pitch + (unsigned)ho * (unsigned)p.x_row_pitch + (unsigned)wo * (unsigned)p.x_pix_pitch +
             (unsigned)csw) * 2u;
    unsigned mk = 0;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      const int dh = p.tsign * (t / 3) - p.pad, dw = p.tsign * (t % 3) - p.pad_w;
      const bool ok = live && (unsigned)(ho + dh) < (unsigned)p.Hi && (unsigned)(wo + dw) < (unsigned)p.Wi;
      mk |= (ok ? 1u : 0u) << t;
    }
    vmk[q] = mk;
    const int n = tile_n * 256 + q * 64 + r0;
    vw[q] = n < p.Co ? ((unsigned)n * (unsigned)p.w_row_pitch + (unsigned)csw) * 2u : ASM_OOB;
  }
  const unsigned tapw = (unsigned)p.Ci * 2u;       // bytes between consecutive taps of a filter row
  auto x_so = [&](const int t) -> unsigned {       // scalar offset of tap t (>= 0 after the base shift)
    const int dh = p.tsign * (t / 3) - p.pad, dw = p.tsign * (t % 3) - p.pad_w;
    return (unsigned)(dh * p.x_row_pitch + dw * p.x_pix_pitch + shift) * 2u;
  };
  auto w_so = [&](const int t) -> unsigned { return (unsigned)(p.wt0 + (t / 3) * p.wtr + (t % 3) * p.wts) * tapw; };

  const int wrow = wave * 8;
  // one half-tile = two LDS-DMA pieces per wave (rows wrow .. wrow + 7 of each 64-row pass)
  auto stage_x = [&](unsigned char* half, const int h, const int t, unsigned so, bool dead) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = h * 2 + j;
      unsigned mk = vmk[q];
      asm volatile("" : "+v"(mk));                 // opaque: keeps the per-tap select HERE (hoisted, it is 36 live registers)
      unsigned vo = (mk & (1u << t)) ? vb[q] : ASM_OOB;
      if (dead) vo = ASM_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(half + (j * 64 + wrow) * ROWB), 16, (int)vo, (int)so, 0, 0);
    }
  };
  auto stage_w = [&](unsigned char* half, const int h, unsigned so, bool dead) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = h * 2 + j;
      unsigned vo = vw[q];
      if (dead) vo = ASM_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(half + (j * 64 + wrow) * ROWB), 16, (int)vo, (int)so, 0, 0);
    }
  };

  f32x16 acc[2][4];                                // [filter half a][pixel half h * 2 + 32-row sub-tile b]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  const int l31 = lane & 31, lhi = lane >> 5;
  // fragment byte offsets inside a half-tile: rows 32 apart share the swizzle term, k-substeps differ by an XOR of 32 bytes
  unsigned fxo[KK], fwo[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const unsigned part = (unsigned)(((kk * 2 + lhi) ^ swz<64>(l31)) << 4);
    fxo[kk] = (unsigned)(wm * 64 + l31) * ROWB + part;
    fwo[kk] = (unsigned)(wn * 32 + l31) * ROWB + part;
  }
  bf16x8 fw0[KK], fw1[KK], fx[2][KK];

  // one step = tap T of the current chunk in LDS buffer CB (schedule: header)
  auto step = [&](auto cbc, auto tc, const unsigned kcb, const bool last) {
    constexpr int CB = decltype(cbc)::value, T = decltype(tc)::value;
    constexpr int T1 = (T + 1) % NTAP, T2 = (T + 2) % NTAP;
    constexpr bool WR1 = T + 1 >= NTAP, WR2 = T + 2 >= NTAP;
    const unsigned k1 = kcb + (WR1 ? 128u : 0u), k2 = kcb + (WR2 ? 128u : 0u);
    const bool dead1 = WR1 && last, dead2 = WR2 && last;
    unsigned char* const cur = smem + CB * TILE;
    unsigned char* const nxt = smem + (CB ^ 1) * TILE;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const int k0 = (ph & 1) * 2;            // k-substeps k0, k0 + 1
      const int h = ph >> 1;                  // pixel half
      if (h == 0) {
#pragma unroll
        for (int kk = k0; kk < k0 + 2; ++kk) {
          fw0[kk] = *reinterpret_cast<const bf16x8*>(cur + OFF_W0 + fwo[kk]);
          fw1[kk] = *reinterpret_cast<const bf16x8*>(cur + OFF_W1 + fwo[kk]);
        }
      }
#pragma unroll
      for (int kk = k0; kk < k0 + 2; ++kk)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          fx[b][kk] = *reinterpret_cast<const bf16x8*>(cur + (h ? OFF_X1 : OFF_X0) + fxo[kk] + b * 32 * ROWB);
      if (ph == 0) stage_w(nxt + OFF_W1, 1, w_so(T1) + k1, dead1);
      if (ph == 1) stage_x(nxt + OFF_X0, 0, T1, x_so(T1) + k1, dead1);
      if (ph == 2) stage_x(nxt + OFF_X1, 1, T1, x_so(T1) + k1, dead1);
      if (ph == 3) stage_w(cur + OFF_W0, 0, w_so(T2) + k2, dead2);
      if (ph == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      if (ph == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = k0; kk < k0 + 2; ++kk)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          acc[0][h * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw0[kk], fx[b][kk], acc[0][h * 2 + b], 0, 0, 0);
          acc[1][h * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw1[kk], fx[b][kk], acc[1][h * 2 + b], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
      if (h == 0) asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
      else asm volatile("" : "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][2]), "+v"(acc[1][3]));
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
  };
  // the nine taps of one chunk; the first step of the chunk sits in LDS buffer P0 (nine steps: the next chunk starts in P0 ^ 1)
  auto chunk9 = [&](auto p0c, const unsigned kcb, const bool last) {
    constexpr int P0 = decltype(p0c)::value;
    step(IC<P0>{}, IC<0>{}, kcb, last);
    step(IC<P0 ^ 1>{}, IC<1>{}, kcb, last);
    step(IC<P0>{}, IC<2>{}, kcb, last);
    step(IC<P0 ^ 1>{}, IC<3>{}, kcb, last);
    step(IC<P0>{}, IC<4>{}, kcb, last);
    step(IC<P0 ^ 1>{}, IC<5>{}, kcb, last);
    step(IC<P0>{}, IC<6>{}, kcb, last);
    step(IC<P0 ^ 1>{}, IC<7>{}, kcb, last);
    step(IC<P0>{}, IC<8>{}, kcb, last);
  };

  // ---- pipeline fill: step 0 whole and filter half 0 of step 1 (what phase 4 of a step -1 would have staged) ----
  {
    const bool dead = p.kchunks * NTAP < 2;      // never (nine taps), kept for the form
    stage_w(smem + OFF_W0, 0, w_so(0), false);
    stage_x(smem + OFF_X0, 0, 0, x_so(0), false);
    stage_w(smem + OFF_W1, 1, w_so(0), false);
    stage_x(smem + OFF_X1, 1, 0, x_so(0), false);
    stage_w(smem + TILE + OFF_W0, 0, w_so(1), dead);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");        // all of step 0 has landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (wm == 1) __builtin_amdgcn_s_barrier();      // the stagger: group 1 runs one barrier behind group 0

  unsigned kcb = 0;
#pragma unroll 1
  for (int kc = 0; kc < p.kchunks; kc += 2) {
    chunk9(IC<0>{}, kcb, kc + 1 >= p.kchunks);
    kcb += 128;
    if (kc + 1 < p.kchunks) {
      chunk9(IC<1>{}, kcb, kc + 2 >= p.kchunks);
      kcb += 128;
    }
  }

  if (wm == 0) __builtin_amdgcn_s_barrier();      // group 0 waits for group 1's last phase
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the zero-fill DMA of the steps past the end
  __syncthreads();                                 // the epilogue reuses the region
  igemm_epilogue<C, 256, 256, 128, 64, 4, 2, false, STATS, PFA, false, true, BNRED, 8>(p, acc, smem, tile_m, tile_n, tid, wm, wn, l31, lhi);
}

template <bool STATS, bool PFA, bool BNRED = false>
int launch8_one(const IGemmArgs& a, hipStream_t st) {
  using C = Cfg<256, 256, 64, 2, 4, false, STATS, 2>;
  constexpr int LDS = cmax(cmax(2 * 4 * 128 * 128, C::EPI), C::RED);
  static_assert(LDS <= 160 * 1024, "lds");
  auto kern = igemm8_kernel<STATS, PFA, BNRED>;
  static bool attr_done[ASM_MAX_DEVICES] = {};
  if (hipError_t e = asm_ensure_dyn_lds(kern, LDS, attr_done); e != hipSuccess)
    ASM_FAIL(ASM_EHIP, "igemm8_kernel: dynamic LDS opt-in: %s", hipGetErrorString(e));
  ASM_LAUNCH(kern, dim3(a.n_blocks), dim3(512), LDS, st, a);
  asm_last_conv_kernel = 8;
  ASM_CHECK_LAUNCH("igemm8_kernel");
  return ASM_OK;
}

}  // namespace

// is the layer one igemm8_kernel covers?
bool asm_igemm8_covers(const IGemmArgs& a, bool out_f32) {
  if (out_f32 || a.R != 3 || a.S != 3 || a.so != 1 || a.sd != 1 || a.y_strided || a.pool_dy) return false;
  if (!((a.tsign > 0 && a.pad == 1 && a.pad_w == 1) || (a.tsign < 0 && a.pad == -1 && a.pad_w == -1))) return false;
  if (a.Ci % 64 || a.Co % 8) return false;
  const long long span = (long long)a.x_bytes + 2ll * (a.x_row_pitch + a.x_pix_pitch);
  return span < (long long)ASM_OOB;
}

// returns 1 when the layer is not one igemm8_kernel covers (the caller goes on to igemm3 / igemm2)
int asm_igemm8_try(IGemmArgs& a, bool out_f32, bool stats, hipStream_t st) {
  if (!asm_igemm8_covers(a, out_f32)) return 1;
  a.n_tiles_n = cdiv(a.Co, 256);
  a.n_blocks = cdiv(a.M, 256) * a.n_tiles_n;
  a.kchunks = a.Ci / 64;
  a.fd_ntn = make_fastdiv((unsigned)a.n_tiles_n);
  const int pfa_env = asm_tune().igemm_pfa;
  const bool pfa = a.addend != nullptr && (pfa_env >= 0 ? pfa_env != 0 : true);
  if (stats && a.red_y) return launch8_one<true, false, true>(a, st);   // the batch-norm backward sums of an input gradient
  if (stats) return launch8_one<true, false>(a, st);
  if (pfa) return launch8_one<false, true>(a, st);
  return launch8_one<false, false>(a, st);
}
