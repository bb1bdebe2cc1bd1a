// igemm1_kernel: the 1x1 convolutions (forward and input gradient) as a GEMM with a RING of LDS stages.
//
//   Y[m][n] = sum_c X[row(m)][c] * Wt[n][c]        fprop: Wt = filter KRSC (R = S = 1), dgrad: Wt = the CRSK copy
//
// conv1 / conv3 of every bottleneck, the projection shortcuts and the BigLittle transition layers
// (nets/resnet_model.py:55-60,76-80, nets/model_helper.py:67-78) are this GEMM with M = N*H*W pixel rows: 159 launches and
// as much time per step as the 3x3 class.  The kernel was written on the hypothesis that igemm2_kernel's two LDS stages leave
// every K step of the 14 x 14 / 7 x 7 layers behind an exposed LDS-DMA round trip (SQ counters, profiles/round4_c_sq_counters.md:
// matrix pipe 9 - 15 % busy, 49 - 65 % of a wave's life parked on s_waitcnt): a step's tiles are requested NS - 1 steps ahead,
// `s_waitcnt vmcnt(P * (NS - 2))` (P = LDS-DMA pieces per wave and step) waits for the OLDEST step only, a raw s_barrier
// publishes it, and the requests of the following steps stay in flight across the barrier.
//
// The sweeps refuted the hypothesis and found something else (profiles/round5_gemm1_sweep*.md): with 3 - 4
// stages of 128 x 128 x 64 tiles (96 - 128 KB of LDS: ONE workgroup per CU) every layer is 20 - 60 % SLOWER than with two stages
// and two workgroups per CU.  A workgroup of these layers spends its life in prologue (address decode, first round trip),
// 2 - 32 K steps and epilogue (LDS staging, 8 - 16 store passes, the statistics' barriers); what hides one workgroup's
// prologue and epilogue is another workgroup's K loop on the same CU.  The configurations that win are the SMALL ones --
// 24 - 72 KB, three to six workgroups per CU -- and the template below exists to provide them per layer (asm_gemm1_try's
// table); beside the weight-gradient streams' kernels of the training step they gain 2.5 x what they gain alone.
// Same tiles, fragment layout, (chunk, k) accumulation order and epilogue as igemm2_kernel: bit-identical results.
#include "common.h"
#include "igemm_common.h"

using namespace asm_igemm;

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_waitcnt vmcnt(P * min(r, RMAX)) for a wave-uniform run-time r >= 0 (the tail of the K loop)
template <int P, int RMAX>
__device__ __forceinline__ void wait_vm_steps(const int r) {
  if constexpr (RMAX <= 0) {
    wait_vm<0>();
  } else {
    if (r >= RMAX) wait_vm<P * RMAX>();
    else wait_vm_steps<P, RMAX - 1>(r);
  }
}

template <int BM, int BN, int BK, int WGM, int WGN, int NS>
struct Cfg1 {
  using C = Cfg<BM, BN, BK, WGM, WGN, false, true, 2>;
  static constexpr int STAGE = (BM + BN) * C::ROWB;
  static constexpr int LDS = cmax(cmax(NS * STAGE, C::EPI), C::RED);
  static constexpr int P = C::XP + C::WP;           // LDS-DMA pieces per wave and step
  static_assert(BN % C::RPP == 0, "every wave issues the same number of pieces (the vmcnt immediates count them)");
  static_assert(C::RPP % 16 == 0, "the source-side swizzle must not depend on the pass");
  static_assert(NS >= 2 && P * (NS - 2) <= 63, "vmcnt is a 6-bit counter");
  static_assert(LDS <= 160 * 1024, "lds");
};

template <int BM, int BN, int BK, int WGM, int WGN, bool STATS, int NS, bool PFA, bool POOL, bool BNRED = false>
__global__ __launch_bounds__(64 * WGM * WGN) void igemm1_kernel(IGemmArgs p) {
  using C = Cfg<BM, BN, BK, WGM, WGN, false, STATS, 2>;
  using C1 = Cfg1<BM, BN, BK, WGM, WGN, NS>;
  constexpr int CPR = C::CPR, RPP = C::RPP, XP = C::XP, WP = C::WP, ROWB = C::ROWB, STAGE = C1::STAGE, P = C1::P;
  constexpr int TM = C::TM, TN = C::TN, WTM = C::WTM, WTN = C::WTN;
  constexpr int KK = BK / 16;
  static_assert(KK >= 2 && KK % 2 == 0, "fragment double buffer: the last group of a step lives in buffer 1");

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
  const int tile_m = (int)fd_div((unsigned)logical, p.fd_ntn);
  const int tile_n = logical - tile_m * p.n_tiles_n;

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);

  const int chunk = tid % CPR;
  const int r0 = tid / CPR;
  const int csw = (chunk ^ swz<BK>(r0)) << 3;   // channel (element) offset of the 16-byte piece this lane fetches
  const bool c_ok = csw < p.Ci;                  // only a single-chunk layer can have a channel tail

  // ---- prologue: per-row byte offsets of the activation tile (stride / parity-class decode as in igemm2), per-row offsets
  // of the filter tile ----
  unsigned vx[XP];
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int m = tile_m * BM + r0 + j * RPP;
    bool ok = c_ok && m < p.M;
    const unsigned img = fd_div((unsigned)m, p.fd_howo);
    const unsigned rem = (unsigned)m - img * (unsigned)p.HoWo;
    const unsigned ho = fd_div(rem, p.fd_wo);
    const unsigned wo = rem - ho * (unsigned)p.Wo;
    int nh = (int)ho * p.so - p.pad, nw = (int)wo * p.so - p.pad_w;
    if (p.sd == 2) {
      ok = ok && (((nh | nw) & 1) == 0);
      nh >>= 1;
      nw >>= 1;
    }
    ok = ok && ((unsigned)nh < (unsigned)p.Hi) && ((unsigned)nw < (unsigned)p.Wi);
    vx[j] = ok ? (img * (unsigned)p.x_img_pitch + (unsigned)nh * (unsigned)p.x_row_pitch +
                  (unsigned)nw * (unsigned)p.x_pix_pitch + (unsigned)csw) * 2u
               : ASM_OOB;
  }
  unsigned vw[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const int n = tile_n * BN + r0 + j * RPP;
    vw[j] = (c_ok && n < p.Co) ? ((unsigned)n * (unsigned)p.w_row_pitch + (unsigned)csw) * 2u : ASM_OOB;
  }

  const int wrow0 = wave * (64 / CPR);
  const unsigned wtap = (unsigned)p.wt0 * (unsigned)p.Ci * 2u;   // the filter tap of a parity class (0 otherwise)
  auto issue = [&](int stage, unsigned kb) {                     // kb = byte offset of the channel chunk
    unsigned char* xs = smem + stage * STAGE;
    unsigned char* ws = xs + BM * ROWB;
#pragma unroll
    for (int j = 0; j < XP; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(xs + (j * RPP + wrow0) * ROWB), 16, (int)vx[j], (int)kb, 0, 0);
#pragma unroll
    for (int j = 0; j < WP; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(ws + (j * RPP + wrow0) * ROWB), 16, (int)vw[j],
                                               (int)(kb + wtap), 0, 0);
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
    fwo[kk] = BM * ROWB + rw_ * ROWB + ((ch ^ swz<BK>(rw_)) << 4);
    fxo[kk] = rx_ * ROWB + ((ch ^ swz<BK>(rx_)) << 4);
  }
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

  // ---- ring: steps 0 .. NS-2 requested up front; step k waits for ITS pieces only, then requests step k + NS - 1 into the
  // stage step k - 1 has just released (every wave's reads of it are in registers before it reaches the barrier) ----
  const int nk = p.kchunks;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) issue(s, (unsigned)(s * BK * 2));
#pragma unroll
  for (int a = 0; a < TN; ++a) fwb[1][a] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};   // the first "previous group" adds 0
#pragma unroll
  for (int b = 0; b < TM; ++b) fxb[1][b] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  int cur = 0;
  unsigned kb_next = (unsigned)((NS - 1) * BK * 2);
#pragma unroll 1
  for (int kc = 0; kc < nk; ++kc) {
    wait_vm_steps<P, NS - 2>(nk - 1 - kc);      // the steps requested after this one may stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // one statement: no LDS access moves across it
    load_frags(cur, 0, 0);
    mma(1);                                     // last group of the previous step, under this step's first fragment reads
    if (kc + NS - 1 < nk) {
      int nst = cur + NS - 1;
      if (nst >= NS) nst -= NS;
      issue(nst, kb_next);
    }
#pragma unroll
    for (int kk = 0; kk + 1 < KK; ++kk) {
      load_frags(cur, kk + 1, (kk + 1) & 1);
      mma(kk & 1);
    }
    cur = cur + 1 == NS ? 0 : cur + 1;
    kb_next += BK * 2;
  }
  mma(1);
  __syncthreads();   // every wave's fragment reads are done: the epilogue reuses the region

  igemm_epilogue<C, BM, BN, WTM, WTN, TM, TN, false, STATS, PFA, POOL, false, BNRED>(p, acc, smem, tile_m, tile_n, tid, wm, wn, l31, lhi);
}

template <int BM, int BN, int BK, int WGM, int WGN, bool STATS, int NS, bool PFA, bool POOL, bool BNRED = false>
int launch1_one(const IGemmArgs& a, hipStream_t st) {
  using C1 = Cfg1<BM, BN, BK, WGM, WGN, NS>;
  auto kern = igemm1_kernel<BM, BN, BK, WGM, WGN, STATS, NS, PFA, POOL, BNRED>;
  static bool attr_done[ASM_MAX_DEVICES] = {};
  if (hipError_t e = asm_ensure_dyn_lds(kern, C1::LDS, attr_done); e != hipSuccess)
    ASM_FAIL(ASM_EHIP, "igemm1_kernel: dynamic LDS opt-in: %s", hipGetErrorString(e));
  ASM_LAUNCH(kern, dim3(a.n_blocks), dim3(64 * WGM * WGN), C1::LDS, st, a);
  asm_last_conv_kernel = 1;
  ASM_CHECK_LAUNCH("igemm1_kernel");
  return ASM_OK;
}

// returns 1 if this tile does not fit the layer
template <int BM, int BN, int BK, int WGM, int WGN, int NS>
int launch1_cfg(IGemmArgs& a, bool stats, hipStream_t st) {
  if (a.Ci % BK != 0 && a.Ci > BK) return 1;
  a.n_tiles_n = cdiv(a.Co, BN);
  a.n_blocks = cdiv(a.M, BM) * a.n_tiles_n;
  a.kchunks = cdiv(a.Ci, BK);
  a.fd_ntn = make_fastdiv((unsigned)a.n_tiles_n);
  const int pfa_env = asm_tune().igemm_pfa;   // the addend-prefetching epilogue, chosen as igemm2 does (launch2_cfg)
  const bool pfa = a.addend != nullptr && !a.y_strided && (pfa_env >= 0 ? pfa_env != 0 : (BM == 256 || a.n_blocks <= 1024));
  if (stats && a.red_y) return launch1_one<BM, BN, BK, WGM, WGN, true, NS, false, false, true>(a, st);
  if (stats) return launch1_one<BM, BN, BK, WGM, WGN, true, NS, false, false>(a, st);
  if (a.pool_dy) return launch1_one<BM, BN, BK, WGM, WGN, false, NS, false, true>(a, st);
  if (pfa) return launch1_one<BM, BN, BK, WGM, WGN, false, NS, true, false>(a, st);
  return launch1_one<BM, BN, BK, WGM, WGN, false, NS, false, false>(a, st);
}

// Per-layer choice (asm_tuning.gemm1 = -1) for the 1x1 stride-1 layers of Assemble-ResNet-50 at batch 256 -- 65 of its 69
// (layer, kind) pairs; everything else (other batch sizes, other networks) stays with igemm2.  Built in three steps, every
// candidate bit-identical to igemm2:
//   1. the same-box stand-alone sweep (tools/gemm1_sweep.py, profiles/round5_gemm1_sweep.md; warm + cold HIP-event time):
//      rows where a configuration beat igemm2's choice by >= 4 %   -> whole step 25.25 -> 24.2 - 24.4 ms
//   2. + rows where the best other configuration TIED igemm2 stand-alone (within 3 %): beside the weight-gradient streams
//      the smaller footprint wins them                            -> -0.15 .. -0.25 ms
//   3. 24 rows re-picked from IN-SITU times (tools/insitu_sweep.py: every configuration forced in turn on all 1x1 layers of
//      an eager side-stream step, HIP events around every launch; a row changed where another configuration was >= 5 %
//      faster in place)                                            -> the same step time, better per-layer numbers
// kind: 0 forward with fused statistics, 1 input gradient, 2 input gradient with a fan-in addend.
struct AutoRow {
  int kind, M, Ci, Co, code;
};
constexpr AutoRow kAuto[] = {
    {0, 802816,   32,   64, 13}, {0, 802816,   32,  128, 13}, {0, 802816,   64,   32, 12}, {0, 802816,   64,  128, 10},
    {0, 802816,  128,  256, 10}, {0, 802816,  256,   64, 12}, {0, 200704,   64,  256, 10}, {0, 200704,  256,   64, 12},
    {0, 200704,  256,  128,  5}, {0, 200704,  256,  256,  5}, {0, 200704,  256,  512, 10}, {0, 200704,  512,  128,  5},
    {0,  50176,  128,  512, 10}, {0,  50176,  256,  512, 10}, {0,  50176,  256, 1024, 10}, {0,  50176,  512,  128, 14},
    {0,  50176,  512,  256, 16}, {0,  50176,  512,  512, 10}, {0,  50176,  512, 1024, 10}, {0,  50176, 1024,  256, 16},
    {0,  50176, 1024,  512, 16}, {0,  50176, 1024, 1024, 14}, {0,  12544,  256, 1024, 16}, {0,  12544,  512, 1024, 14},
    {0,  12544,  512, 2048, 10}, {0,  12544, 1024,  256,  5}, {0,  12544, 2048,  512,  8}, {1, 802816,   32,   64, 12},
    {1, 802816,   64,   32, 12}, {1, 802816,   64,   64, 11}, {1, 802816,   64,  256, 16}, {1, 802816,  128,   32, 13},
    {1, 802816,  128,   64, 13}, {1, 802816,  256,  128,  5}, {1, 200704,   64,  256,  5}, {1, 200704,  128,  256,  5},
    {1, 200704,  128,  512,  5}, {1, 200704,  256,   64, 12}, {1, 200704,  256,  256, 10}, {1, 200704,  512,  256,  5},
    {1,  50176,  128,  512, 10}, {1,  50176,  256,  512, 10}, {1,  50176,  256, 1024, 14}, {1,  50176,  512,  128,  1},
    {1,  50176,  512,  512, 10}, {1,  50176,  512, 1024, 10}, {1,  50176, 1024,  256, 16}, {1,  50176, 1024,  512, 14},
    {1,  50176, 1024, 1024, 10}, {1,  12544,  256, 1024, 14}, {1,  12544,  512, 2048, 10}, {1,  12544, 1024,  256,  8},
    {1,  12544, 1024,  512,  8}, {1,  12544, 2048,  512,  8}, {1,  12544, 2048, 1024, 16}, {2, 802816,   64,  256, 14},
    {2, 200704,   64,  256, 13}, {2, 200704,  128,  256, 12}, {2, 200704,  128,  512, 14}, {2,  50176,  128,  512, 12},
    {2,  50176,  256,  512, 12}, {2,  50176,  256, 1024, 13}, {2,  50176,  512, 1024, 14}, {2,  12544,  256, 1024, 14},
    {2,  12544,  512, 2048, 14},
    // shapes of the other BASELINE configurations that Assemble-ResNet-50 + D does not have: ResNet-50 v1.5 at batch 256 (config 2:
    // stand-alone wins >= 4 % and ties within 3 %) and Assemble-ResNet-152 at batch 128 (config 5 shard: wins >= 4 % only -- its ties
    // LOST in the step); profiles/round5_gemm1_sweep_r50.md, _r152.md
    {0, 802816,   64,  256, 10}, {1, 802816,  256,   64, 11}, {0, 802816,  256,  128, 14}, {1, 802816,  128,  256, 14},
    {2, 802816,  128,  256, 14}, {0, 200704,  128,  512, 10}, {1, 200704,  512,  128,  5}, {0, 200704,  512,  256,  5},
    {1, 200704,  256,  512, 10}, {2, 200704,  256,  512, 14}, {0, 401408,   64,   64, 12}, {1, 100352,  256,   64, 12},
    {0, 100352,  256,   64, 12}, {1, 100352,   64,  256,  5}, {2, 100352,   64,  256, 13}, {0, 401408,   64,  256, 10},
    {0, 401408,  256,  256,  5}, {1, 401408,  256,  256, 14}, {0, 100352,  256,  256, 10}, {1, 100352,  256,  256, 10},
    {0, 401408,  256,   64, 12}, {1, 401408,   64,  256,  5}, {2, 401408,   64,  256, 14}, {0,  25088,  256,  512, 16},
    {1,  25088,  512,  256,  8}, {0, 100352,  256,  128, 14}, {1, 100352,  128,  256, 12}, {2, 100352,  128,  256, 12},
    {0,  25088,  128,  512, 10}, {1,  25088,  512,  128, 11}, {0,  25088,  512,  128,  5}, {1,  25088,  128,  512, 14},
    {2,  25088,  128,  512, 12}, {0, 100352,  256,  512, 10}, {0, 100352,  128,  512, 10}, {1, 100352,  512,  128, 14},
    {1, 100352,  128,  512, 10}, {2, 100352,  128,  512, 12}, {0, 100352,  512,  512, 14}, {1, 100352,  512,  512, 14},
    {1,   6272, 1024,  512,  5}, {0,  25088,  512,  256,  8}, {1,  25088,  256,  512, 14}, {2,  25088,  256,  512, 14},
    {1,   6272, 1024,  256, 12}, {0,   6272, 1024,  256, 11}, {1,   6272,  256, 1024,  8}, {0,  25088,  256, 1024, 10},
    {1,  25088, 1024,  256,  8}, {0,  25088, 1024,  256,  8}, {1,  25088,  256, 1024, 10}, {2,  25088,  256, 1024, 10},
    {1,   6272, 2048, 1024,  8}, {1,   6272, 2048,  512,  5}, {0,   6272, 2048,  512,  5},
};

int auto_cfg(const IGemmArgs& a, bool stats) {
  if (a.so != 1 || a.sd != 1 || a.y_strided || a.pool_dy) return 0;      // the stride-1 layers the sweep covered
  const int kind = a.tsign < 0 ? (a.addend ? 2 : 1) : (stats ? 0 : -1);     // (an input gradient with statistics: asm_conv2d_dgrad_bnred)
  for (const AutoRow& r : kAuto)
    if (r.kind == kind && r.M == a.M && r.Ci == a.Ci && r.Co == a.Co) return r.code;
  return 0;
}

}  // namespace

// returns 1 when the layer stays with igemm2_kernel; a.fd_howo / a.fd_wo are set by the caller
int asm_gemm1_try(IGemmArgs& a, bool out_f32, bool stats, hipStream_t st) {
  if (out_f32 || a.R != 1 || a.S != 1 || a.bn_scale) return 1;
  const int g = asm_tune().gemm1;
  const int code = g < 0 ? auto_cfg(a, stats) : g;
  switch (code) {
    // LDS per workgroup -> workgroups per CU.  Round-5 sweep (tools/gemm1_sweep.py, profiles/round5_gemm1_sweep.md): what
    // pays on these layers is MORE RESIDENT WORKGROUPS (one's prologue / epilogue under the others' K loops), not a deeper
    // ring: 128 x 128 x 64 with 3 or 4 stages (one workgroup per CU) LOSES 20 - 60 % against two stages (two per CU).
    case 1: return launch1_cfg<128, 128, 64, 2, 2, 2>(a, stats, st);   //  64 KB: 2 per CU (igemm2's tile and depth)
    case 5: return launch1_cfg<128, 128, 32, 2, 2, 3>(a, stats, st);   //  48 KB: 3
    case 8: return launch1_cfg<256, 128, 64, 4, 2, 3>(a, stats, st);   // 144 KB: 1 (8 waves)
    case 10: return launch1_cfg<128, 128, 32, 2, 2, 2>(a, stats, st);  //  34 KB: 4
    case 11: return launch1_cfg<128, 64, 64, 2, 2, 2>(a, stats, st);   //  48 KB: 3
    case 12: return launch1_cfg<128, 64, 32, 2, 2, 3>(a, stats, st);   //  36 KB: 4
    case 13: return launch1_cfg<128, 64, 32, 2, 2, 2>(a, stats, st);   //  24 KB: 6
    case 14: return launch1_cfg<256, 128, 32, 4, 2, 3>(a, stats, st);  //  72 KB: 2 (8 waves each)
    case 16: return launch1_cfg<128, 256, 32, 2, 2, 2>(a, stats, st);  //  66 KB: 2 (4 waves of 64 x 128)
    default: return 1;
  }
}
