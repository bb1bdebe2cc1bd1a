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
// Global -> LDS staging goes through registers with raw buffer loads: an out-of-image tap or a
// channel tail simply gets an out-of-range offset and the hardware returns zeros (branch-free
// zero padding).  LDS rows are XOR-swizzled at 16-byte granularity so the ds_read_b128 fragment
// reads are bank-conflict free.
#include "common.h"

namespace {

struct IGemmArgs {
  const void* x;
  const void* w;
  void* y;
  float* stats;
  unsigned x_bytes, w_bytes;
  int M;           // number of output rows
  int Hi, Wi, Ci;  // gathered tensor dims
  int Wo, HoWo;    // output spatial decode
  int Co;          // N dimension (valid)
  int ldy;         // output row stride (elements)
  int R, S;
  int so, sd, tsign, pad;  // num = o*so + tsign*t - pad ; valid iff num % sd == 0 ; idx = num / sd
  int x_img_pitch, x_row_pitch, x_pix_pitch;  // elements
  int w_row_pitch;  // elements between consecutive n rows of Wt (= R*S*Ci)
  int n_tiles_n, n_blocks, kchunks;
};

constexpr int BM = 128;
constexpr int LDS_BYTES = 65536;

template <int BK>
__device__ __forceinline__ int swz(int row) {
  // BK=64: 8 chunks / 128-B row ; BK=32: 4 chunks / 64-B row.  16 rows that are distinct mod 16
  // land on 16 distinct 16-byte slots of the 256-byte bank row.
  return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

template <int BN, int BK, bool OUT_F32, bool STATS>
__global__ __launch_bounds__(256) void igemm_kernel(IGemmArgs p) {
  constexpr int CPR = BK / 8;           // 16-byte chunks per tile row
  constexpr int RPP = 256 / CPR;        // rows staged per pass by the 256 threads
  constexpr int XP = BM / RPP;          // passes for the activation tile
  constexpr int WP = BN / RPP;          // passes for the filter tile
  constexpr int ROWB = BK * 2;          // bytes per tile row
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int TN = BN / 64;           // 32-wide filter tiles per wave
  constexpr int TM = 2;                 // 32-wide pixel tiles per wave
  // LDS is sized per instantiation (main-loop double buffer vs the epilogue's output tile) so that
  // the small-tile variants run 3-6 workgroups per CU: the short-K layers are latency-bound and
  // need the extra waves to hide the global-load round trip of every K-step.
  constexpr int LDO = BN * 2 + 16;  // padded output-tile row (bytes)
  constexpr int EPI = OUT_F32 ? 0 : BM * LDO;
  constexpr int LDS = (2 * STAGE > EPI) ? 2 * STAGE : EPI;
  static_assert(LDS <= LDS_BYTES, "lds");

  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

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
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const int n = tile_n * BN + r0 + j * RPP;
    wb[j] = (n < p.Co) ? n * p.w_row_pitch : -1;
  }

  u32x4 xr[XP], wr[WP];
  int kt_r = 0, kt_s = 0, kt_c = 0;  // tap / channel-chunk cursor of the NEXT tile to load

  auto load_tile = [&]() {
    const int c = kt_c * BK + chunk * 8;
    const bool cok = c < p.Ci;
    const int th = p.tsign * kt_r, tw = p.tsign * kt_s;
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      int nh = bh[j] + th, nw = bw[j] + tw;
      bool ok = cok;
      if (p.sd == 2) {
        ok = ok && (((nh | nw) & 1) == 0);
        nh >>= 1;
        nw >>= 1;
      }
      ok = ok && ((unsigned)nh < (unsigned)p.Hi) && ((unsigned)nw < (unsigned)p.Wi);
      const unsigned off = ((unsigned)xb[j] + (unsigned)nh * (unsigned)p.x_row_pitch +
                            (unsigned)nw * (unsigned)p.x_pix_pitch + (unsigned)c) * 2u;
      xr[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : ASM_OOB, 0, 0);
    }
    const int tap_off = (kt_r * p.S + kt_s) * p.Ci + c;
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const bool ok = cok && (wb[j] >= 0);
      const unsigned off = (unsigned)(wb[j] + tap_off) * 2u;
      wr[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? off : ASM_OOB, 0, 0);
    }
    // advance cursor
    if (++kt_c == p.kchunks) {
      kt_c = 0;
      if (++kt_s == p.S) {
        kt_s = 0;
        ++kt_r;
      }
    }
  };

  auto store_tile = [&](int stage) {
    unsigned char* xs = smem + stage * STAGE;
    unsigned char* ws = xs + BM * ROWB;
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int row = r0 + j * RPP;
      *reinterpret_cast<u32x4*>(xs + row * ROWB + ((chunk ^ swz<BK>(row)) << 4)) = xr[j];
    }
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const int row = r0 + j * RPP;
      *reinterpret_cast<u32x4*>(ws + row * ROWB + ((chunk ^ swz<BK>(row)) << 4)) = wr[j];
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

  load_tile();
  store_tile(0);
  __syncthreads();

  const int l31 = lane & 31, lhi = lane >> 5;
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) load_tile();

    const unsigned char* xs = smem + cur * STAGE;
    const unsigned char* ws = xs + BM * ROWB;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const int ch = kk * 2 + lhi;
      bf16x8 fw[TN], fx[TM];
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        const int row = wn * (BN / 2) + a * 32 + l31;
        fw[a] = *reinterpret_cast<const bf16x8*>(ws + row * ROWB + ((ch ^ swz<BK>(row)) << 4));
      }
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int row = wm * 64 + b * 32 + l31;
        fx[b] = *reinterpret_cast<const bf16x8*>(xs + row * ROWB + ((ch ^ swz<BK>(row)) << 4));
      }
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[a], fx[b], acc[a][b], 0, 0, 0);
    }

    if (kt + 1 < KT) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  // acc[a][b][reg]: n_local = wn*(BN/2) + a*32 + (reg&3) + 8*(reg>>2) + 4*lhi ; m_local = wm*64 + b*32 + l31
  if constexpr (OUT_F32) {
    float* y = reinterpret_cast<float*>(p.y);
    const int co4 = (p.Co + 3) & ~3;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int m = tile_m * BM + wm * 64 + b * 32 + l31;
        if (m < p.M) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = tile_n * BN + wn * (BN / 2) + a * 32 + 8 * g + 4 * lhi;
            if (n < co4) {
              f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
              *reinterpret_cast<f32x4*>(y + (size_t)m * p.ldy + n) = v;
            }
          }
        }
      }
  } else {
    constexpr int CPO = BN / 8;       // 16-byte chunks per output row
    constexpr int RPO = 256 / CPO;    // rows per pass
    constexpr int OP = BM / RPO;      // passes
    static_assert(RPO * BN * 2 * 4 <= BM * LDO, "stats scratch aliases the output tile");
    unsigned char* os = smem;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int ml = wm * 64 + b * 32 + l31;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = wn * (BN / 2) + a * 32 + 8 * g + 4 * lhi;
          u32x2 v;
          v.x = pack2bf(acc[a][b][4 * g], acc[a][b][4 * g + 1]);
          v.y = pack2bf(acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]);
          *reinterpret_cast<u32x2*>(os + ml * LDO + nl * 2) = v;
        }
      }
    __syncthreads();
    bf16_t* y = reinterpret_cast<bf16_t*>(p.y);
    const int oc = tid % CPO, orow = tid / CPO;
    const int n0 = tile_n * BN + oc * 8;
    const int co8 = (p.Co + 7) & ~7;
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
#pragma unroll
    for (int ps = 0; ps < OP; ++ps) {
      const int row = ps * RPO + orow;
      const int m = tile_m * BM + row;
      const u32x4 v = *reinterpret_cast<const u32x4*>(os + row * LDO + oc * 16);
      if (m < p.M && n0 < co8) *reinterpret_cast<u32x4*>(y + (size_t)m * p.ldy + n0) = v;
      if constexpr (STATS) {
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s[e] += f[e];
          ss[e] += f[e] * f[e];
        }
      }
    }
    if constexpr (STATS) {
      float* red = reinterpret_cast<float*>(smem);  // [RPO][2][BN], aliases the (now consumed) output tile
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(orow * 2 + 0) * BN + oc * 8 + e] = s[e];
        red[(orow * 2 + 1) * BN + oc * 8 + e] = ss[e];
      }
      __syncthreads();
      if (tid < 2 * BN) {
        const int which = tid / BN, nl = tid - which * BN;
        float t = 0.f;
#pragma unroll 8
        for (int g = 0; g < RPO; ++g) t += red[(g * 2 + which) * BN + nl];
        const int n = tile_n * BN + nl;
        if (n < p.Co) p.stats[((size_t)tile_m * 2 + which) * p.Co + n] = t;
      }
    }
  }
}

template <int BN, int BK>
int launch_cfg(const IGemmArgs& a, bool out_f32, bool stats, hipStream_t st) {
  dim3 grid(a.n_blocks), block(256);
  if (out_f32)
    hipLaunchKernelGGL((igemm_kernel<BN, BK, true, false>), grid, block, 0, st, a);
  else if (stats)
    hipLaunchKernelGGL((igemm_kernel<BN, BK, false, true>), grid, block, 0, st, a);
  else
    hipLaunchKernelGGL((igemm_kernel<BN, BK, false, false>), grid, block, 0, st, a);
  ASM_CHECK_LAUNCH("igemm_kernel");
  return ASM_OK;
}

int launch(IGemmArgs& a, bool out_f32, bool stats, hipStream_t st) {
  const int bn = (a.Co <= 64) ? 64 : 128;
  // BK=64 halves the barrier count for the MFMA-heavy 3x3 / 7x7 layers; the 1x1 layers are HBM-bound
  // with very short K loops, where the smaller BK=32 footprint (3-6 workgroups per CU) hides latency better
  const int bk = (a.Ci % 64 == 0 && a.R * a.S > 1) ? 64 : 32;
  a.n_tiles_n = cdiv(a.Co, bn);
  a.n_blocks = cdiv(a.M, BM) * a.n_tiles_n;
  a.kchunks = cdiv(a.Ci, bk);
  if (bn == 64 && bk == 64) return launch_cfg<64, 64>(a, out_f32, stats, st);
  if (bn == 64 && bk == 32) return launch_cfg<64, 32>(a, out_f32, stats, st);
  if (bn == 128 && bk == 64) return launch_cfg<128, 64>(a, out_f32, stats, st);
  return launch_cfg<128, 32>(a, out_f32, stats, st);
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

extern "C" int asm_conv2d_stats_blocks(const asm_conv_desc* d) {
  if (!d) return ASM_EINVAL;
  return cdiv(d->N * d->Ho * d->Wo, BM);
}

extern "C" int asm_conv2d_fprop(const asm_conv_desc* d, const void* x, const void* w, void* y,
                                float* stats_partial, void* stream) {
  if (int e = check_desc(d)) return e;
  ASM_REQUIRE(x && w && y, "conv fprop: null pointer");
  const int64_t xelems = (int64_t)d->N * img_pitch(d);
  ASM_REQUIRE(xelems * 2 < (int64_t)ASM_OOB, "conv fprop: input larger than 2 GiB");
  const int ldy = d->ldy ? d->ldy : d->K;
  ASM_REQUIRE(ldy % (d->out_f32 ? 4 : 8) == 0 && ldy >= d->K, "conv fprop: bad ldy %d", ldy);
  ASM_REQUIRE(!(stats_partial && d->out_f32), "conv fprop: fused statistics need bf16 output");
  IGemmArgs a;
  a.x = x; a.w = w; a.y = y; a.stats = stats_partial;
  a.x_bytes = (unsigned)(xelems * 2);
  a.w_bytes = (unsigned)((int64_t)d->K * d->R * d->S * d->C * 2);
  a.M = d->N * d->Ho * d->Wo;
  a.Hi = d->H; a.Wi = d->W; a.Ci = d->C;
  a.Wo = d->Wo; a.HoWo = d->Ho * d->Wo;
  a.Co = d->K; a.ldy = ldy;
  a.R = d->R; a.S = d->S;
  a.so = d->stride; a.sd = 1; a.tsign = 1; a.pad = d->pad;
  a.x_img_pitch = (int)img_pitch(d); a.x_row_pitch = row_pitch(d); a.x_pix_pitch = pix_pitch(d);
  a.w_row_pitch = d->R * d->S * d->C;
  return launch(a, d->out_f32 != 0, stats_partial != nullptr, (hipStream_t)stream);
}

extern "C" int asm_conv2d_dgrad(const asm_conv_desc* d, const void* dy, const void* wt, void* dx,
                                void* stream) {
  if (int e = check_desc(d)) return e;
  ASM_REQUIRE(dy && wt && dx, "conv dgrad: null pointer");
  ASM_REQUIRE(d->K % 8 == 0, "conv dgrad: K=%d must be a multiple of 8 (pad dy)", d->K);
  ASM_REQUIRE(d->x_img_pitch == 0 && d->x_row_pitch == 0 && d->x_pix_pitch == 0 && !d->out_f32,
              "conv dgrad: custom pitches / f32 output not supported");
  const int64_t dyelems = (int64_t)d->N * d->Ho * d->Wo * d->K;
  ASM_REQUIRE(dyelems * 2 < (int64_t)ASM_OOB, "conv dgrad: dy larger than 2 GiB");
  IGemmArgs a;
  a.x = dy; a.w = wt; a.y = dx; a.stats = nullptr;
  a.x_bytes = (unsigned)(dyelems * 2);
  a.w_bytes = (unsigned)((int64_t)d->K * d->R * d->S * d->C * 2);
  a.M = d->N * d->H * d->W;
  a.Hi = d->Ho; a.Wi = d->Wo; a.Ci = d->K;
  a.Wo = d->W; a.HoWo = d->H * d->W;
  a.Co = d->C; a.ldy = d->C;
  a.R = d->R; a.S = d->S;
  // p = (h + pad - r) / stride  when divisible
  a.so = 1; a.sd = d->stride; a.tsign = -1; a.pad = -d->pad;
  a.x_img_pitch = d->Ho * d->Wo * d->K; a.x_row_pitch = d->Wo * d->K; a.x_pix_pitch = d->K;
  a.w_row_pitch = d->R * d->S * d->K;
  return launch(a, false, false, (hipStream_t)stream);
}
