// Shared by the MFMA implicit-GEMM convolution kernels (conv_igemm.hip, conv_gemm1.hip): the argument block, the tile
// configuration and the epilogue (bf16 pack through LDS, coalesced NHWC stores, fused batch-norm statistics, gradient
// fan-in addend with its ReLU mask, pooled-gradient gather, inference batch norm).
#pragma once
#include "common.h"

namespace asm_igemm {

struct IGemmArgs {
  const void* x;
  const void* w;
  void* y;
  const void* addend;  // optional bf16 [M][ldy] added to the result (gradient accumulation, dgrad)
  const uint8_t* addend_mask;  // optional packed ReLU mask of the addend ([M][ldy/8] bytes): addend lanes with a 0 bit count as 0
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
  int m_tile0;     // first row tile of this launch (igemm2 / igemm3 with 128-row tiles: the ragged last round of an igemm8 layer)
  FastDiv fd_howo, fd_wo, fd_ntn;  // m -> (img, ho, wo), block -> (tile_m, tile_n) without integer division (igemm2_kernel)
  // igemm2_kernel only (parity-class decomposition of the stride-2 input gradient, asm_conv2d_dgrad):
  int pad_w;                       // column pad (== pad except in a parity class)
  int wt0, wtr, wts;               // filter tap of loop tap (i, j): wt0 + i * wtr + j * wts   (0, S, 1 normally)
  int y_strided;                   // 1: output row m = (img, ho, wo) goes to y_base + img*y_img + ho*y_row + wo*y_pix
  int y_base, y_img_pitch, y_row_pitch, y_pix_pitch;   // elements
  // inference-mode batch norm folded into the epilogue (asm_conv2d_fprop_bn): y = [relu](acc * scale[n] + shift[n] + addend)
  const float* bn_scale;
  const float* bn_shift;
  int bn_relu;
  // average-pool backward folded into the epilogue of a 1x1 stride-1 input gradient (asm_conv2d_dgrad_pooled): the block
  // input of a projection bottleneck is read by conv1 and by the shortcut's average pool; dx += avgpool_bwd(pool_dy)
  // batch-norm backward sums in the epilogue of an input gradient (asm_conv2d_dgrad_bnred): the tensor this launch writes is the
  // gradient dout of a conv -> BN [-> + shortcut] [-> ReLU] output; with red_y the STATS partials are not (sum y, sum y^2) but
  // (sum dz, sum dz * y), dz = bf16(dout) * [mask bit], y = that layer's pre-BN convolution output at the same element --
  // what rowreduce_kernel<1> (bn.hip) reads dout and y again for.  asm_bn_bwd_finalize_raw turns sum dz * y into sum dz * xhat.
  const void* red_y;           // bf16 [M][ldy] or null
  const uint8_t* red_mask;     // packed ReLU mask [M][ldy / 8] or null (no ReLU: dz = dout)
  const void* pool_dy;   // bf16 [N][pool_Hp][pool_Wp][Co] or null
  int pool_k, pool_stride, pool_pad, pool_Hp, pool_Wp, pool_cv, pool_H;
};

constexpr int STATS_BM = 128;  // rows per statistics partial (asm_conv2d_stats_blocks)

template <int BK>
__device__ __forceinline__ int swz(int row) {
  // BK=64: 8 chunks / 128-B row ; BK=32: 4 chunks / 64-B row.  16 rows that are distinct mod 16
  // land on 16 distinct 16-byte slots of the 256-byte bank row.
  return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// Tile configuration: BM x BN x BK block tile, WGM x WGN waves, each wave (BM/WGM) x (BN/WGN) as
// 32x32 MFMA tiles.  Arithmetic intensity against the L2->LDS path is 2*BM*BN/((BM+BN)*2) FLOP/B:
// 64 for 128x128, 85 for 256x128, 128 for 256x256 -- the MFMA-bound layers need the big tiles, the
// HBM-bound ones the small footprints (more workgroups per CU).
// MODE: 0 = register-staged, 1-deep prefetch; 1 = register-staged, 2-deep prefetch (two register tile sets);
//       2 = LDS-DMA staging (global_load_lds_dwordx4: no VGPR round trip, no ds_write), 1-deep.
template <int BM, int BN, int BK, int WGM, int WGN, bool OUT_F32, bool STATS, int MODE>
struct Cfg {
  static constexpr int NT = 64 * WGM * WGN;
  static constexpr int CPR = BK / 8;            // 16-byte chunks per tile row
  static constexpr int RPP = NT / CPR;          // rows staged per pass
  static constexpr int XP = BM / RPP;           // passes for the activation tile
  static constexpr int WP = cmax(1, BN / RPP);  // passes for the filter tile
  static constexpr int ROWB = BK * 2;
  static constexpr int STAGE = (BM + BN) * ROWB;
  static constexpr int WTM = BM / WGM, WTN = BN / WGN;
  static constexpr int TM = WTM / 32, TN = WTN / 32;
  static constexpr int LDO = BN * 2 + 16;       // padded output-tile row (bytes)
  static constexpr int CPO = BN / 8;            // 16-byte chunks per output row
  static constexpr int RPO = NT / CPO;          // output rows per pass
  static constexpr int OP = BM / RPO;
  static constexpr int EPI = OUT_F32 ? 0 : BM * LDO;
  static constexpr int RED = STATS ? RPO * BN * 2 * 4 : 0;
  static constexpr int LDS = cmax(cmax(2 * STAGE, EPI), RED);
  static_assert(BM % RPP == 0 && (BN % RPP == 0 || BN < RPP), "loader tiling");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0 && BM % RPO == 0, "wave tiling");
  static_assert(LDS <= 160 * 1024, "lds");
  static_assert(!STATS || BM % STATS_BM == 0, "stats granularity");
};

// Epilogue shared by both kernels.  acc[a][b][reg]: n_local = wn*WTN + a*32 + (reg&3) + 8*(reg>>2) + 4*lhi ;
// m_local = wm*WTM + b*32 + l31.
// patch_base >= 0: the tile's 128 rows are an 8 x 16 pixel patch of one image (conv_halo_kernel): row r is pixel
// patch_base + (r >> 4) * W + (r & 15) of the [N*H*W] output.
// SPLIT8 (igemm8_kernel, conv_igemm8.hip): a wave's 128 x 64 outputs are two 64-row pieces (one per 128-row half of the pixel
// tile) by two 32-channel pieces (one per 128-row half of the filter tile): acc[a][b] is filter half a, pixel half b >> 1,
// 32-row sub-tile b & 1.
// BNRED (with STATS; asm_conv2d_dgrad_bnred): the statistics partials are the batch-norm BACKWARD sums (sum dz, sum dz * y) of the
// gradient this launch writes (IGemmArgs::red_y).  A separate instantiation: as a run-time branch in the statistics epilogue it
// took the forward kernels from 60 to 152 VGPRs (4 -> 2 waves per SIMD) and the 1x1 class from 7.6 to 8.5 ms per step.
// RPF (BNRED): how many output passes ahead the y vectors and mask bytes of the batch-norm sums are fetched.  1 for the
// bandwidth-bound 1x1 kernels (registers are occupancy there); 8 for the MFMA kernels, whose accumulators are dead once they are
// in LDS: with one load in flight per lane every pass of the epilogue is a dependent HBM round trip (8 - 16 per workgroup), which
// is what made the 3x3 form cost what it saved when it was first measured.
template <class C, int BM, int BN, int WTM, int WTN, int TM, int TN, bool OUT_F32, bool STATS, bool PFA = false, bool POOL = false,
          bool SPLIT8 = false, bool BNRED = false, int RPF = 1>
__device__ __forceinline__ void igemm_epilogue(const IGemmArgs& p, f32x16 (&acc)[TN][TM], unsigned char* smem,
                                               int tile_m, int tile_n, int tid, int wm, int wn, int l31, int lhi,
                                               int patch_base = -1) {
  // ---------------- epilogue ----------------
  // acc[a][b][reg]: n_local = wn*WTN + a*32 + (reg&3) + 8*(reg>>2) + 4*lhi ; m_local = wm*WTM + b*32 + l31
  if constexpr (OUT_F32) {
    static_assert(!SPLIT8, "igemm8 writes bf16");
    float* y = reinterpret_cast<float*>(p.y);
    const int co4 = (p.Co + 3) & ~3;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int m = tile_m * BM + wm * WTM + b * 32 + l31;
        if (m < p.M) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = tile_n * BN + wn * WTN + a * 32 + 8 * g + 4 * lhi;
            if (n < co4) {
              f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
              *reinterpret_cast<f32x4*>(y + (size_t)m * p.ldy + n) = v;
            }
          }
        }
      }
  } else {
    constexpr int LDO = C::LDO, CPO = C::CPO, RPO = C::RPO, OP = C::OP;
    unsigned char* os = smem;
    bf16_t* y = reinterpret_cast<bf16_t*>(p.y);
    const int oc = tid % CPO, orow = tid / CPO;
    const int n0 = tile_n * BN + oc * 8;
    const int co8 = (p.Co + 7) & ~7;
    auto row_off = [&](int row, int m) -> size_t {
      if (patch_base >= 0) return (size_t)(patch_base + (row >> 4) * p.Wi + (row & 15)) * p.ldy + n0;
      if (p.y_strided) {   // workgroup-uniform: rows of a parity class scatter into the full-resolution tensor
        const unsigned img = fd_div((unsigned)m, p.fd_howo);
        const unsigned rem = (unsigned)m - img * (unsigned)p.HoWo;
        const unsigned ho = fd_div(rem, p.fd_wo);
        const unsigned wo = rem - ho * (unsigned)p.Wo;
        return (size_t)p.y_base + (size_t)img * p.y_img_pitch + (size_t)ho * p.y_row_pitch + (size_t)wo * p.y_pix_pitch + n0;
      }
      return (size_t)m * p.ldy + n0;
    };
    // PFA kernels fetch the gradient fan-in addend (and its ReLU mask) of up to PF output passes BEFORE the passes run --
    // the first group even before the accumulators go to LDS.  In the pass loop every addend load sits behind the previous
    // pass's store (they may alias: in-place accumulation is allowed), i.e. one exposed HBM round trip per pass, 8-16 per
    // workgroup: that is what bounds the input gradients of the small maps (7x7 / 14x14: a few hundred workgroups, nothing
    // to overlap with; measured -20..-25 % there).  On the large maps the extra ~40 registers cost occupancy and the
    // bandwidth-bound 1x1 layers lose 10-30 %, so the launcher picks the variant per layer (launch2_cfg).  A thread only
    // ever re-reads the addresses it writes itself, so in-place accumulation stays exact.
    constexpr bool PF_ON = PFA && !STATS;
    constexpr int PF = !PF_ON ? 1 : (OP < 8 ? OP : 8);
    u32x4 av[PF];
    unsigned amk[PF];
    auto prefetch = [&](int ps0) {
#pragma unroll
      for (int i = 0; i < PF; ++i) {
        const int row = (ps0 + i) * RPO + orow;
        const int m = tile_m * BM + row;
        const bool ok = m < p.M && n0 < co8;
        const size_t yo = row_off(row, ok ? m : 0);
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        av[i] = ok ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.addend) + yo) : z4;
        amk[i] = (ok && p.addend_mask) ? (unsigned)p.addend_mask[yo >> 3] : 0xffu;
      }
    };
    if constexpr (PF_ON) {
      if (p.addend) prefetch(0);      // workgroup-uniform
    }
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int ml = SPLIT8 ? (b >> 1) * 128 + wm * 64 + (b & 1) * 32 + l31 : wm * WTM + b * 32 + l31;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = SPLIT8 ? a * 128 + wn * 32 + 8 * g + 4 * lhi : wn * WTN + a * 32 + 8 * g + 4 * lhi;
          u32x2 v;
          v.x = pack2bf(acc[a][b][4 * g], acc[a][b][4 * g + 1]);
          v.y = pack2bf(acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]);
          *reinterpret_cast<u32x2*>(os + ml * LDO + nl * 2) = v;
        }
      }
    __syncthreads();
    // Average-pool backward gathered in this epilogue (POOL kernels, asm_conv2d_dgrad_pooled): this pixel's share of every
    // window that holds it.  Branch-free: with k <= 2 * stride at most TWO windows per dimension hold a pixel
    // (o_hi = (p + pad) / stride and o_hi - 1), so the four candidate vectors are loaded back to back (invalid ones from a
    // clamped address, weight 0) and waited for once -- a loop over the k x k taps with early-outs issued up to nine
    // dependent L2 round trips per pass and cost more than the scatter pass it replaced -- and the taps of pass ps + 1 are
    // issued before pass ps stores (the store would otherwise fence them: it may alias).
    struct PoolTaps {
      u32x4 v[4];
      float w[4];
    };
    PoolTaps pcur, pnxt;
    auto pool_issue = [&](int ps, PoolTaps& t) {
      const int row = ps * RPO + orow;
      const int m = tile_m * BM + row;
      const bool ok = m < p.M && n0 < co8;
      const unsigned mm = ok ? (unsigned)m : 0u;
      const unsigned img = fd_div(mm, p.fd_howo);
      const unsigned rem = mm - img * (unsigned)p.HoWo;
      const unsigned ph = fd_div(rem, p.fd_wo);
      const unsigned pw = rem - ph * (unsigned)p.Wo;
      const int H = p.pool_H, W = p.Wo;
      const int sh = p.pool_stride >> 1;                     // stride in {1, 2}
      const int th = (int)ph + p.pool_pad, tw = (int)pw + p.pool_pad;
      int oh[2], ow[2];
      float wh[2], ww[2];
      oh[0] = th >> sh; ow[0] = tw >> sh;
      oh[1] = oh[0] - 1; ow[1] = ow[0] - 1;
      const int rh = th - (oh[0] << sh), rw = tw - (ow[0] << sh);
      bool vh[2], vw[2];
      vh[0] = ok && oh[0] < p.pool_Hp;           vw[0] = ow[0] < p.pool_Wp;
      vh[1] = ok && oh[1] >= 0 && oh[1] < p.pool_Hp && rh + p.pool_stride <= p.pool_k - 1;
      vw[1] = ow[1] >= 0 && ow[1] < p.pool_Wp && rw + p.pool_stride <= p.pool_k - 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float ch_ = (float)p.pool_k, cw_ = (float)p.pool_k;
        if (p.pool_cv) {       // workgroup-uniform: the "count only valid taps" SAME rule
          int a_ = 0, b_ = 0;
          for (int q = 0; q < p.pool_k; ++q) {
            a_ += ((unsigned)(oh[i] * p.pool_stride + q - p.pool_pad) < (unsigned)H);
            b_ += ((unsigned)(ow[i] * p.pool_stride + q - p.pool_pad) < (unsigned)W);
          }
          ch_ = (float)(a_ > 0 ? a_ : 1);
          cw_ = (float)(b_ > 0 ? b_ : 1);
        }
        wh[i] = vh[i] ? 1.0f / ch_ : 0.f;
        ww[i] = vw[i] ? 1.0f / cw_ : 0.f;
        oh[i] = vh[i] ? oh[i] : 0;
        ow[i] = vw[i] ? ow[i] : 0;
      }
      const bf16_t* src = reinterpret_cast<const bf16_t*>(p.pool_dy) + (size_t)img * p.pool_Hp * p.pool_Wp * p.ldy +
                          (n0 < co8 ? n0 : 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          t.v[i * 2 + j] = *reinterpret_cast<const u32x4*>(src + (size_t)(oh[i] * p.pool_Wp + ow[j]) * p.ldy);
          t.w[i * 2 + j] = wh[i] * ww[j];
        }
    };
    if constexpr (POOL) pool_issue(0, pcur);
    // statistics partials are per STATS_BM (=128) rows: a 256-row tile emits two of them
    constexpr int SG = STATS ? BM / STATS_BM : 1;
    float s[SG][8], ss[SG][8];
#pragma unroll
    for (int q = 0; q < SG; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) s[q][e] = ss[q][e] = 0.f;
    static_assert(!BNRED || STATS, "the batch-norm backward sums use the statistics epilogue");
    // batch-norm backward sums (BNRED, p.red_y): this thread's y vector and ReLU-mask byte of pass ps, fetched RD passes ahead
    constexpr int RD = !BNRED ? 1 : (RPF < OP ? RPF : OP);
    u32x4 ryv[RD];
    unsigned rmkv[RD];
    auto red_fetch = [&](int ps_, u32x4& yv, unsigned& mk) {
      const int row = ps_ * RPO + orow;
      const int m = tile_m * BM + row;
      const u32x4 z4 = {0u, 0u, 0u, 0u};
      yv = z4;
      mk = 0xffu;
      if (m < p.M && n0 < co8) {
        const size_t ro = row_off(row, m);
        yv = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.red_y) + ro);
        if (p.red_mask) mk = (unsigned)p.red_mask[ro >> 3];
      }
    };
    if constexpr (BNRED) {
#pragma unroll
      for (int i = 0; i < RD; ++i) red_fetch(i, ryv[i], rmkv[i]);
    }
#pragma unroll
    for (int ps = 0; ps < OP; ++ps) {
      if constexpr (PF_ON) {
        if (p.addend && ps > 0 && ps % PF == 0) prefetch(ps);
      }
      if constexpr (POOL) {
        if (ps + 1 < OP) pool_issue(ps + 1, pnxt);
      }
      const int row = ps * RPO + orow;
      const int m = tile_m * BM + row;
      u32x4 ry = ryv[ps % RD];
      unsigned rmk = rmkv[ps % RD];
      if constexpr (BNRED) {
        if (ps + RD < OP) red_fetch(ps + RD, ryv[ps % RD], rmkv[ps % RD]);   // RD passes ahead: lands under the passes between
      }
      u32x4 v = *reinterpret_cast<const u32x4*>(os + row * LDO + oc * 16);
      if (m < p.M && n0 < co8) {
        const size_t yoff = row_off(row, m);
        if (POOL || p.addend || p.bn_scale) {   // workgroup-uniform
          float fv[8];
          unpack8(v, fv);
          if constexpr (POOL) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              float g[8];
              unpack8(pcur.v[t], g);
#pragma unroll
              for (int e = 0; e < 8; ++e) fv[e] += g[e] * pcur.w[t];
            }
          }
          if (p.bn_scale) {             // fused inference BN on the bf16-rounded conv tile (== the two-pass numerics)
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.bn_scale + n0), s1 = *reinterpret_cast<const f32x4*>(p.bn_scale + n0 + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(p.bn_shift + n0), h1 = *reinterpret_cast<const f32x4*>(p.bn_shift + n0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              fv[e] = fv[e] * s0[e] + h0[e];
              fv[e + 4] = fv[e + 4] * s1[e] + h1[e];
            }
          }
          if (p.addend) {
            if constexpr (!PF_ON) {
              av[0] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.addend) + yoff);
              amk[0] = p.addend_mask ? (unsigned)p.addend_mask[yoff >> 3] : 0xffu;
            }
            float fa[8];
            unpack8(av[ps % PF], fa);
            const unsigned mk = amk[ps % PF];     // 0xff without a mask: the addend is a not-yet-masked gradient (dz = dy * [y > 0]) otherwise
#pragma unroll
            for (int e = 0; e < 8; ++e) fv[e] += ((mk >> e) & 1u) ? fa[e] : 0.f;
          }
          if (p.bn_relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) fv[e] = fmaxf(fv[e], 0.f);
          }
          v = pack8(fv);
        }
        *reinterpret_cast<u32x4*>(y + yoff) = v;
      }
      if constexpr (POOL) pcur = pnxt;
      if constexpr (STATS) {
        constexpr int PPG = OP / SG;   // passes per statistics group (rows are pass-major)
        float f[8];
        unpack8(v, f);
        if constexpr (BNRED) {      // batch-norm backward sums of the gradient just written (rows past M: v = 0, ry = 0)
          float fy[8];
          unpack8(ry, fy);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float dz = ((rmk >> e) & 1u) ? f[e] : 0.f;
            s[ps / PPG][e] += dz;
            ss[ps / PPG][e] += dz * fy[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s[ps / PPG][e] += f[e];
            ss[ps / PPG][e] += f[e] * f[e];
          }
        }
      }
    }
    if constexpr (STATS) {
      float* red = reinterpret_cast<float*>(smem);  // [RPO][2][BN], aliases the (now consumed) output tile
#pragma unroll
      for (int q = 0; q < SG; ++q) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          red[(orow * 2 + 0) * BN + oc * 8 + e] = s[q][e];
          red[(orow * 2 + 1) * BN + oc * 8 + e] = ss[q][e];
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * BN; idx += C::NT) {   // (a 256-column tile of a 256-thread workgroup takes two trips)
          const int which = idx / BN, nl = idx - which * BN;
          float t = 0.f;
#pragma unroll 8
          for (int g = 0; g < RPO; ++g) t += red[(g * 2 + which) * BN + nl];
          const int n = tile_n * BN + nl;
          const int mb = tile_m * SG + q;   // 128-row statistics block index
          if (n < p.Co && mb * STATS_BM < p.M) p.stats[((size_t)mb * 2 + which) * p.Co + n] = t;
        }
      }
    }
  }
}

}  // namespace asm_igemm
