// Pooling / resampling kernels (NHWC bf16, one thread = one output pixel x 8 channels).
// All are HBM-bound gathers with fp32 accumulation; backward passes are written in gather form
// (each input pixel sums the outputs whose window covers it) so there are no atomics.
#include "common.h"

namespace {

inline unsigned grid_for(size_t nvec) {
  size_t b = cdivz(nvec, 256);
  return (unsigned)(b ? b : 1);
}

__device__ __forceinline__ u32x4 ldv(const bf16_t* p, size_t elem_off) {
  return *reinterpret_cast<const u32x4*>(p + elem_off);
}

// decode flat vector index -> (n, h, w, vc)
struct Pix {
  int n, h, w, vc;
};
__device__ __forceinline__ Pix decode(size_t i64, int H, int W, int vcols) {
  // 32-bit index math (every launcher bounds the vector count below 2^31): 64-bit div/mod costs ~10x
  const unsigned i = (unsigned)i64;
  Pix p;
  unsigned m = i / (unsigned)vcols;
  p.vc = (int)(i - m * (unsigned)vcols);
  unsigned q = m / (unsigned)W;
  p.w = (int)(m - q * (unsigned)W);
  const unsigned n = q / (unsigned)H;
  p.h = (int)(q - n * (unsigned)H);
  p.n = (int)n;
  return p;
}

// ---- max pool 3x3 / 2, SAME (pad 0 before, up to 1 after) ---------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                          uint8_t* __restrict__ amax, int N, int H, int W, int C,
                                                          int Ho, int Wo, int pad_h, int pad_w) {
  const int vcols = C >> 3;
  const size_t nvec = (size_t)N * Ho * Wo * vcols;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const Pix p = decode(i, Ho, Wo, vcols);
  float best[8];
  int bidx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    best[e] = -INFINITY;
    bidx[e] = 0;
  }
  for (int r = 0; r < 3; ++r) {
    const int ih = p.h * 2 + r - pad_h;
    if ((unsigned)ih >= (unsigned)H) continue;
    for (int s = 0; s < 3; ++s) {
      const int iw = p.w * 2 + s - pad_w;
      if ((unsigned)iw >= (unsigned)W) continue;
      float f[8];
      unpack8(ldv(x, (((size_t)p.n * H + ih) * W + iw) * C + p.vc * 8), f);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (f[e] > best[e]) {  // strict: first maximum in window scan order wins
          best[e] = f[e];
          bidx[e] = r * 3 + s;
        }
    }
  }
  *reinterpret_cast<u32x4*>(y + i * 8) = pack8(best);
  if (amax) {
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lo |= (unsigned)bidx[e] << (8 * e);
      hi |= (unsigned)bidx[e + 4] << (8 * e);
    }
    u32x2 v = {lo, hi};
    *reinterpret_cast<u32x2*>(amax + i * 8) = v;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16_t* __restrict__ dy,
                                                          const uint8_t* __restrict__ amax, bf16_t* __restrict__ dx,
                                                          int N, int H, int W, int C, int Ho, int Wo, int pad_h,
                                                          int pad_w) {
  const int vcols = C >> 3;
  const size_t nvec = (size_t)N * H * W * vcols;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const Pix p = decode(i, H, W, vcols);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  // windows (ho, r) with ho*2 + r - pad_h == h
  for (int r = 0; r < 3; ++r) {
    const int th = p.h + pad_h - r;
    if (th < 0 || (th & 1)) continue;
    const int ho = th >> 1;
    if (ho >= Ho) continue;
    for (int s = 0; s < 3; ++s) {
      const int tw = p.w + pad_w - s;
      if (tw < 0 || (tw & 1)) continue;
      const int wo = tw >> 1;
      if (wo >= Wo) continue;
      const size_t o = (((size_t)p.n * Ho + ho) * Wo + wo) * C + p.vc * 8;
      const u32x2 a = *reinterpret_cast<const u32x2*>(amax + o);
      float g[8];
      unpack8(ldv(dy, o), g);
      const int code = r * 3 + s;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if ((int)((a.x >> (8 * e)) & 0xff) == code) acc[e] += g[e];
        if ((int)((a.y >> (8 * e)) & 0xff) == code) acc[e + 4] += g[e + 4];
      }
    }
  }
  *reinterpret_cast<u32x4*>(dx + i * 8) = pack8(acc);
}

// ---- average pool (zero pad before = pad; divisor k*k or valid count) ------------------------------
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                          int N, int H, int W, int C, int k, int stride, int pad,
                                                          int Ho, int Wo, int count_valid) {
  const int vcols = C >> 3;
  const size_t nvec = (size_t)N * Ho * Wo * vcols;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const Pix p = decode(i, Ho, Wo, vcols);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  int cnt = 0;
  for (int r = 0; r < k; ++r) {
    const int ih = p.h * stride + r - pad;
    if ((unsigned)ih >= (unsigned)H) continue;
    for (int s = 0; s < k; ++s) {
      const int iw = p.w * stride + s - pad;
      if ((unsigned)iw >= (unsigned)W) continue;
      float f[8];
      unpack8(ldv(x, (((size_t)p.n * H + ih) * W + iw) * C + p.vc * 8), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e];
      ++cnt;
    }
  }
  const float inv = 1.0f / (float)(count_valid ? cnt : k * k);
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] *= inv;
  *reinterpret_cast<u32x4*>(y + i * 8) = pack8(acc);
}

__device__ __forceinline__ int valid_taps(int o, int stride, int pad, int k, int L) {
  int c = 0;
  for (int r = 0; r < k; ++r) c += ((unsigned)(o * stride + r - pad) < (unsigned)L);
  return c;
}

// Gather form (no atomics): dx(h, w) = sum over the windows that contain (h, w) of dy / window size.  stride is 1
// or 2 (checked by the host wrapper), so the window test is a shift / mask instead of an integer division, and the
// divisor is a per-launch constant unless the reference's "count only valid taps" SAME rule is on.
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* dx,
                                                          int N, int H, int W, int C, int k, int stride, int pad,
                                                          int Ho, int Wo, int count_valid, const bf16_t* addend) {
  const int vcols = C >> 3;
  const size_t nvec = (size_t)N * H * W * vcols;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const Pix p = decode(i, H, W, vcols);
  const int sh = stride >> 1, msk = stride - 1;   // stride in {1, 2}
  const float inv_full = 1.0f / (float)(k * k);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const bf16_t* src = dy + (size_t)p.n * Ho * Wo * C + p.vc * 8;
  for (int r = 0; r < k; ++r) {
    const int th = p.h + pad - r;
    const int ho = th >> sh;
    if (th < 0 || (th & msk) || ho >= Ho) continue;
    for (int s = 0; s < k; ++s) {
      const int tw = p.w + pad - s;
      const int wo = tw >> sh;
      if (tw < 0 || (tw & msk) || wo >= Wo) continue;
      float g[8];
      unpack8(ldv(src, (size_t)(ho * Wo + wo) * C), g);
      const float inv = count_valid
                            ? 1.0f / (float)(valid_taps(ho, stride, pad, k, H) * valid_taps(wo, stride, pad, k, W))
                            : inv_full;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += g[e] * inv;
    }
  }
  if (addend) {   // same element is read and written by the same thread: aliasing dx is fine
    float f[8];
    unpack8(ldv(addend, i * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += f[e];
  }
  *reinterpret_cast<u32x4*>(dx + i * 8) = pack8(acc);
}

// ---- UpSampling2D((2,2)) backward: 2x2 block sum ---------------------------------------------------
// MASKED: dy is a not-yet-masked gradient and `mask` the packed ReLU mask ([N*H*W][C/8] bytes) of the full-resolution
// tensor: the block sum runs over dy * [bit set], so the masked gradient dz is never written (exact: masking is exact).
template <bool MASKED>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const bf16_t* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                           bf16_t* __restrict__ dx, int N, int Hs, int Ws, int C) {
  const int vcols = C >> 3;
  const size_t nvec = (size_t)N * Hs * Ws * vcols;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const Pix p = decode(i, Hs, Ws, vcols);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const int H = Hs * 2, W = Ws * 2;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float g[8];
      const size_t pix = ((size_t)p.n * H + p.h * 2 + r) * W + p.w * 2 + s;
      unpack8(ldv(dy, pix * C + p.vc * 8), g);
      if (MASKED) {
        const unsigned mk = mask[pix * vcols + p.vc];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = ((mk >> e) & 1u) ? g[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += g[e];
    }
  *reinterpret_cast<u32x4*>(dx + i * 8) = pack8(acc);
}

// ---- blur pool (REFLECT pad (k-1)/2, binomial k x k / sum, stride) --------------------------------
struct BlurCoef {
  float a[8];  // normalised 1-D binomial; 2-D weight = a[r]*a[s] (exact: all factors are dyadic)
};
inline BlurCoef blur_coef(int k) {
  static const int tri[8][7] = {{0}, {1}, {1, 1}, {1, 2, 1}, {1, 3, 3, 1}, {1, 4, 6, 4, 1}, {1, 5, 10, 10, 5, 1},
                                {1, 6, 15, 20, 15, 6, 1}};
  BlurCoef c;
  int sum = 0;
  for (int i = 0; i < k; ++i) sum += tri[k][i];
  for (int i = 0; i < 8; ++i) c.a[i] = i < k ? (float)tri[k][i] / (float)sum : 0.f;
  return c;
}
__device__ __forceinline__ int reflect(int i, int L) {
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  return i;
}

__global__ __launch_bounds__(256) void blur_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N,
                                                       int H, int W, int C, int k, int stride, int Ho, int Wo,
                                                       BlurCoef cf) {
  const int vcols = C >> 3;
  const size_t nvec = (size_t)N * Ho * Wo * vcols;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const Pix p = decode(i, Ho, Wo, vcols);
  const int pad = (k - 1) / 2;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int r = 0; r < k; ++r) {
    const int ih = reflect(p.h * stride + r - pad, H);
    for (int s = 0; s < k; ++s) {
      const int iw = reflect(p.w * stride + s - pad, W);
      float f[8];
      unpack8(ldv(x, (((size_t)p.n * H + ih) * W + iw) * C + p.vc * 8), f);
      const float wgt = cf.a[r] * cf.a[s];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e] * wgt;
    }
  }
  *reinterpret_cast<u32x4*>(y + i * 8) = pack8(acc);
}

// gather-form adjoint: input index h receives from every padded position hp with src(hp) == h
__device__ __forceinline__ int padded_sources(int h, int L, int pad, int* hp) {
  int n = 0;
  hp[n++] = h + pad;
  if (h >= 1 && h <= pad) hp[n++] = pad - h;
  if (h <= L - 2 && h >= L - 1 - pad) hp[n++] = pad + 2 * (L - 1) - h;
  return n;
}

__global__ __launch_bounds__(256) void blur_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int N,
                                                       int H, int W, int C, int k, int stride, int Ho, int Wo,
                                                       BlurCoef cf) {
  const int vcols = C >> 3;
  const size_t nvec = (size_t)N * H * W * vcols;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const Pix p = decode(i, H, W, vcols);
  const int pad = (k - 1) / 2;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  int hps[3], wps[3];
  const int nh = padded_sources(p.h, H, pad, hps);
  const int nw = padded_sources(p.w, W, pad, wps);
  for (int a = 0; a < nh; ++a)
    for (int r = 0; r < k; ++r) {
      const int th = hps[a] - r;
      if (th < 0 || th % stride) continue;
      const int ho = th / stride;
      if (ho >= Ho) continue;
      for (int b = 0; b < nw; ++b)
        for (int s = 0; s < k; ++s) {
          const int tw = wps[b] - s;
          if (tw < 0 || tw % stride) continue;
          const int wo = tw / stride;
          if (wo >= Wo) continue;
          float g[8];
          unpack8(ldv(dy, (((size_t)p.n * Ho + ho) * Wo + wo) * C + p.vc * 8), g);
          const float wgt = cf.a[r] * cf.a[s];
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += g[e] * wgt;
        }
    }
  *reinterpret_cast<u32x4*>(dx + i * 8) = pack8(acc);
}

// The workload's case -- 3 taps (1 2 1) / 4, stride 2, even H and W -- without the generic kernel's nest of tap loops and
// divisibility tests (2.3 TB/s: 27 mostly-failing predicate evaluations per 16 bytes stored): one thread owns a 2 x 2
// quad of dx, which draws on the 2 x 2 neighbourhood dy[i..i+1][j..j+1] only.  Per dimension (h = 2i + u):
//   u = 0: tap 1 of output i                                      -> 1/2 g[i]
//   u = 1: tap 0 of output i + 1 (if it exists), tap 2 of output i -> 1/4 g[i+1] + 1/4 g[i]
//          h = 1 is also what the REFLECT pad shows at padded position 0 (tap 0 of output 0) -> + 1/4 g[0]
// (the far-end reflection lands on tap positions no stride-2 output uses when the extent is even).  The terms are
// added in the generic kernel's order, so the result is the same bits.
__global__ __launch_bounds__(256) void blur3s2_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int N,
                                                          int Hh, int Wh, int C, float a0, float a1, float a2) {
  const int vcols = C >> 3;
  const size_t nq = (size_t)N * Hh * Wh * vcols;
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const Pix p = decode(q, Hh, Wh, vcols);       // p.h, p.w: the quad = the dy pixel (i, j)
  const bool hv = p.h + 1 < Hh, wv = p.w + 1 < Wh;
  float g[2][2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const bool ok = (a == 0 || hv) && (b == 0 || wv);
      const size_t off = (((size_t)p.n * Hh + p.h + (ok ? a : 0)) * Wh + p.w + (ok ? b : 0)) * C + p.vc * 8;
      unpack8(ldv(dy, off), g[a][b]);
    }
  // per dimension and parity: up to three (source, weight) terms in the generic kernel's order
  const int src_odd[3] = {1, 0, 0};
  const float wgt_odd[3] = {a0, a2, a0};
  const bool okh[3] = {hv, true, p.h == 0}, okw[3] = {wv, true, p.w == 0};
  const int W2 = 2 * Wh;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
      for (int th = 0; th < (u ? 3 : 1); ++th) {
        const int sh = u ? src_odd[th] : 0;
        const float wh = u ? wgt_odd[th] : a1;
        if (u && !okh[th]) continue;
#pragma unroll
        for (int tw = 0; tw < (v ? 3 : 1); ++tw) {
          const int sw = v ? src_odd[tw] : 0;
          const float ww = v ? wgt_odd[tw] : a1;
          if (v && !okw[tw]) continue;
          const float wgt = wh * ww;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += g[sh][sw][e] * wgt;
        }
      }
      const size_t off = (((size_t)p.n * 2 * Hh + 2 * p.h + u) * W2 + 2 * p.w + v) * C + p.vc * 8;
      *reinterpret_cast<u32x4*>(dx + off) = pack8(acc);
    }
}

// ---- global average pool: [N, HW, C] -> [N, C]; one block per (n, group of <=32 vector columns) -----
// NT threads = (NT / vcb) row-lanes x vcb vector columns, vcb = min(C/8, 32).  One block per (image, 32 vector
// columns): with 3136 rows per image the 256-thread form walked ~100 dependent trips per lane on 4 waves per CU
// (2.6 TB/s); the 1024-thread form with 4 rows in flight per trip is used for the large maps.
template <bool TWO_BRANCH, int NT>
__global__ __launch_bounds__(NT) void gap_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int HW,
                                                     int Cin, int Cout, int vcb) {
  __shared__ float red[NT][9];
  const int vcols = Cout >> 3;
  const int vcl = threadIdx.x % vcb;
  const int rl = threadIdx.x / vcb;
  const int nrl = NT / vcb;
  const int vc = blockIdx.x * vcb + vcl;
  const int n = blockIdx.y;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (vc < vcols && rl < nrl) {
    constexpr int U = 4;
    const bf16_t* base = x + (size_t)n * HW * Cin + vc * 8;
    int r = rl;
    for (; r + (U - 1) * nrl < HW; r += U * nrl) {
      u32x4 v[U], w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t off = (size_t)(r + u * nrl) * Cin;
        v[u] = ldv(base, off);
        if (TWO_BRANCH) w[u] = ldv(base, off + Cout);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float f[8];
        unpack8(v[u], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f[e];
        if (TWO_BRANCH) {
          unpack8(w[u], f);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
      }
    }
    for (; r < HW; r += nrl) {
      const size_t off = (size_t)r * Cin;
      float f[8];
      unpack8(ldv(base, off), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e];
      if (TWO_BRANCH) {
        unpack8(ldv(base, off + Cout), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  if (rl == 0 && vc < vcols) {
    float o[8];
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
      for (int r = 0; r < nrl; ++r) t += red[r * vcb + vcl][e];
      o[e] = t * inv;
    }
    *reinterpret_cast<u32x4*>(y + (size_t)n * Cout + vc * 8) = pack8(o);
  }
}

__global__ __launch_bounds__(256) void gap_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int N,
                                                      int HW, int C) {
  const int vcols = C >> 3;
  const size_t nvec = (size_t)N * HW * vcols;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  const unsigned iu = (unsigned)i;
  const int vc = (int)(iu % (unsigned)vcols);
  const int n = (int)(iu / ((unsigned)HW * (unsigned)vcols));
  float g[8];
  unpack8(ldv(dy, (size_t)n * C + vc * 8), g);
  const float inv = 1.0f / (float)HW;
#pragma unroll
  for (int e = 0; e < 8; ++e) g[e] *= inv;
  *reinterpret_cast<u32x4*>(dx + i * 8) = pack8(g);
}

}  // namespace

#define POOL_ARGS_OK(name) ASM_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, name ": bad shape")

static inline void same_pad(int in, int k, int s, int* out, int* before) {
  *out = (in + s - 1) / s;
  int total = (*out - 1) * s + k - in;
  if (total < 0) total = 0;
  *before = total / 2;
}

extern "C" int asm_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* argmax, int N, int H, int W, int C,
                                    void* stream) {
  POOL_ARGS_OK("maxpool_fwd");
  ASM_REQUIRE(x && y, "maxpool_fwd: null pointer");
  int Ho, Wo, ph, pw;
  same_pad(H, 3, 2, &Ho, &ph);
  same_pad(W, 3, 2, &Wo, &pw);
  const size_t nvec = (size_t)N * Ho * Wo * (C / 8);
  ASM_REQUIRE(nvec < 0x7fffffffull, "pool: tensor too large for 32-bit indexing");
  ASM_LAUNCH(maxpool_fwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)y, argmax, N, H, W, C, Ho, Wo, ph, pw);
  ASM_CHECK_LAUNCH("maxpool_fwd");
  return ASM_OK;
}

extern "C" int asm_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int N, int H, int W, int C,
                                    void* stream) {
  POOL_ARGS_OK("maxpool_bwd");
  ASM_REQUIRE(dy && argmax && dx, "maxpool_bwd: null pointer");
  int Ho, Wo, ph, pw;
  same_pad(H, 3, 2, &Ho, &ph);
  same_pad(W, 3, 2, &Wo, &pw);
  const size_t nvec = (size_t)N * H * W * (C / 8);
  ASM_REQUIRE(nvec < 0x7fffffffull, "pool: tensor too large for 32-bit indexing");
  ASM_LAUNCH(maxpool_bwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     argmax, (bf16_t*)dx, N, H, W, C, Ho, Wo, ph, pw);
  ASM_CHECK_LAUNCH("maxpool_bwd");
  return ASM_OK;
}

extern "C" int asm_avgpool_fwd(const void* x, void* y, int N, int H, int W, int C, int k, int stride, int pad,
                               int Ho, int Wo, int count_valid, void* stream) {
  POOL_ARGS_OK("avgpool_fwd");
  ASM_REQUIRE(x && y && k >= 1 && k <= 7 && stride >= 1 && pad >= 0 && Ho > 0 && Wo > 0, "avgpool_fwd: bad arguments");
  const size_t nvec = (size_t)N * Ho * Wo * (C / 8);
  ASM_REQUIRE(nvec < 0x7fffffffull, "pool: tensor too large for 32-bit indexing");
  ASM_LAUNCH(avgpool_fwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)y, N, H, W, C, k, stride, pad, Ho, Wo, count_valid);
  ASM_CHECK_LAUNCH("avgpool_fwd");
  return ASM_OK;
}

extern "C" int asm_avgpool_bwd(const void* dy, void* dx, int N, int H, int W, int C, int k, int stride, int pad,
                               int Ho, int Wo, int count_valid, const void* addend, void* stream) {
  POOL_ARGS_OK("avgpool_bwd");
  ASM_REQUIRE(dy && dx && k >= 1 && k <= 7 && pad >= 0 && Ho > 0 && Wo > 0, "avgpool_bwd: bad arguments");
  ASM_REQUIRE(stride == 1 || stride == 2, "avgpool_bwd: stride %d not supported (the shortcut pools use 1 and 2)", stride);
  const size_t nvec = (size_t)N * H * W * (C / 8);
  ASM_REQUIRE(nvec < 0x7fffffffull, "pool: tensor too large for 32-bit indexing");
  ASM_LAUNCH(avgpool_bwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     (bf16_t*)dx, N, H, W, C, k, stride, pad, Ho, Wo, count_valid, (const bf16_t*)addend);
  ASM_CHECK_LAUNCH("avgpool_bwd");
  return ASM_OK;
}

extern "C" int asm_upsample2x_bwd(const void* dy, void* dx, int N, int Hs, int Ws, int C, void* stream) {
  ASM_REQUIRE(dy && dx && N > 0 && Hs > 0 && Ws > 0 && C > 0 && C % 8 == 0, "upsample2x_bwd: bad arguments");
  const size_t nvec = (size_t)N * Hs * Ws * (C / 8);
  ASM_REQUIRE(nvec < 0x7fffffffull, "pool: tensor too large for 32-bit indexing");
  ASM_LAUNCH(upsample_bwd_kernel<false>, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     nullptr, (bf16_t*)dx, N, Hs, Ws, C);
  ASM_CHECK_LAUNCH("upsample2x_bwd");
  return ASM_OK;
}

extern "C" int asm_upsample2x_bwd_masked(const void* dy, const uint8_t* relu_mask, void* dx, int N, int Hs, int Ws, int C,
                                         void* stream) {
  ASM_REQUIRE(dy && relu_mask && dx && N > 0 && Hs > 0 && Ws > 0 && C > 0 && C % 8 == 0, "upsample2x_bwd_masked: bad arguments");
  const size_t nvec = (size_t)N * Hs * Ws * (C / 8);
  ASM_REQUIRE(nvec < 0x7fffffffull, "pool: tensor too large for 32-bit indexing");
  ASM_LAUNCH(upsample_bwd_kernel<true>, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     relu_mask, (bf16_t*)dx, N, Hs, Ws, C);
  ASM_CHECK_LAUNCH("upsample2x_bwd_masked");
  return ASM_OK;
}

static int blur_out(int in, int k, int stride) { return (in + 2 * ((k - 1) / 2) - k) / stride + 1; }

extern "C" int asm_blurpool_fwd(const void* x, void* y, int N, int H, int W, int C, int k, int stride, void* stream) {
  POOL_ARGS_OK("blurpool_fwd");
  ASM_REQUIRE(x && y && k >= 2 && k <= 7 && stride >= 1, "blurpool_fwd: filter size %d not supported", k);
  ASM_REQUIRE((k - 1) / 2 < H && (k - 1) / 2 < W, "blurpool_fwd: REFLECT pad needs pad < size");
  const int Ho = blur_out(H, k, stride), Wo = blur_out(W, k, stride);
  const size_t nvec = (size_t)N * Ho * Wo * (C / 8);
  ASM_REQUIRE(nvec < 0x7fffffffull, "pool: tensor too large for 32-bit indexing");
  ASM_LAUNCH(blur_fwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)y, N, H, W, C, k, stride, Ho, Wo, blur_coef(k));
  ASM_CHECK_LAUNCH("blurpool_fwd");
  return ASM_OK;
}

extern "C" int asm_blurpool_bwd(const void* dy, void* dx, int N, int H, int W, int C, int k, int stride,
                                void* stream) {
  POOL_ARGS_OK("blurpool_bwd");
  ASM_REQUIRE(dy && dx && k >= 2 && k <= 7 && stride >= 1, "blurpool_bwd: filter size %d not supported", k);
  const int Ho = blur_out(H, k, stride), Wo = blur_out(W, k, stride);
  const size_t nvec = (size_t)N * H * W * (C / 8);
  ASM_REQUIRE(nvec < 0x7fffffffull, "pool: tensor too large for 32-bit indexing");
  if (k == 3 && stride == 2 && H % 2 == 0 && W % 2 == 0 && H >= 4 && W >= 4) {
    const BlurCoef cf = blur_coef(3);
    ASM_LAUNCH(blur3s2_bwd_kernel, dim3(grid_for(nvec / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (bf16_t*)dx, N, H / 2, W / 2, C, cf.a[0], cf.a[1], cf.a[2]);
  } else {
    ASM_LAUNCH(blur_bwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (bf16_t*)dx, N, H, W, C, k, stride, Ho, Wo, blur_coef(k));
  }
  ASM_CHECK_LAUNCH("blurpool_bwd");
  return ASM_OK;
}

extern "C" int asm_gap_fwd(const void* x, void* y, int N, int HW, int C, void* stream) {
  ASM_REQUIRE(x && y && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "gap_fwd: bad arguments");
  const int vcb = C / 8 < 32 ? C / 8 : 32;
  if (HW >= 512)
    ASM_LAUNCH((gap_fwd_kernel<false, 1024>), dim3(cdiv(C / 8, vcb), N), dim3(1024), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)y, HW, C, C, vcb);
  else
    ASM_LAUNCH((gap_fwd_kernel<false, 256>), dim3(cdiv(C / 8, vcb), N), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)y, HW, C, C, vcb);
  ASM_CHECK_LAUNCH("gap_fwd");
  return ASM_OK;
}

extern "C" int asm_sk_gap(const void* f, void* s, int N, int HW, int F, void* stream) {
  ASM_REQUIRE(f && s && N > 0 && HW > 0 && F > 0 && F % 8 == 0, "sk_gap: bad arguments");
  const int vcb = F / 8 < 32 ? F / 8 : 32;
  if (HW >= 512)
    ASM_LAUNCH((gap_fwd_kernel<true, 1024>), dim3(cdiv(F / 8, vcb), N), dim3(1024), 0, (hipStream_t)stream,
                       (const bf16_t*)f, (bf16_t*)s, HW, 2 * F, F, vcb);
  else
    ASM_LAUNCH((gap_fwd_kernel<true, 256>), dim3(cdiv(F / 8, vcb), N), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)f, (bf16_t*)s, HW, 2 * F, F, vcb);
  ASM_CHECK_LAUNCH("sk_gap");
  return ASM_OK;
}

extern "C" int asm_gap_bwd(const void* dy, void* dx, int N, int HW, int C, void* stream) {
  ASM_REQUIRE(dy && dx && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "gap_bwd: bad arguments");
  const size_t nvec = (size_t)N * HW * (C / 8);
  ASM_REQUIRE(nvec < 0x7fffffffull, "pool: tensor too large for 32-bit indexing");
  ASM_LAUNCH(gap_bwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     (bf16_t*)dx, N, HW, C);
  ASM_CHECK_LAUNCH("gap_bwd");
  return ASM_OK;
}
