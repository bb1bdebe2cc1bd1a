// Element-wise, loss, input-pipeline, optimiser, filter-layout and debug kernels.
#include "common.h"
#include "../../include/asm_hip_debug.h"

// ---- error plumbing ------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void asm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* asm_last_error(void) { return g_err; }
extern "C" int asm_abi_version(void) { return ASM_ABI_VERSION; }

namespace {

inline unsigned ew_grid(size_t n) {
  size_t b = cdivz(n, 256);
  return (unsigned)(b < 8192 ? (b ? b : 1) : 8192);
}

// ---- element-wise ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_fwd_kernel(const bf16_t* x, bf16_t* y, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + i * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
    *reinterpret_cast<u32x4*>(y + i * 8) = pack8(f);
  }
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const bf16_t* dy, const bf16_t* y, bf16_t* dx, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float g[8], f[8];
    unpack8(*reinterpret_cast<const u32x4*>(dy + i * 8), g);
    unpack8(*reinterpret_cast<const u32x4*>(y + i * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = f[e] > 0.f ? g[e] : 0.f;
    *reinterpret_cast<u32x4*>(dx + i * 8) = pack8(g);
  }
}
__global__ __launch_bounds__(256) void mask_apply_kernel(const bf16_t* dy, const uint8_t* mask, bf16_t* dx, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float g[8];
    unpack8(*reinterpret_cast<const u32x4*>(dy + i * 8), g);
    const unsigned mk = mask[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = ((mk >> e) & 1u) ? g[e] : 0.f;
    *reinterpret_cast<u32x4*>(dx + i * 8) = pack8(g);
  }
}
__global__ __launch_bounds__(256) void add_kernel(const bf16_t* a, const bf16_t* b, bf16_t* o, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float f[8], g[8];
    unpack8(*reinterpret_cast<const u32x4*>(a + i * 8), f);
    unpack8(*reinterpret_cast<const u32x4*>(b + i * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += g[e];
    *reinterpret_cast<u32x4*>(o + i * 8) = pack8(f);
  }
}
__global__ void bias_add_kernel(float* y, const float* bias, int M, int C, int ldy) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const int m = (int)(i / C), c = (int)(i - (size_t)m * C);
  y[(size_t)m * ldy + c] += bias[c];
}
// dbias[c] = sum_m dz[m][c]; one block per 256 columns, rows strided over threads... (tiny: M<=512)
// 32 columns x 8 row lanes per block, four independent chains per lane, fixed-order LDS tree (bit-reproducible).  The first
// version was one thread per column walking all M rows as a chain of dependent loads: 61 us for a 256 x 1001 matrix.
__global__ __launch_bounds__(256) void bias_grad_kernel(const bf16_t* __restrict__ dz, int M, int C, int ld,
                                                        float* __restrict__ dbias) {
  __shared__ float red[8][32];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    int m = ry;
    for (; m + 24 < M; m += 32) {
      s0 += bf2f(dz[(size_t)m * ld + c]);
      s1 += bf2f(dz[(size_t)(m + 8) * ld + c]);
      s2 += bf2f(dz[(size_t)(m + 16) * ld + c]);
      s3 += bf2f(dz[(size_t)(m + 24) * ld + c]);
    }
    for (; m < M; m += 8) s0 += bf2f(dz[(size_t)m * ld + c]);
  }
  red[ry][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ry == 0 && c < C) {
    float t = red[0][cx];
#pragma unroll
    for (int r = 1; r < 8; ++r) t += red[r][cx];
    dbias[c] = t;
  }
}
__global__ void cast_kernel(const float* x, bf16_t* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = f2bf(x[i]);
}
__global__ void widen_kernel(const bf16_t* x, float* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = bf2f(x[i]);
}

// ---- softmax cross-entropy (+label smoothing, +KD), one block per row --------------------------------
__device__ float block_reduce(float v, bool is_max, float* sh) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, int ld,
                                                         const float* __restrict__ targets,
                                                         const float* __restrict__ teacher, int B, int C, float eps,
                                                         float T, float loss_scale, float* __restrict__ loss_rows,
                                                         bf16_t* __restrict__ dlogits, int ld_out) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const float* z = logits + (size_t)b * ld;
  const float* y = targets + (size_t)b * C;
  const float* t = teacher ? teacher + (size_t)b * C : nullptr;
  const float invT = t ? 1.0f / T : 1.0f;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, z[c]);
  mx = block_reduce(mx, true, sh);
  float se = 0.f, seT = 0.f, sy = 0.f, syz = 0.f, st = 0.f, stz = 0.f;
  const float smooth = eps / (float)C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float d = z[c] - mx;
    se += __expf(d);
    const float yy = y[c] * (1.0f - eps) + smooth;
    sy += yy;
    syz += yy * d;
    if (t) {
      seT += __expf(d * invT);
      st += t[c];
      stz += t[c] * d * invT;
    }
  }
  se = block_reduce(se, false, sh);
  sy = block_reduce(sy, false, sh);
  syz = block_reduce(syz, false, sh);
  float loss = sy * __logf(se) - syz;  // -sum yy * (d - log se)
  float lseT = 0.f;
  if (t) {
    seT = block_reduce(seT, false, sh);
    st = block_reduce(st, false, sh);
    stz = block_reduce(stz, false, sh);
    lseT = __logf(seT);
    loss += T * T * (st * lseT - stz);
  }
  if (threadIdx.x == 0) loss_rows[b] = loss;
  if (dlogits) {
    const float k = loss_scale / (float)B;
    const float inv_se = 1.0f / se, inv_seT = t ? 1.0f / seT : 0.f;
    bf16_t* dz = dlogits + (size_t)b * ld_out;
    for (int c = threadIdx.x; c < ld_out; c += 256) {
      float g = 0.f;
      if (c < C) {
        const float d = z[c] - mx;
        const float yy = y[c] * (1.0f - eps) + smooth;
        g = __expf(d) * inv_se * sy - yy;
        if (t) g += T * (__expf(d * invT) * inv_seT * st - t[c]);  // T^2 * (1/T)
        g *= k;
      }
      dz[c] = f2bf(g);
    }
  }
}

__global__ void onehot_kernel(const int32_t* labels, float* out, int B, int C) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)B * C) return;
  const int b = (int)(i / C), c = (int)(i - (size_t)b * C);
  out[i] = labels[b] == c ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* x, float* y, int B, int C, float inv_temp) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const float* z = x + (size_t)b * C;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, z[c] * inv_temp);
  mx = block_reduce(mx, true, sh);
  float se = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) se += __expf(z[c] * inv_temp - mx);
  se = block_reduce(se, false, sh);
  const float inv = 1.0f / se;
  for (int c = threadIdx.x; c < C; c += 256) y[(size_t)b * C + c] = __expf(z[c] * inv_temp - mx) * inv;
}

__global__ __launch_bounds__(256) void mean_kernel(const float* x, int n, float* out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i];
  s = block_reduce(s, false, sh);
  if (threadIdx.x == 0) *out = s / (float)n;
}

// ---- input: mixup + mean subtraction + cast + stem halo ------------------------------------------------
// out [Bout][H+6][W+6][4] bf16; one thread per output pixel (8 bytes)
__global__ __launch_bounds__(256) void mixup_meansub_kernel(const void* __restrict__ images, int is_u8, int Bin,
                                                            int Bout, int H, int W, int mixup_type,
                                                            const float* __restrict__ lam1,
                                                            const float* __restrict__ lam2, bf16_t* __restrict__ out) {
  const int Hp = H + 6, Wp = W + 6;
  const size_t total = (size_t)Bout * Hp * Wp;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int wp = (int)(i % Wp);
  const size_t t = i / Wp;
  const int hp = (int)(t % Hp);
  const int b = (int)(t / Hp);
  const int h = hp - 3, w = wp - 3;
  u32x2 o = {0u, 0u};
  if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
    int ia = b, ib = -1;
    float la = 1.0f;
    if (mixup_type == 1) {
      ia = b; ib = Bout + b; la = lam1[b];
    } else if (mixup_type == 2) {
      const int half = Bin / 2;
      if (b < half) { ia = b; ib = half + b; la = lam1[b]; }
      else { ia = b - half; ib = Bin - 1 - (b - half); la = lam2[b - half]; }  // x3 = reverse(x2)
    }
    float va[3], vb[3] = {0.f, 0.f, 0.f};
    const size_t pa = (((size_t)ia * H + h) * W + w) * 3;
    if (is_u8) {
      const uint8_t* p = reinterpret_cast<const uint8_t*>(images);
      for (int c = 0; c < 3; ++c) va[c] = (float)p[pa + c];
      if (ib >= 0) {
        const size_t pb = (((size_t)ib * H + h) * W + w) * 3;
        for (int c = 0; c < 3; ++c) vb[c] = (float)p[pb + c];
      }
    } else {
      const float* p = reinterpret_cast<const float*>(images);
      for (int c = 0; c < 3; ++c) va[c] = p[pa + c];
      if (ib >= 0) {
        const size_t pb = (((size_t)ib * H + h) * W + w) * 3;
        for (int c = 0; c < 3; ++c) vb[c] = p[pb + c];
      }
    }
    const float mean[3] = {123.68f, 116.78f, 103.94f};
    float r[3];
    for (int c = 0; c < 3; ++c) {
      const float xa = va[c] - mean[c];
      r[c] = ib >= 0 ? la * xa + (1.0f - la) * (vb[c] - mean[c]) : xa;
    }
    o.x = pack2bf(r[0], r[1]);
    o.y = pack2bf(r[2], 0.f);
  }
  *reinterpret_cast<u32x2*>(out + i * 4) = o;
}

// uint8 input with W % 4 == 0: one 64-thread block per output row, a thread converts four pixels from three aligned dword
// loads per source image (the per-pixel form above issues three byte loads per pixel: 2.6 TB/s).  Same arithmetic.
__global__ __launch_bounds__(64) void mixup_meansub_rows_kernel(const uint8_t* __restrict__ images, int Bin, int Bout, int H,
                                                                int W, int mixup_type, const float* __restrict__ lam1,
                                                                const float* __restrict__ lam2, bf16_t* __restrict__ out) {
  const int Hp = H + 6, Wp = W + 6;
  const int b = blockIdx.x / Hp, hp = blockIdx.x - b * Hp;
  const int h = hp - 3;
  u32x2* orow = reinterpret_cast<u32x2*>(out + (size_t)blockIdx.x * Wp * 4);
  const u32x2 z = {0u, 0u};
  if ((unsigned)h >= (unsigned)H) {
    for (int wp = threadIdx.x; wp < Wp; wp += 64) orow[wp] = z;
    return;
  }
  if (threadIdx.x < 3) {
    orow[threadIdx.x] = z;
    orow[W + 3 + threadIdx.x] = z;
  }
  int ia = b, ib = -1;
  float la = 1.0f;
  if (mixup_type == 1) {
    ia = b; ib = Bout + b; la = lam1[b];
  } else if (mixup_type == 2) {
    const int half = Bin / 2;
    if (b < half) { ia = b; ib = half + b; la = lam1[b]; }
    else { ia = b - half; ib = Bin - 1 - (b - half); la = lam2[b - half]; }
  }
  const unsigned* pa = reinterpret_cast<const unsigned*>(images + ((size_t)ia * H + h) * W * 3);
  const unsigned* pb = ib >= 0 ? reinterpret_cast<const unsigned*>(images + ((size_t)ib * H + h) * W * 3) : nullptr;
  const float mean[3] = {123.68f, 116.78f, 103.94f};
  for (int q = threadIdx.x; q < (W >> 2); q += 64) {
    unsigned a[3], bb[3] = {0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] = pa[q * 3 + k];
    if (pb) {
#pragma unroll
      for (int k = 0; k < 3; ++k) bb[k] = pb[q * 3 + k];
    }
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      float r[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int byte = px * 3 + c;
        const float xa = (float)((a[byte >> 2] >> ((byte & 3) * 8)) & 0xffu) - mean[c];
        const float xb = (float)((bb[byte >> 2] >> ((byte & 3) * 8)) & 0xffu) - mean[c];
        r[c] = pb ? la * xa + (1.0f - la) * xb : xa;
      }
      u32x2 o;
      o.x = pack2bf(r[0], r[1]);
      o.y = pack2bf(r[2], 0.f);
      orow[3 + q * 4 + px] = o;
    }
  }
}

__global__ __launch_bounds__(256) void stem_pad_kernel(const void* __restrict__ x, int is_f32, bf16_t* __restrict__ out,
                                                       int N, int H, int W) {
  const int Hp = H + 6, Wp = W + 6;
  const size_t total = (size_t)N * Hp * Wp;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int wp = (int)(i % Wp);
  const size_t t = i / Wp;
  const int hp = (int)(t % Hp);
  const int b = (int)(t / Hp);
  const int h = hp - 3, w = wp - 3;
  u32x2 o = {0u, 0u};
  if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
    const size_t p = (((size_t)b * H + h) * W + w) * 3;
    float r[3];
    if (is_f32) {
      const float* xf = reinterpret_cast<const float*>(x);
      for (int c = 0; c < 3; ++c) r[c] = xf[p + c];
    } else {
      const bf16_t* xb = reinterpret_cast<const bf16_t*>(x);
      for (int c = 0; c < 3; ++c) r[c] = bf2f(xb[p + c]);
    }
    o.x = pack2bf(r[0], r[1]);
    o.y = pack2bf(r[2], 0.f);
  }
  *reinterpret_cast<u32x2*>(out + i * 4) = o;
}

__global__ void mixup_labels_kernel(const float* y, int Bin, int Bout, int C, int mixup_type, const float* lam1,
                                    const float* lam2, float* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)Bout * C) return;
  const int b = (int)(i / C), c = (int)(i - (size_t)b * C);
  int ia = b, ib = -1;
  float la = 1.0f;
  if (mixup_type == 1) {
    ia = b; ib = Bout + b; la = lam1[b];
  } else if (mixup_type == 2) {
    const int half = Bin / 2;
    if (b < half) { ia = b; ib = half + b; la = lam1[b]; }
    else { ia = b - half; ib = Bin - 1 - (b - half); la = lam2[b - half]; }
  }
  float v = y[(size_t)ia * C + c];
  if (ib >= 0) v = la * v + (1.0f - la) * y[(size_t)ib * C + c];
  out[i] = v;
}

// ---- optimiser ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, float* __restrict__ a,
                                                  const float* __restrict__ g, bf16_t* __restrict__ wb, size_t n,
                                                  float lr, float mom, float wd, float gs) {
  const size_t nv = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
    // fp32 master weights, momentum and gradients are touched once per step, here: streamed past the caches (the bf16
    // shadow below is what the next forward pass reads)
    f32x4 wv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w + i * 4));
    f32x4 av = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a + i * 4));
    const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + i * 4));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gv[e] * gs + wd * wv[e];
      av[e] = mom * av[e] + gg;
      wv[e] = wv[e] - lr * av[e];
    }
    __builtin_nontemporal_store(wv, reinterpret_cast<f32x4*>(w + i * 4));
    __builtin_nontemporal_store(av, reinterpret_cast<f32x4*>(a + i * 4));
    if (wb) {
      u32x2 o = {pack2bf(wv[0], wv[1]), pack2bf(wv[2], wv[3])};
      *reinterpret_cast<u32x2*>(wb + i * 4) = o;
    }
  }
  // tail
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = nv * 4 + threadIdx.x;
    const float gg = g[i] * gs + wd * w[i];
    a[i] = mom * a[i] + gg;
    w[i] = w[i] - lr * a[i];
    if (wb) wb[i] = f2bf(w[i]);
  }
}

// ---- filter layouts -----------------------------------------------------------------------------------
// KRSC -> CRSK (bf16).  One thread per output element; tiny tensors.
__global__ void filter_transpose_kernel(const bf16_t* w, bf16_t* wt, int K, int RS, int C, int ldk) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n = (size_t)K * RS * C;
  if (i >= n) return;
  // i indexes wt[c][t][k]
  const int k = (int)(i % K);
  const size_t r = i / K;
  const int t = (int)(r % RS);
  const int c = (int)(r / RS);
  wt[((size_t)c * RS + t) * ldk + k] = w[((size_t)k * RS + t) * C + c];
}
// all KRSC -> CRSK copies of a model in ONE launch: table[l] = {src_off, dst_off, K, RS, C, ldk, elem_begin, 0}
// (offsets in elements inside the two flat bf16 arenas); thread e handles flat source element e.
__global__ __launch_bounds__(256) void filter_transpose_batched_kernel(const bf16_t* __restrict__ w,
                                                                       bf16_t* __restrict__ wt,
                                                                       const int* __restrict__ table, int nl,
                                                                       long long total) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  int lo = 0, hi = nl - 1;
  while (lo < hi) {  // last layer whose elem_begin <= e
    const int mid = (lo + hi + 1) >> 1;
    if ((long long)table[mid * 8 + 6] <= e) lo = mid; else hi = mid - 1;
  }
  const int* t = table + lo * 8;
  const int K = t[2], RS = t[3], C = t[4], ldk = t[5];
  const int local = (int)(e - t[6]);
  const int c = local % C;
  const int r = local / C;
  const int tap = r % RS;
  const int k = r / RS;
  wt[(size_t)t[1] + ((size_t)c * RS + tap) * ldk + k] = w[(size_t)t[0] + local];
}

// Tiled form of the same batched transpose: table[l][7] = first 64x64 tile of layer l; a layer has
// RS * ceil(K/64) * ceil(C/64) tiles.  Rows of 128 bytes are read along c, turned through LDS, and written along k
// (the one-thread-per-element kernel above scatters 2-byte stores ldk*2 bytes apart: 0.4 TB/s on a 42 M-parameter net).
__global__ __launch_bounds__(256) void filter_transpose_tiled_kernel(const bf16_t* __restrict__ w,
                                                                     bf16_t* __restrict__ wt,
                                                                     const int* __restrict__ table, int nl) {
  __shared__ unsigned short tile[64][66];
  const int b = blockIdx.x;
  int lo = 0, hi = nl - 1;
  while (lo < hi) {  // last layer whose tile_begin <= b  (block-uniform: scalar loads)
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * 8 + 7] <= b) lo = mid; else hi = mid - 1;
  }
  const int* t = table + lo * 8;
  const int K = t[2], RS = t[3], C = t[4], ldk = t[5];
  const int tc = (C + 63) >> 6, tk = (K + 63) >> 6;
  int local = b - t[7];
  const int ct = local % tc;
  local /= tc;
  const int kt = local % tk;
  const int tap = local / tk;
  const int k0 = kt * 64, c0 = ct * 64;
  const bf16_t* src = w + (size_t)t[0];
  bf16_t* dst = wt + (size_t)t[1];
  const int tid = threadIdx.x;
  const int ch = (tid & 7) * 8, r0 = tid >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = r0 + 32 * i;            // k within the tile
    u32x4 v = {0u, 0u, 0u, 0u};
    if (k0 + row < K && c0 + ch < C)        // C % 8 == 0
      v = *reinterpret_cast<const u32x4*>(src + ((size_t)(k0 + row) * RS + tap) * C + c0 + ch);
    unsigned short* d = &tile[row][ch];
    d[0] = (unsigned short)(v.x & 0xffffu); d[1] = (unsigned short)(v.x >> 16);
    d[2] = (unsigned short)(v.y & 0xffffu); d[3] = (unsigned short)(v.y >> 16);
    d[4] = (unsigned short)(v.z & 0xffffu); d[5] = (unsigned short)(v.z >> 16);
    d[6] = (unsigned short)(v.w & 0xffffu); d[7] = (unsigned short)(v.w >> 16);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int crow = r0 + 32 * i;           // c within the tile
    if (c0 + crow < C && k0 + ch < ldk) {   // ldk % 8 == 0; columns K..ldk-1 are written as zeros
      u32x4 v;
      v.x = (unsigned)tile[ch + 0][crow] | ((unsigned)tile[ch + 1][crow] << 16);
      v.y = (unsigned)tile[ch + 2][crow] | ((unsigned)tile[ch + 3][crow] << 16);
      v.z = (unsigned)tile[ch + 4][crow] | ((unsigned)tile[ch + 5][crow] << 16);
      v.w = (unsigned)tile[ch + 6][crow] | ((unsigned)tile[ch + 7][crow] << 16);
      *reinterpret_cast<u32x4*>(dst + ((size_t)(c0 + crow) * RS + tap) * ldk + k0 + ch) = v;
    }
  }
}

__global__ void stem_pack_kernel(const float* w, bf16_t* wp, int K, int ks, int L) {  // [K][k][k][3] -> [K][k][L]
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K * ks * L) return;
  const int j = i % L, r = (i / L) % ks, k = i / (L * ks);
  const int s = j >> 2, c = j & 3;
  float v = 0.f;
  if (s < ks && c < 3) v = w[((k * ks + r) * ks + s) * 3 + c];
  wp[i] = f2bf(v);
}
__global__ void stem_unpack_kernel(const float* dwp, float* dw, int K, int ks, int L) {  // [K][k][L] -> [K][k][k][3]
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K * ks * ks * 3) return;
  const int c = i % 3, s = (i / 3) % ks, r = (i / (3 * ks)) % ks, k = i / (3 * ks * ks);
  dw[i] = dwp[(k * ks + r) * L + s * 4 + c];
}

// ---- debug: semantics probe of ds_read_b64_tr_b16 (the wgrad operand read) -----------------------------
__global__ void tr_probe_kernel(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[256];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = (short)i;
  __syncthreads();
  typedef __attribute__((ext_vector_type(4))) short s4;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + l * 4));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

// ---- debug direct convolutions ----------------------------------------------------------------------------
struct NaiveArgs {
  int N, H, W, C, K, R, S, stride, pad, Ho, Wo, ldy, out_f32;
  long long img_pitch;
  int row_pitch, pix_pitch;
};
__global__ void naive_fprop_kernel(const bf16_t* x, const bf16_t* w, void* y, NaiveArgs a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)a.N * a.Ho * a.Wo * a.K;
  if (i >= total) return;
  const int k = (int)(i % a.K);
  size_t m = i / a.K;
  const int wo = (int)(m % a.Wo);
  const int ho = (int)((m / a.Wo) % a.Ho);
  const int n = (int)(m / ((size_t)a.Wo * a.Ho));
  float acc = 0.f;
  for (int r = 0; r < a.R; ++r) {
    const int ih = ho * a.stride + r - a.pad;
    if ((unsigned)ih >= (unsigned)a.H) continue;
    for (int s = 0; s < a.S; ++s) {
      const int iw = wo * a.stride + s - a.pad;
      if ((unsigned)iw >= (unsigned)a.W) continue;
      const bf16_t* xp = x + (size_t)n * a.img_pitch + (size_t)ih * a.row_pitch + (size_t)iw * a.pix_pitch;
      const bf16_t* wp = w + (((size_t)k * a.R + r) * a.S + s) * a.C;
      for (int c = 0; c < a.C; ++c) acc += bf2f(xp[c]) * bf2f(wp[c]);
    }
  }
  if (a.out_f32) reinterpret_cast<float*>(y)[m * a.ldy + k] = acc;
  else reinterpret_cast<bf16_t*>(y)[m * a.ldy + k] = f2bf(acc);
}
__global__ void naive_dgrad_kernel(const bf16_t* dy, const bf16_t* w, bf16_t* dx, NaiveArgs a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)a.N * a.H * a.W * a.C;
  if (i >= total) return;
  const int c = (int)(i % a.C);
  size_t m = i / a.C;
  const int wi = (int)(m % a.W);
  const int hi = (int)((m / a.W) % a.H);
  const int n = (int)(m / ((size_t)a.W * a.H));
  float acc = 0.f;
  for (int r = 0; r < a.R; ++r) {
    const int th = hi + a.pad - r;
    if (th < 0 || th % a.stride) continue;
    const int ho = th / a.stride;
    if (ho >= a.Ho) continue;
    for (int s = 0; s < a.S; ++s) {
      const int tw = wi + a.pad - s;
      if (tw < 0 || tw % a.stride) continue;
      const int wo = tw / a.stride;
      if (wo >= a.Wo) continue;
      const bf16_t* gp = dy + (((size_t)n * a.Ho + ho) * a.Wo + wo) * a.K;
      for (int k = 0; k < a.K; ++k) acc += bf2f(gp[k]) * bf2f(w[(((size_t)k * a.R + r) * a.S + s) * a.C + c]);
    }
  }
  dx[i] = f2bf(acc);
}
__global__ void naive_wgrad_kernel(const bf16_t* x, const bf16_t* dy, float* dw, NaiveArgs a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)a.K * a.R * a.S * a.C;
  if (i >= total) return;
  const int c = (int)(i % a.C);
  const int s = (int)((i / a.C) % a.S);
  const int r = (int)((i / ((size_t)a.C * a.S)) % a.R);
  const int k = (int)(i / ((size_t)a.C * a.S * a.R));
  float acc = 0.f;
  for (int n = 0; n < a.N; ++n)
    for (int ho = 0; ho < a.Ho; ++ho) {
      const int ih = ho * a.stride + r - a.pad;
      if ((unsigned)ih >= (unsigned)a.H) continue;
      for (int wo = 0; wo < a.Wo; ++wo) {
        const int iw = wo * a.stride + s - a.pad;
        if ((unsigned)iw >= (unsigned)a.W) continue;
        acc += bf2f(dy[(((size_t)n * a.Ho + ho) * a.Wo + wo) * a.ldy + k]) *
               bf2f(x[(size_t)n * a.img_pitch + (size_t)ih * a.row_pitch + (size_t)iw * a.pix_pitch + c]);
      }
    }
  dw[i] = acc;
}
NaiveArgs naive_args(const asm_conv_desc* d) {
  NaiveArgs a;
  a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->C; a.K = d->K; a.R = d->R; a.S = d->S;
  a.stride = d->stride; a.pad = d->pad; a.Ho = d->Ho; a.Wo = d->Wo;
  a.ldy = d->ldy ? d->ldy : d->K; a.out_f32 = d->out_f32;
  a.img_pitch = d->x_img_pitch ? d->x_img_pitch : (long long)d->H * d->W * d->C;
  a.row_pitch = d->x_row_pitch ? d->x_row_pitch : d->W * d->C;
  a.pix_pitch = d->x_pix_pitch ? d->x_pix_pitch : d->C;
  return a;
}

}  // namespace

#define VEC8_OK(name) ASM_REQUIRE(n % 8 == 0, name ": element count must be a multiple of 8")

extern "C" int asm_relu_fwd(const void* x, void* y, size_t n, void* stream) {
  ASM_REQUIRE(x && y, "relu_fwd: null pointer");
  VEC8_OK("relu_fwd");
  ASM_LAUNCH(relu_fwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)y, n / 8);
  ASM_CHECK_LAUNCH("relu_fwd");
  return ASM_OK;
}
extern "C" int asm_relu_bwd(const void* dy, const void* y, void* dx, size_t n, void* stream) {
  ASM_REQUIRE(dy && y && dx, "relu_bwd: null pointer");
  VEC8_OK("relu_bwd");
  ASM_LAUNCH(relu_bwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     (const bf16_t*)y, (bf16_t*)dx, n / 8);
  ASM_CHECK_LAUNCH("relu_bwd");
  return ASM_OK;
}
extern "C" int asm_mask_apply(const void* dy, const uint8_t* mask, void* dx, size_t n, void* stream) {
  ASM_REQUIRE(dy && mask && dx, "mask_apply: null pointer");
  VEC8_OK("mask_apply");
  ASM_LAUNCH(mask_apply_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, mask,
                     (bf16_t*)dx, n / 8);
  ASM_CHECK_LAUNCH("mask_apply");
  return ASM_OK;
}
extern "C" int asm_add_bf16(const void* a, const void* b, void* out, size_t n, void* stream) {
  ASM_REQUIRE(a && b && out, "add_bf16: null pointer");
  VEC8_OK("add_bf16");
  ASM_LAUNCH(add_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                     (const bf16_t*)b, (bf16_t*)out, n / 8);
  ASM_CHECK_LAUNCH("add_bf16");
  return ASM_OK;
}
extern "C" int asm_bias_add_f32(float* y, const float* bias, int M, int C, int ldy, void* stream) {
  ASM_REQUIRE(y && bias && M > 0 && C > 0 && ldy >= C, "bias_add: bad arguments");
  ASM_LAUNCH(bias_add_kernel, dim3((unsigned)cdivz((size_t)M * C, 256)), dim3(256), 0, (hipStream_t)stream, y,
                     bias, M, C, ldy);
  ASM_CHECK_LAUNCH("bias_add");
  return ASM_OK;
}
extern "C" int asm_bias_grad_bf16(const void* dz, int M, int C, int ld, float* dbias, void* stream) {
  ASM_REQUIRE(dz && dbias && M > 0 && C > 0 && ld >= C, "bias_grad: bad arguments");
  ASM_LAUNCH(bias_grad_kernel, dim3(cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dz, M, C,
                     ld, dbias);
  ASM_CHECK_LAUNCH("bias_grad");
  return ASM_OK;
}
extern "C" int asm_cast_f32_to_bf16(const float* x, void* y, size_t n, void* stream) {
  ASM_REQUIRE(x && y, "cast: null pointer");
  if (n == 0) return ASM_OK;
  ASM_LAUNCH(cast_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n);
  ASM_CHECK_LAUNCH("cast");
  return ASM_OK;
}

extern "C" int asm_cast_bf16_to_f32(const void* x, float* y, size_t n, void* stream) {
  ASM_REQUIRE(x && y, "cast: null pointer");
  if (n == 0) return ASM_OK;
  ASM_LAUNCH(widen_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, n);
  ASM_CHECK_LAUNCH("cast_bf16_to_f32");
  return ASM_OK;
}

extern "C" int asm_softmax_ce(const float* logits, int ld, const float* targets, const float* teacher, int B, int C,
                              float label_smoothing, float kd_temp, float loss_scale, float* loss_rows, void* dlogits,
                              int ld_out, void* stream) {
  ASM_REQUIRE(logits && targets && loss_rows && B > 0 && C > 0 && ld >= C, "softmax_ce: bad arguments");
  ASM_REQUIRE(!teacher || kd_temp > 0.f, "softmax_ce: teacher given but kd_temp <= 0");
  ASM_REQUIRE(!dlogits || ld_out >= C, "softmax_ce: bad ld_out");
  ASM_LAUNCH(softmax_ce_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ld, targets, teacher, B, C,
                     label_smoothing, kd_temp, loss_scale, loss_rows, (bf16_t*)dlogits, ld_out);
  ASM_CHECK_LAUNCH("softmax_ce");
  return ASM_OK;
}
extern "C" int asm_onehot(const int32_t* labels, float* out, int B, int C, void* stream) {
  ASM_REQUIRE(labels && out && B > 0 && C > 0, "onehot: bad arguments");
  ASM_LAUNCH(onehot_kernel, dim3((unsigned)cdivz((size_t)B * C, 256)), dim3(256), 0, (hipStream_t)stream,
                     labels, out, B, C);
  ASM_CHECK_LAUNCH("onehot");
  return ASM_OK;
}
extern "C" int asm_softmax_rows(const float* x, float* y, int B, int C, float inv_temp, void* stream) {
  ASM_REQUIRE(x && y && B > 0 && C > 0, "softmax_rows: bad arguments");
  ASM_LAUNCH(softmax_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, y, B, C, inv_temp);
  ASM_CHECK_LAUNCH("softmax_rows");
  return ASM_OK;
}
extern "C" int asm_mean_f32(const float* x, int n, float* out, void* stream) {
  ASM_REQUIRE(x && out && n > 0, "mean: bad arguments");
  ASM_LAUNCH(mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, out);
  ASM_CHECK_LAUNCH("mean");
  return ASM_OK;
}

extern "C" int asm_mixup_meansub(const void* images, int is_u8, int Bin, int H, int W, int mixup_type,
                                 const float* lam1, const float* lam2, void* out, void* stream) {
  ASM_REQUIRE(images && out && Bin > 0 && H > 0 && W > 0, "mixup_meansub: bad arguments");
  ASM_REQUIRE(mixup_type >= 0 && mixup_type <= 2, "mixup_meansub: mixup_type must be 0, 1 or 2");
  ASM_REQUIRE(mixup_type == 0 || (lam1 && Bin % 2 == 0), "mixup_meansub: mixup needs lam1 and an even batch");
  ASM_REQUIRE(mixup_type != 2 || lam2, "mixup_meansub: mixup_type 2 needs lam2");
  const int Bout = mixup_type == 1 ? Bin / 2 : Bin;
  const size_t total = (size_t)Bout * (H + 6) * (W + 6);
  if (is_u8 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(images) & 3) == 0)
    ASM_LAUNCH(mixup_meansub_rows_kernel, dim3((unsigned)(Bout * (H + 6))), dim3(64), 0, (hipStream_t)stream,
               (const uint8_t*)images, Bin, Bout, H, W, mixup_type, lam1, lam2, (bf16_t*)out);
  else
    ASM_LAUNCH(mixup_meansub_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, (hipStream_t)stream, images,
               is_u8, Bin, Bout, H, W, mixup_type, lam1, lam2, (bf16_t*)out);
  ASM_CHECK_LAUNCH("mixup_meansub");
  return ASM_OK;
}
extern "C" int asm_stem_pad_input(const void* x, int x_is_f32, void* xp, int N, int H, int W, void* stream) {
  ASM_REQUIRE(x && xp && N > 0 && H > 0 && W > 0, "stem_pad_input: bad arguments");
  const size_t total = (size_t)N * (H + 6) * (W + 6);
  ASM_LAUNCH(stem_pad_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, (hipStream_t)stream, x, x_is_f32,
                     (bf16_t*)xp, N, H, W);
  ASM_CHECK_LAUNCH("stem_pad_input");
  return ASM_OK;
}
extern "C" int asm_mixup_labels(const float* y, int Bin, int C, int mixup_type, const float* lam1, const float* lam2,
                                float* out, void* stream) {
  ASM_REQUIRE(y && out && Bin > 0 && C > 0, "mixup_labels: bad arguments");
  ASM_REQUIRE(mixup_type >= 0 && mixup_type <= 2, "mixup_labels: mixup_type must be 0, 1 or 2");
  ASM_REQUIRE(mixup_type == 0 || (lam1 && Bin % 2 == 0), "mixup_labels: mixup needs lam1 and an even batch");
  ASM_REQUIRE(mixup_type != 2 || lam2, "mixup_labels: mixup_type 2 needs lam2");
  const int Bout = mixup_type == 1 ? Bin / 2 : Bin;
  ASM_LAUNCH(mixup_labels_kernel, dim3((unsigned)cdivz((size_t)Bout * C, 256)), dim3(256), 0,
                     (hipStream_t)stream, y, Bin, Bout, C, mixup_type, lam1, lam2, out);
  ASM_CHECK_LAUNCH("mixup_labels");
  return ASM_OK;
}

extern "C" int asm_sgd_momentum(float* w, float* accum, const float* grad, void* w_bf16, size_t n, float lr,
                                float momentum, float weight_decay, float grad_scale, void* stream) {
  ASM_REQUIRE(w && accum && grad, "sgd_momentum: null pointer");
  if (n == 0) return ASM_OK;
  ASM_LAUNCH(sgd_kernel, dim3(ew_grid(cdivz(n, 4))), dim3(256), 0, (hipStream_t)stream, w, accum, grad,
                     (bf16_t*)w_bf16, n, lr, momentum, weight_decay, grad_scale);
  ASM_CHECK_LAUNCH("sgd_momentum");
  return ASM_OK;
}

extern "C" int asm_filter_transpose(const void* w_krsc, void* w_crsk, int K, int R, int S, int C, int ldk,
                                    void* stream) {
  ASM_REQUIRE(w_krsc && w_crsk && K > 0 && R > 0 && S > 0 && C > 0 && (ldk == 0 || ldk >= K),
              "filter_transpose: bad arguments");
  if (ldk == 0) ldk = K;
  const size_t n = (size_t)K * R * S * C;
  ASM_LAUNCH(filter_transpose_kernel, dim3((unsigned)cdivz(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)w_krsc, (bf16_t*)w_crsk, K, R * S, C, ldk);
  ASM_CHECK_LAUNCH("filter_transpose");
  return ASM_OK;
}
extern "C" int asm_stem_pack_filter(const float* w_krsc3, void* w_packed, int K, int ksize, void* stream) {
  ASM_REQUIRE(w_krsc3 && w_packed && K > 0 && (ksize == 3 || ksize == 7), "stem_pack_filter: bad arguments");
  const int L = (4 * ksize + 7) & ~7;
  ASM_LAUNCH(stem_pack_kernel, dim3(cdiv(K * ksize * L, 256)), dim3(256), 0, (hipStream_t)stream, w_krsc3,
                     (bf16_t*)w_packed, K, ksize, L);
  ASM_CHECK_LAUNCH("stem_pack_filter");
  return ASM_OK;
}
extern "C" int asm_stem_unpack_grad(const float* dw_packed, float* dw_krsc3, int K, int ksize, void* stream) {
  ASM_REQUIRE(dw_packed && dw_krsc3 && K > 0 && (ksize == 3 || ksize == 7), "stem_unpack_grad: bad arguments");
  const int L = (4 * ksize + 7) & ~7;
  ASM_LAUNCH(stem_unpack_kernel, dim3(cdiv(K * ksize * ksize * 3, 256)), dim3(256), 0, (hipStream_t)stream,
                     dw_packed, dw_krsc3, K, ksize, L);
  ASM_CHECK_LAUNCH("stem_unpack_grad");
  return ASM_OK;
}

extern "C" int asm_conv2d_fprop_naive(const asm_conv_desc* d, const void* x, const void* w, void* y, void* stream) {
  ASM_REQUIRE(d && x && w && y, "fprop_naive: null pointer");
  NaiveArgs a = naive_args(d);
  const size_t total = (size_t)a.N * a.Ho * a.Wo * a.K;
  ASM_LAUNCH(naive_fprop_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)w, y, a);
  ASM_CHECK_LAUNCH("fprop_naive");
  return ASM_OK;
}
extern "C" int asm_conv2d_dgrad_naive(const asm_conv_desc* d, const void* dy, const void* w_krsc, void* dx,
                                      void* stream) {
  ASM_REQUIRE(d && dy && w_krsc && dx, "dgrad_naive: null pointer");
  NaiveArgs a = naive_args(d);
  const size_t total = (size_t)a.N * a.H * a.W * a.C;
  ASM_LAUNCH(naive_dgrad_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (const bf16_t*)w_krsc, (bf16_t*)dx, a);
  ASM_CHECK_LAUNCH("dgrad_naive");
  return ASM_OK;
}
extern "C" int asm_conv2d_wgrad_naive(const asm_conv_desc* d, const void* x, const void* dy, float* dw, void* stream) {
  ASM_REQUIRE(d && x && dy && dw, "wgrad_naive: null pointer");
  NaiveArgs a = naive_args(d);
  const size_t total = (size_t)a.K * a.R * a.S * a.C;
  ASM_LAUNCH(naive_wgrad_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)dy, dw, a);
  ASM_CHECK_LAUNCH("wgrad_naive");
  return ASM_OK;
}

extern "C" int asm_debug_tr_probe(void* out256_i16, void* stream) {
  ASM_REQUIRE(out256_i16, "tr_probe: null pointer");
  ASM_LAUNCH(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (short*)out256_i16);
  ASM_CHECK_LAUNCH("tr_probe");
  return ASM_OK;
}

extern "C" int asm_filter_transpose_tiled(const void* w_arena, void* wt_arena, const int32_t* table, int nlayers,
                                          int total_tiles, void* stream) {
  ASM_REQUIRE(w_arena && wt_arena && table && nlayers > 0 && total_tiles > 0, "filter_transpose_tiled: bad arguments");
  ASM_LAUNCH(filter_transpose_tiled_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)w_arena, (bf16_t*)wt_arena, (const int*)table, nlayers);
  ASM_CHECK_LAUNCH("filter_transpose_tiled");
  return ASM_OK;
}

extern "C" int asm_filter_transpose_batched(const void* w_arena, void* wt_arena, const int32_t* table, int nlayers,
                                            long long total_elems, void* stream) {
  ASM_REQUIRE(w_arena && wt_arena && table && nlayers > 0 && total_elems > 0, "filter_transpose_batched: bad arguments");
  ASM_LAUNCH(filter_transpose_batched_kernel, dim3((unsigned)cdivz((size_t)total_elems, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)w_arena, (bf16_t*)wt_arena, (const int*)table, nlayers,
                     total_elems);
  ASM_CHECK_LAUNCH("filter_transpose_batched");
  return ASM_OK;
}
