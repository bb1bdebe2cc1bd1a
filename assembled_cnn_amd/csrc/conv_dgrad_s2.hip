// Input gradient of a 3x3 / stride-2 / pad-1 convolution in ONE launch (gfx950), for the two narrow high-resolution
// layers of the Assemble-ResNet stems and BigLittle branches (112 x 112 x 64 -> 56 x 56 x 64 and the like).
//
//   dx(2hh + ph, 2ww + pw, c) = sum over the taps (r, s) with r = ph + 1 (mod 2), s = pw + 1 (mod 2) of
//                               dy(hh + (ph + 1 - r) / 2, ww + (pw + 1 - s) / 2, k) * w(k, r, s, c)
//
// asm_conv2d_dgrad already splits dx into its four (ph, pw) parity classes (1 / 2 / 2 / 4 taps instead of 9 passes over
// 75 % zeros), but as FOUR launches of the generic gather-GEMM: each re-reads dy from HBM / L2, writes every other pixel
// of dx, and is a 1-4 step K loop whose prologue, DMA round trips and epilogue are all exposed -- 0.30 ms in situ for
// 118 GFLOP and 0.46 GB (198 TFLOP/s at 1.7 TB/s: neither bound).
// Here a workgroup owns an 8 x 8 patch of dy (-> the 16 x 16 patch of dx above it):
//   * the WHOLE filter slice of a wave (its 32 input channels x 9 taps x 64 output channels) lives in REGISTERS for the
//     kernel's lifetime (36 fragments = 144 VGPRs; persistent workgroups): no filter traffic per patch at all;
//   * the 9 x 9 dy halo of the patch is staged in LDS once (row pitch 144 bytes: conflict-free ds_read_b128 at every
//     shift), the next patch's halo is prefetched into registers under the MFMAs;
//   * the four classes accumulate side by side (4 x 16 accumulator registers) and the dx patch leaves through LDS as whole
//     2 KB rows, with the gradient fan-in addend (and its packed ReLU mask) added on the way out.
// The products are accumulated in exactly the order of the parity-class launches (taps of a class in filter order, 16
// output channels at a time), so the result is BIT-IDENTICAL to them (tests/test_gpu_conv.py).
#include "common.h"

namespace {

struct S2Args {
  const void* dy;          // [N][Ho][Wo][64] bf16
  const void* wt;          // [64 c][3][3][64 k] bf16 (CRSK)
  void* dx;                // [N][2 Ho][2 Wo][64] bf16
  const void* addend;      // optional, dx-shaped
  const uint8_t* mask;     // optional packed ReLU mask of the addend ([N * H * W][8] bytes)
  unsigned dy_bytes;
  int N, Ho, Wo;           // dy geometry (multiples of 8)
  int patches, ppi, ppr;   // total, per image, per patch row
};

constexpr int S2_RB = 64 * 2 + 16;           // halo pixel pitch (bytes)
constexpr int S2_HALO = 81 * S2_RB;          // 9 x 9 pixels
constexpr int S2_OB = 64 * 2 + 16;           // output staging pixel pitch
constexpr int S2_STAGE = 256 * S2_OB;        // 16 x 16 pixels
constexpr int S2_LDS = S2_HALO + S2_STAGE;   // 48.5 KB

// class (ph, pw) -> its taps in filter order: (filter tap index, dy row shift, dy column shift)
struct S2Tap { int t, dh, dw; };
__device__ constexpr S2Tap s2_taps[4][4] = {
    {{4, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}},            // (0, 0): (1, 1)
    {{3, 0, 1}, {5, 0, 0}, {0, 0, 0}, {0, 0, 0}},            // (0, 1): (1, 0), (1, 2)
    {{1, 1, 0}, {7, 0, 0}, {0, 0, 0}, {0, 0, 0}},            // (1, 0): (0, 1), (2, 1)
    {{0, 1, 1}, {2, 1, 0}, {6, 0, 1}, {8, 0, 0}},            // (1, 1): (0, 0), (0, 2), (2, 0), (2, 2)
};
__device__ constexpr int s2_ntaps[4] = {1, 2, 2, 4};

// One workgroup per CU (the filter slice + accumulators + two patches of prefetch need ~330 of the 512 registers of a
// lane): nothing overlaps a workgroup's phases from outside, so every global load of patch i + 1 -- the dy halo AND the
// addend / mask vectors its copy-out will need -- is issued before patch i is multiplied, and the stores are fire and
// forget: the only waits left are on data that has had a whole patch time to arrive.
__global__ __launch_bounds__(256, 1) void dgrad_s2_kernel(S2Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* hs = smem;                  // dy halo of the current patch
  unsigned char* os = smem + S2_HALO;        // dx patch on its way out
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;   // pixel half (32 of the 64 dy pixels), input-channel half (32 of 64)
  const int l31 = lane & 31, lhi = lane >> 5;

  // this workgroup's run of patches (XCD-contiguous, as the other persistent kernels)
  int t_begin, t_end;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per = p.patches / nb, extra = p.patches - per * nb;
    t_begin = logical * per + (logical < extra ? logical : extra);
    t_end = t_begin + per + (logical < extra ? 1 : 0);
  }

  // ---- the wave's filter slice, once: fw[t][kk] = wt[c = 32 wn + l31][t][16 kk + 8 lhi .. + 8] ----
  bf16x8 fw[9][4];
  {
    const bf16_t* w = reinterpret_cast<const bf16_t*>(p.wt) + (size_t)(wn * 32 + l31) * (9 * 64) + lhi * 8;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) fw[t][kk] = *reinterpret_cast<const bf16x8*>(w + t * 64 + kk * 16);
  }

  const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dy_bytes);
  const int H = 2 * p.Ho, W = 2 * p.Wo;
  const int ck = tid & 7;
  // halo vector i of a thread: pixel hp = i / 8 of the 9 x 9 halo, 16-byte chunk i % 8
  constexpr int HP = (81 * 8 + 255) / 256;   // 3
  u32x4 hv[HP], av[8], avn[8];
  unsigned amk[8], amkn[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) amk[q] = amkn[q] = 0xffu;
  auto patch_pix0 = [&](int patch) -> size_t {
    const int img = patch / p.ppi, rem = patch - img * p.ppi;
    const int py = rem / p.ppr, px = rem - py * p.ppr;
    return ((size_t)img * H + (size_t)py * 16) * W + (size_t)px * 16;
  };
  // everything patch `patch` will read from global memory: its dy halo (-> hv) and, with a fan-in addend, the eight
  // vectors + mask bytes of this thread's copy-out passes (-> a, m)
  auto prefetch = [&](int patch, u32x4 (&a)[8], unsigned (&m)[8]) {
    const int img = patch / p.ppi, rem = patch - img * p.ppi;
    const int py = rem / p.ppr, px = rem - py * p.ppr;
    const int hh0 = py * 8, ww0 = px * 8;
#pragma unroll
    for (int k = 0; k < HP; ++k) {
      const int i = k * 256 + tid;
      const int hp = i >> 3, c8 = i & 7;
      const int hy = hp / 9, hx = hp - hy * 9;
      const int gy = hh0 + hy, gx = ww0 + hx;
      const bool ok = i < 81 * 8 && gy < p.Ho && gx < p.Wo;          // the row / column past the map reads as zeros
      const unsigned off = ((((unsigned)img * (unsigned)p.Ho + (unsigned)gy) * (unsigned)p.Wo + (unsigned)gx) * 64u + (unsigned)c8 * 8u) * 2u;
      hv[k] = __builtin_amdgcn_raw_buffer_load_b128(rdy, ok ? off : ASM_OOB, 0, 0);
    }
    if (p.addend) {      // workgroup-uniform
      const size_t pix0 = ((size_t)img * H + (size_t)py * 16) * W + (size_t)px * 16;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int op = q * 32 + (tid >> 3);
        const size_t gp = pix0 + (size_t)(op >> 4) * W + (op & 15);
        a[q] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p.addend) + gp * 64 + ck * 8);
        if (p.mask) m[q] = (unsigned)p.mask[gp * 8 + ck];
      }
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int k = 0; k < HP; ++k) {
      const int i = k * 256 + tid;
      if (i < 81 * 8) *reinterpret_cast<u32x4*>(hs + (i >> 3) * S2_RB + (i & 7) * 16) = hv[k];
    }
  };

  // byte offset (inside the halo) of this lane's dy pixel at shift (0, 0): pixel i = 32 wm + l31 of the 8 x 8 patch
  const int pi = wm * 32 + l31;
  const unsigned xb00 = (unsigned)(((pi >> 3) * 9 + (pi & 7)) * S2_RB + lhi * 16);

  if (t_begin < t_end) prefetch(t_begin, avn, amkn);
#pragma unroll 1
  for (int patch = t_begin; patch < t_end; ++patch) {
    __syncthreads();                       // previous patch: every wave is done with the halo and the copy-out with `os`
    store_halo();
#pragma unroll
    for (int q = 0; q < 8; ++q) {          // this patch's addend vectors (loaded one patch ago)
      av[q] = avn[q];
      amk[q] = amkn[q];
    }
    __syncthreads();
    if (patch + 1 < t_end) prefetch(patch + 1, avn, amkn);     // in flight under this patch's MFMAs and stores

    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    // class by class, a class's taps in filter order, 16 output channels at a time: the parity-class launches' order
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < s2_ntaps[c]) {
          const S2Tap tp = s2_taps[c][j];
          const unsigned xb = xb00 + (unsigned)((tp.dh * 9 + tp.dw) * S2_RB);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 fx = *reinterpret_cast<const bf16x8*>(hs + xb + kk * 32);
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[tp.t][kk], fx, acc[c], 0, 0, 0);
          }
        }

    // ---- dx patch -> LDS: pixel (2 (i / 8) + ph, 2 (i % 8) + pw) of the 16 x 16 patch, channels 32 wn + ... ----
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ph = c >> 1, pw = c & 1;
      const int opix = (2 * (pi >> 3) + ph) * 16 + 2 * (pi & 7) + pw;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 v;
        v.x = pack2bf(acc[c][4 * g], acc[c][4 * g + 1]);
        v.y = pack2bf(acc[c][4 * g + 2], acc[c][4 * g + 3]);
        *reinterpret_cast<u32x2*>(os + opix * S2_OB + (wn * 32 + 8 * g + 4 * lhi) * 2) = v;
      }
    }
    __syncthreads();
    // ---- copy-out: 256 pixels x 8 chunks of 16 bytes = 8 passes; a dx row of the patch is 2 KB contiguous ----
    {
      const size_t pix0 = patch_pix0(patch);
      bf16_t* dx = reinterpret_cast<bf16_t*>(p.dx);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int op = q * 32 + (tid >> 3);
        const size_t gp = pix0 + (size_t)(op >> 4) * W + (op & 15);
        u32x4 v = *reinterpret_cast<const u32x4*>(os + op * S2_OB + ck * 16);
        if (p.addend) {
          float fv[8], fa[8];
          unpack8(v, fv);
          unpack8(av[q], fa);
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] += ((amk[q] >> e) & 1u) ? fa[e] : 0.f;
          v = pack8(fv);
        }
        *reinterpret_cast<u32x4*>(dx + gp * 64 + ck * 8) = v;
      }
    }
  }
}

}  // namespace

// returns 1 when the layer is not one this kernel covers (the caller falls back to the parity-class launches)
int asm_dgrad_s2_try(const asm_conv_desc* d, const void* dy, const void* wt, const void* addend, const uint8_t* addend_mask,
                     void* dx, void* stream) {
  if (asm_tune().dgrad_parity < 2) return 1;
  if (d->R != 3 || d->S != 3 || d->stride != 2 || d->pad != 1 || d->C != 64 || d->K != 64) return 1;
  if (d->H != 2 * d->Ho || d->W != 2 * d->Wo || d->Ho % 8 || d->Wo % 8) return 1;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(wt) | reinterpret_cast<uintptr_t>(dx) |
       reinterpret_cast<uintptr_t>(addend)) & 15) return 1;
  S2Args a;
  a.dy = dy; a.wt = wt; a.dx = dx; a.addend = addend; a.mask = addend_mask;
  a.dy_bytes = (unsigned)((size_t)d->N * d->Ho * d->Wo * 64 * 2);
  a.N = d->N; a.Ho = d->Ho; a.Wo = d->Wo;
  a.ppr = d->Wo / 8; a.ppi = a.ppr * (d->Ho / 8); a.patches = a.ppi * d->N;
  const int grid = a.patches < 256 ? a.patches : 256;      // one persistent workgroup per CU
  ASM_LAUNCH(dgrad_s2_kernel, dim3(grid), dim3(256), S2_LDS, (hipStream_t)stream, a);
  asm_last_conv_kernel = 5;
  ASM_CHECK_LAUNCH("dgrad_s2_kernel");
  return ASM_OK;
}
