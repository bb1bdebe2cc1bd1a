// Probe (round 6): a PERSISTENT, filter-resident streaming kernel for the 1x1 layers whose filter fits LDS.
//
//   Y[m][n] = sum_c X[m][c] * W[n][c]  (+ addend[m][n])        bf16 in, fp32 accumulate, bf16 out
//
// What the library runs for these layers (igemm1 / igemm2: one workgroup per 128 x 128 tile, 3 - 6 workgroups per CU) reaches
// 4.0 - 4.8 TB/s of in + out traffic on them (tools/conv_bench.py); a copy reaches 6.3.  The question the probe answers: does a
// workgroup that STAYS -- filter loaded once, X tiles streamed through a ring by a loader wave, Y tiles stored straight from
// the accumulators while the next tiles' loads are in flight -- get closer?
//   * one workgroup = 1 loader wave + NC consumer waves, one per CU (LDS: filter K x C + NS ring slots of BM x C);
//   * loader: per tile, LDS-DMA of the BM x C tile into slot i % NS (lane-linear destination, XOR swizzle on the source),
//     counted vmcnt (it issues loads only, so the count is exact), ONE s_barrier per tile hands slot i to the consumers;
//   * consumer wave w: rows 32 w .. 32 w + 31 of the tile against all K channels: fragments by ds_read_b128, MFMA 32x32x16 with
//     the filter as the A operand (an accumulator quad = 4 consecutive channels of one pixel), then bf16 pack, one
//     v_permlane32_swap per channel-group pair (lanes l and l + 32 hold the two halves of 8 consecutive channels) and 16-byte
//     global stores straight from registers: no LDS round trip, no barrier in the epilogue; the optional addend is loaded with
//     the same addresses at the top of the tile;
//   * stores and loads of one wave share vmcnt on gfx950 and complete out of order with respect to each other, which is why
//     the loads live in a wave of their own.
// Checked against fp64 sums on a sample of rows.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/stream1x1_probe.hip
// -o tools/probes/bin/stream1x1_probe ; run: stream1x1_probe [M]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lptr_t;

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);            \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }
__device__ __forceinline__ float bflo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// 16-byte-slot XOR swizzle of an LDS row of RB bytes so that 16 rows distinct mod 16 read at one chunk index cover the 16
// slots of the 256-byte bank row
template <int RB>
__device__ __forceinline__ int swz(int row) {
  return RB >= 256 ? (row & 15) : ((row >> 1) & 7);
}

struct Args {
  const void* x;
  const void* w;
  void* y;
  const void* addend;
  unsigned x_bytes, w_bytes;
  int M, n_tiles;
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int C, int K, int BM, int NS, bool ADDEND>
__global__ __launch_bounds__(64 * (BM / 32 + 1)) void stream1x1_kernel(Args p) {
  constexpr int NC = BM / 32;                    // consumer waves
  constexpr int RB = C * 2;                      // row bytes of X and W rows in LDS
  constexpr int CPR = RB / 16;                   // 16-byte chunks per row
  constexpr int WBYTES = K * RB, SLOT = BM * RB;
  constexpr int RPI = 1024 / RB;                 // rows per 1 KB DMA instruction
  constexpr int PPT = BM / RPI;                  // DMA instructions per tile
  constexpr int KK = C / 16, TN = K / 32;
  static_assert(WBYTES + NS * SLOT <= 160 * 1024, "lds");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const wl = smem;
  unsigned char* const ring = smem + WBYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;

  // contiguous run of tiles per workgroup
  const int per = p.n_tiles / gridDim.x, extra = p.n_tiles - per * gridDim.x;
  const int bid = blockIdx.x;
  const int t_begin = bid * per + (bid < extra ? bid : extra);
  const int t_count = per + (bid < extra ? 1 : 0);

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);

  // ---- filter: every wave loads a share (K rows x RB bytes, lane-linear 1 KB pieces, swizzle on the source) ----
  {
    constexpr int NW = NC + 1;
    const int r_in = lane / CPR, ch = lane % CPR;      // row within the piece, chunk position
    for (int pc = wave; pc < WBYTES / 1024; pc += NW) {
      const int row = pc * RPI + r_in;
      const unsigned off = (unsigned)row * RB + (unsigned)((ch ^ (swz<RB>(row) % CPR)) << 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(wl + pc * 1024), 16, (int)off, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (wave == NC) {
    // ================= loader =================
    const int r_in = lane / CPR, ch = lane % CPR;
    auto issue = [&](int i) {      // tile t_begin + i -> slot i % NS
      const int tile = t_begin + i;
      unsigned char* dst = ring + (i % NS) * SLOT;
#pragma unroll
      for (int pc = 0; pc < PPT; ++pc) {
        const int row = pc * RPI + r_in;                 // row within the tile
        const long long m = (long long)tile * BM + row;
        const unsigned off = (i < t_count && m < p.M) ? (unsigned)(m * RB) + (unsigned)((ch ^ (swz<RB>(row) % CPR)) << 4) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(dst + pc * 1024), 16, (int)off, 0, 0, 0);
      }
    };
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) issue(i);
    wait_vm<PPT*(NS - 2)>();                          // tile 0 has landed
    __builtin_amdgcn_s_barrier();                        // B_0
#pragma unroll 1
    for (int i = 0; i < t_count; ++i) {
      issue(i + NS - 1);                                 // into the slot the consumers left before B_i (zero-fill past the end)
      wait_vm<PPT*(NS - 2)>();                        // tile i + 1 has landed
      __builtin_amdgcn_s_barrier();                      // B_{i+1}
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ================= consumers =================
  unsigned fxo[KK], fwo[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int ch = kk * 2 + lhi;
    fxo[kk] = (unsigned)((wave * 32 + l31) * RB + ((ch ^ (swz<RB>(l31) % CPR)) << 4));
    fwo[kk] = (unsigned)(l31 * RB + ((ch ^ (swz<RB>(l31) % CPR)) << 4));
  }
  bf16_t* const y = reinterpret_cast<bf16_t*>(p.y);
  const bf16_t* const ad = reinterpret_cast<const bf16_t*>(p.addend);
  __builtin_amdgcn_s_barrier();                          // B_0
#pragma unroll 1
  for (int i = 0; i < t_count; ++i) {
    const unsigned char* xs = ring + (i % NS) * SLOT;
    const long long m = (long long)(t_begin + i) * BM + wave * 32 + l31;
    const bool ok = m < p.M;
    // this lane stores, for channel-group pair (2q, 2q + 1) of n-tile a, 8 channels at column a * 32 + 16 q + 8 lhi
    const size_t rowoff = (size_t)(ok ? m : 0) * K + 8 * lhi;
    u32x4 av[ADDEND ? TN * 2 : 1];
    if constexpr (ADDEND) {
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) av[a * 2 + q] = *reinterpret_cast<const u32x4*>(ad + rowoff + a * 32 + q * 16);
    }
    f32x16 acc[TN];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const bf16x8 fx = *reinterpret_cast<const bf16x8*>(xs + fxo[kk]);
#pragma unroll
      for (int a = 0; a < TN; ++a) {
        const bf16x8 fw = *reinterpret_cast<const bf16x8*>(wl + fwo[kk] + a * 32 * RB);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fx, acc[a], 0, 0, 0);
      }
    }
    // every fragment read of this slot is in registers once the MFMAs above have their operands; the barrier below is what lets
    // the loader overwrite it
    // ---- stores straight from the accumulators ----
    // acc[a][4 g + j]: channel a * 32 + 8 g + 4 lhi + j of pixel l31.  Pair (g = 2q, 2q + 1): after the half swap the lower lanes
    // hold channels 8 (2q) .. + 7, the upper lanes channels 8 (2q + 1) .. + 7 of their pixel.
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        unsigned lo0 = pack2bf(acc[a][8 * q + 0], acc[a][8 * q + 1]), lo1 = pack2bf(acc[a][8 * q + 2], acc[a][8 * q + 3]);
        unsigned hi0 = pack2bf(acc[a][8 * q + 4], acc[a][8 * q + 5]), hi1 = pack2bf(acc[a][8 * q + 6], acc[a][8 * q + 7]);
        // group 2q = (lo0, lo1): this lane's channels 4 lhi .. + 3 of 8 (2q) ..; group 2q + 1 = (hi0, hi1)
        auto r0 = __builtin_amdgcn_permlane32_swap(lo0, hi0, false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(lo1, hi1, false, false);
        // lower lanes: [own group 2q | upper lane's group 2q]; upper lanes: [lower lane's group 2q + 1 | own group 2q + 1]
        u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
        if constexpr (ADDEND) {
          const u32x4 ad4 = av[a * 2 + q];
          v.x = pack2bf(bflo(v.x) + bflo(ad4.x), bfhi(v.x) + bfhi(ad4.x));
          v.y = pack2bf(bflo(v.y) + bflo(ad4.y), bfhi(v.y) + bfhi(ad4.y));
          v.z = pack2bf(bflo(v.z) + bflo(ad4.z), bfhi(v.z) + bfhi(ad4.z));
          v.w = pack2bf(bflo(v.w) + bflo(ad4.w), bfhi(v.w) + bfhi(ad4.w));
        }
        if (ok) *reinterpret_cast<u32x4*>(y + rowoff + a * 32 + q * 16) = v;
      }
    __builtin_amdgcn_s_barrier();                        // B_{i+1}
  }
}

static float bf2f_h(bf16_t b) {
  unsigned u = (unsigned)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static bf16_t f2bf_h(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

template <int C, int K, int BM, int NS, bool ADDEND>
void run(int M, int wg_per_cu, int iters) {
  std::vector<bf16_t> hx((size_t)M * C), hw((size_t)K * C), ha((size_t)M * K);
  unsigned s = 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
  };
  for (auto& v : hx) v = f2bf_h(rnd());
  for (auto& v : hw) v = f2bf_h(rnd() * 0.125f);
  for (auto& v : ha) v = f2bf_h(rnd());
  void *dx, *dw, *dy, *da;
  CK(hipMalloc(&dx, hx.size() * 2));
  CK(hipMalloc(&dw, hw.size() * 2));
  CK(hipMalloc(&dy, (size_t)M * K * 2));
  CK(hipMalloc(&da, (size_t)M * K * 2));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dy, 0xff, (size_t)M * K * 2));
  Args a;
  a.x = dx; a.w = dw; a.y = dy; a.addend = ADDEND ? da : nullptr;
  a.x_bytes = (unsigned)(hx.size() * 2); a.w_bytes = (unsigned)(hw.size() * 2);
  a.M = M; a.n_tiles = (M + BM - 1) / BM;
  constexpr int LDS = K * C * 2 + NS * BM * C * 2;
  auto kern = stream1x1_kernel<C, K, BM, NS, ADDEND>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  int grid = cus * wg_per_cu;
  if (grid > a.n_tiles) grid = a.n_tiles;
  const int nthr = 64 * (BM / 32 + 1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(nthr), LDS, 0, a);
  CK(hipDeviceSynchronize());
  // check a sample of rows against fp64
  std::vector<bf16_t> hy((size_t)M * K);
  CK(hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost));
  double worst = 0;
  int bad = 0;
  for (int t = 0; t < 4096; ++t) {
    const int m = (int)(((long long)t * 2654435761u) % M);
    for (int n = 0; n < K; ++n) {
      double r = 0;
      for (int c = 0; c < C; ++c) r += (double)bf2f_h(hx[(size_t)m * C + c]) * (double)bf2f_h(hw[(size_t)n * C + c]);
      double ref = (double)bf2f_h(f2bf_h((float)r));
      if (ADDEND) ref = (double)bf2f_h(f2bf_h((float)(ref + (double)bf2f_h(ha[(size_t)m * K + n]))));
      const double got = bf2f_h(hy[(size_t)m * K + n]);
      const double err = fabs(got - ref) / (fabs(ref) + 1.0);
      if (err > worst) worst = err;
      if (err > 2e-2) ++bad;
    }
  }
  // also the last rows (ragged tail) exactly
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(nthr), LDS, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  const double io = 2.0 * ((double)M * C + (double)M * K), all = io + (ADDEND ? 2.0 * M * K : 0.0);
  printf("C=%d K=%d M=%d BM=%d NS=%d addend=%d wg/cu=%d grid=%d lds=%d KB: %.1f us  %.2f TB/s in+out  %.2f TB/s all  worst rel err %.2e bad %d\n", C, K, M, BM,
         NS, (int)ADDEND, wg_per_cu, grid, LDS / 1024, ms * 1e3, io / ms / 1e9, all / ms / 1e9, worst, bad);
  CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dy)); CK(hipFree(da));
}

int main(int argc, char** argv) {
  const int iters = 20;
  // the three shapes of VERDICT round 5 item 2 (batch 256): 56x56x256 -> 64, 28x28x64 -> 256, 56x56x64 -> 256 input gradient + addend
  for (int wg = 1; wg <= 2; ++wg) {
    run<256, 64, 64, 3, false>(256 * 56 * 56, wg, iters);
    run<256, 64, 64, 2, false>(256 * 56 * 56, wg, iters);
    run<64, 256, 128, 3, false>(256 * 28 * 28, wg, iters);
    run<64, 256, 128, 4, false>(256 * 28 * 28, wg, iters);
    run<64, 256, 128, 3, true>(256 * 56 * 56, wg, iters);
    run<64, 256, 128, 4, true>(256 * 56 * 56, wg, iters);
  }
  return 0;
}
