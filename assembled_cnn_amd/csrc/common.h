// Shared device/host helpers for libasm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <tuple>
#include <utility>
#include "../../include/asm_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- error plumbing -------------------------------------------------------------------------
void asm_set_error(const char* fmt, ...);
// launches of this thread that have not reached their ASM_CHECK_LAUNCH yet (see below); any error return resets it, so a
// path that leaves between ASM_LAUNCH and the check cannot leave later argument checks blind to stale sticky errors
inline thread_local int asm_unchecked_launches = 0;
#define ASM_FAIL(code, ...)       \
  do {                            \
    asm_set_error(__VA_ARGS__);   \
    asm_unchecked_launches = 0;   \
    return (code);                \
  } while (0)
// hipGetLastError() is STICKY per host thread across ALL HIP users of the process: an error some other component left
// behind (PyTorch probing devices / peers, a collective library, ...) would be reported by ASM_CHECK_LAUNCH as the
// failure of the next kernel this library launches ("cast: no ROCm-capable device is detected" on a healthy GPU).  The
// argument checks at the top of an entry point are where that stale state is dropped -- but ONLY while this thread has no
// launch of its own waiting for its ASM_CHECK_LAUNCH (asm_unchecked_launches, counted by ASM_LAUNCH): an ASM_REQUIRE that
// a later edit places after a launch can therefore never swallow that launch's error.
// which convolution kernel family the last forward / input-gradient call of this thread launched (tests assert the plan a
// shape gets: asm_debug_last_conv_kernel, include/asm_hip_debug.h): 0 igemm_kernel, 1 igemm1 (conv_gemm1), 2 igemm2, 3 igemm3,
// 4 conv_halo, 5 dgrad_s2, 8 igemm8
inline thread_local int asm_last_conv_kernel = -1;
void asm_count_launch();             // plan.hip: process-wide kernel-launch counter (asm_launch_count)
// tape.hip: while THIS host thread records a launch tape (asm_tape_begin .. asm_tape_end) every launch is also written
// down -- kernel, geometry, stream and a copy of its arguments converted to the kernel's parameter types -- so that
// asm_tape_replay can issue the same launches again without the host code that derived them.
inline thread_local bool asm_tape_on = false;
void asm_tape_add_launch(const void* fn, dim3 grid, dim3 block, unsigned shmem, hipStream_t stream, void** args,
                         const size_t* sizes, const size_t* aligns, int nargs);
// device-to-device copy (from != nullptr) or byte fill, seen by a tape that is being recorded
hipError_t asm_fill_async(void* dst, const void* from, int value, size_t bytes, hipStream_t stream);
template <typename... P, typename... A>
inline void asm_launch(void (*kernel)(P...), dim3 grid, dim3 block, unsigned shmem, hipStream_t stream, A&&... a) {
  static_assert(sizeof...(P) == sizeof...(A), "ASM_LAUNCH: argument count differs from the kernel's parameter count");
  if (asm_tape_on) {
    auto params = std::tuple<P...>{static_cast<P>(a)...};      // exactly what the launch below converts them to
    std::apply(
        [&](auto&... p) {
          void* ptrs[] = {(void*)&p..., nullptr};
          const size_t sizes[] = {sizeof(p)..., 0}, aligns[] = {alignof(decltype(p))..., 0};
          asm_tape_add_launch((const void*)kernel, grid, block, shmem, stream, ptrs, sizes, aligns, (int)sizeof...(P));
        },
        params);
  }
  hipLaunchKernelGGL(kernel, grid, block, shmem, stream, static_cast<P>(a)...);
}
#define ASM_LAUNCH(...)             \
  do {                              \
    ++asm_unchecked_launches;       \
    asm_count_launch();             \
    asm_launch(__VA_ARGS__);        \
  } while (0)
#define ASM_REQUIRE(cond, ...)                               \
  do {                                                       \
    if (asm_unchecked_launches == 0) (void)hipGetLastError(); \
    if (!(cond)) ASM_FAIL(ASM_EINVAL, __VA_ARGS__);          \
  } while (0)
#define ASM_CHECK_LAUNCH(name)                                                       \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    asm_unchecked_launches = 0;                                                      \
    if (e__ != hipSuccess) ASM_FAIL(ASM_EHIP, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---- bf16 <-> f32 ----------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((unsigned)b) << 16); }
// round-to-nearest-even; lowers to v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}
__device__ __forceinline__ float bflo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// 8 bf16 (one 16-byte vector) <-> 8 floats
__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
  f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
  f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

// ---- wave / block reductions -------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- invariant integer division (n < 2^31) --------------------------------------------------------
struct FastDiv {
  unsigned mul, shr, d;
};
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) {
    f.mul = 0;
    f.shr = 0;
    return f;
  }
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  f.shr = s;
  f.mul = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
  return f;
}
__device__ __forceinline__ unsigned fd_div(unsigned n, const FastDiv& f) {
  return (__umulhi(n, f.mul) + n) >> f.shr;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// Tuning / test knobs: an explicit, caller-set struct (asm_tuning, include/asm_hip.h; asm_set_tuning / asm_get_tuning in
// plan.hip).  The library reads NO environment variable: the host (assembled_cnn_amd/ops.py) maps its ASM_* variables onto
// the struct, so a C caller sees exactly the heuristics the header documents.  Read on every call (never cached).
extern "C" const asm_tuning* asm_tuning_current(void);
static inline const asm_tuning& asm_tune() { return *asm_tuning_current(); }
// > 64 KiB of dynamic LDS needs an explicit per-function opt-in, and the attribute is per DEVICE: `done` is the
// caller's static per-device flag table (idempotent; a racing second call just repeats the same setting).
#define ASM_MAX_DEVICES 16
template <class Kern>
static inline hipError_t asm_ensure_dyn_lds(Kern kern, int lds_bytes, bool (&done)[ASM_MAX_DEVICES]) {
  if (lds_bytes <= 64 * 1024) return hipSuccess;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const int slot = (dev >= 0 && dev < ASM_MAX_DEVICES) ? dev : -1;
  if (slot >= 0 && done[slot]) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e == hipSuccess && slot >= 0) done[slot] = true;
  return e;
}
// compute units of the current device (cached per device; 256 on MI355X)
static inline int asm_num_cus() {
  static int cached[ASM_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ASM_MAX_DEVICES) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cached[dev];
}
static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

// buffer resource for raw (stride 0) access: out-of-range offsets load 0
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
#define ASM_OOB 0x80000000u
