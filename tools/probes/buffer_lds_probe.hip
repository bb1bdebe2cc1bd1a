// Probe: does `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer resource) zero-fill the LDS destination of
// lanes whose voffset is out of range?  (decides whether the conv kernels can mask padded taps with an OOB offset)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const void* x, unsigned bytes, const int* offs, int so, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* w = reinterpret_cast<unsigned*>(smem);
  for (int i = 0; i < 4; ++i) w[threadIdx.x * 4 + i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, bytes, 0x00020000);
  int vo = offs[threadIdx.x];
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(smem + (threadIdx.x >> 6) * 1024), 16, vo, so, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = w[threadIdx.x * 4 + i];
}
int main() {
  const int NT = 256, NW = 4096;
  std::vector<unsigned> hx(NW);
  for (int i = 0; i < NW; ++i) hx[i] = 0x1000000u + i;
  std::vector<int> ho(NT);
  for (int t = 0; t < NT; ++t) ho[t] = (t % 3 == 0) ? (int)0x80000000u : ((t * 7) % 200) * 16;
  unsigned *dx, *dout; int* doffs;
  hipMalloc(&dx, NW * 4); hipMalloc(&dout, NT * 16); hipMalloc(&doffs, NT * 4);
  hipMemcpy(dx, hx.data(), NW * 4, hipMemcpyHostToDevice);
  hipMemcpy(doffs, ho.data(), NT * 4, hipMemcpyHostToDevice);
  const int so = 256;
  hipLaunchKernelGGL(k, dim3(1), dim3(NT), NT * 16, 0, dx, NW * 4, doffs, so, dout);
  std::vector<unsigned> out(NT * 4);
  hipMemcpy(out.data(), dout, NT * 16, hipMemcpyDeviceToHost);
  int bad = 0, zero = 0, untouched = 0;
  for (int t = 0; t < NT; ++t)
    for (int i = 0; i < 4; ++i) {
      unsigned v = out[t * 4 + i];
      if (t % 3 == 0) { zero += v == 0; untouched += v == 0xdeadbeefu; if (v != 0) ++bad; }
      else if (v != 0x1000000u + (ho[t] + so) / 4 + i) ++bad;
    }
  printf("buffer_load_lds probe: bad=%d  oob lanes: zero-filled words=%d untouched words=%d\n", bad, zero, untouched);
  return bad != 0;
}
