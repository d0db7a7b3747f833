// Probe: does buffer_load_dwordx4 ... lds (a) land at M0 + lane*16, (b) zero-fill lanes whose offset fails the range
// check, (c) honour LDS destinations above 64 KiB?   build: hipcc --offload-arch=gfx950 -O2 bufload_lds_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
extern __shared__ __attribute__((aligned(16))) char smem[];
__global__ void k(const unsigned* g, unsigned* o, int nbytes, int soff, int lds_off) {
  int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 256; i += 64) ((unsigned*)(smem + lds_off))[i] = 0xdeadbeef;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  unsigned voff = (lane & 1) ? 0xFFFFFFFFu : lane * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + lds_off), 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int i = threadIdx.x; i < 256; i += 64) o[i] = ((unsigned*)(smem + lds_off))[i];
}
int main() {
  std::vector<unsigned> h(4096); for (int i = 0; i < 4096; ++i) h[i] = i;
  unsigned *g, *o; hipMalloc(&g, 16384); hipMalloc(&o, 1024); hipMemcpy(g, h.data(), 16384, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int offs[3] = {0, 1024, 100 * 1024};
  for (int t = 0; t < 3; ++t) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, g, o, 16384, 256, offs[t]);
    std::vector<unsigned> r(256); hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    printf("lds_off %d:", offs[t]); for (int i = 0; i < 24; ++i) printf(" %x", r[i]); printf("\n");
  }
  return 0;
}
