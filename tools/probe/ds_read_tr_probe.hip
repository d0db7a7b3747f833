// Probe of ds_read_b64_tr_b16: lane 4q+p of each 16-lane group supplies the address of row q, columns 4p..4p+3 of a
// 4 x 16 block of 16-bit elements; lane i then receives column i of the 4 rows (row q in element q).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* o) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i;      // element (row, col) of a [64][64] tile = row*64 + col
  __syncthreads();
  int lane = threadIdx.x, grp = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  auto ptr = (__attribute__((address_space(3))) s16x4*)(lds + (4 * grp + q) * 64 + 4 * p);   // rows 4grp..4grp+3, cols 0..15
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
  for (int e = 0; e < 4; ++e) o[lane * 4 + e] = v[e];
}
int main() {
  short* o; hipMalloc(&o, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
  short h[256]; hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
    int grp = lane >> 4, i = lane & 15, expect = (4 * grp + e) * 64 + i;    // row 4grp+e, column i
    if (h[lane * 4 + e] != expect) ++bad;
  }
  printf("lane 0: %d %d %d %d | lane 5: %d %d %d %d | lane 21: %d %d %d %d | mismatches %d\n", h[0], h[1], h[2], h[3], h[20], h[21], h[22], h[23],
         h[84], h[85], h[86], h[87], bad);
  return 0;
}
