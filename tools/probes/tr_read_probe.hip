#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((address_space(3))) s4_t lds_s4_t;
__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  int idx;   // element index supplied by this lane
  if (mode == 0) idx = l * 4;                                   // linear: lane l -> 8-byte piece l
  else idx = ((l >> 4) * 4 + ((l & 15) >> 2)) * 64 + (l & 3) * 4; // rows of 64 elements (128 B): row = 4g + (i>>2), col chunk (i&3)*4
  s4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(sm + idx));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
