#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* g, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[2048];
  // two wave-level DMA loads into different LDS regions; lane l's 16 bytes land at base + l*16
  __builtin_amdgcn_global_load_lds(g + threadIdx.x * 4, lds + 256 + (threadIdx.x >> 6) * 256, 16, 0, 0);
  __builtin_amdgcn_global_load_lds(g + 1024 + threadIdx.x * 4, lds + 1280 + (threadIdx.x >> 6) * 256, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  volatile float* v = lds;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = v[i];
}
int main() {
  float *g, *o; hipMalloc(&g, 4096 * 4); hipMalloc(&o, 2048 * 4);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)i; hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemset(o, 0, 2048 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, g, o);
  float r[2048]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 512; ++i) { if (r[256 + i] != (float)i) ++bad; if (r[1280 + i] != (float)(1024 + i)) ++bad; }
  printf("bad %d  r[256..259] %g %g %g %g  r[1280] %g r[0] %g %s\n", bad, r[256], r[257], r[258], r[259], r[1280], r[0], hipGetErrorString(hipGetLastError()));
  return 0;
}
