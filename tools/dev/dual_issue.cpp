// Dev microbenchmark: can a gfx950 SIMD retire fp32 VALU FMAs (v_pk_fma_f32) in the shadow of v_mfma_f32_32x32x2_f32?
//   hipcc --offload-arch=gfx950 -O3 -o tools/dev/dual_issue tools/dev/dual_issue.cpp && tools/dev/dual_issue
// R packed FMAs are interleaved with every MFMA (an MFMA occupies the matrix pipe for 64 cycles; a v_pk_fma_f32 the VALU for 4+).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int R, bool MFMA>
__global__ __launch_bounds__(256) void k(long iters, float* sink) {
  f32x16 acc[4] = {};
  f32x2 v[16];
  for (int i = 0; i < 16; ++i) v[i] = f32x2{(float)threadIdx.x * 1e-3f + i, 1.0f};
  const float a = threadIdx.x * 1e-6f, b = 1.0f + threadIdx.x * 1e-7f;
  const f32x2 m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
  for (long it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MFMA) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int q = (j * R + r) & 15;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v[q]) : "v"(v[q]), "v"(m), "v"(c));
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
  if (s == 123.456f) sink[0] = s;
}

template <int R, bool MFMA>
void run(int waves_per_simd, float* sink) {
  const long iters = 20000;
  const int blocks = 256 * waves_per_simd;           // 4 waves per block -> one wave per SIMD per block
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<R, MFMA>), dim3(blocks), dim3(256), 0, 0, 1000, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<R, MFMA>), dim3(blocks), dim3(256), 0, 0, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * 4, mf = MFMA ? waves * iters * 4 * 4096.0 : 0, vf = waves * iters * 4 * R * 256.0;
  printf("R=%2d mfma=%d waves/SIMD=%d  %8.3f ms  MFMA %6.1f TF/s  VALU %6.1f TF/s  sum %6.1f\n", R, (int)MFMA, waves_per_simd, ms,
         mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
}

int main() {
  float* sink; hipMalloc(&sink, 16);
  for (int w : {1, 2, 4}) {
    run<0, true>(w, sink);
    run<16, false>(w, sink);
    run<2, true>(w, sink);
    run<4, true>(w, sink);
    run<8, true>(w, sink);
    run<12, true>(w, sink);
    run<16, true>(w, sink);
  }
  return 0;
}
