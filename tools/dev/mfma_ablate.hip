// Dev experiment (standalone): what costs the conv kernel its last 20 % of MFMA issue rate?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/abl tools/dev/mfma_ablate.hip && /tmp/abl
// Variants of a 4-wave workgroup loop with the conv kernel's 128x128 / KB=16 structure:
//   0 pure MFMA   1 + ds_read_b128 operands   2 + barrier per stage   3 + ds_write_b128 per stage   4 + global loads
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#ifndef WMAP
#define WMAP 0
#endif
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int KB = 16, RS = KB + 4, BM = 128, BN = 128;

template <int V>
__global__ __launch_bounds__(256) void k(long stages, const float* __restrict__ g, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  for (int i = tid; i < 2 * (BM + BN) * RS; i += 256) lds[i] = 1.0f + i * 1e-7f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int frag_off = (lane & 31) * RS + 4 * (lane >> 5);
  float4 ra[2], rb[2];
  const float* gp = g + ((long)blockIdx.x * 256 + tid) * 4;
  float4 af0[2], bf0[2];
  af0[0] = af0[1] = bf0[0] = bf0[1] = make_float4(1.f, 1.0001f, 0.9999f, 1.f);
  float4 bnext[2][2];
  if (V == 6) {
    for (int h = 0; h < 2; ++h) for (int ni = 0; ni < 2; ++ni)
      bnext[h][ni] = *reinterpret_cast<const float4*>(g + ((long)blockIdx.x * 4096 + (wave & 1) * 1024 + (h * 2 + ni) * 256 + lane * 4) % (1 << 24));
  }
  for (long s = 0; s < stages; ++s) {
    const int buf = s & 1;
    float4 bcur[2][2];
    if (V == 6) {
      for (int h = 0; h < 2; ++h) for (int ni = 0; ni < 2; ++ni) bcur[h][ni] = bnext[h][ni];
      for (int h = 0; h < 2; ++h) for (int ni = 0; ni < 2; ++ni)      // next stage's B fragments: 4 contiguous 1 KB wave loads
        bnext[h][ni] = *reinterpret_cast<const float4*>(g + (((s + 1) * 65536 + (long)blockIdx.x * 4096 + (wave & 1) * 1024 + (h * 2 + ni) * 256 + lane * 4) % (1 << 24)));
      for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const float4*>(gp + ((s * 4 + i) & 1023) * 262144L % (1 << 24));
    }
    if (V == 4) {
      for (int i = 0; i < 2; ++i) {
        ra[i] = *reinterpret_cast<const float4*>(gp + ((s * 4 + i) & 1023) * 262144L % (1 << 24));
        rb[i] = *reinterpret_cast<const float4*>(gp + ((s * 4 + 2 + i) & 1023) * 262144L % (1 << 24));
      }
    } else {
      ra[0] = ra[1] = rb[0] = rb[1] = af0[0];
    }
    const float* a = &lds[buf * (BM + BN) * RS + wm0 * RS + frag_off];
    const float* b = &lds[buf * (BM + BN) * RS + (BM + wn0) * RS + frag_off];
    if (V == 5) {   // direct global -> LDS (no VGPR staging, no ds_write): rows unpadded (KB floats), chunk slot XOR-swizzled
      float* dst = &lds[(buf ^ 1) * (BM + BN) * KB];
      for (int i = 0; i < 2; ++i) {
        const float* src = gp + ((s * 4 + i) & 1023) * 262144L % (1 << 24);
        __builtin_amdgcn_global_load_lds(src, dst + (i * 256 + wave * 64) * 4, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src + 64, dst + BM * KB + (i * 256 + wave * 64) * 4, 16, 0, 0);
      }
    }
#pragma unroll
    for (int h = 0; h < KB / 8; ++h) {
      float4 af[2], bf[2];
      if (V == 5) {
        const int r = lane & 31, q = 2 * h + (lane >> 5);
        const float* base = &lds[buf * (BM + BN) * KB];
        for (int mi = 0; mi < 2; ++mi) {
          const int row = wm0 + mi * 32 + r;
          af[mi] = *reinterpret_cast<const float4*>(base + (row * 4 + (q ^ ((row >> 2) & 3))) * 4);
        }
        for (int ni = 0; ni < 2; ++ni) {
          const int row = wn0 + ni * 32 + r;
          bf[ni] = *reinterpret_cast<const float4*>(base + BM * KB + (row * 4 + (q ^ ((row >> 2) & 3))) * 4);
        }
      } else if (V == 6) {
        for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const float4*>(a + mi * 32 * RS + h * 8);
        for (int ni = 0; ni < 2; ++ni) bf[ni] = bcur[h][ni];
      } else if (V >= 1) {
        for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const float4*>(a + mi * 32 * RS + h * 8);
        for (int ni = 0; ni < 2; ++ni) bf[ni] = *reinterpret_cast<const float4*>(b + ni * 32 * RS + h * 8);
      } else {
        af[0] = af0[0]; af[1] = af0[1]; bf[0] = bf0[0]; bf[1] = bf0[1];
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].x, bf[ni].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].y, bf[ni].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].z, bf[ni].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].w, bf[ni].w, acc[mi][ni], 0, 0, 0);
        }
    }
    if (V == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (V == 6) {
      float* wa = &lds[(buf ^ 1) * (BM + BN) * RS];
      for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 2) + i * 64;
        *reinterpret_cast<float4*>(wa + row * RS + (tid & 3) * 4) = ra[i];
      }
    }
    if (V >= 3 && V < 5) {
      float* wa = &lds[(buf ^ 1) * (BM + BN) * RS];
      for (int i = 0; i < 2; ++i) {
#if WMAP == 1
        // lanes of a 16-lane group hit rows g, g+4, g+8, g+12: (5*row + chunk) mod 16 all distinct
        const int row = (wave * 16 + ((lane >> 2) & 3) * 4 + (lane >> 4)) + i * 64;
#elif WMAP == 2
        // 8-lane groups -> rows g, g+8
        const int row = (wave * 16 + ((lane >> 2) & 1) * 8 + (lane >> 3)) + i * 64;
#else
        const int row = (tid >> 2) + i * 64;
#endif
        *reinterpret_cast<float4*>(wa + row * RS + (tid & 3) * 4) = ra[i];
        *reinterpret_cast<float4*>(wa + (BM + row) * RS + (tid & 3) * 4) = rb[i];
      }
    }
    if (V >= 2) __syncthreads();
  }
  float t = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) t += acc[a][b][r];
  if (t == 123.456f) sink[0] = t;
}

template <int V>
void run(int blocks, long stages, const float* g, float* sink, int lds_bytes) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), lds_bytes, 0, stages / 8, g, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), lds_bytes, 0, stages, g, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double fl = (double)blocks * 4 * stages * 32 * 4096.0;
  printf("variant %d  blocks %4d  lds %6d B: %7.1f TF/s  (%.2f ms)  %s\n", V, blocks, lds_bytes, fl / (ms * 1e-3) / 1e12, ms,
         hipGetErrorString(hipGetLastError()));
}


template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void k2(long stages, const float* __restrict__ g, float* sink) {
  constexpr int BM2 = 64 * WM, BN2 = 64 * WN, NT = 64 * WM * WN;
  constexpr int APT = BM2 * 4 / NT, BPT = BN2 * 4 / NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 64, wn0 = (wave % WN) * 64;
  for (int i = tid; i < 2 * (BM2 + BN2) * RS; i += NT) lds[i] = 1.0f + i * 1e-7f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int frag_off = (lane & 31) * RS + 4 * (lane >> 5);
  float4 ra[APT], rb[BPT];
  const float* gp = g + ((long)blockIdx.x * NT + tid) * 4;
  for (long s = 0; s < stages; ++s) {
    const int buf = s & 1;
    for (int i = 0; i < APT; ++i) ra[i] = *reinterpret_cast<const float4*>(gp + ((s * 8 + i) & 1023) * 262144L % (1 << 24));
    for (int i = 0; i < BPT; ++i) rb[i] = *reinterpret_cast<const float4*>(gp + ((s * 8 + 4 + i) & 1023) * 262144L % (1 << 24));
    const float* a = &lds[buf * (BM2 + BN2) * RS + wm0 * RS + frag_off];
    const float* b = &lds[buf * (BM2 + BN2) * RS + (BM2 + wn0) * RS + frag_off];
#pragma unroll
    for (int h = 0; h < KB / 8; ++h) {
      float4 af[2], bf[2];
      for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const float4*>(a + mi * 32 * RS + h * 8);
      for (int ni = 0; ni < 2; ++ni) bf[ni] = *reinterpret_cast<const float4*>(b + ni * 32 * RS + h * 8);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].x, bf[ni].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].y, bf[ni].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].z, bf[ni].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].w, bf[ni].w, acc[mi][ni], 0, 0, 0);
        }
    }
    float* wa = &lds[(buf ^ 1) * (BM2 + BN2) * RS];
    for (int i = 0; i < APT; ++i) *reinterpret_cast<float4*>(wa + ((tid >> 2) + i * (NT / 4)) * RS + (tid & 3) * 4) = ra[i];
    for (int i = 0; i < BPT; ++i) *reinterpret_cast<float4*>(wa + (BM2 + (tid >> 2) + i * (NT / 4)) * RS + (tid & 3) * 4) = rb[i];
    __syncthreads();
  }
  float t = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) t += acc[a][b][r];
  if (t == 123.456f) sink[0] = t;
}

template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void k4(long stages, const float* __restrict__ g, float* sink) {
  constexpr int BM2 = 64 * WM, BN2 = 64 * WN, NT = 64 * WM * WN;
  constexpr int APT = BM2 * 4 / NT, BPT = BN2 * 4 / NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 64, wn0 = (wave % WN) * 64;
  for (int i = tid; i < 3 * (BM2 + BN2) * RS; i += NT) lds[i] = 1.0f + i * 1e-7f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int frag_off = (lane & 31) * RS + 4 * (lane >> 5);
  float4 ra[APT], rb[BPT];
  const float* gp = g + ((long)blockIdx.x * NT + tid) * 4;
  for (long s = 0; s < stages; ++s) {
    const int buf = s % 3, wbuf = (s + 2) % 3;
    for (int i = 0; i < APT; ++i) ra[i] = *reinterpret_cast<const float4*>(gp + ((s * 8 + i) & 1023) * 262144L % (1 << 24));
    for (int i = 0; i < BPT; ++i) rb[i] = *reinterpret_cast<const float4*>(gp + ((s * 8 + 4 + i) & 1023) * 262144L % (1 << 24));
    const float* a = &lds[buf * (BM2 + BN2) * RS + wm0 * RS + frag_off];
    const float* b = &lds[buf * (BM2 + BN2) * RS + (BM2 + wn0) * RS + frag_off];
#pragma unroll
    for (int h = 0; h < KB / 8; ++h) {
      if (h == 1) __syncthreads();       // everyone has finished stage s-1 (its buffer is this stage's write target)
      float4 af[2], bf[2];
      for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const float4*>(a + mi * 32 * RS + h * 8);
      for (int ni = 0; ni < 2; ++ni) bf[ni] = *reinterpret_cast<const float4*>(b + ni * 32 * RS + h * 8);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].x, bf[ni].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].y, bf[ni].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].z, bf[ni].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].w, bf[ni].w, acc[mi][ni], 0, 0, 0);
        }
    }
    float* wa = &lds[wbuf * (BM2 + BN2) * RS];
    for (int i = 0; i < APT; ++i) *reinterpret_cast<float4*>(wa + ((tid >> 2) + i * (NT / 4)) * RS + (tid & 3) * 4) = ra[i];
    for (int i = 0; i < BPT; ++i) *reinterpret_cast<float4*>(wa + (BM2 + (tid >> 2) + i * (NT / 4)) * RS + (tid & 3) * 4) = rb[i];
  }
  float t = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) t += acc[a][b][r];
  if (t == 123.456f) sink[0] = t;
}

template <int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(64 * WM * WN) void k3(long stages, const float* __restrict__ g, float* sink) {
  constexpr int BM2 = 32 * MI * WM, BN2 = 32 * NI * WN, NT = 64 * WM * WN;
  constexpr int APT = BM2 * 4 / NT, BPT = BN2 * 4 / NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * 32 * MI, wn0 = (wave % WN) * 32 * NI;
  for (int i = tid; i < 2 * (BM2 + BN2) * RS; i += NT) lds[i] = 1.0f + i * 1e-7f;
  __syncthreads();
  f32x16 acc[MI][NI];
  for (int a = 0; a < MI; ++a) for (int b = 0; b < NI; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int frag_off = (lane & 31) * RS + 4 * (lane >> 5);
  float4 ra[APT], rb[BPT];
  const float* gp = g + ((long)blockIdx.x * NT + tid) * 4;
  for (long s = 0; s < stages; ++s) {
    const int buf = s & 1;
    for (int i = 0; i < APT; ++i) ra[i] = *reinterpret_cast<const float4*>(gp + ((s * 8 + i) & 1023) * 262144L % (1 << 24));
    for (int i = 0; i < BPT; ++i) rb[i] = *reinterpret_cast<const float4*>(gp + ((s * 8 + 4 + i) & 1023) * 262144L % (1 << 24));
    const float* a = &lds[buf * (BM2 + BN2) * RS + wm0 * RS + frag_off];
    const float* b = &lds[buf * (BM2 + BN2) * RS + (BM2 + wn0) * RS + frag_off];
#pragma unroll
    for (int h = 0; h < KB / 8; ++h) {
      float4 af[MI], bf[NI];
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const float4*>(a + mi * 32 * RS + h * 8);
      for (int ni = 0; ni < NI; ++ni) bf[ni] = *reinterpret_cast<const float4*>(b + ni * 32 * RS + h * 8);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].x, bf[ni].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].y, bf[ni].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].z, bf[ni].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].w, bf[ni].w, acc[mi][ni], 0, 0, 0);
        }
    }
    float* wa = &lds[(buf ^ 1) * (BM2 + BN2) * RS];
    for (int i = 0; i < APT; ++i) *reinterpret_cast<float4*>(wa + ((tid >> 2) + i * (NT / 4)) * RS + (tid & 3) * 4) = ra[i];
    for (int i = 0; i < BPT; ++i) *reinterpret_cast<float4*>(wa + (BM2 + (tid >> 2) + i * (NT / 4)) * RS + (tid & 3) * 4) = rb[i];
    __syncthreads();
  }
  float t = 0.f;
  for (int a = 0; a < MI; ++a) for (int b = 0; b < NI; ++b) for (int r = 0; r < 16; ++r) t += acc[a][b][r];
  if (t == 123.456f) sink[0] = t;
}

template <int WM, int WN>
void run2(int blocks, long stages, const float* g, float* sink) {
  const int lds_bytes = 2 * 64 * (WM + WN) * RS * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k2<WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k2<WM, WN>), dim3(blocks), dim3(64 * WM * WN), lds_bytes, 0, stages / 8, g, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k2<WM, WN>), dim3(blocks), dim3(64 * WM * WN), lds_bytes, 0, stages, g, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double fl = (double)blocks * WM * WN * stages * 32 * 4096.0;
  printf("tile %3dx%3d (%d waves)  blocks %4d  lds %6d B: %7.1f TF/s  %s\n", 64 * WM, 64 * WN, WM * WN, blocks, lds_bytes,
         fl / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

template <int WM, int WN, int MI, int NI>
void run3(int blocks, long stages, const float* g, float* sink) {
  const int lds_bytes = 2 * 32 * (MI * WM + NI * WN) * RS * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k3<WM, WN, MI, NI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k3<WM, WN, MI, NI>), dim3(blocks), dim3(64 * WM * WN), lds_bytes, 0, stages / 8, g, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k3<WM, WN, MI, NI>), dim3(blocks), dim3(64 * WM * WN), lds_bytes, 0, stages, g, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double fl = (double)blocks * WM * WN * stages * (MI * NI * 8) * 4096.0;
  printf("wave tile %3dx%3d, block %3dx%3d (%d waves)  blocks %4d  lds %6d B: %7.1f TF/s  %s\n", 32 * MI, 32 * NI, 32 * MI * WM,
         32 * NI * WN, WM * WN, blocks, lds_bytes, fl / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

template <int WM, int WN>
void run4(int blocks, long stages, const float* g, float* sink) {
  const int lds_bytes = 3 * 64 * (WM + WN) * RS * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k4<WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k4<WM, WN>), dim3(blocks), dim3(64 * WM * WN), lds_bytes, 0, stages / 8, g, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k4<WM, WN>), dim3(blocks), dim3(64 * WM * WN), lds_bytes, 0, stages, g, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double fl = (double)blocks * WM * WN * stages * 32 * 4096.0;
  printf("3-buffer pipeline, tile %3dx%3d (%d waves)  blocks %4d  lds %6d B: %7.1f TF/s  %s\n", 64 * WM, 64 * WN, WM * WN, blocks,
         lds_bytes, fl / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

template <int V>
double spin(int blocks, long stages, const float* g, float* sink, int lds, double secs) {      // launches k<V> back to back for `secs`; TFLOP/s
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<V><<<blocks, 256, lds>>>(stages, g, sink); hipDeviceSynchronize();
  long n = 0; float ms = 0.f;
  hipEventRecord(a);
  do {
    for (int i = 0; i < 20; ++i) k<V><<<blocks, 256, lds>>>(stages, g, sink);
    n += 20;
    hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
  } while (ms < secs * 1e3);
  return (double)n * blocks * 4 * stages * 32 * 4096.0 / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  float *g, *sink;
  hipMalloc(&g, (1L << 25) * 4); hipMemset(g, 0, (1L << 25) * 4); hipMalloc(&sink, 64);
  const long stages = 4000;
  const int lds = 2 * (BM + BN) * RS * 4;           // 40960 B -> 3-4 workgroups per CU
  if (argc >= 4 && argv[1][0] == 'p') {             // power ladder (tools/dev/power_ladder.py): `power <variant> <seconds>`
    const int v = atoi(argv[2]); const double secs = atof(argv[3]);
    double tf = 0;
    switch (v) {
      case 0: tf = spin<0>(768, stages, g, sink, lds, secs); break;
      case 1: tf = spin<1>(768, stages, g, sink, lds, secs); break;
      case 2: tf = spin<2>(768, stages, g, sink, lds, secs); break;
      case 3: tf = spin<3>(768, stages, g, sink, lds, secs); break;
      case 4: tf = spin<4>(768, stages, g, sink, lds, secs); break;
      case 5: tf = spin<5>(768, stages, g, sink, lds, secs); break;
      default: tf = spin<6>(768, stages, g, sink, lds, secs); break;
    }
    printf("variant %d: %.1f TFLOP/s\n", v, tf);
    return 0;
  }
  printf("WMAP %d\n", WMAP);
  run4<2, 2>(512, stages, g, sink); run4<2, 2>(768, stages, g, sink); run4<4, 2>(256, stages, g, sink); run4<4, 2>(512, stages, g, sink);
  run2<2, 2>(512, stages, g, sink);
  run3<2, 2, 2, 4>(512, stages, g, sink); run3<2, 2, 2, 4>(768, stages, g, sink);
  run3<2, 2, 4, 2>(512, stages, g, sink); run3<2, 2, 4, 4>(256, stages, g, sink); run3<2, 2, 4, 4>(512, stages, g, sink);
  run3<2, 2, 2, 2>(768, stages, g, sink); run3<2, 2, 1, 2>(1024, stages, g, sink);
  run2<2, 2>(768, stages, g, sink); run2<2, 2>(1024, stages, g, sink);
  run2<4, 2>(256, stages, g, sink); run2<4, 2>(512, stages, g, sink);
  run2<2, 4>(512, stages, g, sink);
  run2<4, 4>(256, stages, g, sink);
  run2<2, 1>(1536, stages, g, sink); run2<1, 1>(3072, stages, g, sink);
  for (int blocks : {768}) {
    run<2>(blocks, stages, g, sink, lds); run<3>(blocks, stages, g, sink, lds); run<4>(blocks, stages, g, sink, lds);
    run<6>(blocks, stages, g, sink, lds);
    run<6>(1024, stages, g, sink, lds);
    run<4>(1024, stages, g, sink, lds);
    run<6>(768, stages, g, sink, 24 * 1024);
  }
  return 0;
}
