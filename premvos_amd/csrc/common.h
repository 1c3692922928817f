// Shared helpers for libpremvos_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/premvos_hip.h"

namespace premvos {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PREMVOS_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return PREMVOS_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Four registers with an arbitrary (but, for the compiler, defined) content and no instruction behind them: the destination of a
// load that only the in-range lanes execute.  `t = zero; if (ok) t = load;` would be the obvious spelling, but hipcc turns it into
// load + select, and a select behind a load makes the wave wait for that load before it issues the next one (measured on the
// depthwise kernels: 3.5 instead of 4.6-5.8 TB/s); out-of-range lanes are masked when the value is USED instead.  (Where even this
// costs -- the row-ahead depthwise tile kernel -- the register is left unwritten and provably never read for those lanes.)
__device__ inline float4 arbitrary4() {
  float4 t;
  asm("" : "=v"(t.x), "=v"(t.y), "=v"(t.z), "=v"(t.w));     // (not volatile: free to move, nothing to order)
  return t;
}

// A 16-byte load whose result is meant to live in a register ARRAY element: `arr[i] = *reinterpret_cast<const float4*>(p)` is an
// aggregate copy into the array, which hipcc leaves in scratch memory (measured: the staging registers of a GEMM loop went through
// scratch_store / scratch_load, 75 instead of 115 TFLOP/s); building the value from its components keeps the array in VGPRs.
__device__ inline float4 ld4(const float* p) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  return make_float4(t.x, t.y, t.z, t.w);
}

// The dispatcher places workgroup `id` of a launch on XCD id % 8 (eight XCDs, a private L2 each).  Returns the position of
// workgroup `id` in an order in which every XCD owns ONE contiguous run of the `nwg` work items: neighbours in the work raster
// (tiles sharing halos, the N tiles of one row of A) then share an L2.  A bijection on [0, nwg): pure speed, any order is correct.
__device__ inline int xcd_contiguous(int id, int nwg) {
  const int xcd = id & 7, local = id >> 3, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// The resident split layout "S8" (round 4, csrc/conv_bf16x3_s8.hip): every group of EIGHT channels of a pixel is the 32 bytes
// {hi(8 x bf16), lo(8 x bf16)} -- a 16-byte half is exactly one operand of v_mfma_f32_32x32x16_bf16, so the consumer stages it by
// LDS-DMA and reads fragments with one ds_read_b128, no re-pairing.  x = hi + lo, hi = bf16(x) (round to nearest even), lo = bf16(x - hi).
__device__ __forceinline__ uint2 bf16_hi4(const float4 r, float4& rest) {
  using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
  const bf16x4 hi = {(__bf16)r.x, (__bf16)r.y, (__bf16)r.z, (__bf16)r.w};
  rest = make_float4(r.x - (float)hi[0], r.y - (float)hi[1], r.z - (float)hi[2], r.w - (float)hi[3]);
  return __builtin_bit_cast(uint2, hi);
}
__device__ __forceinline__ uint2 bf16_rn4(const float4 r) {
  using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
  const bf16x4 b = {(__bf16)r.x, (__bf16)r.y, (__bf16)r.z, (__bf16)r.w};
  return __builtin_bit_cast(uint2, b);
}
// eight channels (one whole group) -> its 32 bytes at `dst` (32-byte aligned): two 16-byte stores
__device__ __forceinline__ void store_split8(char* dst, const float4 v0, const float4 v1) {
  float4 l0, l1;
  const uint2 h0 = bf16_hi4(v0, l0), h1 = bf16_hi4(v1, l1);
  const uint2 q0 = bf16_rn4(l0), q1 = bf16_rn4(l1);
  *reinterpret_cast<uint4*>(dst) = make_uint4(h0.x, h0.y, h1.x, h1.y);
  *reinterpret_cast<uint4*>(dst + 16) = make_uint4(q0.x, q0.y, q1.x, q1.y);
}
// the inverse: the 32 bytes of a group (a = hi half, b = lo half, as loaded) -> its eight floats hi + lo in (a, b)
__device__ __forceinline__ void join_split8(float4& a, float4& b) {
  const unsigned h[4] = {__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w)};
  const unsigned l[4] = {__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), __float_as_uint(b.w)};
  float v[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {      // a bf16 is the upper half of the float with the same value
    v[2 * j] = __uint_as_float(h[j] << 16) + __uint_as_float(l[j] << 16);
    v[2 * j + 1] = __uint_as_float(h[j] & 0xffff0000u) + __uint_as_float(l[j] & 0xffff0000u);
  }
  a = make_float4(v[0], v[1], v[2], v[3]);
  b = make_float4(v[4], v[5], v[6], v[7]);
}
// four channels = HALF a group (producers whose threads own four channels: depthwise convs, Winograd output transforms):
// `pixel` = the pixel's first byte (channel 0 of the window), cg = index of the 4-channel unit; two 8-byte stores
__device__ __forceinline__ void store_split4(char* pixel, const int cg, const float4 v) {
  float4 l;
  const uint2 h = bf16_hi4(v, l);
  char* g = pixel + (cg >> 1) * 32 + (cg & 1) * 8;
  *reinterpret_cast<uint2*>(g) = h;
  *reinterpret_cast<uint2*>(g + 16) = bf16_rn4(l);
}

int conv_desc_check(const premvos_conv_desc& d);      // conv_igemm_f32.hip: what every dense-conv entry requires of a descriptor

}  // namespace premvos

#define PV_REQUIRE(cond, ...) \
  do {                        \
    if (!(cond)) return premvos::fail(PREMVOS_EINVAL, __VA_ARGS__); \
  } while (0)
