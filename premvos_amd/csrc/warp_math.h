// Backward bilinear warp of one pixel x 4 channels with the validity mask folded in (PWCDCNet.warp,
// models/PWCNet.py:140-176, on torch-0.2 grid_sample: bilinear, zeros padding, align_corners=True).
// The float sequence mirrors the reference: grid = 2*(x+u)/max(W-1,1)-1, grid_sample's un-normalisation
// ((g+1)/2)*(W-1), floor, corner weights (x1-ix)*(y1-iy) ...; the mask is the sum of the in-bounds corner weights
// thresholded at 0.9999 (:171-174).  Shared by warp_kernel (flow_ops.hip) and the fused warp + cost-volume kernel
// (corr_tile.hip) so both produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace premvos {

// img: NHWC image base already offset to the first of the 4 channels; (u, v) = flow * scale at (px, py).
__device__ __forceinline__ float4 warp_sample4(const float* __restrict__ img, int ps, float u, float v, int px, int py,
                                               int h, int w) {
#pragma clang fp contract(off)
  const float wm = (float)(w - 1 > 1 ? w - 1 : 1), hm = (float)(h - 1 > 1 ? h - 1 : 1);
  const float gx = 2.0f * ((float)px + u) / wm - 1.0f;
  const float gy = 2.0f * ((float)py + v) / hm - 1.0f;
  const float ix = ((gx + 1.f) / 2.f) * (float)(w - 1);
  const float iy = ((gy + 1.f) / 2.f) * (float)(h - 1);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const float fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
  const float wnw = (fx1 - ix) * (fy1 - iy), wne = (ix - fx0) * (fy1 - iy);
  const float wsw = (fx1 - ix) * (iy - fy0), wse = (ix - fx0) * (iy - fy0);
  // float compare keeps huge |flow| (beyond int range) out of bounds without UB
  const bool x0ok = fx0 >= 0.f && fx0 <= (float)(w - 1), x1ok = fx1 >= 0.f && fx1 <= (float)(w - 1);
  const bool y0ok = fy0 >= 0.f && fy0 <= (float)(h - 1), y1ok = fy1 >= 0.f && fy1 <= (float)(h - 1);
  const int x0 = x0ok ? (int)fx0 : 0, x1 = x1ok ? (int)fx1 : 0;
  const int y0 = y0ok ? (int)fy0 : 0, y1 = y1ok ? (int)fy1 : 0;
  // Branch-free taps: an out-of-bounds corner reads pixel (0,0)-clamped memory and is then replaced by zeros, so it adds
  // exactly +0 to the sums (the weights are >= 0) -- the same bits as skipping it, with all four loads in flight at once.
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float msum = 0.f;
  const bool ok[4] = {y0ok && x0ok, y0ok && x1ok, y1ok && x0ok, y1ok && x1ok};
  const int ty[4] = {y0, y0, y1, y1}, tx[4] = {x0, x1, x0, x1};
  const float wt[4] = {wnw, wne, wsw, wse};
  float4 vv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) vv[t] = *reinterpret_cast<const float4*>(img + ((long)ty[t] * w + tx[t]) * ps);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float4 q = ok[t] ? vv[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    acc.x += q.x * wt[t];
    acc.y += q.y * wt[t];
    acc.z += q.z * wt[t];
    acc.w += q.w * wt[t];
    msum += ok[t] ? wt[t] : 0.f;
  }
  if (!(msum >= 0.9999f)) acc = make_float4(0.f, 0.f, 0.f, 0.f);  // NaN-safe: mask<0.9999 -> 0
  return acc;
}

}  // namespace premvos
