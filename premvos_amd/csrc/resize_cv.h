// cv2.resize(INTER_LINEAR) coordinate/coefficient rules shared by the flow and proposal
// pre-processing kernels (restated from OpenCV imgproc resize.cpp; see oracle/cv_resize_oracle.py).
#pragma once
#include <hip/hip_runtime.h>

namespace premvos {

// source taps s0,s1 and the fractional weight of s1 for destination index d
__device__ inline void cv_lin_coef_f(int d, double scale, int ssize, int* s0, int* s1, float* f1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  *s0 = s;
  *s1 = s + 1 < ssize ? s + 1 : ssize - 1;
  *f1 = f;
}

// uint8 path: 11-bit fixed-point coefficients, cvRound = round half to even
__device__ inline void cv_lin_coef(int d, double scale, int ssize, int* s0, int* s1, short* a0, short* a1) {
  float f;
  cv_lin_coef_f(d, scale, ssize, s0, s1, &f);
  *a0 = (short)__float2int_rn((1.f - f) * 2048.f);
  *a1 = (short)__float2int_rn(f * 2048.f);
}

// one uint8 channel of the fixed-point bilinear resize (HResizeLinear + VResizeLinear)
__device__ inline int cv_resize_u8_px(const uint8_t* im, int w, int cn, int ch, int x0, int x1, int y0, int y1,
                                      short a0, short a1, short b0, short b1) {
  const int r0 = im[((long)y0 * w + x0) * cn + ch] * a0 + im[((long)y0 * w + x1) * cn + ch] * a1;
  const int r1 = im[((long)y1 * w + x0) * cn + ch] * a0 + im[((long)y1 * w + x1) * cn + ch] * a1;
  int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
  return v < 0 ? 0 : v > 255 ? 255 : v;
}

}  // namespace premvos
