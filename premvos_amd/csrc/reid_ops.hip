// ReID embedding net (code/ReID_net): the two operations its wide pre-activation ResNet needs next to
// the dense convs (premvos_conv2d_f32), max-pool and the FC layers (1x1 convs):
//   * per-box crops     datasets/Similarity/DAVIS_Forward_Feed.py:62-96, Similarity.py:288-297
//   * BatchNorm + ReLU on a tensor that is ALSO consumed raw (the unit's identity shortcut): NetworkLayers.py:171-173
// Both are HBM-bound element-wise passes.
#include "common.h"

namespace {

__device__ inline void tf_lerp(int d, float scale, int in_size, int* lo, int* hi, float* t) {   // TF1 legacy coordinates
  const float s = (float)d * scale;
  const float f = floorf(s);
  *lo = (int)f;
  const int c = (int)ceilf(s);
  *hi = c < in_size - 1 ? c : in_size - 1;
  *t = s - f;
}

// boxes: int32 [n][4] = (x, y, w, h) already context-expanded / rounded / clipped by the host.  zero_small: boxes with
// min(w, h) <= 10 yield an all-zero image BEFORE normalisation (the in-merge feed dataset), as do empty boxes.
__global__ __launch_bounds__(256) void reid_input_kernel(const uint8_t* __restrict__ frame, int H, int W,
                                                         const int* __restrict__ boxes, int n, int S, int zero_small,
                                                         float* __restrict__ out) {
  const long total = (long)n * S * S;
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = idx % S, y = (idx / S) % S, p = idx / ((long)S * S);
    int bx = boxes[p * 4], by = boxes[p * 4 + 1], bw = boxes[p * 4 + 2], bh = boxes[p * 4 + 3];
    // tensor slicing [y:y+h, x:x+w] clips to the image
    int x1 = bx + bw, y1 = by + bh;
    bx = bx < 0 ? 0 : bx; by = by < 0 ? 0 : by;
    x1 = x1 > W ? W : x1; y1 = y1 > H ? H : y1;
    const int wc = x1 - bx, hc = y1 - by;
    float v[3] = {0.f, 0.f, 0.f};
    const bool small = zero_small && (bw < bh ? bw : bh) <= 10;
    if (!small && wc > 0 && hc > 0) {
      int ylo, yhi, xlo, xhi;
      float ty, tx;
      tf_lerp(y, (float)hc / (float)S, hc, &ylo, &yhi, &ty);
      tf_lerp(x, (float)wc / (float)S, wc, &xlo, &xhi, &tx);
      const uint8_t* f00 = frame + ((long)(by + ylo) * W + bx + xlo) * 3;
      const uint8_t* f01 = frame + ((long)(by + ylo) * W + bx + xhi) * 3;
      const uint8_t* f10 = frame + ((long)(by + yhi) * W + bx + xlo) * 3;
      const uint8_t* f11 = frame + ((long)(by + yhi) * W + bx + xhi) * 3;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        // in-merge feed: image / 255 (DAVIS_Forward_Feed.py:27); batch stage: tf.image.convert_image_dtype = cast * (1 / 255)
        // (Util/Reader.py:162) -- not the same float for 39 % of the byte values
        const float r255 = 1.0f / 255.0f;
        const float tl = zero_small ? (float)f00[ch] / 255.f : (float)f00[ch] * r255;
        const float tr = zero_small ? (float)f01[ch] / 255.f : (float)f01[ch] * r255;
        const float bl = zero_small ? (float)f10[ch] / 255.f : (float)f10[ch] * r255;
        const float br = zero_small ? (float)f11[ch] / 255.f : (float)f11[ch] * r255;
        const float top = tl + (tr - tl) * tx, bot = bl + (br - bl) * tx;
        v[ch] = top + (bot - top) * ty;
      }
    }
    *reinterpret_cast<float4*>(out + idx * 4) =
        make_float4((v[0] - mean[0]) / stdv[0], (v[1] - mean[1]) / stdv[1], (v[2] - mean[2]) / stdv[2], 0.f);
  }
}

// out = max(in * scale[c] + shift[c], 0) (relu != 0) over NHWC pixels; float4 over channels.
__global__ __launch_bounds__(256) void scale_shift_relu_kernel(const float* __restrict__ in, int in_ps, long npix, int c4,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift, float* __restrict__ out,
                                                               int out_ps, int relu) {
  const long total = npix * c4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = idx % c4;
    const long p = idx / c4;
    const float4 v = *reinterpret_cast<const float4*>(in + p * in_ps + cg * 4);
    const float4 s = *reinterpret_cast<const float4*>(scale + cg * 4);
    const float4 t = *reinterpret_cast<const float4*>(shift + cg * 4);
    float4 r = make_float4(v.x * s.x + t.x, v.y * s.y + t.y, v.z * s.z + t.z, v.w * s.w + t.w);
    if (relu) {
      r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f);
    }
    *reinterpret_cast<float4*>(out + p * out_ps + cg * 4) = r;
  }
}

inline int grid_for(long total) {
  long g = (total + 255) / 256;
  return (int)(g > 65535L * 16 ? 65535L * 16 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int premvos_reid_input_u8(const uint8_t* frame_rgb, int32_t h, int32_t w, const int32_t* boxes_xywh, int32_t n,
                                     int32_t size, int32_t zero_small, float* out, void* stream) {
  PV_REQUIRE(frame_rgb && boxes_xywh && out, "reid_input: null pointer");
  PV_REQUIRE(h > 0 && w > 0 && n > 0 && size > 0, "reid_input: bad dims");
  PV_REQUIRE(premvos::aligned16(out), "reid_input: out must be 16-byte aligned");
  hipLaunchKernelGGL(reid_input_kernel, dim3(grid_for((long)n * size * size)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), frame_rgb, h, w, boxes_xywh, n, size, zero_small, out);
  return premvos::check_launch("reid_input");
}

extern "C" int premvos_scale_shift_relu_f32(const float* in, int32_t in_ps, int64_t npix, int32_t c, const float* scale,
                                            const float* shift, float* out, int32_t out_ps, int32_t relu, void* stream) {
  PV_REQUIRE(in && scale && shift && out, "scale_shift_relu: null pointer");
  PV_REQUIRE(npix > 0 && c > 0 && c % 4 == 0 && in_ps % 4 == 0 && out_ps % 4 == 0 && in_ps >= c && out_ps >= c,
             "scale_shift_relu: C and strides must be multiples of 4");
  PV_REQUIRE(premvos::aligned16(in) && premvos::aligned16(out) && premvos::aligned16(scale) && premvos::aligned16(shift),
             "scale_shift_relu: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(scale_shift_relu_kernel, dim3(grid_for((long)npix * (c / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, in_ps, (long)npix, c / 4, scale, shift, out, out_ps, relu);
  return premvos::check_launch("scale_shift_relu");
}
