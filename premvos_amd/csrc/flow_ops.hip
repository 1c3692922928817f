// HBM-bound kernels of the PWC-Net flow path (gfx950): cost volume, bilinear warp + validity
// mask, NCHW<->NHWC edge transposes, and the flow driver's pre/post processing.
// Reference call sites: see include/premvos_hip.h.
#include "common.h"
#include "resize_cv.h"
#include "warp_math.h"

using premvos::cv_lin_coef;
using premvos::cv_lin_coef_f;

namespace premvos {
int corr81_tile(const float* f1, int f1_ps, const float* f2, int f2_ps, const float* flow, int flow_ps, float fscale,
                float* out, int out_ps, int n, int h, int w, int c, float slope, int copy_f1, hipStream_t s);   // corr_tile.hip
}

namespace {

// ------------------------------------------------------------------------------------------
// Cost volume, NHWC in / NHWC slice out (general md; md = 4 runs the LDS-tiled kernel of corr_tile.hip).  One thread per (pixel, displacement) with the
// displacement index fastest: the 81 outputs of a pixel are one coalesced 324-byte run, the
// f1 row is a wave-wide broadcast and neighbouring displacements read neighbouring f2 pixels
// (16-byte loads, served from L1/L2: each f2 pixel is touched by 81 displacements).
// Follows corr_cuda_kernel.cu:59-127 (index math, mean over C) for k=1, s1=s2=1, pad=md.
constexpr int CORR_PIX = 16;  // pixels per 256-thread block

__global__ __launch_bounds__(256) void corr_nhwc_kernel(const float* __restrict__ f1, int f1_ps,
                                                        const float* __restrict__ f2, int f2_ps,
                                                        float* __restrict__ out, int out_ps, int npix, int h,
                                                        int w, int c, int md, float slope, int copy_f1) {
  const int d = 2 * md + 1, d2 = d * d;
  const int per_pix = d2 + (copy_f1 ? c : 0);
  const int pix0 = blockIdx.x * CORR_PIX;
  const float inv_c = 1.0f;  // (division below keeps the reference's  sum / (float)sumelems)
  (void)inv_c;
  for (int o = threadIdx.x; o < CORR_PIX * per_pix; o += 256) {
    const int pl = o / per_pix, e = o - pl * per_pix;
    const int pix = pix0 + pl;
    if (pix >= npix) break;
    const float* a = f1 + (long)pix * f1_ps;
    if (e >= d2) {  // fused torch.cat((corr, c1, ...)) copy
      out[(long)pix * out_ps + e] = a[e - d2];
      continue;
    }
    const int hw = h * w;
    const int n = pix / hw, rem = pix - n * hw;
    const int y = rem / w, x = rem - y * w;
    const int dy = e / d - md, dx = e - (e / d) * d - md;
    const int y2 = y + dy, x2 = x + dx;
    float acc = 0.f;
    if ((unsigned)y2 < (unsigned)h && (unsigned)x2 < (unsigned)w) {
      const float* b = f2 + ((long)n * hw + (long)y2 * w + x2) * f2_ps;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int k = 0; k < c; k += 4) {
        const float4 av = *reinterpret_cast<const float4*>(a + k);
        const float4 bv = *reinterpret_cast<const float4*>(b + k);
        s0 += av.x * bv.x;
        s1 += av.y * bv.y;
        s2 += av.z * bv.z;
        s3 += av.w * bv.w;
      }
      acc = ((s0 + s1) + (s2 + s3)) / (float)c;
    }
    if (acc < 0.f) acc *= slope;
    out[(long)pix * out_ps + e] = acc;
  }
}

// General op-level form on NCHW tensors: the full argument list of corr_cuda_forward
// (corr_cuda.c:7-45), multiply type.  One thread per output element.
__global__ __launch_bounds__(256) void corr_nchw_kernel(const float* __restrict__ in1,
                                                        const float* __restrict__ in2,
                                                        float* __restrict__ out, int n, int c, int h, int w,
                                                        int pad, int ks, int md, int s1, int s2, int oh, int ow,
                                                        int r, int dd) {
  const long total = (long)n * dd * dd * oh * ow;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int ox = idx % ow;
    const int oy = (idx / ow) % oh;
    const int tc = (idx / ((long)ow * oh)) % (dd * dd);
    const int item = idx / ((long)ow * oh * dd * dd);
    // coordinates in the zero-padded frame (corr_cuda_kernel.cu:67-69,94-95)
    const int x1 = ox * s1 + md, y1 = oy * s1 + md;
    const int x2 = x1 + (tc % dd - r) * s2, y2 = y1 + (tc / dd - r) * s2;
    float sum = 0.f;
    for (int j = 0; j < ks; ++j)
      for (int i = 0; i < ks; ++i) {
        const int ya = y1 + j - pad, xa = x1 + i - pad, yb = y2 + j - pad, xb = x2 + i - pad;
        if ((unsigned)ya >= (unsigned)h || (unsigned)xa >= (unsigned)w || (unsigned)yb >= (unsigned)h ||
            (unsigned)xb >= (unsigned)w)
          continue;
        const float* a = in1 + ((long)item * c * h + ya) * w + xa;
        const float* b = in2 + ((long)item * c * h + yb) * w + xb;
        for (int ch = 0; ch < c; ++ch) sum += a[(long)ch * h * w] * b[(long)ch * h * w];
      }
    out[idx] = sum / (float)(ks * ks * c);
  }
}

// ------------------------------------------------------------------------------------------
// Backward bilinear warp x validity mask (PWCNet.py:140-176).  The float sequence mirrors
// the reference: grid = 2*(x+u)/max(W-1,1)-1, then grid_sample's align_corners=True
// un-normalisation ((g+1)/2)*(W-1), floor, corner weights (x1-ix)*(y1-iy) ...; the mask is
// the sum of the in-bounds corner weights thresholded at 0.9999 (warp_math.h).  One thread per (pixel, 4 ch).
#pragma clang fp contract(off)   // from here to the end of the file (warp, cv2-exact pre/post processing)
__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ x, int x_ps,
                                                   const float* __restrict__ flow, int flow_ps, float fscale,
                                                   float* __restrict__ out, int out_ps, int npix, int h, int w,
                                                   int c4) {
  const long total = (long)npix * c4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int pix = idx / c4, cg = idx - (long)pix * c4;
    const int hw = h * w;
    const int n = pix / hw, rem = pix - n * hw;
    const int py = rem / w, px = rem - py * w;
    const float u = flow[(long)pix * flow_ps] * fscale, v = flow[(long)pix * flow_ps + 1] * fscale;
    const float4 acc = premvos::warp_sample4(x + (long)n * hw * x_ps + cg * 4, x_ps, u, v, px, py, h, w);
    *reinterpret_cast<float4*>(out + (long)pix * out_ps + cg * 4) = acc;
  }
}

// ------------------------------------------------------------------------------------------
// Edge transposes through a 32x33 LDS tile (pixels x channels).
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int out_ps, int c, int hw, int cwrite) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int ch = c0 + i, p = p0 + tx;
    tile[i][tx] = (ch < c && p < hw) ? in[((long)n * c + ch) * hw + p] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int p = p0 + i, ch = c0 + tx;
    if (p < hw && ch < cwrite) out[((long)n * hw + p) * out_ps + ch] = tile[tx][i];
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, int in_ps,
                                                           float* __restrict__ out, int c, int hw) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int p = p0 + i, ch = c0 + tx;
    tile[i][tx] = (p < hw && ch < c) ? in[((long)n * hw + p) * in_ps + ch] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int ch = c0 + i, p = p0 + tx;
    if (ch < c && p < hw) out[((long)n * c + ch) * hw + p] = tile[tx][i];
  }
}

// ------------------------------------------------------------------------------------------
// cv2.resize(uint8, INTER_LINEAR) restated: 11-bit fixed-point coefficients, horizontal pass
// to int, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2   (OpenCV imgproc
// resize.cpp: HResizeLinear / VResizeLinear<uchar,int,short,FixedPtCast<..,22>>; third-party,
// absent from /root/reference and from this image -> parity unpinned).
__global__ __launch_bounds__(256) void flow_preprocess_kernel(const uint8_t* __restrict__ im1,
                                                              const uint8_t* __restrict__ im2, int batch, int h,
                                                              int w, float* __restrict__ out, int h_, int w_) {
  const long total = 2L * batch * h_ * w_;
  const double sx = 1.0 / ((double)w_ / (double)w), sy = 1.0 / ((double)h_ / (double)h);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = idx % w_, y = (idx / w_) % h_, n = idx / ((long)w_ * h_);
    const uint8_t* im = (n >= batch ? im2 : im1) + (long)(n % batch) * h * w * 3;
    int x0, x1, y0, y1;
    short a0, a1, b0, b1;
    cv_lin_coef(x, sx, w, &x0, &x1, &a0, &a1);
    cv_lin_coef(y, sy, h, &y0, &y1, &b0, &b1);
    float px[4];
    for (int ch = 0; ch < 3; ++ch) {
      int v;
      if (h_ == h && w_ == w) {
        v = im[((long)y * w + x) * 3 + ch];  // cv2.resize to the same size is a copy
      } else {
        v = premvos::cv_resize_u8_px(im, w, 3, ch, x0, x1, y0, y1, a0, a1, b0, b1);
      }
      px[2 - ch] = (float)((double)v / 255.0);  // RGB -> BGR, 1.0*im/255.0 in double then .float()
    }
    px[3] = 0.f;
    *reinterpret_cast<float4*>(out + idx * 4) = make_float4(px[0], px[1], px[2], px[3]);
  }
}

// cv2.resize(float32, INTER_LINEAR) of (20*flow2) to (w,h), then u *= w/w_, v *= h/h_.
__global__ __launch_bounds__(256) void flow_postprocess_kernel(const float* __restrict__ flow2, int ps,
                                                               int batch, int h4, int w4,
                                                               float* __restrict__ out, int h, int w, int h_,
                                                               int w_) {
  const long total = (long)batch * h * w;
  const double sx = 1.0 / ((double)w / (double)w4), sy = 1.0 / ((double)h / (double)h4);
  const float ku = (float)((double)w / (double)w_), kv = (float)((double)h / (double)h_);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = idx % w, y = (idx / w) % h;
    const float* fl = flow2 + (idx / ((long)w * h)) * h4 * w4 * ps;
    int x0, x1, y0, y1;
    float fx, fy;
    cv_lin_coef_f(x, sx, w4, &x0, &x1, &fx);
    cv_lin_coef_f(y, sy, h4, &y0, &y1, &fy);
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    float r[2];
    for (int ch = 0; ch < 2; ++ch) {
      const float s00 = fl[((long)y0 * w4 + x0) * ps + ch] * 20.0f, s01 = fl[((long)y0 * w4 + x1) * ps + ch] * 20.0f;
      const float s10 = fl[((long)y1 * w4 + x0) * ps + ch] * 20.0f, s11 = fl[((long)y1 * w4 + x1) * ps + ch] * 20.0f;
      const float r0 = s00 * a0 + s01 * a1, r1 = s10 * a0 + s11 * a1;
      r[ch] = r0 * b0 + r1 * b1;
    }
    out[idx * 2] = r[0] * ku;
    out[idx * 2 + 1] = r[1] * kv;
  }
}

// (round 5: a FLAT grid -- one item per thread -- moves 6.1 TB/s where a grid-stride loop over 4 ... 8 k resident workgroups moves 4.3 ... 4.7,
// tools/dev/hbm_copy_variants.hip; the cap only bounds the grid dimension)
inline int grid_for(long total, int per_block = 256, int cap = 1 << 22) {
  long g = (total + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : g > cap ? cap : g);
}

}  // namespace

extern "C" int premvos_corr_fwd_f32(const float* f1, int32_t f1_ps, const float* f2, int32_t f2_ps, float* out,
                                    int32_t out_ps, int32_t n, int32_t h, int32_t w, int32_t c, int32_t md,
                                    float slope, int32_t copy_f1, void* stream) {
  PV_REQUIRE(f1 && f2 && out, "corr: null pointer");
  PV_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && md >= 0, "corr: bad dims");
  PV_REQUIRE(c % 4 == 0 && f1_ps % 4 == 0 && f2_ps % 4 == 0 && f1_ps >= c && f2_ps >= c,
             "corr: C and pixel strides must be multiples of 4");
  PV_REQUIRE(premvos::aligned16(f1) && premvos::aligned16(f2), "corr: inputs must be 16-byte aligned");
  const int d2 = (2 * md + 1) * (2 * md + 1);
  PV_REQUIRE(out_ps >= d2 + (copy_f1 ? c : 0), "corr: out_ps too small");
  if (md == 4)   // the PWC-Net instantiation: LDS-tiled kernel (corr_tile.hip)
    return premvos::corr81_tile(f1, f1_ps, f2, f2_ps, nullptr, 0, 0.f, out, out_ps, n, h, w, c, slope, copy_f1,
                                static_cast<hipStream_t>(stream));
  const int npix = n * h * w;
  hipLaunchKernelGGL(corr_nhwc_kernel, dim3(premvos::cdiv(npix, CORR_PIX)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), f1, f1_ps, f2, f2_ps, out, out_ps, npix, h, w, c, md, slope,
                     copy_f1);
  return premvos::check_launch("corr_nhwc");
}

extern "C" int premvos_warp_corr_fwd_f32(const float* f1, int32_t f1_ps, const float* x2, int32_t x2_ps,
                                         const float* flow, int32_t flow_ps, float flow_scale, float* out,
                                         int32_t out_ps, int32_t n, int32_t h, int32_t w, int32_t c, int32_t md,
                                         float slope, int32_t copy_f1, void* stream) {
  PV_REQUIRE(f1 && x2 && flow && out, "warp_corr: null pointer");
  PV_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0, "warp_corr: bad dims");
  PV_REQUIRE(md == 4, "warp_corr: only md = 4 (the PWC-Net instantiation, PWCNet.py:69) is implemented");
  PV_REQUIRE(c % 4 == 0 && f1_ps % 4 == 0 && x2_ps % 4 == 0 && f1_ps >= c && x2_ps >= c,
             "warp_corr: C and pixel strides must be multiples of 4");
  PV_REQUIRE(premvos::aligned16(f1) && premvos::aligned16(x2), "warp_corr: inputs must be 16-byte aligned");
  PV_REQUIRE(flow_ps >= 2, "warp_corr: flow needs 2 channels");
  PV_REQUIRE(out_ps >= 81 + (copy_f1 ? c : 0), "warp_corr: out_ps too small");
  return premvos::corr81_tile(f1, f1_ps, x2, x2_ps, flow, flow_ps, flow_scale, out, out_ps, n, h, w, c, slope, copy_f1,
                              static_cast<hipStream_t>(stream));
}

extern "C" int premvos_corr_nchw_fwd_f32(const float* in1, const float* in2, float* out, int32_t n, int32_t c,
                                         int32_t h, int32_t w, int32_t pad_size, int32_t kernel_size,
                                         int32_t max_displacement, int32_t stride1, int32_t stride2,
                                         int32_t corr_type_multiply, void* stream) {
  PV_REQUIRE(in1 && in2 && out, "corr_nchw: null pointer");
  PV_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, "corr_nchw: bad dims");
  PV_REQUIRE(kernel_size >= 1 && (kernel_size & 1) && stride1 >= 1 && stride2 >= 1 && pad_size >= 0 &&
                 max_displacement >= 0,
             "corr_nchw: bad geometry");
  PV_REQUIRE(corr_type_multiply == 1, "corr_nchw: only the multiply type is implemented (PWCNet.py:69)");
  const int kr = (kernel_size - 1) / 2, border = max_displacement + kr;
  const int ph = h + 2 * pad_size, pw = w + 2 * pad_size;
  const int ow = (pw - 2 * border + stride1 - 1) / stride1, oh = (ph - 2 * border + stride1 - 1) / stride1;
  PV_REQUIRE(ow > 0 && oh > 0, "corr_nchw: empty output");
  const int r = max_displacement / stride2, dd = 2 * r + 1;
  const long total = (long)n * dd * dd * oh * ow;
  hipLaunchKernelGGL(corr_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), in1,
                     in2, out, n, c, h, w, pad_size, kernel_size, max_displacement, stride1, stride2, oh, ow, r, dd);
  return premvos::check_launch("corr_nchw");
}

extern "C" int premvos_warp_fwd_f32(const float* x, int32_t x_ps, const float* flow, int32_t flow_ps,
                                    float flow_scale, float* out, int32_t out_ps, int32_t n, int32_t h, int32_t w,
                                    int32_t c, void* stream) {
  PV_REQUIRE(x && flow && out, "warp: null pointer");
  PV_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0, "warp: bad dims");
  PV_REQUIRE(c % 4 == 0 && x_ps % 4 == 0 && out_ps % 4 == 0 && x_ps >= c && out_ps >= c && flow_ps >= 2,
             "warp: C and pixel strides must be multiples of 4");
  PV_REQUIRE(premvos::aligned16(x) && premvos::aligned16(out), "warp: x/out must be 16-byte aligned");
  const long total = (long)n * h * w * (c / 4);
  hipLaunchKernelGGL(warp_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, x_ps,
                     flow, flow_ps, flow_scale, out, out_ps, n * h * w, h, w, c / 4);
  return premvos::check_launch("warp");
}

extern "C" int premvos_nchw_to_nhwc_f32(const float* in, float* out, int32_t out_ps, int32_t n, int32_t c,
                                        int32_t h, int32_t w, void* stream) {
  PV_REQUIRE(in && out, "nchw_to_nhwc: null pointer");
  PV_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && out_ps >= c, "nchw_to_nhwc: bad dims");
  int cwrite = (c + 3) / 4 * 4;  // zero the pad channels up to the float4 boundary if they exist
  if (cwrite > out_ps) cwrite = out_ps;
  dim3 grid(premvos::cdiv(h * w, 32), premvos::cdiv(cwrite, 32), n);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in, out, out_ps, c,
                     h * w, cwrite);
  return premvos::check_launch("nchw_to_nhwc");
}

extern "C" int premvos_nhwc_to_nchw_f32(const float* in, int32_t in_ps, float* out, int32_t n, int32_t c,
                                        int32_t h, int32_t w, void* stream) {
  PV_REQUIRE(in && out, "nhwc_to_nchw: null pointer");
  PV_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && in_ps >= c, "nhwc_to_nchw: bad dims");
  dim3 grid(premvos::cdiv(h * w, 32), premvos::cdiv(c, 32), n);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), in, in_ps, out, c,
                     h * w);
  return premvos::check_launch("nhwc_to_nchw");
}

extern "C" int premvos_flow_preprocess_u8(const uint8_t* im1, const uint8_t* im2, int32_t batch, int32_t h, int32_t w,
                                          float* out, int32_t h_, int32_t w_, void* stream) {
  PV_REQUIRE(im1 && im2 && out, "flow_preprocess: null pointer");
  PV_REQUIRE(batch > 0 && h > 0 && w > 0 && h_ > 0 && w_ > 0, "flow_preprocess: bad dims");
  PV_REQUIRE(premvos::aligned16(out), "flow_preprocess: out must be 16-byte aligned");
  hipLaunchKernelGGL(flow_preprocess_kernel, dim3(grid_for(2L * batch * h_ * w_)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), im1, im2, batch, h, w, out, h_, w_);
  return premvos::check_launch("flow_preprocess");
}

extern "C" int premvos_flow_postprocess_f32(const float* flow2, int32_t flow_ps, int32_t batch, int32_t h4,
                                            int32_t w4, float* out, int32_t h, int32_t w, int32_t h_, int32_t w_,
                                            void* stream) {
  PV_REQUIRE(flow2 && out, "flow_postprocess: null pointer");
  PV_REQUIRE(batch > 0 && flow_ps >= 2 && h4 > 0 && w4 > 0 && h > 0 && w > 0 && h_ > 0 && w_ > 0,
             "flow_postprocess: bad dims");
  hipLaunchKernelGGL(flow_postprocess_kernel, dim3(grid_for((long)batch * h * w)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), flow2, flow_ps, batch, h4, w4, out, h, w, h_, w_);
  return premvos::check_launch("flow_postprocess");
}

// Host utility (no GPU): CRC-32C (Castagnoli) of a buffer, the checksum of TF tensor-bundle checkpoints
// (premvos_amd/weights.py reads/writes them without TensorFlow).  Byte-wise table, ~1 GB/s.
extern "C" uint32_t premvos_crc32c_host(const void* data, int64_t n) {
  struct Table {
    uint32_t v[256];
    Table() {
      for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        v[i] = c;
      }
    }
  };
  static const Table tbl;                       // built once, thread-safe (the loaders of two engines may run side by side)
  const uint32_t* table = tbl.v;
  const uint8_t* p = static_cast<const uint8_t*>(data);
  uint32_t c = 0xFFFFFFFFu;
  for (int64_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
