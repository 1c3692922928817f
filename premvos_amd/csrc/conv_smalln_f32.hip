// Direct convolution for layers with ONE or TWO output channels (PWC-Net's predict_flow / dc_conv7 heads,
// models/PWCNet.py:30-31,131; DeepLab's 2-class logits, network/deeplab/model.py:601-661).  On the implicit-GEMM kernel
// such a layer fills 2 of the 32 columns of an MFMA tile (measured 4-5 TFLOP/s); it is really a per-pixel dot product
// over K = taps x cin, i.e. HBM/L1-bound work: 16 lanes share one output pixel, each lane walks the pixel's channels in
// float4 steps (256 contiguous bytes per pixel per step), the <= 2 weight rows sit in LDS, the 16 partial sums are
// combined with a butterfly of wave shuffles in a fixed order (deterministic).
#include "common.h"

namespace {

constexpr int LPP = 16;   // lanes per output pixel

template <int NOUT>
__global__ __launch_bounds__(256) void conv_smalln_kernel(const premvos_conv_desc p) {
  extern __shared__ __attribute__((aligned(16))) float wl[];          // [NOUT][k_pad]
  for (int i = threadIdx.x * 4; i < NOUT * p.k_pad; i += 256 * 4)
    *reinterpret_cast<float4*>(wl + i) = *reinterpret_cast<const float4*>(p.wgt + (long)(i / p.k_pad) * p.k_pad + i % p.k_pad);
  __syncthreads();
  const int sub = threadIdx.x & (LPP - 1);
  const long M = (long)p.n * p.ho * p.wo;
  const long g0 = ((long)blockIdx.x * 256 + threadIdx.x) / LPP;
  const long gstride = (long)gridDim.x * 256 / LPP;
  const int hw = p.ho * p.wo;
  for (long m = g0; m < M; m += gstride) {
    const int n = (int)(m / hw), rem = (int)(m - (long)n * hw);
    const int oy = rem / p.wo, ox = rem - oy * p.wo;
    const float* img = p.in + (long)n * p.h * p.w * p.in_ps;
    float acc[NOUT];
#pragma unroll
    for (int co = 0; co < NOUT; ++co) acc[co] = 0.f;
    for (int ky = 0; ky < p.kh; ++ky) {
      const int iy = oy * p.sh - p.pt + ky * p.dh;
      if ((unsigned)iy >= (unsigned)p.h) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        const int ix = ox * p.sw - p.pl + kx * p.dw;
        if ((unsigned)ix >= (unsigned)p.w) continue;
        const float* src = img + ((long)iy * p.w + ix) * p.in_ps;
        const float* wt = wl + (ky * p.kw + kx) * p.cin_pad;
        // 4 independent 16-byte loads in flight per lane (the loop is latency-bound otherwise: one L2 round trip per step)
        for (int c = sub * 4; c < p.cin_pad; c += 4 * LPP * 4) {
          float4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int cu = c + u * LPP * 4;
            if (cu < p.cin_pad) v[u] = *reinterpret_cast<const float4*>(src + cu);   // (never read otherwise; no select behind the load)
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int cu = c + u * LPP * 4;
            if (cu >= p.cin_pad) break;
#pragma unroll
            for (int co = 0; co < NOUT; ++co) {
              const float4 w = *reinterpret_cast<const float4*>(wt + co * p.k_pad + cu);
              acc[co] += v[u].x * w.x + v[u].y * w.y + v[u].z * w.z + v[u].w * w.w;
            }
          }
        }
      }
    }
#pragma unroll
    for (int co = 0; co < NOUT; ++co) {
#pragma unroll
      for (int off = LPP / 2; off > 0; off >>= 1) acc[co] += __shfl_xor(acc[co], off, LPP);
    }
    if (sub == 0) {
#pragma unroll
      for (int co = 0; co < NOUT; ++co) {
        if (co >= p.cout) break;
        float v = acc[co] + (p.bias != nullptr ? p.bias[co] : 0.f);
        if (p.res != nullptr) v += p.res[m * p.res_ps + co];
        if (p.act == PREMVOS_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (p.act == PREMVOS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
        else if (p.act == PREMVOS_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
        p.out[m * p.out_ps + co] = v;
      }
    }
  }
}

}  // namespace

namespace premvos {

bool conv_smalln_applicable(const premvos_conv_desc& d) {
  return d.precision == PREMVOS_PREC_F32 && d.cout <= 2 && d.out_mode == PREMVOS_OUT_NHWC &&
         (long)d.cout * d.k_pad * (long)sizeof(float) <= 150 * 1024 && d.kh * d.kw * d.cin_pad >= 32;
}

int conv_smalln(const premvos_conv_desc& d, hipStream_t s) {
  const long M = (long)d.n * d.ho * d.wo;
  const int lds = d.cout * d.k_pad * (int)sizeof(float);
  long blocks = (M * LPP + 255) / 256;
  if (blocks > 256L * 12) blocks = 256L * 12;          // persistent-ish: <= 12 workgroups per CU, grid-stride over pixels
  if (blocks < 1) blocks = 1;
  if (d.cout == 1) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_smalln_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(conv_smalln_kernel<1>, dim3((unsigned)blocks), dim3(256), lds, s, d);
  } else {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_smalln_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(conv_smalln_kernel<2>, dim3((unsigned)blocks), dim3(256), lds, s, d);
  }
  return check_launch("conv_smalln_f32");
}

}  // namespace premvos
