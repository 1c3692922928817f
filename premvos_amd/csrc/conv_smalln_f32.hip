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

// 3x3 / stride 1 / dilation 1 heads (predict_flow2..6 read a 500-600-channel concat buffer for TWO outputs, PWCNet.py:131): the
// per-pixel form above fetches a pixel's channels once per TAP -- 9 x 568 x 4 B = 20 KB per output pixel, 9.4 GB per launch at level
// 2, served by L1 / L2 at ~11 TB/s: cache-bandwidth-bound at 0.82 ms where HBM needs 0.2.  Here 16 lanes own a TY x TX tile of
// outputs and walk the (TY + 2) x (TX + 2) input patch ONCE per 64-channel step: every loaded float4 feeds all the (<= 9) outputs
// whose window covers its pixel (2.25 loads per output for 4 x 4 tiles instead of 9).  The 3 x NOUT weight float4s of a tap row
// are read from LDS per (patch row, output row) pair.  Fixed summation order (channel step, patch row, patch column; 16-lane butterfly at the end): deterministic.
template <int NOUT, int TY, int TX>
__global__ __launch_bounds__(256, 2) void conv_smalln_tile_kernel(const premvos_conv_desc p, const int tiles_y, const int tiles_x) {
  extern __shared__ __attribute__((aligned(16))) float wl[];          // [NOUT][k_pad]
  for (int i = threadIdx.x * 4; i < NOUT * p.k_pad; i += 256 * 4)
    *reinterpret_cast<float4*>(wl + i) = *reinterpret_cast<const float4*>(p.wgt + (long)(i / p.k_pad) * p.k_pad + i % p.k_pad);
  __syncthreads();
  const int sub = threadIdx.x & (LPP - 1);
  const long ntile = (long)p.n * tiles_y * tiles_x;
  const long g0 = ((long)blockIdx.x * 256 + threadIdx.x) / LPP;
  const long gstride = (long)gridDim.x * 256 / LPP;
  for (long t = g0; t < ntile; t += gstride) {
    const int n = (int)(t / (tiles_y * tiles_x)), rem = (int)(t - (long)n * tiles_y * tiles_x);
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int oy0 = ty * TY, ox0 = tx * TX;
    const float* img = p.in + (long)n * p.h * p.w * p.in_ps;
    float acc[TY][TX][NOUT];
#pragma unroll
    for (int a = 0; a < TY; ++a)
#pragma unroll
      for (int b = 0; b < TX; ++b)
#pragma unroll
        for (int co = 0; co < NOUT; ++co) acc[a][b][co] = 0.f;
    // patch rows / columns outside the map: clamped addresses (always readable), the loaded value replaced by zero
    int xoff[TX + 2];
    float mx[TX + 2], my[TY + 2];                          // 1 inside the map, 0 outside (selects the loaded value or 0)
#pragma unroll
    for (int q = 0; q < TX + 2; ++q) {
      const int ix = ox0 - 1 + q;
      mx[q] = (unsigned)ix < (unsigned)p.w ? 1.f : 0.f;
      xoff[q] = min(max(ix, 0), p.w - 1) * p.in_ps;
    }
#pragma unroll
    for (int r = 0; r < TY + 2; ++r) my[r] = (unsigned)(oy0 - 1 + r) < (unsigned)p.h ? 1.f : 0.f;
#pragma unroll 1
    for (int c = sub * 4; c < p.cin_pad; c += 4 * LPP) {
#pragma unroll
      for (int r = 0; r < TY + 2; ++r) {
        const float* rowp = img + (long)min(max(oy0 - 1 + r, 0), p.h - 1) * p.w * p.in_ps + c;
        float4 v[TX + 2];
        // (xoff / mx pass through an empty asm: otherwise the optimiser hoists all 36 addresses and 36 mask products of the patch
        //  out of the channel loop -- ~110 loop-invariant registers -- and the loop spills)
#pragma unroll
        for (int q = 0; q < TX + 2; ++q) {                   // the row's requests go out together
          int xo = xoff[q];
          asm volatile("" : "+v"(xo));
          v[q] = premvos::ld4(rowp + xo);
        }
#pragma unroll
        for (int q = 0; q < TX + 2; ++q) {
          float m = mx[q];
          asm volatile("" : "+v"(m));
          const bool in = m * my[r] != 0.f;                  // selected, not multiplied: 0 * Inf / NaN of a clamped border pixel
          v[q] = in ? v[q] : make_float4(0.f, 0.f, 0.f, 0.f);  // must not reach a sum that true zero padding leaves finite
        }
#pragma unroll
        for (int a = 0; a < TY; ++a) {
          const int ky = r - a;                              // patch row r is tap row ky of output row a
          if (ky < 0 || ky > 2) continue;
          float4 w[3][NOUT];
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int co = 0; co < NOUT; ++co) w[kx][co] = *reinterpret_cast<const float4*>(wl + co * p.k_pad + (ky * 3 + kx) * p.cin_pad + c);
#pragma unroll
          for (int q = 0; q < TX + 2; ++q)
#pragma unroll
            for (int b = 0; b < TX; ++b) {
              const int kx = q - b;
              if (kx < 0 || kx > 2) continue;
#pragma unroll
              for (int co = 0; co < NOUT; ++co) {
                const float4 ww = w[kx][co];
                acc[a][b][co] += v[q].x * ww.x + v[q].y * ww.y + v[q].z * ww.z + v[q].w * ww.w;
              }
            }
          asm volatile("" ::: "memory");                     // (one tap row's weights in registers at a time)
          __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" ::: "memory");                        // one patch row in flight at a time: neither the optimiser nor the
        __builtin_amdgcn_sched_barrier(0);                   // scheduler may pull the next row's requests up (36 float4 would spill)
      }
    }
#pragma unroll
    for (int a = 0; a < TY; ++a)
#pragma unroll
      for (int b = 0; b < TX; ++b)
#pragma unroll
        for (int co = 0; co < NOUT; ++co) {
#pragma unroll
          for (int off = LPP / 2; off > 0; off >>= 1) acc[a][b][co] += __shfl_xor(acc[a][b][co], off, LPP);
        }
    // lane `sub` writes output (sub / TX, sub % TX) of the tile (TY * TX <= 16): select its sums without a runtime register index
    float mine[NOUT];
#pragma unroll
    for (int co = 0; co < NOUT; ++co) mine[co] = 0.f;
#pragma unroll
    for (int a = 0; a < TY; ++a)
#pragma unroll
      for (int b = 0; b < TX; ++b)
#pragma unroll
        for (int co = 0; co < NOUT; ++co) mine[co] = sub == a * TX + b ? acc[a][b][co] : mine[co];
    const int oy = oy0 + sub / TX, ox = ox0 + sub % TX;
    if (sub < TY * TX && oy < p.ho && ox < p.wo) {
      const long m = ((long)n * p.ho + oy) * p.wo + ox;
#pragma unroll
      for (int co = 0; co < NOUT; ++co) {
        if (co >= p.cout) break;
        float v = mine[co] + (p.bias != nullptr ? p.bias[co] : 0.f);
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
  // 3x3 / stride 1 / "same" heads: the tiled form (stage_k == 1 keeps the per-pixel form: developer A/B, tests)
  // (chosen by the layer's geometry per IMAGE, never by the batch: a ragged last chunk must get the bits of a full one.
  //  Measured, tools/dev/smalln_ab.py: 828 -> 322 us on predict_flow2 (16 x 128 x 224 x 565), 228 -> 93 on level 3, 115 -> 36 on
  //  dc_conv7; maps below ~100 tiles stay on the per-pixel form, which has 16x more independent work items)
  constexpr int TY = 4, TX = 4;
  const int tiles_y = cdiv((int)d.ho, TY), tiles_x = cdiv((int)d.wo, TX);
  const bool tiled = d.kh == 3 && d.kw == 3 && d.sh == 1 && d.sw == 1 && d.dh == 1 && d.dw == 1 && d.pt == 1 && d.pl == 1 && d.ho == d.h &&
                     d.wo == d.w && d.cout == 2 && d.stage_k != 1 && tiles_y * tiles_x >= 100;
  if (tiled) {
    long tb = ((long)d.n * tiles_y * tiles_x * LPP + 255) / 256;
    if (tb > 256L * 12) tb = 256L * 12;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_smalln_tile_kernel<2, TY, TX>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((conv_smalln_tile_kernel<2, TY, TX>), dim3((unsigned)tb), dim3(256), lds, s, d, tiles_y, tiles_x);
    return check_launch("conv_smalln_f32 (tiled)");
  }
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
