// Non-GEMM kernels of the refinement_net forward (gfx950): per-box input assembly (guidance mask, crop,
// TF-legacy bilinear / nearest resize, normalisation chain), depthwise 3x3 (+atrous) conv with folded
// BatchNorm, bilinear resize (both TF conventions), 1x1 -> HxW broadcast, and the SegmentationSoftmax
// eval branch (logit up-sampling, softmax, argmax, un-crop, zero-pad, conf_score reduction).
// Reference call sites: see include/premvos_hip.h.
#include "common.h"

#pragma clang fp contract(off)

namespace {

// (round 5: a FLAT grid -- one item per thread -- moves 6.1 TB/s where a grid-stride loop over 4 ... 8 k resident workgroups moves 4.3 ... 4.7,
// tools/dev/hbm_copy_variants.hip; the cap only bounds the grid dimension)
inline int grid_for(long total, int per_block = 256, int cap = 1 << 22) {
  long g = (total + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : g > cap ? cap : g);
}

// TF1 legacy resize coordinate: src = dst * scale, lower = floor, upper = min(ceil, in-1)
__device__ inline void tf_lerp(int d, float scale, int in_size, int* lo, int* hi, float* t) {
  const float s = (float)d * scale;
  const float f = floorf(s);
  *lo = (int)f;
  const int c = (int)ceilf(s);
  *hi = c < in_size - 1 ? c : in_size - 1;
  *t = s - f;
}

// ------------------------------------------------------------------------------------------
// Per-box network input (a19-a21): frame uint8 RGB -> /255; guidance = 1 inside round(box);
// crop = round(box) +- 50 px clipped; image bilinear (legacy) / guidance nearest (legacy) to S x S;
// (x-mean)/std, then DeepLab's own  unnormalize*255 -> (2/255)x-1  chain.  NHWC [P][S][S][4].
__global__ __launch_bounds__(256) void refine_input_kernel(const uint8_t* __restrict__ frame, int H, int W,
                                                           const float* __restrict__ boxes,
                                                           const int* __restrict__ count, int P, int S,
                                                           float* __restrict__ out, int* __restrict__ crops) {
  const long total = (long)P * S * S;
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  const int n = *count < P ? *count : P;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = idx % S, y = (idx / S) % S, p = idx / ((long)S * S);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < n) {
      const float* b = boxes + p * 4;   // y0 x0 y1 x1
      const int by0 = (int)rintf(b[0]), bx0 = (int)rintf(b[1]), by1 = (int)rintf(b[2]), bx1 = (int)rintf(b[3]);
      const int cy0 = by0 - 50 > 0 ? by0 - 50 : 0, cx0 = bx0 - 50 > 0 ? bx0 - 50 : 0;
      const int cy1 = by1 + 50 < H ? by1 + 50 : H, cx1 = bx1 + 50 < W ? bx1 + 50 : W;
      const int hc = cy1 - cy0, wc = cx1 - cx0;
      if (x == 0 && y == 0) {
        crops[p * 4 + 0] = cy0; crops[p * 4 + 1] = cx0; crops[p * 4 + 2] = cy1; crops[p * 4 + 3] = cx1;
      }
      if (hc > 0 && wc > 0) {
        int ylo, yhi, xlo, xhi;
        float ty, tx;
        tf_lerp(y, (float)hc / (float)S, hc, &ylo, &yhi, &ty);
        tf_lerp(x, (float)wc / (float)S, wc, &xlo, &xhi, &tx);
        const uint8_t* f00 = frame + ((long)(cy0 + ylo) * W + cx0 + xlo) * 3;
        const uint8_t* f01 = frame + ((long)(cy0 + ylo) * W + cx0 + xhi) * 3;
        const uint8_t* f10 = frame + ((long)(cy0 + yhi) * W + cx0 + xlo) * 3;
        const uint8_t* f11 = frame + ((long)(cy0 + yhi) * W + cx0 + xhi) * 3;
        float v[4];
        for (int ch = 0; ch < 3; ++ch) {
          const float tl = (float)f00[ch] / 255.f, tr = (float)f01[ch] / 255.f;
          const float bl = (float)f10[ch] / 255.f, br = (float)f11[ch] / 255.f;
          const float top = tl + (tr - tl) * tx, bot = bl + (br - bl) * tx;
          float r = top + (bot - top) * ty;
          r = (r - mean[ch]) / stdv[ch];                 // Normalization.normalize
          r = (r * stdv[ch] + mean[ch]) * 255.f;         // DeepLabV3Plus.py:12-14
          v[ch] = (float)(2.0 / 255.0) * r - 1.0f;       // feature_extractor.py:114-116
        }
        // nearest (legacy): src = min(floor(dst*in/out), in-1)
        int gy = (int)floorf((float)y * ((float)hc / (float)S)), gx = (int)floorf((float)x * ((float)wc / (float)S));
        gy = (gy < hc - 1 ? gy : hc - 1) + cy0;
        gx = (gx < wc - 1 ? gx : wc - 1) + cx0;
        const int gy0 = by0 > 0 ? by0 : 0, gx0 = bx0 > 0 ? bx0 : 0;
        const float g = (gy >= gy0 && gy < by1 && gx >= gx0 && gx < bx1) ? 1.f : 0.f;
        v[3] = (float)(2.0 / 255.0) * ((g * 1.f + 0.f) * 255.f) - 1.0f;
        o = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else if (x == 0 && y == 0) {
      crops[p * 4 + 0] = crops[p * 4 + 1] = crops[p * 4 + 2] = crops[p * 4 + 3] = 0;
    }
    *reinterpret_cast<float4*>(out + idx * 4) = o;
  }
}

// The store of one (pixel, 4-channel unit) result: floats, or -- PREMVOS_ACT_SPLIT8_BF16 (bit 9 of `act`) -- half of a group of the
// resident S8 layout of the bf16x3 mode (common.h: store_split4; csrc/conv_bf16x3_s8.hip reads it by LDS-DMA)
constexpr int DW_PAIRED = 0x400;     // internal flag of the launcher: an even number of 4-channel units per pixel -- lanes 2j and 2j + 1 own the
                                     // two halves of one S8 group and can swap a half by DPP: ONE 16-byte store per lane instead of two 8-byte ones
__device__ __forceinline__ void store_unit(float* pixel, const int cg, const float4 r, const int act) {
  if (act & PREMVOS_ACT_SPLIT8_BF16) {
    if (act & DW_PAIRED) {
      float4 l;
      const uint2 h = premvos::bf16_hi4(r, l);
      const uint2 lo = premvos::bf16_rn4(l);
      const bool odd = cg & 1;
      const uint2 send = odd ? h : lo;                   // the even lane stores the hi half {h_even, h_odd}, the odd lane the lo half
      uint2 recv;                                        // quad_perm [1, 0, 3, 2]: neighbours swap (both lanes of a pair are always active together)
      recv.x = (unsigned)__builtin_amdgcn_mov_dpp((int)send.x, 0xB1, 0xF, 0xF, true);
      recv.y = (unsigned)__builtin_amdgcn_mov_dpp((int)send.y, 0xB1, 0xF, 0xF, true);
      char* g = reinterpret_cast<char*>(pixel) + (cg >> 1) * 32;
      *reinterpret_cast<uint4*>(g + (odd ? 16 : 0)) = odd ? make_uint4(recv.x, recv.y, lo.x, lo.y) : make_uint4(h.x, h.y, recv.x, recv.y);
    } else {
      premvos::store_split4(reinterpret_cast<char*>(pixel), cg, r);
    }
  } else {
    *reinterpret_cast<float4*>(pixel + cg * 4) = r;
  }
}

// ------------------------------------------------------------------------------------------
// Depthwise 3x3 conv (+stride, +atrous) with folded BatchNorm: out = act(sum_taps w*relu?(in) + bias).
// HBM-bound; one thread per (output pixel, 4 channels), weights [9][C] (BN scale folded in).
template <bool PRE_RELU>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ in, int in_ps, int n, int h, int w,
                                                        int c4, const float* __restrict__ wgt,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int out_ps, int ho, int wo, int stride, int dil, int pt, int pl,
                                                        int act, int cpad) {
  const long total = (long)n * ho * wo * c4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = idx % c4;
    const long pix = idx / c4;
    const int ox = pix % wo, oy = (pix / wo) % ho, b = pix / ((long)wo * ho);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int iy = oy * stride - pt + j * dil;
      if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int ix = ox * stride - pl + i * dil;
        if ((unsigned)ix >= (unsigned)w) continue;
        float4 v = *reinterpret_cast<const float4*>(in + (((long)b * h + iy) * w + ix) * in_ps + cg * 4);
        if (PRE_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        const float4 k = *reinterpret_cast<const float4*>(wgt + (long)(j * 3 + i) * cpad + cg * 4);
        acc.x += v.x * k.x; acc.y += v.y * k.y; acc.z += v.z * k.z; acc.w += v.w * k.w;
      }
    }
    if (bias != nullptr) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + cg * 4);
      acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
    }
    if ((act & 0xff) == PREMVOS_ACT_RELU) {
      acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    }
    store_unit(out + pix * out_ps, cg, acc, act);
  }
}

// Row-tiled variant for dilation 1: one thread produces TW consecutive output pixels of a row for 4 channels,
// so the 3 x (TW*stride+2) input window and the 9 weight vectors are loaded once per TW outputs
// (2.7x fewer load instructions than the per-pixel kernel at TW = 4; the kernel is L2/TA-bound, not HBM-bound).
template <bool PRE_RELU, int STRIDE, int TW>
__global__ __launch_bounds__(256) void dwconv3x3_row_kernel(const float* __restrict__ in, int in_ps, int n, int h, int w,
                                                            int c4, const float* __restrict__ wgt,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int out_ps, int ho, int wo, int pt, int pl, int act, int cpad) {
  constexpr int NCOL = (TW - 1) * STRIDE + 3;
  const int xt = (wo + TW - 1) / TW;
  const long total = (long)n * ho * xt * c4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = idx % c4;
    long t = idx / c4;
    const int tx = t % xt;
    t /= xt;
    const int oy = t % ho, b = t / ho;
    const int ox0 = tx * TW;
    float4 k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = *reinterpret_cast<const float4*>(wgt + (long)i * cpad + cg * 4);
    float4 acc[TW];
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias != nullptr) bv = *reinterpret_cast<const float4*>(bias + cg * 4);
#pragma unroll
    for (int o = 0; o < TW; ++o) acc[o] = bv;
    const int ix0 = ox0 * STRIDE - pl, iy0 = oy * STRIDE - pt;
    // all 3 x NCOL taps are requested before the first one is used; taps outside the map are not loaded (lanes masked off)
    // and zeroed at use -- a select right behind a load would make the wave wait for it before issuing the next load
    float4 raw[3][NCOL];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float* rowp = in + (((long)b * h + iy0 + j) * w) * in_ps + cg * 4;
#pragma unroll
      for (int cx = 0; cx < NCOL; ++cx) {
        raw[j][cx] = premvos::arbitrary4();
        if ((unsigned)(iy0 + j) < (unsigned)h && (unsigned)(ix0 + cx) < (unsigned)w)
          raw[j][cx] = *reinterpret_cast<const float4*>(rowp + (long)(ix0 + cx) * in_ps);
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if ((unsigned)(iy0 + j) >= (unsigned)h) continue;
      float4 v[NCOL];
#pragma unroll
      for (int cx = 0; cx < NCOL; ++cx) {
        const bool ok = (unsigned)(ix0 + cx) < (unsigned)w;
        const float4 t = raw[j][cx];
        if constexpr (PRE_RELU)
          v[cx] = make_float4(ok ? fmaxf(t.x, 0.f) : 0.f, ok ? fmaxf(t.y, 0.f) : 0.f, ok ? fmaxf(t.z, 0.f) : 0.f, ok ? fmaxf(t.w, 0.f) : 0.f);
        else                                  // no max(x, -inf) identity here: fmaxf(NaN, -inf) = -inf would hide a NaN activation
          v[cx] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
      }
#pragma unroll
      for (int o = 0; o < TW; ++o)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float4 a = v[o * STRIDE + i], kk = k[j * 3 + i];
          acc[o].x += a.x * kk.x; acc[o].y += a.y * kk.y; acc[o].z += a.z * kk.z; acc[o].w += a.w * kk.w;
        }
    }
#pragma unroll
    for (int o = 0; o < TW; ++o) {
      const int ox = ox0 + o;
      if (ox >= wo) break;
      float4 r = acc[o];
      if ((act & 0xff) == PREMVOS_ACT_RELU) {
        r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f);
      }
      store_unit(out + (((long)b * ho + oy) * wo + ox) * out_ps, cg, r, act);
    }
  }
}

// Register-tiled variant for dilation 1: one thread owns a TR x TW block of output pixels for 4 channels and walks the
// (TR-1)*STRIDE+3 input rows once; every loaded row feeds all the output rows it touches, and an output row is stored
// as soon as its third input row has been consumed (so only ~3 accumulator rows are live).  Per output this is
// NROW*NCOL/(TR*TW) float4 loads (1.9 at 8x4, 1.96 at 5x5) against 4.5 for the row kernel; accumulation order per
// output (bias, then taps in (ky,kx) order) is the same as in the other two kernels -> identical bits.
template <bool PRE_RELU, int STRIDE, int TW, int TR, bool AHEAD>
__global__ __launch_bounds__(256) void dwconv3x3_tile_kernel(const float* __restrict__ in, int in_ps, int n, int h, int w,
                                                             int c4, const float* __restrict__ wgt,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             int out_ps, int ho, int wo, int pt, int pl, int act,
                                                             int cpad, int dil) {
  // dil > 1 (atrous, stride 1): the outputs of a tile are `dil` apart, i.e. the tile lives on one of the dil x dil
  // sub-lattices of the output, on which the dilated conv is an ordinary 3x3 conv; tile index -> (residue, position).
  constexpr int NCOL = (TW - 1) * STRIDE + 3;
  constexpr int NROW = (TR - 1) * STRIDE + 3;
  const int xt = dil * (((wo + dil - 1) / dil + TW - 1) / TW), yt = dil * (((ho + dil - 1) / dil + TR - 1) / TR);
  const long total = (long)n * yt * xt * c4;
  // Workgroup b runs on XCD b % 8 (private L2 each): give every XCD a contiguous run of the (image, tile row, tile column,
  // channel group) raster, so that the tiles which share halo rows / columns fetch them through ONE L2.
  // (measured: +8 % on the 25x25 maps with 5x5 tiles, a loss on the large entry-flow maps, where the plain order keeps the
  // eight XCDs on neighbouring DRAM pages)
  const long vblock = TW == 5 ? premvos::xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  for (long idx = vblock * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = idx % c4;
    long t = idx / c4;
    const int tx = t % xt;
    t /= xt;
    const int ty = t % yt, b = t / yt;
    const int ox0 = (tx % dil) + (tx / dil) * TW * dil, oy0 = (ty % dil) + (ty / dil) * TR * dil;
    float4 k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = *reinterpret_cast<const float4*>(wgt + (long)i * cpad + cg * 4);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias != nullptr) bv = *reinterpret_cast<const float4*>(bias + cg * 4);
    float4 acc[TR][TW];
#pragma unroll
    for (int o = 0; o < TR; ++o)
#pragma unroll
      for (int q = 0; q < TW; ++q) acc[o][q] = bv;
    const int ix0 = ox0 * STRIDE - pl, iy0 = oy0 * STRIDE - pt;
    const float* imgp = in + ((long)b * h * w) * in_ps + cg * 4;
    // Rows are loaded one ahead of the row being consumed (two rows of 16-byte loads per lane in flight).  Out-of-image taps
    // are not loaded (their lanes are masked off) and are zeroed when the row is consumed: a select right after the load would
    // make the wave wait for it before issuing the next one.
    auto row_ok = [&](int j) { return (unsigned)(iy0 + j * dil) < (unsigned)h; };
    auto col_ok = [&](int cx) { return (unsigned)(ix0 + cx * dil) < (unsigned)w; };
    auto load_row = [&](int j, float4* v) {
      const float* rowp = imgp + ((long)(iy0 + j * dil) * w) * in_ps;
#pragma unroll
      for (int cx = 0; cx < NCOL; ++cx) {
        // (measured: giving the masked-off lanes a defined arbitrary content helps the variant without a row in flight --
        //  wide atrous layers 461 -> 380 us -- and costs the one with it 126 -> 207 us; there an out-of-range tap's register is
        //  simply never read, see the conditional at the use)
        if constexpr (!AHEAD) v[cx] = premvos::arbitrary4();
        if (row_ok(j) && col_ok(cx)) v[cx] = *reinterpret_cast<const float4*>(rowp + (long)(ix0 + cx * dil) * in_ps);
      }
    };
    float4 vbuf[2][NCOL];
    load_row(0, vbuf[0]);
#pragma unroll
    for (int j = 0; j < NROW; ++j) {
      if (AHEAD && j + 1 < NROW) load_row(j + 1, vbuf[(j + 1) & 1]);
      if (!AHEAD && j > 0) load_row(j, vbuf[j & 1]);
      float4 v[NCOL];
#pragma unroll
      for (int cx = 0; cx < NCOL; ++cx) {
        const bool ok = row_ok(j) && col_ok(cx);
        const float4& t = vbuf[j & 1][cx];                    // (read only where ok: the conditional below)
        if constexpr (PRE_RELU)
          v[cx] = make_float4(ok ? fmaxf(t.x, 0.f) : 0.f, ok ? fmaxf(t.y, 0.f) : 0.f, ok ? fmaxf(t.z, 0.f) : 0.f, ok ? fmaxf(t.w, 0.f) : 0.f);
        else                                                  // (NaN / Inf activations propagate unchanged, as in the reference)
          v[cx] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
      }
#pragma unroll
      for (int o = 0; o < TR; ++o) {
        const int kj = j - o * STRIDE;             // compile-time after unrolling
        if (kj < 0 || kj > 2) continue;
#pragma unroll
        for (int q = 0; q < TW; ++q)
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const float4 a = v[q * STRIDE + i], kk = k[kj * 3 + i];
            acc[o][q].x += a.x * kk.x; acc[o][q].y += a.y * kk.y; acc[o][q].z += a.z * kk.z; acc[o][q].w += a.w * kk.w;
          }
        if (kj == 2 && oy0 + o * dil < ho) {       // this output row is complete
          float* orow = out + (((long)b * ho + oy0 + o * dil) * wo) * out_ps;
#pragma unroll
          for (int q = 0; q < TW; ++q) {
            if (ox0 + q * dil >= wo) break;
            float4 r = acc[o][q];
            if ((act & 0xff) == PREMVOS_ACT_RELU) {
              r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f);
            }
            store_unit(orow + (long)(ox0 + q * dil) * out_ps, cg, r, act);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// tf.image.resize_bilinear (TF1): align_corners=True  src = dst*(in-1)/(out-1);  False (legacy) src = dst*in/out
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, int in_ps, int n, int h,
                                                              int w, int c4, float* __restrict__ out, int out_ps,
                                                              int ho, int wo, int align) {
  const long total = (long)n * ho * wo * c4;
  const float sy = (align && ho > 1) ? (float)(h - 1) / (float)(ho - 1) : (float)h / (float)ho;
  const float sx = (align && wo > 1) ? (float)(w - 1) / (float)(wo - 1) : (float)w / (float)wo;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = idx % c4;
    const long pix = idx / c4;
    const int ox = pix % wo, oy = (pix / wo) % ho, b = pix / ((long)wo * ho);
    int ylo, yhi, xlo, xhi;
    float ty, tx;
    tf_lerp(oy, sy, h, &ylo, &yhi, &ty);
    tf_lerp(ox, sx, w, &xlo, &xhi, &tx);
    const float* base = in + (long)b * h * w * in_ps + cg * 4;
    const float4 tl = *reinterpret_cast<const float4*>(base + ((long)ylo * w + xlo) * in_ps);
    const float4 tr = *reinterpret_cast<const float4*>(base + ((long)ylo * w + xhi) * in_ps);
    const float4 bl = *reinterpret_cast<const float4*>(base + ((long)yhi * w + xlo) * in_ps);
    const float4 br = *reinterpret_cast<const float4*>(base + ((long)yhi * w + xhi) * in_ps);
    float4 o;
    float top, bot;
    top = tl.x + (tr.x - tl.x) * tx; bot = bl.x + (br.x - bl.x) * tx; o.x = top + (bot - top) * ty;
    top = tl.y + (tr.y - tl.y) * tx; bot = bl.y + (br.y - bl.y) * tx; o.y = top + (bot - top) * ty;
    top = tl.z + (tr.z - tl.z) * tx; bot = bl.z + (br.z - bl.z) * tx; o.z = top + (bot - top) * ty;
    top = tl.w + (tr.w - tl.w) * tx; bot = bl.w + (br.w - bl.w) * tx; o.w = top + (bot - top) * ty;
    *reinterpret_cast<float4*>(out + pix * out_ps + cg * 4) = o;
  }
}

// [N][1][1][C] -> every pixel of [N][H][W] (ASPP image-level feature; a bilinear resize of a 1x1 map)
__global__ __launch_bounds__(256) void broadcast_kernel(const float* __restrict__ in, int in_ps, int n, int hw, int c4,
                                                        float* __restrict__ out, int out_ps) {
  const long total = (long)n * hw * c4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = idx % c4;
    const long pix = idx / c4;
    const int b = pix / hw;
    *reinterpret_cast<float4*>(out + pix * out_ps + cg * 4) =
        *reinterpret_cast<const float4*>(in + (long)b * in_ps + cg * 4);
  }
}

// ------------------------------------------------------------------------------------------
// SegmentationSoftmax eval branch, stage 1: logits [P][lh][lw][2] --legacy bilinear--> S x S, softmax
// foreground probability and argmax class (ties -> class 0, like tf.argmax).
__global__ __launch_bounds__(256) void seg_softmax_kernel(const float* __restrict__ logits, int ps, int P, int lh,
                                                          int lw, int S, float* __restrict__ prob,
                                                          uint8_t* __restrict__ cls) {
  const long total = (long)P * S * S;
  const float sy = (float)lh / (float)S, sx = (float)lw / (float)S;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = idx % S, y = (idx / S) % S, p = idx / ((long)S * S);
    int ylo, yhi, xlo, xhi;
    float ty, tx;
    tf_lerp(y, sy, lh, &ylo, &yhi, &ty);
    tf_lerp(x, sx, lw, &xlo, &xhi, &tx);
    const float* base = logits + (long)p * lh * lw * ps;
    float l[2];
    for (int ch = 0; ch < 2; ++ch) {
      const float tl = base[((long)ylo * lw + xlo) * ps + ch], tr = base[((long)ylo * lw + xhi) * ps + ch];
      const float bl = base[((long)yhi * lw + xlo) * ps + ch], br = base[((long)yhi * lw + xhi) * ps + ch];
      const float top = tl + (tr - tl) * tx, bot = bl + (br - bl) * tx;
      l[ch] = top + (bot - top) * ty;
    }
    const float m = fmaxf(l[0], l[1]);
    const float e0 = expf(l[0] - m), e1 = expf(l[1] - m);
    prob[idx] = e1 / (e0 + e1);
    cls[idx] = l[1] > l[0] ? 1 : 0;
  }
}

// stage 2: un-crop.  mask = nearest(cls) / posterior = legacy-bilinear(prob) resized to the crop, zero padded
// to the frame; per-block partial sums of the conf_score integrand (2p-1 inside the mask, 1-2p outside)
// in a fixed order (deterministic), reduced by seg_conf_kernel.
constexpr int UNCROP_PIX = 2048;   // frame pixels per block

__global__ __launch_bounds__(256) void seg_uncrop_kernel(const float* __restrict__ prob,
                                                         const uint8_t* __restrict__ cls, int S,
                                                         const int* __restrict__ crops, const int* __restrict__ count,
                                                         int P, int H, int W, uint8_t* __restrict__ mask,
                                                         float* __restrict__ post, double* __restrict__ partial,
                                                         int nblk) {
  __shared__ double red[256];
  const int p = blockIdx.y, blk = blockIdx.x;
  const int n = *count < P ? *count : P;
  const long hw = (long)H * W;
  double s = 0.0;
  if (p < n) {
    const int cy0 = crops[p * 4], cx0 = crops[p * 4 + 1], cy1 = crops[p * 4 + 2], cx1 = crops[p * 4 + 3];
    const int hc = cy1 - cy0, wc = cx1 - cx0;
    const float* pr = prob + (long)p * S * S;
    const uint8_t* cl = cls + (long)p * S * S;
    for (long i = (long)blk * UNCROP_PIX + threadIdx.x; i < (long)(blk + 1) * UNCROP_PIX && i < hw; i += 256) {
      const int y = i / W, x = i - (long)y * W;
      uint8_t mv = 0;
      float pv = 0.f;
      if (y >= cy0 && y < cy1 && x >= cx0 && x < cx1) {
        const int yy = y - cy0, xx = x - cx0;
        const float sy = (float)S / (float)hc, sx = (float)S / (float)wc;
        int ny = (int)floorf((float)yy * sy), nx = (int)floorf((float)xx * sx);
        ny = ny < S - 1 ? ny : S - 1;
        nx = nx < S - 1 ? nx : S - 1;
        mv = cl[(long)ny * S + nx];
        int ylo, yhi, xlo, xhi;
        float ty, tx;
        tf_lerp(yy, sy, S, &ylo, &yhi, &ty);
        tf_lerp(xx, sx, S, &xlo, &xhi, &tx);
        const float tl = pr[(long)ylo * S + xlo], tr = pr[(long)ylo * S + xhi];
        const float bl = pr[(long)yhi * S + xlo], br = pr[(long)yhi * S + xhi];
        const float top = tl + (tr - tl) * tx, bot = bl + (br - bl) * tx;
        pv = top + (bot - top) * ty;
      }
      mask[(long)p * hw + i] = mv;
      if (post != nullptr) post[(long)p * hw + i] = pv;
      const float c = mv ? pv : 1.f - pv;       // FewShotSegmentationForwarder.py:144-147
      s += (double)(2.f * c - 1.f);
    }
  } else {
    for (long i = (long)blk * UNCROP_PIX + threadIdx.x; i < (long)(blk + 1) * UNCROP_PIX && i < hw; i += 256) {
      mask[(long)p * hw + i] = 0;
      if (post != nullptr) post[(long)p * hw + i] = 0.f;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(long)p * nblk + blk] = red[0];
}

__global__ void seg_conf_kernel(const double* __restrict__ partial, int nblk, int P, long hw, float* __restrict__ conf) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += partial[(long)p * nblk + i];
  conf[p] = (float)(s / (double)hw);
}

}  // namespace

extern "C" int premvos_refine_input_u8(const uint8_t* frame_rgb, int32_t h, int32_t w, const float* boxes_y0x0y1x1,
                                       const int32_t* count, int32_t max_boxes, int32_t size, float* out,
                                       int32_t* crop_boxes, void* stream) {
  PV_REQUIRE(frame_rgb && boxes_y0x0y1x1 && count && out && crop_boxes, "refine_input: null pointer");
  PV_REQUIRE(h > 0 && w > 0 && max_boxes > 0 && size > 1, "refine_input: bad dims");
  PV_REQUIRE(premvos::aligned16(out), "refine_input: out must be 16-byte aligned");
  hipLaunchKernelGGL(refine_input_kernel, dim3(grid_for((long)max_boxes * size * size)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), frame_rgb, h, w, boxes_y0x0y1x1, count, max_boxes, size, out,
                     crop_boxes);
  return premvos::check_launch("refine_input");
}

extern "C" int premvos_dwconv3x3_f32(const float* in, int32_t in_ps, int32_t n, int32_t h, int32_t w, int32_t c,
                                     const float* wgt, const float* bias, int32_t c_pad, float* out, int32_t out_ps,
                                     int32_t ho, int32_t wo, int32_t stride, int32_t dilation, int32_t pt, int32_t pl,
                                     int32_t pre_relu, int32_t act, void* stream) {
  PV_REQUIRE(in && wgt && out, "dwconv3x3: null pointer");
  PV_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && ho > 0 && wo > 0 && stride > 0 && dilation > 0, "dwconv3x3: bad dims");
  PV_REQUIRE(c_pad % 4 == 0 && c_pad >= c && in_ps % 4 == 0 && out_ps % 4 == 0 && in_ps >= c_pad && out_ps >= c_pad,
             "dwconv3x3: channel count / strides must be padded to multiples of 4");
  PV_REQUIRE(premvos::aligned16(in) && premvos::aligned16(out) && premvos::aligned16(wgt) &&
                 (bias == nullptr || premvos::aligned16(bias)),
             "dwconv3x3: pointers must be 16-byte aligned");
  PV_REQUIRE((act & 0xff) == PREMVOS_ACT_NONE || (act & 0xff) == PREMVOS_ACT_RELU, "dwconv3x3: bad activation");
  PV_REQUIRE((act & ~0xff & ~PREMVOS_ACT_SPLIT8_BF16) == 0, "dwconv3x3: unknown output-layout flags");
  PV_REQUIRE(!(act & PREMVOS_ACT_SPLIT8_BF16) || (out_ps % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 31u) == 0),
             "dwconv3x3: an S8 output needs out_ps %% 8 == 0 and a 32-byte aligned channel window");
  if ((act & PREMVOS_ACT_SPLIT8_BF16) && (c_pad / 4) % 2 == 0) act |= DW_PAIRED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (stride == 1 && (wo + dilation - 1) / dilation >= (dilation == 1 ? 8 : 3) &&
      (ho + dilation - 1) / dilation >= (dilation == 1 ? 8 : 3)) {   // register-tiled fast path (stride 2: row kernel wins)
    // tile: 5x5 when both extents are multiples of 5 and small (the 25x25 maps of the middle/exit flow), else 8 rows
    // x 4 columns, 4x4 when that leaves the chip short of threads
    const int c4 = c_pad / 4;
    int tr = 8, tw = 4;
    if (dilation > 1) tr = 4;                       // sub-lattices of an atrous layer are short: 4x4 tiles
    else if (ho % 5 == 0 && wo % 5 == 0 && ho <= 50) tr = tw = 5;
    else if ((long)n * ((ho + 7) / 8) * ((wo + 3) / 4) * c4 < 256L * 1024) tr = 4;
    const long tot = (long)n * (dilation * (((ho + dilation - 1) / dilation + tr - 1) / tr)) *
                     (dilation * (((wo + dilation - 1) / dilation + tw - 1) / tw)) * c4;
    const dim3 g(grid_for(tot)), b(256);
#define PV_DW_TILE(PR, ST, TW_, TR_)                                                                                  \
  do {                                                                                                                \
    if (dilation <= 4)                                                                                                \
      hipLaunchKernelGGL((dwconv3x3_tile_kernel<PR, ST, TW_, TR_, true>), g, b, 0, s, in, in_ps, n, h, w, c4, wgt, bias, \
                         out, out_ps, ho, wo, pt, pl, act, c_pad, dilation);                                          \
    else /* wide atrous: most taps of the short sub-lattices fall outside the map; a second row in flight only costs registers */ \
      hipLaunchKernelGGL((dwconv3x3_tile_kernel<PR, ST, TW_, TR_, false>), g, b, 0, s, in, in_ps, n, h, w, c4, wgt, bias, \
                         out, out_ps, ho, wo, pt, pl, act, c_pad, dilation);                                          \
  } while (0)
#define PV_DW_SHAPE(PR, ST)                                                                                           \
  do {                                                                                                                \
    if (tr == 5) PV_DW_TILE(PR, ST, 5, 5);                                                                            \
    else if (tr == 8) PV_DW_TILE(PR, ST, 4, 8);                                                                       \
    else PV_DW_TILE(PR, ST, 4, 4);                                                                                    \
  } while (0)
    if (pre_relu) PV_DW_SHAPE(true, 1);
    else PV_DW_SHAPE(false, 1);
#undef PV_DW_SHAPE
#undef PV_DW_TILE
    return premvos::check_launch("dwconv3x3_tile");
  }
  if (dilation == 1 && (stride == 1 || stride == 2) && wo >= 8) {   // row-tiled path (stride 2, short maps)
    constexpr int TW = 4;
    const long tot = (long)n * ho * ((wo + TW - 1) / TW) * (c_pad / 4);
    const dim3 g(grid_for(tot)), b(256);
#define PV_DW_ROW(PR, ST)                                                                                            \
  hipLaunchKernelGGL((dwconv3x3_row_kernel<PR, ST, TW>), g, b, 0, s, in, in_ps, n, h, w, c_pad / 4, wgt, bias, out, \
                     out_ps, ho, wo, pt, pl, act, c_pad)
    if (pre_relu && stride == 1) PV_DW_ROW(true, 1);
    else if (pre_relu) PV_DW_ROW(true, 2);
    else if (stride == 1) PV_DW_ROW(false, 1);
    else PV_DW_ROW(false, 2);
#undef PV_DW_ROW
    return premvos::check_launch("dwconv3x3_row");
  }
  const long total = (long)n * ho * wo * (c_pad / 4);
  if (pre_relu)
    hipLaunchKernelGGL(dwconv3x3_kernel<true>, dim3(grid_for(total)), dim3(256), 0, s, in, in_ps, n, h, w, c_pad / 4,
                       wgt, bias, out, out_ps, ho, wo, stride, dilation, pt, pl, act, c_pad);
  else
    hipLaunchKernelGGL(dwconv3x3_kernel<false>, dim3(grid_for(total)), dim3(256), 0, s, in, in_ps, n, h, w, c_pad / 4,
                       wgt, bias, out, out_ps, ho, wo, stride, dilation, pt, pl, act, c_pad);
  return premvos::check_launch("dwconv3x3");
}

extern "C" int premvos_resize_bilinear_f32(const float* in, int32_t in_ps, int32_t n, int32_t h, int32_t w, int32_t c,
                                           float* out, int32_t out_ps, int32_t ho, int32_t wo, int32_t align_corners,
                                           void* stream) {
  PV_REQUIRE(in && out, "resize_bilinear: null pointer");
  PV_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && ho > 0 && wo > 0, "resize_bilinear: bad dims");
  PV_REQUIRE(c % 4 == 0 && in_ps % 4 == 0 && out_ps % 4 == 0 && in_ps >= c && out_ps >= c,
             "resize_bilinear: C and strides must be multiples of 4");
  PV_REQUIRE(premvos::aligned16(in) && premvos::aligned16(out), "resize_bilinear: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for((long)n * ho * wo * (c / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, in_ps, n, h, w, c / 4, out, out_ps, ho, wo, align_corners);
  return premvos::check_launch("resize_bilinear");
}

extern "C" int premvos_broadcast_pixel_f32(const float* in, int32_t in_ps, int32_t n, int32_t c, float* out,
                                           int32_t out_ps, int32_t h, int32_t w, void* stream) {
  PV_REQUIRE(in && out, "broadcast_pixel: null pointer");
  PV_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, "broadcast_pixel: bad dims");
  PV_REQUIRE(c % 4 == 0 && in_ps % 4 == 0 && out_ps % 4 == 0 && in_ps >= c && out_ps >= c,
             "broadcast_pixel: C and strides must be multiples of 4");
  PV_REQUIRE(premvos::aligned16(in) && premvos::aligned16(out), "broadcast_pixel: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(broadcast_kernel, dim3(grid_for((long)n * h * w * (c / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, in_ps, n, h * w, c / 4, out, out_ps);
  return premvos::check_launch("broadcast_pixel");
}

extern "C" int64_t premvos_refine_output_workspace_bytes(int32_t max_boxes, int32_t size, int32_t h, int32_t w) {
  const long nblk = ((long)h * w + UNCROP_PIX - 1) / UNCROP_PIX;
  return (int64_t)max_boxes * size * size * (sizeof(float) + 1) + (int64_t)max_boxes * nblk * sizeof(double) + 256;
}

extern "C" int premvos_refine_output_f32(const float* logits, int32_t logits_ps, int32_t lh, int32_t lw,
                                         const int32_t* crop_boxes, const int32_t* count, int32_t max_boxes,
                                         int32_t size, int32_t h, int32_t w, uint8_t* mask, float* posterior,
                                         float* conf_score, void* workspace, void* stream) {
  PV_REQUIRE(logits && crop_boxes && count && mask && conf_score && workspace, "refine_output: null pointer");
  PV_REQUIRE(logits_ps >= 2 && lh > 0 && lw > 0 && max_boxes > 0 && size > 1 && h > 0 && w > 0,
             "refine_output: bad dims");
  PV_REQUIRE(premvos::aligned16(workspace), "refine_output: workspace must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long ss = (long)max_boxes * size * size;
  const int nblk = (int)(((long)h * w + UNCROP_PIX - 1) / UNCROP_PIX);
  char* ws = static_cast<char*>(workspace);
  double* partial = reinterpret_cast<double*>(ws);
  float* prob = reinterpret_cast<float*>(ws + (((long)max_boxes * nblk * sizeof(double) + 255) / 256) * 256);
  uint8_t* cls = reinterpret_cast<uint8_t*>(prob + ss);
  hipLaunchKernelGGL(seg_softmax_kernel, dim3(grid_for(ss)), dim3(256), 0, s, logits, logits_ps, max_boxes, lh, lw, size,
                     prob, cls);
  int rc = premvos::check_launch("seg_softmax");
  if (rc) return rc;
  hipLaunchKernelGGL(seg_uncrop_kernel, dim3(nblk, max_boxes), dim3(256), 0, s, prob, cls, size, crop_boxes, count,
                     max_boxes, h, w, mask, posterior, partial, nblk);
  rc = premvos::check_launch("seg_uncrop");
  if (rc) return rc;
  hipLaunchKernelGGL(seg_conf_kernel, dim3((max_boxes + 63) / 64), dim3(64), 0, s, partial, nblk, max_boxes, (long)h * w,
                     conf_score);
  return premvos::check_launch("seg_conf");
}
