// 3x3 / stride-1 dense convolution by Winograd F(4x4, 3x3) on the fp32 matrix pipe: 4x fewer multiplies than the implicit GEMM
// (F(2x2,3x3) of conv_wino_f32.hip: 2.25x) for the K-rich layers -- the RPN 3x3 (1024 -> 1024), ResNet conv2 of the conv4 / conv5
// bottlenecks (256 -> 256 on the 46x83 map, 512 -> 512 on 1600 7x7 RoI maps).  Chosen per layer by the plan-time autotuner
// (tile_hint 4) where it measures faster than the implicit GEMM and both F(2x2,3x3) kernels.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        per 4x4 output tile, 6x6 input patch d, 3x3 filter g   (Lavin & Gray 2016,
//                                                interpolation points 0, +-1, +-2, inf)
//
// 36 components are too many for the slab-free form (16 output-pixel accumulator sets per wave), and the rows of this B^T have
// three to four non-zeros (a fused input transform would gather up to 16 pixels per staged value), so the three steps are three
// launches around workspace slabs:
//   wino4_input_kernel   V[c][tile][k]   = (B^T d B)[c]          one thread per (tile, 4 channels): 36 16-byte loads (the 6x6
//                                                                 patches of neighbouring tiles overlap in L2), 36 16-byte stores
//   wino4_gemm_kernel    M[c][tile][n]   = V[c][tile][:] . U[c][n][:]    36 independent GEMMs in ONE launch; the LDS-staged,
//                                                                 double-buffered v_mfma_f32_32x32x2_f32 loop of the other conv
//                                                                 kernels with plain operands; an XCD owns whole components, so
//                                                                 the U[c] it multiplies by stays in its L2
//   wino4_output_kernel  Y = A^T M A, + bias (+ folded BatchNorm) + residual + activation -> the tile's 4x4 pixels
// Slab traffic: 2.25x the input (V) + 2.25x the output (M), each written and read once -- small against the 4x fewer MFMA cycles
// when K and N are in the hundreds, which is why the tuner only sees this candidate there.
//
// fp32 throughout.  Rounding: the transforms scale by up to 100 (B^T) and 1/576 (G) before the products, ~1e-5 relative to the
// output scale against ~1e-6 for F(2x2,3x3) (tests/test_gpu_conv_wino.py: 2e-4 bar against the fp64 convolution).
#include "common.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 operator+(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float s, const float4& a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }

// t = B^T d for one column (or row) of six values:
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void bt6(const float4& d0, const float4& d1, const float4& d2, const float4& d3, const float4& d4,
                                    const float4& d5, float4 (&t)[6]) {
  const float4 a = d4 - 4.f * d2, b = d3 - 4.f * d1;            // shared by rows 1, 2
  const float4 c = d4 - d2, e = 2.f * (d3 - d1);                // shared by rows 3, 4
  t[0] = 4.f * d0 - 5.f * d2 + d4;
  t[1] = a + b;
  t[2] = a - b;
  t[3] = c + e;
  t[4] = c - e;
  t[5] = 4.f * d1 - 5.f * d3 + d5;
}

// V[c = 6 i + j][tile][v_c0 + k] = (B^T d B)[i][j] of the tile's 6x6 patch (origin 4 ty - pad_top, 4 tx - pad_left) for the channels
// k in [0, kcount) of the layer's input window, zero outside the map and for k >= cin_pad.  Vp = row pitch of the slab: the layer's
// own Kp (v_c0 = 0, kcount = Kp) in the self-contained form; the pitch of a DenseNet level's shared slab in the kept-slab form
// (premvos_conv_wino4_slab_f32), where only the channels the previous layer added are transformed.
__global__ __launch_bounds__(256) void wino4_input_kernel(const premvos_conv_desc p, float* __restrict__ V, const int tiles_y,
                                                          const int tiles_x, const int Vp, const int v_c0, const int kcount) {
  // atrous layers (dilation d = p.dh = p.dw, PWCNet.py:266 dc_conv2..5): the d x d interleaved sub-grids are independent dense 3x3
  // problems -- tile (phase (py, px), ty, tx) covers outputs ((4 ty + a) d + py, (4 tx + b) d + px) and reads the inputs d apart
  const int d = p.dh, tpp = tiles_y * tiles_x, tpi = d * d * tpp;
  const int Mt = p.n * tpi, kg = kcount / 4;
  const long total = (long)Mt * kg, cstride = (long)Mt * Vp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int m = (int)(idx / kg), k = (int)(idx - (long)m * kg) * 4;
    const int n = m / tpi, rem = m - n * tpi;
    const int ph = rem / tpp, tt = rem - ph * tpp;
    const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
    const int y0 = 4 * ty * d + ph / d - p.pt, x0 = 4 * tx * d + ph % d - p.pl;
    const bool kok = k < p.cin_pad;
    const float* img = p.in + (long)n * p.h * p.w * p.in_ps + k;
    float4 t[6][6];                                             // t[j][i] = (B^T d)[i][j]: the column pass, stored transposed
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      float4 dd[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const int y = y0 + a * d, x = x0 + b * d;
        const bool ok = kok && (unsigned)y < (unsigned)p.h && (unsigned)x < (unsigned)p.w;
        dd[a] = ok ? *reinterpret_cast<const float4*>(img + ((long)y * p.w + x) * p.in_ps) : f4(0.f);
      }
      bt6(dd[0], dd[1], dd[2], dd[3], dd[4], dd[5], t[b]);
    }
    float* dst = V + (long)m * Vp + v_c0 + k;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float4 v[6];
      bt6(t[0][i], t[1][i], t[2][i], t[3][i], t[4][i], t[5][i], v);
#pragma unroll
      for (int j = 0; j < 6; ++j) *reinterpret_cast<float4*>(dst + (long)(6 * i + j) * cstride) = v[j];
    }
  }
}

// M[c][tile][col] = sum_k V[c][tile][v_c0 + k] * U[c][col][k]; rows past Mt / cout_pad are computed on clamped addresses and never
// read.  Vp = row pitch of V (Kp, or a shared slab's pitch with the layer's window at channel v_c0).
template <int BM, int BN, int WM, int WN, int KB>
__global__ __launch_bounds__(256) void wino4_gemm_kernel(const float* __restrict__ V, const float* __restrict__ U,
                                                            float* __restrict__ Ms, const int Mt, const int Kp, const int cout_pad,
                                                            const int m_tiles, const int n_tiles, const int Vp, const int v_c0) {
  constexpr int NT = 256, RS = KB + 4, KU = KB / 4;
  static_assert(WM * WN == 4, "four waves per workgroup");
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NTL = WTN / 32;
  constexpr int A_PER_T = BM * KU / NT, B_PER_T = BN * KU / NT;
  constexpr int BUF = (BM + BN) * RS;
  static_assert(BM * KU % NT == 0 && BN * KU % NT == 0, "staging must divide evenly");
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  float(*lds)[BUF] = reinterpret_cast<float(*)[BUF]>(lds_dyn);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * WTM, wn0 = (wave % WN) * WTN;
  // component-major order, XCD-contiguous: an XCD works through whole components (its U[c] and the V[c] rows its column tiles
  // share stay in that XCD's L2; measured: row-major / column-major tiles inside a component and no XCD remap are within 1 %,
  // component-fastest is 5 % slower)
  const int v = premvos::xcd_contiguous(blockIdx.x, gridDim.x);
  const int per_comp = m_tiles * n_tiles;
  const int comp = v / per_comp, rest = v - comp * per_comp;
  const int tile_m = rest / n_tiles, tile_n = rest - tile_m * n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int j4 = (tid % KU) * 4;

  const float* arow[A_PER_T];
  const float* brow[B_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int m = min(m0 + (tid / KU) + i * (NT / KU), Mt - 1);
    arow[i] = V + ((long)comp * Mt + m) * Vp + v_c0 + j4;
  }
#pragma unroll
  for (int i = 0; i < B_PER_T; ++i) {
    const int c = min(n0 + (tid / KU) + i * (NT / KU), cout_pad - 1);
    brow[i] = U + ((long)comp * cout_pad + c) * Kp + j4;
  }
  float4 ra[A_PER_T], rb[B_PER_T];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) ra[i] = premvos::ld4(arow[i] + kt * KB);
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) rb[i] = premvos::ld4(brow[i] + kt * KB);
  };
  auto lstore = [&](int buf) {
    float* a = &lds[buf][0];
    float* b = &lds[buf][BM * RS];
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) *reinterpret_cast<float4*>(a + ((tid / KU) + i * (NT / KU)) * RS + j4) = ra[i];
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) *reinterpret_cast<float4*>(b + ((tid / KU) + i * (NT / KU)) * RS + j4) = rb[i];
  };

  f32x16 acc[MT][NTL];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int KT = Kp / KB;
  gload(0);
  lstore(0);
  __syncthreads();
  const int frag_off = (lane & 31) * RS + 4 * (lane >> 5);
#ifdef PV_DBG_W4_OLDLOOP         // developer A/B builds: the loop as it was everywhere (fragments read right before their MFMAs)
  constexpr bool FRAG_PF = false;
#else
  constexpr bool FRAG_PF = KB == 16;   // measured: +3 ... +8 % with 16-deep stages, -5 % with 32-deep ones (which keep the old loop)
#endif
  if constexpr (!FRAG_PF) {
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) gload(kt + 1);
    const float* a = &lds[buf][wm0 * RS + frag_off];
    const float* b = &lds[buf][(BM + wn0) * RS + frag_off];
#pragma unroll
    for (int h = 0; h < KB / 8; ++h) {
      float4 af[MT], bf[NTL];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) af[mi] = *reinterpret_cast<const float4*>(a + mi * 32 * RS + h * 8);
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni) bf[ni] = *reinterpret_cast<const float4*>(b + ni * 32 * RS + h * 8);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTL; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].x, bf[ni].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].y, bf[ni].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].z, bf[ni].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].w, bf[ni].w, acc[mi][ni], 0, 0, 0);
        }
    }
    if (kt + 1 < KT) lstore(buf ^ 1);
    __syncthreads();
  }
  } else {
  // The loop of conv_igemm_f32.hip (round 3): two fragment sets (the next 8-deep group's LDS reads in flight under the current
  // group's MFMAs; the first group of the next stage requested right behind the barrier) and a scheduling fence behind the global
  // requests.  Same products in the same order.
  constexpr int H = KB / 8;
  static_assert(H % 2 == 0, "the two fragment sets alternate per 8-deep group");
  float4 af[2][MT], bf[2][NTL];
  auto ldfrag = [&](const int set, const int buf, const int h) {
    const float* a = &lds[buf][wm0 * RS + frag_off] + h * 8;
    const float* b = &lds[buf][(BM + wn0) * RS + frag_off] + h * 8;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) af[set][mi] = *reinterpret_cast<const float4*>(a + mi * 32 * RS);
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni) bf[set][ni] = *reinterpret_cast<const float4*>(b + ni * 32 * RS);
  };
  auto mfma_rows = [&](const int set, const int mi0, const int mi1) {
#pragma unroll
    for (int mi = mi0; mi < mi1; ++mi)
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].x, bf[set][ni].x, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].y, bf[set][ni].y, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].z, bf[set][ni].z, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].w, bf[set][ni].w, acc[mi][ni], 0, 0, 0);
      }
  };
  ldfrag(0, 0, 0);
  for (int kt = 0; kt + 1 < KT; ++kt) {
    const int buf = kt & 1;
    gload(kt + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h + 1 < H; ++h) {
      ldfrag((h + 1) & 1, buf, h + 1);
      mfma_rows(h & 1, 0, MT);
    }
    mfma_rows((H - 1) & 1, 0, MT - 1);
    lstore(buf ^ 1);
    __syncthreads();
    ldfrag(0, buf ^ 1, 0);
    mfma_rows((H - 1) & 1, MT - 1, MT);
  }
  {
    const int buf = (KT - 1) & 1;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      if (h + 1 < H) ldfrag((h + 1) & 1, buf, h + 1);
      mfma_rows(h & 1, 0, MT);
    }
    __syncthreads();
  }
  }

  // raw tile -> Ms[comp][tile][ncols], staged through the (now idle) operand LDS so that every lane stores 16 bytes
  const int ncols = n_tiles * BN;
  float* dst = Ms + (long)comp * Mt * ncols;
  constexpr int EP = BN + 4;
  static_assert(WTM * EP <= 2 * BUF, "the staged wave row must fit the operand buffers");
  float* stg = lds_dyn;
#pragma unroll 1
  for (int wr = 0; wr < WM; ++wr) {
    if (wave / WN == wr) {
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            stg[row * EP + wn0 + ni * 32 + (lane & 31)] = acc[mi][ni][r];
          }
    }
    __syncthreads();
    constexpr int C4 = BN / 4, UNITS = WTM * C4;
#pragma unroll
    for (int u = tid; u < UNITS; u += NT) {
      const int row = u / C4, c4 = u - row * C4;
      const int m = m0 + wr * WTM + row;
      if (m < Mt) *reinterpret_cast<float4*>(dst + (long)m * ncols + n0 + c4 * 4) = *reinterpret_cast<const float4*>(&stg[row * EP + c4 * 4]);
    }
    if (wr + 1 < WM) __syncthreads();
  }
}

// s = A^T m for six values: A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void at6(const float4& m0, const float4& m1, const float4& m2, const float4& m3, const float4& m4,
                                    const float4& m5, float4 (&s)[4]) {
  const float4 p12 = m1 + m2, d12 = m1 - m2, p34 = m3 + m4, d34 = m3 - m4;
  s[0] = m0 + p12 + p34;
  s[1] = d12 + 2.f * d34;
  s[2] = p12 + 4.f * p34;
  s[3] = d12 + 8.f * d34 + m5;
}

// Y = A^T M A per (tile, 4 couts) + bias + residual + activation -> the tile's 4x4 pixels.
template <bool WIDE>
__global__ __launch_bounds__(256) void wino4_output_kernel(const premvos_conv_desc p, const float* __restrict__ Ms, const int tiles_y,
                                                           const int tiles_x, const int ncols) {
  const int d = p.dh, tpp = tiles_y * tiles_x, tpi = d * d * tpp;
  const int Mt = p.n * tpi, c4 = p.cout / 4;
  const long total = (long)Mt * c4, cstride = (long)Mt * ncols;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int m = (int)(idx / c4), col = (int)(idx - (long)m * c4) * 4;
    const float* q = Ms + (long)m * ncols + col;
    float4 s[6][4];                                             // s[j][a] = (A^T M)[a][j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float4 mm[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) mm[i] = *reinterpret_cast<const float4*>(q + (long)(6 * i + j) * cstride);
      at6(mm[0], mm[1], mm[2], mm[3], mm[4], mm[5], s[j]);
    }
    const int n = m / tpi, rem = m - n * tpi;
    const int ph = rem / tpp, tt = rem - ph * tpp;
    const int ty = tt / tiles_x, tx = tt - ty * tiles_x;
    const float4 bv = p.bias != nullptr ? *reinterpret_cast<const float4*>(p.bias + col) : f4(0.f);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float4 y[4];
      at6(s[0][a], s[1][a], s[2][a], s[3][a], s[4][a], s[5][a], y);
      const int oy = (4 * ty + a) * d + ph / d;
      if (oy >= p.ho) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ox = (4 * tx + b) * d + ph % d;
        if (ox >= p.wo) continue;
        const long pix = ((long)n * p.ho + oy) * p.wo + ox;
        float4 v = y[b] + bv;
        if (p.res != nullptr) {
          if constexpr (WIDE) {
            v = v + *reinterpret_cast<const float4*>(p.res + pix * p.res_ps + col);
          } else {
            const float* r = p.res + pix * p.res_ps + col;
            v = v + make_float4(r[0], r[1], r[2], r[3]);
          }
        }
        const int act = p.act & 0xff;
        if (act == PREMVOS_ACT_RELU) {
          v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        } else if (act == PREMVOS_ACT_LEAKY) {
          v = make_float4(v.x > 0.f ? v.x : v.x * p.slope, v.y > 0.f ? v.y : v.y * p.slope, v.z > 0.f ? v.z : v.z * p.slope,
                          v.w > 0.f ? v.w : v.w * p.slope);
        } else if (act == PREMVOS_ACT_SIGMOID) {
          v = make_float4(1.f / (1.f + expf(-v.x)), 1.f / (1.f + expf(-v.y)), 1.f / (1.f + expf(-v.z)), 1.f / (1.f + expf(-v.w)));
        }
        float* o = p.out + pix * p.out_ps + col;
        if constexpr (WIDE) {
          *reinterpret_cast<float4*>(o) = v;
        } else {
          o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        }
      }
    }
  }
}

// columns of a GEMM workgroup: 32 for the 32-cout estimator layers of PWC-Net (PWCNet.py:96-101 conv*_4; needs 32-deep stages: a
// 32 x 16 weight stage is half a request per thread), else 64 / 128
inline int wino4_bn(const premvos_conv_desc& d) {
  const int kp = premvos::cdiv((int)d.cin_pad, 16) * 16;
  if (d.cout <= 32 && kp % 32 == 0 && (d.stage_k & 16) == 0) return 32;
  return d.cout <= 64 ? 64 : 128;
}

struct Geo {
  int ty, tx, kp, bn, n_tiles;
  long mt;
};

inline Geo geometry(const premvos_conv_desc& d) {
  Geo g;
  g.ty = (premvos::cdiv((int)d.ho, (int)d.dh) + 3) / 4;      // per phase of an atrous layer (dilation 1: one phase)
  g.tx = (premvos::cdiv((int)d.wo, (int)d.dw) + 3) / 4;
  g.mt = (long)d.n * d.dh * d.dw * g.ty * g.tx;
  g.kp = premvos::cdiv((int)d.cin_pad, 16) * 16;
  g.bn = wino4_bn(d);
  g.n_tiles = premvos::cdiv(d.cout, g.bn);
  return g;
}

template <int BM, int BN, int WM, int WN, int KB>
int launch_gemm(const premvos_conv_desc& d, const Geo& g, const float* V, const int Vp, const int v_c0, float* Ms, hipStream_t s) {
  const int m_tiles = premvos::cdiv((int)g.mt, BM);
  constexpr int LDS_BYTES = 2 * (BM + BN) * (KB + 4) * (int)sizeof(float);
  static const bool attr_done = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_gemm_kernel<BM, BN, WM, WN, KB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    return true;
  }();
  (void)attr_done;
  hipLaunchKernelGGL((wino4_gemm_kernel<BM, BN, WM, WN, KB>), dim3(36 * m_tiles * g.n_tiles), dim3(256), LDS_BYTES, s, V, d.wgt_wino4, Ms,
                     (int)g.mt, g.kp, (int)d.cout_pad, m_tiles, g.n_tiles, Vp, v_c0);
  return premvos::check_launch("wino4_gemm");
}

}  // namespace

namespace premvos {

bool conv_wino4_applicable(const premvos_conv_desc& d) {
  return d.wgt_wino4 != nullptr && d.kh == 3 && d.kw == 3 && d.sh == 1 && d.sw == 1 && d.dh == d.dw && d.dh >= 1 && d.dh <= 64 &&
         d.out_mode == PREMVOS_OUT_NHWC && d.precision == PREMVOS_PREC_F32 && d.cout % 4 == 0 && d.k_pad >= d.cin_pad &&
         d.ho == d.h + 2 * d.pt - 2 * d.dh && d.wo == d.w + 2 * d.pl - 2 * d.dw && d.pt >= 0 && d.pl >= 0;
}

long conv_wino4_workspace_bytes(const premvos_conv_desc& d) {
  const Geo g = geometry(d);
  return 36L * g.mt * ((long)g.kp + (long)g.n_tiles * g.bn) * (long)sizeof(float);
}

long conv_wino4_m_bytes(const premvos_conv_desc& d) {
  const Geo g = geometry(d);
  return 36L * g.mt * (long)g.n_tiles * g.bn * (long)sizeof(float);
}

// The three launches.  V: slab of pitch Vp with the layer's window at channel v_c0; channels [0, t_cn) of the window are
// transformed now (the rest of [0, kp) is already there); Ms: 36 x tiles x (n_tiles x bn) floats.
static int wino4_run(const premvos_conv_desc& d, const Geo& g, float* V, const int Vp, const int v_c0, const int t_cn, float* Ms,
                     hipStream_t s) {
  if (g.mt >= (1L << 26)) return fail(PREMVOS_EINVAL, "conv2d(winograd 4x4): too many tiles");
  if (d.stage_k & ~(64 | 16)) return fail(PREMVOS_EINVAL, "conv2d(winograd 4x4): stage_k %d is not a block id (0, 16, 64, 80)", d.stage_k);
  if (t_cn > 0) {
    const long total = g.mt * (t_cn / 4);
    const int grid = (int)(total / 256 < 1 ? 1 : total / 256 > (1 << 20) ? (1 << 20) : (total + 255) / 256);
    hipLaunchKernelGGL(wino4_input_kernel, dim3(grid), dim3(256), 0, s, d, V, g.ty, g.tx, Vp, v_c0, t_cn);
    const int rc = check_launch("wino4_input");
    if (rc) return rc;
  }
  // stage_k: bit 6 (64) = 64 instead of 128 tile rows per workgroup, bit 4 (16) = 16- instead of 32-deep stages (three
  // instead of two workgroups per CU: the better trade for short K, where a tile's first loads and epilogue weigh most)
  const bool bm64 = (d.stage_k & 64) != 0, k32 = g.kp % 32 == 0 && (d.stage_k & 16) == 0;
  int rc;
  if (g.bn == 32) {
    rc = launch_gemm<128, 32, 4, 1, 32>(d, g, V, Vp, v_c0, Ms, s);      // (four waves of 32 x 32: the only block of this width)
  } else if (g.bn == 64) {
    rc = bm64 ? (k32 ? launch_gemm<64, 64, 2, 2, 32>(d, g, V, Vp, v_c0, Ms, s) : launch_gemm<64, 64, 2, 2, 16>(d, g, V, Vp, v_c0, Ms, s))
              : (k32 ? launch_gemm<128, 64, 2, 2, 32>(d, g, V, Vp, v_c0, Ms, s) : launch_gemm<128, 64, 2, 2, 16>(d, g, V, Vp, v_c0, Ms, s));
  } else {
    rc = bm64 ? (k32 ? launch_gemm<64, 128, 2, 2, 32>(d, g, V, Vp, v_c0, Ms, s) : launch_gemm<64, 128, 2, 2, 16>(d, g, V, Vp, v_c0, Ms, s))
              : (k32 ? launch_gemm<128, 128, 2, 2, 32>(d, g, V, Vp, v_c0, Ms, s) : launch_gemm<128, 128, 2, 2, 16>(d, g, V, Vp, v_c0, Ms, s));
  }
  if (rc) return rc;
  const long total = g.mt * (d.cout / 4);
  const int grid = (int)(total / 256 > (1 << 20) ? (1 << 20) : (total + 255) / 256);
  const bool wide = (d.out_ps & 3) == 0 && aligned16(d.out) && (d.res == nullptr || ((d.res_ps & 3) == 0 && aligned16(d.res))) &&
                    (d.bias == nullptr || aligned16(d.bias));
  if (wide) {
    hipLaunchKernelGGL(wino4_output_kernel<true>, dim3(grid), dim3(256), 0, s, d, Ms, g.ty, g.tx, g.n_tiles * g.bn);
  } else {
    hipLaunchKernelGGL(wino4_output_kernel<false>, dim3(grid), dim3(256), 0, s, d, Ms, g.ty, g.tx, g.n_tiles * g.bn);
  }
  return check_launch("wino4_output");
}

// tile_hint 4; stage_k picks the GEMM block: 0 / 64 / 16 / 80.  Self-contained form: V and M slabs in the layer's workspace.
int conv_wino4(const premvos_conv_desc& d, hipStream_t s) {
  const Geo g = geometry(d);
  if (d.workspace == nullptr || d.workspace_bytes < conv_wino4_workspace_bytes(d))
    return fail(PREMVOS_EINVAL, "conv2d(winograd 4x4): needs %ld workspace bytes", conv_wino4_workspace_bytes(d));
  return wino4_run(d, g, d.workspace, g.kp, 0, g.kp, d.workspace + 36L * g.mt * g.kp, s);
}

}  // namespace premvos

// Kept-slab form for DenseNet blocks (PWCNet.py:201-264: layer i reads the concat of everything layers 0 ... i-1 produced, so
// the self-contained form re-transforms the same channels up to four times).  The caller owns one V slab per concat buffer:
// V[36][tiles][v_pitch], slab channel = channel of the concat buffer.  This layer's input window starts at slab channel v_c0;
// channels [0, t_cn) of the window are transformed by this call (0 = everything it reads is already in the slab), the GEMM reads
// [v_c0, v_c0 + Kp) with Kp = cin_pad rounded up to 16 -- the caller keeps [cin_pad, Kp) zero by transforming them once
// (t_cn = Kp on the first layer of a level zero-fills them: every window of a level ends at the same channel).  Same V values,
// same GEMM, same output transform as premvos_conv2d_f32 with tile_hint 4: bit-identical results.  d->workspace holds the M slab.
extern "C" int premvos_conv_wino4_slab_f32(const premvos_conv_desc* dp, float* vslab, int64_t vslab_bytes, int32_t v_pitch, int32_t v_c0,
                                          int32_t t_cn, void* stream) {
  using namespace premvos;
  if (dp == nullptr || vslab == nullptr) return fail(PREMVOS_EINVAL, "conv_wino4_slab: null argument");
  const premvos_conv_desc& d = *dp;
  if (const int rc = conv_desc_check(d)) return rc;
  if (!conv_wino4_applicable(d)) return fail(PREMVOS_EINVAL, "conv_wino4_slab: the layer is not a 3x3 / stride-1 fp32 layer with F(4x4) weights");
  const Geo g = geometry(d);
  if ((v_pitch & 3) || (v_c0 & 3) || (t_cn & 3) || v_c0 < 0 || t_cn < 0 || t_cn > g.kp || v_c0 + g.kp > v_pitch || !aligned16(vslab))
    return fail(PREMVOS_EINVAL, "conv_wino4_slab: window [%d, +%d) / transform count %d do not fit a slab of pitch %d (multiples of 4)", v_c0,
                g.kp, t_cn, v_pitch);
  if (vslab_bytes < 36L * g.mt * v_pitch * (long)sizeof(float))
    return fail(PREMVOS_EINVAL, "conv_wino4_slab: slab of %ld bytes, needs %ld", (long)vslab_bytes, 36L * g.mt * v_pitch * (long)sizeof(float));
  if (d.workspace == nullptr || d.workspace_bytes < conv_wino4_m_bytes(d))
    return fail(PREMVOS_EINVAL, "conv_wino4_slab: needs %ld workspace bytes for the M slab", conv_wino4_m_bytes(d));
  return wino4_run(d, g, vslab, v_pitch, v_c0, t_cn, d.workspace, static_cast<hipStream_t>(stream));
}
