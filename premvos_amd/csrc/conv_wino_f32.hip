// 3x3 / stride-1 dense convolution by Winograd F(2x2, 3x3) on the fp32 matrix pipe: 2.25x fewer multiplies than the implicit
// GEMM of conv_igemm_f32.hip for the layers where that kernel is contraction-bound (ResNet conv2 of every bottleneck, the RPN
// 3x3, PWC-Net's DenseNet estimators and dc_conv1).  Same data path as the GEMM kernel (NHWC input slice with a pixel stride,
// LDS-staged operands, v_mfma_f32_32x32x2_f32, XCD-aware tile order), different algebra:
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// * the filter transform U = G g G^T is done once by the host (ops.pack_conv: 16 matrices [cout_pad][k_pad], k = cin);
// * the INPUT transform is fused into the A-operand staging: every row of B^T has exactly two non-zeros (+-1), so component
//   (xi, nu) of a patch is +-d[i1][j1] +- d[i1][j2] +- d[i2][j1] +- d[i2][j2] -- four 16-byte loads and three adds per staged
//   float4 instead of one load; no transformed-input tensor ever exists in HBM;
// * the 16 components are 16 independent GEMMs M_c[tile][cout] = V_c[tile][cin] * U_c[cout][cin]; a workgroup computes one
//   128x128 tile of ONE component and writes it raw to the workspace slab of that component (the split-K slab layout);
//   consecutive workgroups of an XCD walk the 16 components of the same (tile rows, cout columns) block, so the 4x4 patches
//   they all gather stay in that XCD's L2;
// * wino_output_kernel applies A^T . A, bias (+ folded BatchNorm), residual, activation and writes the 2x2 pixels.
//
// fp32 throughout (exact products on the MFMA pipe); against the direct convolution the result differs by the usual Winograd
// rounding (~1e-6 relative: sums of up to four inputs / nine weights are formed before the multiply).
// Reference call sites replaced: the same framework conv calls as premvos_conv2d_f32 (include/premvos_hip.h).
#include "common.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int KB = 16, RS = KB + 4, KU = KB / 4;

// row r of B^T = +d[I1[r]] + S2[r] * d[I2[r]]
__device__ __constant__ int W_I1[4] = {0, 1, 2, 1};
__device__ __constant__ int W_I2[4] = {2, 2, 1, 3};
__device__ __constant__ float W_S2[4] = {-1.f, 1.f, -1.f, -1.f};

// per-component select (a float4 ?: compiles to a scratch-memory select on this toolchain)
__device__ __forceinline__ float4 sel4(bool c, const float4& v) {
  return make_float4(c ? v.x : 0.f, c ? v.y : 0.f, c ? v.z : 0.f, c ? v.w : 0.f);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 3) void wino_gemm_kernel(const premvos_conv_desc p, const float* __restrict__ wgt_wino,
                                                        float* __restrict__ ws, const int tiles_y, const int tiles_x,
                                                        const int m_tiles, const int n_tiles) {
  constexpr int NT = 256;
  static_assert(WM * WN == 4, "four waves per workgroup");
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NTL = WTN / 32;
  constexpr int A_PER_T = BM * KU / NT, B_PER_T = (BN * KU + NT - 1) / NT;
  constexpr int BUF = (BM + BN) * RS;
  static_assert(BM * KU % NT == 0, "staging must divide evenly");
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  float(*lds)[BUF] = reinterpret_cast<float(*)[BUF]>(lds_dyn);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * WTM, wn0 = (wave % WN) * WTN;
  const int Mt = p.n * tiles_y * tiles_x;
  int tile_m, tile_n, comp;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3, q = nwg >> 3, r = nwg & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    comp = v & 15;
    const int rest = v >> 4;
    tile_n = rest % n_tiles;
    tile_m = rest / n_tiles;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int xi = comp >> 2, nu = comp & 3;
  const int i1 = W_I1[xi], i2 = W_I2[xi], j1 = W_I1[nu], j2 = W_I2[nu];
  const float si = W_S2[xi], sj = W_S2[nu];

  const int j4 = (tid % KU) * 4;
  const float* src[A_PER_T][4];
  bool ok[A_PER_T][4];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int row = (tid / KU) + i * (NT / KU);
    const int m = m0 + row;
    const bool rok = m < Mt;
    const int mm = rok ? m : 0;
    const int tpi = tiles_y * tiles_x;
    const int n = mm / tpi, rem = mm - n * tpi;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int y0 = 2 * ty - p.pt, x0 = 2 * tx - p.pl;
    const float* img = p.in + (long)n * p.h * p.w * p.in_ps;
    const int ys[2] = {y0 + i1, y0 + i2}, xs[2] = {x0 + j1, x0 + j2};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bool in = rok && (unsigned)ys[a] < (unsigned)p.h && (unsigned)xs[b] < (unsigned)p.w;
        ok[i][a * 2 + b] = in;
        src[i][a * 2 + b] = img + (in ? ((long)ys[a] * p.w + xs[b]) * p.in_ps : 0);
      }
  }
  const float* wrow[B_PER_T];
  bool wok[B_PER_T];
#pragma unroll
  for (int i = 0; i < B_PER_T; ++i) {
    const int row = (tid / KU) + i * (NT / KU);
    wok[i] = row < BN && n0 + row < p.cout_pad;
    wrow[i] = wgt_wino + ((long)comp * p.cout_pad + (wok[i] ? n0 + row : 0)) * p.k_pad + j4;
  }

  float4 ra[A_PER_T][4], rb[B_PER_T];
  auto gload = [&](int kt) {
    const int c = kt * KB + j4;
    const bool kok = c < p.cin_pad;
    const int cc = kok ? c : 0;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) ra[i][t] = *reinterpret_cast<const float4*>(src[i][t] + cc);
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) rb[i] = *reinterpret_cast<const float4*>(wrow[i] + kt * KB);
  };
  auto lstore = [&](int buf, int kt) {
    const bool kok = kt * KB + j4 < p.cin_pad;
    float* a = &lds[buf][0];
    float* b = &lds[buf][BM * RS];
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int row = (tid / KU) + i * (NT / KU);
      const float4 d11 = sel4(kok && ok[i][0], ra[i][0]), d12 = sel4(kok && ok[i][1], ra[i][1]);
      const float4 d21 = sel4(kok && ok[i][2], ra[i][2]), d22 = sel4(kok && ok[i][3], ra[i][3]);
      float4 v;      // (d[i1][j1] + sj d[i1][j2]) + si (d[i2][j1] + sj d[i2][j2])
      v.x = (d11.x + sj * d12.x) + si * (d21.x + sj * d22.x);
      v.y = (d11.y + sj * d12.y) + si * (d21.y + sj * d22.y);
      v.z = (d11.z + sj * d12.z) + si * (d21.z + sj * d22.z);
      v.w = (d11.w + sj * d12.w) + si * (d21.w + sj * d22.w);
      *reinterpret_cast<float4*>(a + row * RS + j4) = v;
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      const int row = (tid / KU) + i * (NT / KU);
      if (BN * KU % NT == 0 || row < BN) *reinterpret_cast<float4*>(b + row * RS + j4) = sel4(wok[i] && kt * KB + j4 < p.k_pad, rb[i]);
    }
  };

  f32x16 acc[MT][NTL];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int KT = (p.k_pad + KB - 1) / KB;
  gload(0);
  lstore(0, 0);
  __syncthreads();
  const int frag_off = (lane & 31) * RS + 4 * (lane >> 5);
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
    if (kt + 1 < KT) lstore(buf ^ 1, kt + 1);
    __syncthreads();
  }

  // raw component tile -> ws[comp][tile][ncols]
  const int ncols = n_tiles * BN;
  float* dst = ws + (long)comp * Mt * ncols;
#pragma unroll
  for (int ni = 0; ni < NTL; ++ni) {
    const int col = n0 + wn0 + ni * 32 + (lane & 31);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < Mt) dst[(long)m * ncols + col] = acc[mi][ni][r];
      }
  }
}

// A^T of F(2x2,3x3): output row a of a tile takes component row xi with weight W_AT[a][xi]
__device__ __constant__ float W_AT[2][4] = {{1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, -1.f}};

// The slab-free variant: one workgroup owns a (BM tiles x BN couts) block for ALL 16 components.  It walks the components one
// after the other through the same double-buffered K loop (16 * KT stages, the prefetch runs across component boundaries), keeps
// the raw product of the current component in one set of MFMA accumulators and, at the end of each component, adds it with its
// A^T (x) A^T weight (0 / +1 / -1) into the four output-pixel accumulators of the 2x2 tile.  Bias, residual and activation are
// applied from registers and the 2x2 pixels are written once: no workspace, no second kernel, no 16x slab round trip.  The price
// is registers (4 + 1 accumulator sets), so the per-wave tile is 64x32 / 32x64 and two waves share a SIMD.
template <int BM, int BN, int WM, int WN, int KB>
__global__ __launch_bounds__(64 * WM * WN, 2) void wino_fused_kernel(const premvos_conv_desc p, const float* __restrict__ wgt_wino,
                                                                     const int tiles_y, const int tiles_x, const int m_tiles,
                                                                     const int n_tiles) {
  constexpr int NT = 64 * WM * WN, RS = KB + 4, KU = KB / 4;      // (shadow the file-scope 16-deep stage constants)
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NTL = WTN / 32;
  constexpr int A_PER_T = BM * KU / NT, B_PER_T = (BN * KU + NT - 1) / NT;
  constexpr int BUF = (BM + BN) * RS;
  static_assert(BM * KU % NT == 0 && A_PER_T >= 1, "staging must divide evenly");
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  float(*lds)[BUF] = reinterpret_cast<float(*)[BUF]>(lds_dyn);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * WTM, wn0 = (wave % WN) * WTN;
  const int tpi = tiles_y * tiles_x;
  const int Mt = p.n * tpi;
  const int dil = p.dh;                                     // == p.dw (checked by the launcher)
  int tile_m, tile_n;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3, q = nwg >> 3, r = nwg & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    tile_n = v % n_tiles;
    tile_m = v / n_tiles;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int j4 = (tid % KU) * 4;

  // per staged row: element offset of the 4x4 patch origin and which of its rows / columns lie inside the image
  long boff[A_PER_T];
  unsigned vy[A_PER_T], vx[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int m = m0 + (tid / KU) + i * (NT / KU);
    const bool rok = m < Mt;
    const int mm = rok ? m : 0;
    const int n = mm / tpi, rem = mm - n * tpi;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    // atrous layers (dilation d, stride 1): the conv is an ordinary 3x3 conv on each of the d x d sub-lattices of the map;
    // tile row ty addresses (residue ty % d, lattice tile ty / d), its outputs and its 4x4 patch are d pixels apart
    const int y0 = (ty % dil) + 2 * (ty / dil) * dil - p.pt, x0 = (tx % dil) + 2 * (tx / dil) * dil - p.pl;
    boff[i] = (((long)n * p.h + y0) * p.w + x0) * p.in_ps;
    vy[i] = vx[i] = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      vy[i] |= (rok && (unsigned)(y0 + t * dil) < (unsigned)p.h) ? 1u << t : 0u;
      vx[i] |= ((unsigned)(x0 + t * dil) < (unsigned)p.w) ? 1u << t : 0u;
    }
  }
  unsigned wbase[B_PER_T];                   // element offsets (the launcher refuses tensors past 2^30 elements)
  bool wok[B_PER_T];
#pragma unroll
  for (int i = 0; i < B_PER_T; ++i) {
    const int row = (tid / KU) + i * (NT / KU);
    wok[i] = row < BN && n0 + row < p.cout_pad;
    wbase[i] = (unsigned)(wok[i] ? n0 + row : 0) * p.k_pad + j4;
  }
  const unsigned wcomp = (unsigned)p.cout_pad * p.k_pad;

  // the component being staged
  unsigned src[A_PER_T][4];
  unsigned okm[A_PER_T];
  unsigned wrow[B_PER_T];
  float si = 0.f, sj = 0.f;
  auto set_comp = [&](int comp) {
    const int xi = comp >> 2, nu = comp & 3;
    const int ia[2] = {W_I1[xi], W_I2[xi]}, jb[2] = {W_I1[nu], W_I2[nu]};
    si = W_S2[xi];
    sj = W_S2[nu];
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      okm[i] = 0;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const bool in = ((vy[i] >> ia[a]) & 1u) && ((vx[i] >> jb[b]) & 1u);
          okm[i] |= in ? 1u << (a * 2 + b) : 0u;
          src[i][a * 2 + b] = in ? (unsigned)(boff[i] + ((long)ia[a] * dil * p.w + jb[b] * dil) * p.in_ps) : 0u;
        }
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) wrow[i] = wbase[i] + comp * wcomp;
  };

  float4 ra[A_PER_T][4], rb[B_PER_T];
  auto gload = [&](int kt) {
    const int c = kt * KB + j4;
    const int cc = c < p.cin_pad ? c : 0;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) ra[i][t] = *reinterpret_cast<const float4*>(p.in + (size_t)(src[i][t] + (unsigned)cc));
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i)
      rb[i] = *reinterpret_cast<const float4*>(wgt_wino + (size_t)(wrow[i] + (unsigned)(c < p.k_pad ? kt * KB : 0)));
  };
  auto lstore = [&](int buf, int kt) {
    const bool kok = kt * KB + j4 < p.cin_pad;
    float* a = &lds[buf][0];
    float* b = &lds[buf][BM * RS];
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int row = (tid / KU) + i * (NT / KU);
      const float4 d11 = sel4(kok && (okm[i] & 1u), ra[i][0]), d12 = sel4(kok && (okm[i] & 2u), ra[i][1]);
      const float4 d21 = sel4(kok && (okm[i] & 4u), ra[i][2]), d22 = sel4(kok && (okm[i] & 8u), ra[i][3]);
      float4 v;
      v.x = (d11.x + sj * d12.x) + si * (d21.x + sj * d22.x);
      v.y = (d11.y + sj * d12.y) + si * (d21.y + sj * d22.y);
      v.z = (d11.z + sj * d12.z) + si * (d21.z + sj * d22.z);
      v.w = (d11.w + sj * d12.w) + si * (d21.w + sj * d22.w);
      *reinterpret_cast<float4*>(a + row * RS + j4) = v;
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      const int row = (tid / KU) + i * (NT / KU);
      if (BN * KU % NT == 0 || row < BN) *reinterpret_cast<float4*>(b + row * RS + j4) = sel4(wok[i] && kt * KB + j4 < p.k_pad, rb[i]);
    }
  };

  f32x16 M[MT][NTL], Y[2][2][MT][NTL];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        M[mi][ni][r] = 0.f;
        Y[0][0][mi][ni][r] = Y[0][1][mi][ni][r] = Y[1][0][mi][ni][r] = Y[1][1][mi][ni][r] = 0.f;
      }

  const int KT = (p.k_pad + KB - 1) / KB;
  const int total = 16 * KT;
  set_comp(0);
  gload(0);
  lstore(0, 0);
  __syncthreads();
  const int frag_off = (lane & 31) * RS + 4 * (lane >> 5);
  int comp_c = 0, kt_c = 0, comp_n = 0, kt_n = 0;            // the stage being multiplied / the stage being staged
  for (int it = 0; it < total; ++it) {
    const int buf = it & 1;
    const bool has_next = it + 1 < total;
    if (has_next) {
      if (++kt_n == KT) {
        kt_n = 0;
        set_comp(++comp_n);
      }
      gload(kt_n);
    }
    const float* a = &lds[buf][wm0 * RS + frag_off];
    const float* b = &lds[buf][(BM + wn0) * RS + frag_off];
#pragma unroll
    for (int h = 0; h < KB / 8; ++h) {
      // the next stage goes to LDS half way through this one: its ds_writes retire under the remaining MFMAs
      if (h == KB / 16 && has_next) lstore(buf ^ 1, kt_n);
      float4 af[MT], bf[NTL];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) af[mi] = *reinterpret_cast<const float4*>(a + mi * 32 * RS + h * 8);
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni) bf[ni] = *reinterpret_cast<const float4*>(b + ni * 32 * RS + h * 8);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTL; ++ni) {
          M[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].x, bf[ni].x, M[mi][ni], 0, 0, 0);
          M[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].y, bf[ni].y, M[mi][ni], 0, 0, 0);
          M[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].z, bf[ni].z, M[mi][ni], 0, 0, 0);
          M[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].w, bf[ni].w, M[mi][ni], 0, 0, 0);
        }
    }
    if (++kt_c == KT) {                      // component finished: Y[a][b] += A^T[a][xi] * A^T[b][nu] * M
      const float ca[2] = {W_AT[0][comp_c >> 2], W_AT[1][comp_c >> 2]}, cb[2] = {W_AT[0][comp_c & 3], W_AT[1][comp_c & 3]};
#pragma unroll
      for (int ya = 0; ya < 2; ++ya)
#pragma unroll
        for (int yb = 0; yb < 2; ++yb) {
          const float c = ca[ya] * cb[yb];
          if (c != 0.f) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
              for (int ni = 0; ni < NTL; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[ya][yb][mi][ni][r] += c * M[mi][ni][r];
          }
        }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTL; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) M[mi][ni][r] = 0.f;
      kt_c = 0;
      ++comp_c;
    }
    __syncthreads();
  }

  // Wide epilogue (as in conv_igemm_f32.hip): one output position (ya, yb) of one wave row at a time goes through the idle operand
  // LDS, then every lane handles 16 bytes of a pixel's channel run: bias, residual (16-byte load), activation, 16-byte store.
  {
    const bool wide = (p.out_ps & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15u) == 0 &&
                      (p.res == nullptr || ((p.res_ps & 3) == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15u) == 0)) &&
                      (p.bias == nullptr || (reinterpret_cast<uintptr_t>(p.bias) & 15u) == 0);       // (cout % 4 == 0 is a precondition)
    constexpr int EP = BN + 4;
    static_assert(WTM * EP <= 2 * BUF, "the staged wave row must fit the operand buffers");
    if (wide) {                                                  // kernel-uniform
      float* stg = lds_dyn;
      // a thread's 16-byte units share their four columns when the thread count is a multiple of the units per row: ONE bias
      // request per kernel instead of one (waited for on the spot) per unit (conv_igemm_f32.hip, late round 3)
      constexpr bool BIAS_ONCE = NT % (BN / 4) == 0;
      float4 bias_once = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (BIAS_ONCE) {
        const int colb = n0 + (tid % (BN / 4)) * 4;
        if (p.bias != nullptr && colb < p.cout) bias_once = premvos::ld4(p.bias + colb);
      }
#pragma unroll 1
      for (int wr = 0; wr < WM; ++wr)
#pragma unroll
        for (int ya = 0; ya < 2; ++ya)
#pragma unroll
          for (int yb = 0; yb < 2; ++yb) {
            __syncthreads();                                       // the previous block has been read (first: the K loop is over)
            if (wave / WN == wr) {
#pragma unroll
              for (int ni = 0; ni < NTL; ++ni)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                  for (int r = 0; r < 16; ++r)
                    stg[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EP + wn0 + ni * 32 + (lane & 31)] = Y[ya][yb][mi][ni][r];
            }
            __syncthreads();
            constexpr int C4 = BN / 4, UNITS = WTM * C4;
#pragma unroll
            for (int u = tid; u < UNITS; u += NT) {
              const int row = u / C4, c4 = u - row * C4;
              const int m = m0 + wr * WTM + row, col = n0 + c4 * 4;
              if (m >= Mt || col >= p.cout) continue;
              const int n = m / tpi, rem = m - n * tpi;
              const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
              const int oy = (ty % dil) + (2 * (ty / dil) + ya) * dil, ox = (tx % dil) + (2 * (tx / dil) + yb) * dil;
              if (oy >= p.ho || ox >= p.wo) continue;
              const long pix = ((long)n * p.ho + oy) * p.wo + ox;
              float4 v = *reinterpret_cast<const float4*>(&stg[row * EP + c4 * 4]);
              if constexpr (BIAS_ONCE) {
                v.x += bias_once.x; v.y += bias_once.y; v.z += bias_once.z; v.w += bias_once.w;
              } else if (p.bias != nullptr) {
                const float4 bb = *reinterpret_cast<const float4*>(p.bias + col);
                v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
              }
              if (p.res != nullptr) {
                const float4 rv = *reinterpret_cast<const float4*>(p.res + pix * p.res_ps + col);
                v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
              }
              if (p.act == PREMVOS_ACT_RELU) {
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
              } else if (p.act == PREMVOS_ACT_LEAKY) {
                v.x = v.x > 0.f ? v.x : v.x * p.slope; v.y = v.y > 0.f ? v.y : v.y * p.slope;
                v.z = v.z > 0.f ? v.z : v.z * p.slope; v.w = v.w > 0.f ? v.w : v.w * p.slope;
              } else if (p.act == PREMVOS_ACT_SIGMOID) {
                v.x = 1.f / (1.f + expf(-v.x)); v.y = 1.f / (1.f + expf(-v.y)); v.z = 1.f / (1.f + expf(-v.z)); v.w = 1.f / (1.f + expf(-v.w));
              }
              *reinterpret_cast<float4*>(p.out + pix * p.out_ps + col) = v;
            }
          }
      return;
    }
  }
  // scalar form (unaligned channel windows): bias + residual + activation, the 2x2 pixels of every tile row this lane holds
  float bv[NTL];
  int cols[NTL];
#pragma unroll
  for (int ni = 0; ni < NTL; ++ni) {
    cols[ni] = n0 + wn0 + ni * 32 + (lane & 31);
    bv[ni] = (p.bias != nullptr && cols[ni] < p.cout) ? p.bias[cols[ni]] : 0.f;
  }
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m >= Mt) continue;
      const int n = m / tpi, rem = m - n * tpi;
      const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
#pragma unroll
      for (int ya = 0; ya < 2; ++ya)
#pragma unroll
        for (int yb = 0; yb < 2; ++yb) {
          const int oy = (ty % dil) + (2 * (ty / dil) + ya) * dil, ox = (tx % dil) + (2 * (tx / dil) + yb) * dil;
          if (oy >= p.ho || ox >= p.wo) continue;
          const long pix = ((long)n * p.ho + oy) * p.wo + ox;
#pragma unroll
          for (int ni = 0; ni < NTL; ++ni) {
            if (cols[ni] >= p.cout) continue;
            float v = Y[ya][yb][mi][ni][r] + bv[ni];
            if (p.res != nullptr) v += p.res[pix * p.res_ps + cols[ni]];
            if (p.act == PREMVOS_ACT_RELU) v = v > 0.f ? v : 0.f;
            else if (p.act == PREMVOS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
            else if (p.act == PREMVOS_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
            p.out[pix * p.out_ps + cols[ni]] = v;
          }
        }
    }
}

// Y = A^T M A per (tile, 4 couts), A^T = [[1,1,1,0],[0,1,-1,-1]], + bias + residual + activation -> the tile's 2x2 pixels.
__global__ __launch_bounds__(256) void wino_output_kernel(const premvos_conv_desc p, const float* __restrict__ ws,
                                                          const int tiles_y, const int tiles_x, const int ncols) {
  const int Mt = p.n * tiles_y * tiles_x, c4 = p.cout / 4;
  const long total = (long)Mt * c4;
  const long cstride = (long)Mt * ncols;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int m = idx / c4, col = (idx - (long)m * c4) * 4;
    const float* q = ws + (long)m * ncols + col;
    float4 s0[4], s1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 a = *reinterpret_cast<const float4*>(q + (0 * 4 + j) * cstride), b = *reinterpret_cast<const float4*>(q + (1 * 4 + j) * cstride);
      const float4 c = *reinterpret_cast<const float4*>(q + (2 * 4 + j) * cstride), d = *reinterpret_cast<const float4*>(q + (3 * 4 + j) * cstride);
      s0[j] = make_float4(a.x + b.x + c.x, a.y + b.y + c.y, a.z + b.z + c.z, a.w + b.w + c.w);
      s1[j] = make_float4(b.x - c.x - d.x, b.y - c.y - d.y, b.z - c.z - d.z, b.w - c.w - d.w);
    }
    float4 y[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float4* s = a ? s1 : s0;
      y[a][0] = make_float4(s[0].x + s[1].x + s[2].x, s[0].y + s[1].y + s[2].y, s[0].z + s[1].z + s[2].z, s[0].w + s[1].w + s[2].w);
      y[a][1] = make_float4(s[1].x - s[2].x - s[3].x, s[1].y - s[2].y - s[3].y, s[1].z - s[2].z - s[3].z, s[1].w - s[2].w - s[3].w);
    }
    const int tpi = tiles_y * tiles_x;
    const int n = m / tpi, rem = m - n * tpi;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const float4 bv = p.bias != nullptr ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int oy = 2 * ty + a, ox = 2 * tx + b;
        if (oy >= p.ho || ox >= p.wo) continue;
        const long pix = ((long)n * p.ho + oy) * p.wo + ox;
        float v[4] = {y[a][b].x + bv.x, y[a][b].y + bv.y, y[a][b].z + bv.z, y[a][b].w + bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (p.res != nullptr) v[e] += p.res[pix * p.res_ps + col + e];
          if (p.act == PREMVOS_ACT_RELU) v[e] = v[e] > 0.f ? v[e] : 0.f;
          else if (p.act == PREMVOS_ACT_LEAKY) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
          else if (p.act == PREMVOS_ACT_SIGMOID) v[e] = 1.f / (1.f + expf(-v[e]));
          p.out[pix * p.out_ps + col + e] = v[e];
        }
      }
  }
}

}  // namespace

namespace premvos {

// the slab-free kernel also takes atrous layers (dilation d, pad d: 'SAME'): Winograd on the d x d sub-lattices
bool conv_wino_fused_applicable(const premvos_conv_desc& d) {
  return d.wgt_wino != nullptr && d.kh == 3 && d.kw == 3 && d.sh == 1 && d.sw == 1 && d.dh == d.dw && d.dh >= 1 &&
         d.out_mode == PREMVOS_OUT_NHWC && d.precision == PREMVOS_PREC_F32 && d.cout % 4 == 0 && d.k_pad >= d.cin_pad &&
         d.ho == d.h + 2 * d.pt - 2 * d.dh && d.wo == d.w + 2 * d.pl - 2 * d.dw && d.pt >= 0 && d.pl >= 0 &&
         (d.dh == 1 || (d.pt == d.dh && d.pl == d.dw));
}

bool conv_wino_applicable(const premvos_conv_desc& d) {
  return d.wgt_wino != nullptr && d.kh == 3 && d.kw == 3 && d.sh == 1 && d.sw == 1 && d.dh == 1 && d.dw == 1 &&
         d.out_mode == PREMVOS_OUT_NHWC && d.precision == PREMVOS_PREC_F32 && d.cout % 4 == 0 && d.k_pad >= d.cin_pad &&
         d.ho == d.h + 2 * d.pt - 2 && d.wo == d.w + 2 * d.pl - 2 && d.pt >= 0 && d.pl >= 0;
}

static inline void wino_geometry(const premvos_conv_desc& d, int bn, int* ty, int* tx, long* mt, int* n_tiles) {
  *ty = (d.ho + 1) / 2;
  *tx = (d.wo + 1) / 2;
  *mt = (long)d.n * *ty * *tx;
  *n_tiles = cdiv(d.cout, bn);
}

static inline int wino_bn(const premvos_conv_desc& d) { return d.cout <= 32 ? 32 : d.cout <= 64 ? 64 : 128; }

long conv_wino_workspace_bytes(const premvos_conv_desc& d) {
  int ty, tx, nt;
  long mt;
  const int bn = wino_bn(d);
  wino_geometry(d, bn, &ty, &tx, &mt, &nt);
  return 16L * mt * nt * bn * (long)sizeof(float);
}

template <int BM, int BN, int WM, int WN>
static int launch_wino(const premvos_conv_desc& d, hipStream_t s) {
  int ty, tx, n_tiles;
  long mt;
  wino_geometry(d, BN, &ty, &tx, &mt, &n_tiles);
  premvos_conv_desc w = d;
  w.k_pad = cdiv((int)d.cin_pad, 16) * 16;              // the transformed filters are packed with k = cin only
  const int m_tiles = cdiv((int)mt, BM);
  constexpr int LDS_BYTES = 2 * (BM + BN) * RS * (int)sizeof(float);
  static const bool attr_done = [] {            // once per instantiation, thread-safe
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_gemm_kernel<BM, BN, WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS_BYTES);
    return true;
  }();
  (void)attr_done;
  hipLaunchKernelGGL((wino_gemm_kernel<BM, BN, WM, WN>), dim3(m_tiles * n_tiles * 16), dim3(256), LDS_BYTES, s, w, d.wgt_wino,
                     d.workspace, ty, tx, m_tiles, n_tiles);
  int rc = check_launch("wino_gemm");
  if (rc) return rc;
  const long total = mt * (d.cout / 4);
  int g = (int)((total + 255) / 256);
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(wino_output_kernel, dim3(g), dim3(256), 0, s, d, d.workspace, ty, tx, n_tiles * BN);
  return check_launch("wino_output");
}

template <int BM, int BN, int WM, int WN, int KB>
static int launch_wino_fused(const premvos_conv_desc& d, hipStream_t s) {
  // virtual tile rows / columns: the dil residues of a coordinate interleaved, ceil(ceil(extent / dil) / 2) lattice tiles each
  const int ty = d.dh * ((cdiv(d.ho, d.dh) + 1) / 2), tx = d.dw * ((cdiv(d.wo, d.dw) + 1) / 2);
  const long mt = (long)d.n * ty * tx;
  premvos_conv_desc w = d;
  w.k_pad = cdiv((int)d.cin_pad, 16) * 16;              // (a 32-deep stage zero-fills past it)
  const int m_tiles = cdiv((int)mt, BM), n_tiles = cdiv(d.cout, BN);
  constexpr int LDS_BYTES = 2 * (BM + BN) * (KB + 4) * (int)sizeof(float);
  static const bool attr_done = [] {            // once per instantiation, thread-safe
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fused_kernel<BM, BN, WM, WN, KB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS_BYTES);
    return true;
  }();
  (void)attr_done;
  hipLaunchKernelGGL((wino_fused_kernel<BM, BN, WM, WN, KB>), dim3(m_tiles * n_tiles), dim3(64 * WM * WN), LDS_BYTES, s, w, d.wgt_wino, ty, tx,
                     m_tiles, n_tiles);
  return check_launch("wino_fused");
}

// tile_hint 3: the slab-free kernel; stage_k picks the block (tile rows x couts / waves / stage depth): 0 = 128x128 / 8 / 16,
// 2 = 64x128 / 4 / 16, 3 = 64x64 / 4 / 16, 4 = 64x128 / 4 / 32, 5 = 64x64 / 4 / 32, 6 = 128x128 / 8 / 32 (measured and dropped:
// 128x64 / 4 / 16, 128x32 / 4 / 32, 128x64 / 8 / 32 -- never the fastest on the pipeline's layers)
int conv_wino_fused(const premvos_conv_desc& d, hipStream_t s) {
  if ((long)d.n * (d.dh * ((cdiv(d.ho, d.dh) + 1) / 2)) * (d.dw * ((cdiv(d.wo, d.dw) + 1) / 2)) >= (1L << 27))
    return fail(PREMVOS_EINVAL, "conv2d(winograd): too many tiles");
  if ((long)d.n * d.h * d.w * d.in_ps >= (1L << 30) || 16L * d.cout_pad * (cdiv((int)d.cin_pad, 16) * 16) >= (1L << 30))
    return fail(PREMVOS_EINVAL, "conv2d(winograd, fused): 32-bit operand offsets need tensors below 2^30 elements");
  switch (d.stage_k) {
    case 0: return launch_wino_fused<128, 128, 2, 4, 16>(d, s);
    case 2: return launch_wino_fused<64, 128, 2, 2, 16>(d, s);
    case 3: return launch_wino_fused<64, 64, 2, 2, 16>(d, s);
    case 4: return launch_wino_fused<64, 128, 2, 2, 32>(d, s);
    case 5: return launch_wino_fused<64, 64, 2, 2, 32>(d, s);
    case 6: return launch_wino_fused<128, 128, 2, 4, 32>(d, s);
    default: return fail(PREMVOS_EINVAL, "conv2d(winograd, fused): stage_k %d is not a block id (0, 2..6)", d.stage_k);
  }
}

int conv_wino(const premvos_conv_desc& d, hipStream_t s) {
  if (d.workspace == nullptr || d.workspace_bytes < conv_wino_workspace_bytes(d))
    return fail(PREMVOS_EINVAL, "conv2d(winograd): needs %ld workspace bytes", conv_wino_workspace_bytes(d));
  if ((long)d.n * ((d.ho + 1) / 2) * ((d.wo + 1) / 2) >= (1L << 27)) return fail(PREMVOS_EINVAL, "conv2d(winograd): too many tiles");
  // stage_k == 64 (set by the plan-time autotuner) selects 64-tile-row workgroups: twice as many, half as long -- the better
  // choice when the 128-row grid ends in a mostly empty last round of workgroups
  const bool bm64 = d.stage_k == 64;
  switch (wino_bn(d)) {
    case 32: return launch_wino<128, 32, 4, 1>(d, s);
    case 64: return bm64 ? launch_wino<64, 64, 2, 2>(d, s) : launch_wino<128, 64, 2, 2>(d, s);
    default: return bm64 ? launch_wino<64, 128, 2, 2>(d, s) : launch_wino<128, 128, 2, 2>(d, s);
  }
}

}  // namespace premvos
