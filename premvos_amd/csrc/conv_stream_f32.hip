// Pointwise (1x1, stride 1) fp32 conv with a SHORT K (cin = 64 or 128) and cout a multiple of 128: the HBM-bound layers of the
// three nets -- ResNet group0 / group1 1x1 convs (proposal_net/basemodel.py:49-59), the Xception entry flow's pointwise halves
// (refinement_net/network/deeplab/core/xception.py:154-178).  premvos_conv2d_f32 with tile_hint 5.
//
// Why not the implicit-GEMM kernel (conv_igemm_f32.hip): with K = 64 a tile is four 16-deep stages -- prologue, epilogue and
// their workgroup barriers are most of its life (24 % epilogue at K = 128, profiles/r03_k_sweep.txt) and the layers run at
// 3.3 ... 4.0 TB/s of algorithmic traffic.  Here the weight tile of a workgroup (128 couts x K) is read into LDS ONCE, the
// workgroup is persistent (one per CU, eight waves) and walks M; each wave owns 32 rows x 128 couts, reads its A fragments
// straight from global memory in MFMA order (no LDS, no sharing: a lane needs 16 bytes of its row per 8-deep group) and stages its
// own output through a wave-private LDS region -- after the weights are in, there is not a single workgroup barrier.
//
// Arithmetic: v_mfma_f32_32x32x2_f32, the products of an output element summed in the same order as in conv_igemm_f32.hip
// (8-deep groups in ascending k, steps x, y, z, w; bias, then residual, then activation): BIT-IDENTICAL results, so the choice
// between the two kernels is an order-neutral knob (premvos_amd/ops.py: numerics_key).
#include "common.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int NT = 512, WAVES = 8, BN = 128, ROWS = 32 * WAVES;      // rows of output pixels per workgroup step
constexpr int SC = 64 + 4;                                            // staged row pitch (floats): 32 rows x 64 columns per pass

template <int KG, int ACT, bool HAS_RES>     // 8-deep groups: K = 8 * KG (KG = 8 or 16); activation and residual compiled in (no branches)
__global__ __launch_bounds__(NT, 1) void conv_stream_f32_kernel(const premvos_conv_desc p, const int m_steps, const int n_tiles) {
  constexpr int K = 8 * KG, RS = K + 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* bt = lds;                                   // [BN][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* stg = lds + BN * RS + wave * (32 * SC);     // this wave's staging block
  const long M = (long)p.n * p.ho * p.wo;
  // the column tiles of one row block run on the SAME XCD (the dispatcher places workgroup b on XCD b % 8): they read the same A
  // rows at about the same time, so that block crosses the fabric once
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int tile_n = q % n_tiles, wg = xcd + 8 * (q / n_tiles), wgs = gridDim.x / n_tiles;
  const int n0 = tile_n * BN;

  // the weight tile, once
  for (int u = tid; u < BN * (K / 4); u += NT) {
    const int row = u / (K / 4), j = u - row * (K / 4);
    *reinterpret_cast<float4*>(bt + row * RS + j * 4) = premvos::ld4(p.wgt + (long)(n0 + row) * p.k_pad + j * 4);
  }
  __syncthreads();

  const float* bfrag = bt + (lane & 31) * RS + 4 * (lane >> 5);
  // a lane's eight 16-byte units of an epilogue pass share their four columns: the bias is read once per kernel
  float4 bias_v[2];
#pragma unroll
  for (int half = 0; half < 2; ++half)
    bias_v[half] = p.bias != nullptr ? premvos::ld4(p.bias + n0 + half * 64 + (lane & 15) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int arow = 32 * wave + (lane & 31);
  const int acol = 4 * (lane >> 5);
  auto a_ptr = [&](int step) {
    long m = (long)step * ROWS + arow;
    m = m < M ? m : M - 1;                             // rows past M: clamped, multiplied, never stored
    return p.in + m * p.in_ps + acol;
  };
  float4 a[KG];
  int step = wg;
  if (step < m_steps) {
    const float* ap = a_ptr(step);
#pragma unroll
    for (int g = 0; g < KG; ++g) a[g] = premvos::ld4(ap + 8 * g);
  }
  for (; step < m_steps; step += wgs) {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      float4 b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const float4*>(bfrag + j * 32 * RS + 8 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].x, b[j].x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].y, b[j].y, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].z, b[j].z, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g].w, b[j].w, acc[j], 0, 0, 0);
      }
    }
    // the next step's A fragments go out before this step's epilogue (their registers are free now)
    const int nstep = step + wgs;
    if (nstep < m_steps) {
      const float* ap = a_ptr(nstep);
#pragma unroll
      for (int g = 0; g < KG; ++g) a[g] = premvos::ld4(ap + 8 * g);
    }
    // epilogue, wave-private: two passes of 64 columns through this wave's LDS block, 16 bytes per lane out
    const long mrow0 = (long)step * ROWS + 32 * wave;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          stg[row * SC + jj * 32 + (lane & 31)] = acc[half * 2 + jj][r];
        }
      // (wave-private block: a wave's LDS instructions execute in order, so the reads below see the writes above without a
      //  barrier or a fence -- a fence would also wait for the global stores and the A prefetch; only the compiler is held)
      __builtin_amdgcn_wave_barrier();
      const int c4 = lane & 15, col = n0 + half * 64 + c4 * 4;
      float4 rv[8];
      if constexpr (HAS_RES) {                          // all eight residual requests go out before anything waits for one
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          long m = mrow0 + (lane >> 4) + 4 * i;
          m = m < M ? m : M - 1;
          rv[i] = premvos::ld4(p.res + m * p.res_ps + col);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = (lane >> 4) + 4 * i;
        const long m = mrow0 + row;
        float4 v = *reinterpret_cast<const float4*>(stg + row * SC + c4 * 4);
        v.x += bias_v[half].x; v.y += bias_v[half].y; v.z += bias_v[half].z; v.w += bias_v[half].w;
        if constexpr (HAS_RES) { v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w; }
        if constexpr (ACT == PREMVOS_ACT_RELU) {
          v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        } else if constexpr (ACT == PREMVOS_ACT_LEAKY) {
          v.x = v.x > 0.f ? v.x : v.x * p.slope; v.y = v.y > 0.f ? v.y : v.y * p.slope;
          v.z = v.z > 0.f ? v.z : v.z * p.slope; v.w = v.w > 0.f ? v.w : v.w * p.slope;
        }
        if (m < M) *reinterpret_cast<float4*>(p.out + m * p.out_ps + col) = v;
      }
      // (wave-private block: a wave's LDS instructions execute in order, so the reads below see the writes above without a
      //  barrier or a fence -- a fence would also wait for the global stores and the A prefetch; only the compiler is held)
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <int KG, int ACT, bool HAS_RES>
int launch(const premvos_conv_desc& d, hipStream_t s) {
  constexpr int K = 8 * KG;
  constexpr int LDS_BYTES = (BN * (K + 4) + WAVES * 32 * SC) * (int)sizeof(float);
  static const bool attr_done = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stream_f32_kernel<KG, ACT, HAS_RES>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    return true;
  }();
  (void)attr_done;
  const long M = (long)d.n * d.ho * d.wo;
  const int m_steps = (int)((M + ROWS - 1) / ROWS), n_tiles = d.cout / BN;
  // one persistent workgroup per CU, the column tiles side by side.  The kernel's (XCD, column tile, row walker) mapping is a
  // bijection only when the grid is a multiple of 8 * n_tiles: wgs is rounded DOWN to a multiple of 8 (cout = 384: 80 walkers x 3
  // column tiles = 240 workgroups; 85 x 3 would leave rows of the third column tile uncomputed)
  int wgs = (256 / n_tiles) & ~7;
  if (wgs < 8) wgs = 8;
  while (wgs > 8 && wgs - 8 >= m_steps) wgs -= 8;
  hipLaunchKernelGGL((conv_stream_f32_kernel<KG, ACT, HAS_RES>), dim3(wgs * n_tiles), dim3(NT), LDS_BYTES, s, d, m_steps, n_tiles);
  return premvos::check_launch("conv_stream_f32");
}

template <int KG>
int launch_kg(const premvos_conv_desc& d, hipStream_t s) {
  const bool r = d.res != nullptr;
  switch (d.act) {
    case PREMVOS_ACT_RELU: return r ? launch<KG, PREMVOS_ACT_RELU, true>(d, s) : launch<KG, PREMVOS_ACT_RELU, false>(d, s);
    case PREMVOS_ACT_LEAKY: return r ? launch<KG, PREMVOS_ACT_LEAKY, true>(d, s) : launch<KG, PREMVOS_ACT_LEAKY, false>(d, s);
    default: return r ? launch<KG, PREMVOS_ACT_NONE, true>(d, s) : launch<KG, PREMVOS_ACT_NONE, false>(d, s);
  }
}

}  // namespace

namespace premvos {

bool conv_stream_applicable(const premvos_conv_desc& d) {
  return d.precision == PREMVOS_PREC_F32 && d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.pt == 0 && d.pl == 0 &&
         d.ho == d.h && d.wo == d.w && d.out_mode == PREMVOS_OUT_NHWC && (d.k_pad == 64 || d.k_pad == 128) && d.cin_pad == d.k_pad &&
         d.cout % BN == 0 && d.cout <= 512 && (d.out_ps & 3) == 0 && (d.in_ps & 3) == 0 && aligned16(d.in) && aligned16(d.out) &&
         aligned16(d.wgt) && (d.res == nullptr || ((d.res_ps & 3) == 0 && aligned16(d.res))) &&
         (d.bias == nullptr || aligned16(d.bias)) && (d.act == PREMVOS_ACT_NONE || d.act == PREMVOS_ACT_RELU || d.act == PREMVOS_ACT_LEAKY);
}

int conv_stream(const premvos_conv_desc& d, hipStream_t s) { return d.k_pad == 64 ? launch_kg<8>(d, s) : launch_kg<16>(d, s); }

}  // namespace premvos
