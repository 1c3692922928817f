// Dense convolution as implicit GEMM on the gfx950 bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulate),
// for activations and outputs that STAY fp32 in HBM.  Two precisions of the same kernel:
//
//   NPASS = 1  "bf16"    a*b ~= hi(a)*hi(b)                       (configs[2]/[4] of BASELINE.json: bf16 compute)
//   NPASS = 3  "bf16x3"  a*b ~= hi(a)hi(b) + hi(a)lo(b) + lo(a)hi(b)   with x = hi + lo, hi = bf16(x), lo = bf16(x - hi)
//                        (drops only lo*lo ~ 2^-18 relative: fp32-class results at 3/16 of the fp32-MFMA cost)
//
// Same data path as conv_igemm_f32.hip (NHWC gather with zero fill, LDS double buffer, fused epilogue, channel
// window / pixel-shuffle outputs, deterministic split-K) with these differences: the k stage is 32 deep; the fp32
// activations are split into bf16 hi/lo with v_cvt_pk_bf16_f32 while they are staged into LDS; weights arrive
// pre-split ([cout_pad][k_pad32] bf16 hi and lo); LDS rows are 32 bf16 + 8 pad = 80 B so the ds_read_b128 fragment
// reads (8 consecutive k of one row per lane) stay bank-conflict free.
#include "common.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

namespace premvos {
int launch_splitk_reduce(const premvos_conv_desc& d, int splits, int ncols, hipStream_t s, int m_begin);   // conv_igemm_f32.hip
}

namespace {

constexpr int BK = 32;            // k granularity of the packed bf16 weights (k_pad % 32 == 0)

// KB = k depth of one LDS stage (16 or 32).  LDS rows are KB bf16 + 16 B pad (48 B / 80 B = 12 / 20 dwords, both 4 x odd:
// 16 consecutive rows hit 16 different 16-byte bank slots).  The 16-deep stage halves the LDS per workgroup so twice
// as many workgroups are resident per CU -- this kernel is latency-bound (SQ_WAIT_ANY 0.6), not MFMA-bound.
template <int BM, int BN, int WM, int WN, int NPASS, bool PIXSHUF, bool SPLITK, int KB>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_bf16_kernel(const premvos_conv_desc p, const int kt_per) {
  constexpr int NT = 64 * WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NTL = WTN / 32;
  constexpr int PARTS = NPASS == 3 ? 2 : 1;                  // hi (+ lo)
  constexpr int RSB = KB * 2 + 16;                           // LDS row stride in bytes
  constexpr int KU = KB / 4;                                 // float4 units per row (A)
  constexpr int BC = KB / 8;                                 // 16-byte bf16 chunks per row and part (B)
  constexpr int ROWS_PER_PASS = NT / KU;
  constexpr int A_PER_T = (BM + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
  constexpr int B_PER_T = (BN + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
  constexpr int PART_BYTES_A = BM * RSB, PART_BYTES_B = BN * RSB;
  constexpr int BUF_BYTES = PARTS * (PART_BYTES_A + PART_BYTES_B);

  extern __shared__ __attribute__((aligned(16))) char lds[];   // 2 * BUF_BYTES (up to 80 KB: dynamic)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * WTM, wn0 = (wave % WN) * WTN;
  const int M = p.n * p.ho * p.wo;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  const int j = tid % KU;          // float4 unit inside the stage (A); 16-byte bf16 chunk j % BC of part j / BC (B)
  const float* rowbase[A_PER_T];
  int iy0[A_PER_T], ix0[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int row = (tid / KU) + i * ROWS_PER_PASS;
    const int m = m0 + row;
    const bool ok = (row < BM) && (m < M);
    const int mm = ok ? m : 0;
    const int hw = p.ho * p.wo;
    const int n = mm / hw, rem = mm - n * hw;
    const int oy = rem / p.wo, ox = rem - oy * p.wo;
    rowbase[i] = p.in + (long)n * p.h * p.w * p.in_ps;
    iy0[i] = ok ? oy * p.sh - p.pt : -(1 << 28);
    ix0[i] = ox * p.sw - p.pl;
  }
  const int KT_all = p.k_pad / KB;
  const int kt_begin = SPLITK ? blockIdx.z * kt_per : 0;
  const int kt_end = SPLITK ? (kt_begin + kt_per < KT_all ? kt_begin + kt_per : KT_all) : KT_all;
  int kh, kw, c;
  {
    const int k0 = kt_begin * KB + j * 4;
    const int tap = k0 / p.cin_pad;
    c = k0 - tap * p.cin_pad;
    kh = tap / p.kw;
    kw = tap - kh * p.kw;
  }
  // weights: hi array at p.wgt, lo array at p.wgt_lo, both [cout_pad][k_pad] bf16
  const bool b_active = (j / BC) < PARTS;
  const unsigned short* wbase = reinterpret_cast<const unsigned short*>((j / BC) ? p.wgt_lo : p.wgt);
  const unsigned short* wrow[B_PER_T];
  bool wok[B_PER_T];
#pragma unroll
  for (int i = 0; i < B_PER_T; ++i) {
    const int row = (tid / KU) + i * ROWS_PER_PASS;
    wok[i] = b_active && (row < BN) && (n0 + row < p.cout_pad);
    wrow[i] = wbase + (long)(wok[i] ? n0 + row : 0) * p.k_pad + (j % BC) * 8;
  }

  float4 ra[A_PER_T];
  uint4 rb[B_PER_T];
  auto gload = [&](int kt) {
    const bool tap_ok = kh < p.kh;
    const int dy = kh * p.dh, dx = kw * p.dw;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int iy = iy0[i] + dy, ix = ix0[i] + dx;
      const bool ok = tap_ok && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w;
      ra[i] = ok ? *reinterpret_cast<const float4*>(rowbase[i] + ((long)iy * p.w + ix) * p.in_ps + c)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i)
      rb[i] = wok[i] ? *reinterpret_cast<const uint4*>(wrow[i] + (long)kt * KB) : make_uint4(0, 0, 0, 0);
    c += KB;
    while (c >= p.cin_pad) {
      c -= p.cin_pad;
      if (++kw == p.kw) { kw = 0; ++kh; }
    }
  };
  auto lstore = [&](int buf) {
    char* base = lds + buf * BUF_BYTES;
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int row = (tid / KU) + i * ROWS_PER_PASS;
      if (BM % ROWS_PER_PASS == 0 || row < BM) {
        const float4 v = ra[i];
        const bf16x4 hi = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        *reinterpret_cast<uint2*>(base + row * RSB + j * 8) = __builtin_bit_cast(uint2, hi);
        if constexpr (NPASS == 3) {
          const bf16x4 lo = {(__bf16)(v.x - (float)hi[0]), (__bf16)(v.y - (float)hi[1]), (__bf16)(v.z - (float)hi[2]),
                             (__bf16)(v.w - (float)hi[3])};
          *reinterpret_cast<uint2*>(base + PART_BYTES_A + row * RSB + j * 8) = __builtin_bit_cast(uint2, lo);
        }
      }
    }
    char* bb = base + PARTS * PART_BYTES_A + (j / BC) * PART_BYTES_B;
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      const int row = (tid / KU) + i * ROWS_PER_PASS;
      if (b_active && (BN % ROWS_PER_PASS == 0 || row < BN)) *reinterpret_cast<uint4*>(bb + row * RSB + (j % BC) * 16) = rb[i];
    }
  };

  f32x16 acc[MT][NTL];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int KT = kt_end - kt_begin;
  gload(kt_begin);
  lstore(0);
  __syncthreads();

  const int frag_off = (lane & 31) * RSB + (lane >> 5) * 16;   // 8 consecutive k (16 B) of row lane&31
#ifdef PV_DBG_BF16_OLDLOOP       // developer A/B builds: fragments read right before their MFMAs, everywhere
  constexpr bool FRAG_PF = false;
#else
  // bf16x3 only: measured +2.5 % on the mixed-bf16x3 pipeline, -8 % on mixed-bf16 (one MFMA per product: a block is too short)
  constexpr bool FRAG_PF = NPASS == 3 && (KB / 16) % 2 == 0;      // (a 16-deep stage is a single block: nothing to alternate)
#endif
  if constexpr (!FRAG_PF) {
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) gload(kt_begin + kt + 1);
    const char* base = lds + buf * BUF_BYTES;
    const char* a_hi = base + wm0 * RSB + frag_off;
    const char* b_hi = base + PARTS * PART_BYTES_A + wn0 * RSB + frag_off;
#pragma unroll
    for (int s = 0; s < KB / 16; ++s) {    // 16-deep MFMA k blocks of the stage
      bf16x8 ah[MT], al[MT], bh[NTL], bl[NTL];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        ah[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a_hi + mi * 32 * RSB + s * 32));
        if constexpr (NPASS == 3)
          al[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a_hi + PART_BYTES_A + mi * 32 * RSB + s * 32));
      }
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni) {
        bh[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(b_hi + ni * 32 * RSB + s * 32));
        if constexpr (NPASS == 3)
          bl[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(b_hi + PART_BYTES_B + ni * 32 * RSB + s * 32));
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTL; ++ni) {
          if constexpr (NPASS == 3) {      // small cross terms first, then the leading term
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
          }
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    if (kt + 1 < KT) lstore(buf ^ 1);
    __syncthreads();
  }

  } else {
  // The loop of conv_igemm_f32.hip (late round 3): two fragment sets -- the LDS reads of the next 16-deep block are in flight
  // under the current block's MFMAs, the first block of the next stage is requested right behind the barrier -- and a raised
  // issue priority over the MFMA run.  Same products in the same order.
  constexpr int S = KB / 16;
  bf16x8 ah[2][MT], al[2][MT], bh[2][NTL], bl[2][NTL];
  auto ldfrag = [&](const int set, const int buf, const int blk) {
    const char* base = lds + buf * BUF_BYTES;
    const char* a_hi = base + wm0 * RSB + frag_off + blk * 32;
    const char* b_hi = base + PARTS * PART_BYTES_A + wn0 * RSB + frag_off + blk * 32;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      ah[set][mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a_hi + mi * 32 * RSB));
      if constexpr (NPASS == 3) al[set][mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a_hi + PART_BYTES_A + mi * 32 * RSB));
    }
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni) {
      bh[set][ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(b_hi + ni * 32 * RSB));
      if constexpr (NPASS == 3) bl[set][ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(b_hi + PART_BYTES_B + ni * 32 * RSB));
    }
  };
  auto mfma_rows = [&](const int set, const int mi0, const int mi1) {
#pragma unroll
    for (int mi = mi0; mi < mi1; ++mi)
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni) {
        if constexpr (NPASS == 3) {      // small cross terms first, then the leading term
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[set][mi], bh[set][ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][mi], bl[set][ni], acc[mi][ni], 0, 0, 0);
        }
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[set][mi], bh[set][ni], acc[mi][ni], 0, 0, 0);
      }
  };
  ldfrag(0, 0, 0);
  for (int kt = 0; kt + 1 < KT; ++kt) {
    const int buf = kt & 1;
    gload(kt_begin + kt + 1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int blk = 0; blk + 1 < S; ++blk) {
      ldfrag((blk + 1) & 1, buf, blk + 1);
      mfma_rows(blk & 1, 0, MT);
    }
    mfma_rows((S - 1) & 1, 0, MT - 1);
    __builtin_amdgcn_s_setprio(0);
    lstore(buf ^ 1);
    __syncthreads();
    ldfrag(0, buf ^ 1, 0);
    mfma_rows((S - 1) & 1, MT - 1, MT);
  }
  {
    const int buf = (KT - 1) & 1;
#pragma unroll
    for (int blk = 0; blk < S; ++blk) {
      if (blk + 1 < S) ldfrag((blk + 1) & 1, buf, blk + 1);
      mfma_rows(blk & 1, 0, MT);
    }
    __syncthreads();
  }
  }

  if constexpr (SPLITK) {
    const int ncols = gridDim.y * BN;
    float* ws = p.workspace + (long)blockIdx.z * M * ncols;
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni) {
      const int col = n0 + wn0 + ni * 32 + (lane & 31);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < M) ws[(long)m * ncols + col] = acc[mi][ni][r];
        }
    }
    return;
  }
#pragma unroll
  for (int ni = 0; ni < NTL; ++ni) {
    const int col = n0 + wn0 + ni * 32 + (lane & 31);
    const bool colok = col < p.cout;
    const float bv = (p.bias != nullptr && colok) ? p.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      // residual rows of this 32x32 tile are fetched as 16 independent loads BEFORE they are consumed: a load-use
      // chain per element made the short-K bottleneck layers (ResNet conv3 + shortcut) latency-bound at ~1 TB/s
      float rv[16];
      if (p.res != nullptr) {     // wave-uniform
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          rv[r] = (colok && m < M) ? p.res[(long)m * p.res_ps + col] : 0.f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int m = m0 + row;
        if (!colok || m >= M) continue;
        float v = acc[mi][ni][r] + bv + rv[r];
        if (p.act == PREMVOS_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (p.act == PREMVOS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
        else if (p.act == PREMVOS_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
        if constexpr (PIXSHUF) {
          const int hw = p.ho * p.wo;
          const int n = m / hw, rem = m - n * hw;
          const int oy = rem / p.wo, ox = rem - oy * p.wo;
          const int phase = col / p.cout_ps, co = col - phase * p.cout_ps;
          const long opix = ((long)n * 2 * p.ho + 2 * oy + (phase >> 1)) * (2 * p.wo) + 2 * ox + (phase & 1);
          p.out[opix * p.out_ps + co] = v;
        } else {
          p.out[(long)m * p.out_ps + col] = v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the next tile's 16 residual loads from being hoisted (VGPR budget)
    }
  }
}

inline void pick_tile(const premvos_conv_desc& d, int* bm, int* bn) {
  const int M = d.n * d.ho * d.wo;
  if (d.tile_hint) {
    *bm = d.tile_hint >> 16;
    *bn = d.tile_hint & 0xffff;
    return;
  }
  *bn = d.cout <= 32 ? 32 : d.cout <= 64 ? 64 : 128;
  const long blocks128 = (long)premvos::cdiv(M, 128) * premvos::cdiv(d.cout, *bn);
  *bm = blocks128 >= 512 ? 128 : 64;
}

inline int pick_kb(const premvos_conv_desc& d) { return d.stage_k == 32 ? 32 : 16; }

inline int pick_splits(const premvos_conv_desc& d, int bm, int bn) {
  if (d.split_k > 0) return d.split_k;
  if (d.split_k < 0) return 1;
  const long tiles = (long)premvos::cdiv(d.n * d.ho * d.wo, bm) * premvos::cdiv(d.cout, bn);
  const int KT = d.k_pad / 32;
  if (tiles >= 384 || KT < 8) return 1;
  long s = (768 + tiles - 1) / tiles;
  if (s > KT / 4) s = KT / 4;
  if (s > 32) s = 32;
  return s < 2 ? 1 : (int)s;
}

template <typename K>
inline void allow_lds(K kernel, int bytes) {   // > 64 KB of LDS per workgroup needs the attribute (once per kernel)
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int BM, int BN, int WM, int WN, int NPASS, int KB>
int launch_cfg(const premvos_conv_desc& d, hipStream_t s) {
  const int M = d.n * d.ho * d.wo;
  dim3 grid(premvos::cdiv(M, BM), premvos::cdiv(d.cout, BN));
  dim3 block(64 * WM * WN);
  constexpr int LDS_BYTES = 2 * (NPASS == 3 ? 2 : 1) * (BM + BN) * (KB * 2 + 16);
  static const bool attr_done = [] {            // once per instantiation, thread-safe
    allow_lds(conv_igemm_bf16_kernel<BM, BN, WM, WN, NPASS, false, true, KB>, LDS_BYTES);
    allow_lds(conv_igemm_bf16_kernel<BM, BN, WM, WN, NPASS, true, false, KB>, LDS_BYTES);
    allow_lds(conv_igemm_bf16_kernel<BM, BN, WM, WN, NPASS, false, false, KB>, LDS_BYTES);
    return true;
  }();
  (void)attr_done;
  int splits = pick_splits(d, BM, BN);
  const int KT = d.k_pad / KB;
  if (splits > 1) {
    const int kt_per = premvos::cdiv(KT, splits);
    splits = premvos::cdiv(KT, kt_per);
    const int ncols = grid.y * BN;
    const long need = (long)splits * M * ncols * sizeof(float);
    if (splits > 1 && d.workspace != nullptr && (long)d.workspace_bytes >= need) {
      grid.z = splits;
      hipLaunchKernelGGL((conv_igemm_bf16_kernel<BM, BN, WM, WN, NPASS, false, true, KB>), grid, block, LDS_BYTES, s, d, kt_per);
      int rc = premvos::check_launch("conv_igemm_bf16(split-k)");
      if (rc) return rc;
      return premvos::launch_splitk_reduce(d, splits, ncols, s, 0);
    }
    if (d.split_k > 0) return premvos::fail(PREMVOS_EINVAL, "conv2d: split_k=%d needs %ld workspace bytes", d.split_k, need);
  }
  if (d.out_mode == PREMVOS_OUT_PIXSHUF2)
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<BM, BN, WM, WN, NPASS, true, false, KB>), grid, block, LDS_BYTES, s, d, 0);
  else
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<BM, BN, WM, WN, NPASS, false, false, KB>), grid, block, LDS_BYTES, s, d, 0);
  return premvos::check_launch("conv_igemm_bf16");
}

template <int NPASS, int KB>
int dispatch_kb(const premvos_conv_desc& d, hipStream_t s) {
  int bm, bn;
  pick_tile(d, &bm, &bn);
  switch ((bm << 16) | bn) {
    case (128 << 16) | 128: return launch_cfg<128, 128, 2, 2, NPASS, KB>(d, s);
    case (128 << 16) | 64: return launch_cfg<128, 64, 2, 2, NPASS, KB>(d, s);
    case (128 << 16) | 32: return launch_cfg<128, 32, 4, 1, NPASS, KB>(d, s);
    case (64 << 16) | 128: return launch_cfg<64, 128, 2, 2, NPASS, KB>(d, s);
    case (64 << 16) | 64: return launch_cfg<64, 64, 2, 2, NPASS, KB>(d, s);
    case (64 << 16) | 32: return launch_cfg<64, 32, 2, 1, NPASS, KB>(d, s);
    default: return premvos::fail(PREMVOS_EINVAL, "conv2d(bf16): no tile config %dx%d", bm, bn);
  }
}

template <int NPASS>
int dispatch(const premvos_conv_desc& d, hipStream_t s) {
  return pick_kb(d) == 32 ? dispatch_kb<NPASS, 32>(d, s) : dispatch_kb<NPASS, 16>(d, s);
}

}  // namespace

namespace premvos {

int conv2d_bf16(const premvos_conv_desc& d, hipStream_t s) {
  if (d.k_pad % BK != 0) return fail(PREMVOS_EINVAL, "conv2d(bf16): k_pad must be a multiple of 32");
  if (d.precision == PREMVOS_PREC_BF16X3) {
    if (d.wgt_lo == nullptr) return fail(PREMVOS_EINVAL, "conv2d(bf16x3): wgt_lo is NULL");
    return dispatch<3>(d, s);
  }
  return dispatch<1>(d, s);
}

long conv2d_bf16_workspace_bytes(const premvos_conv_desc& d) {
  int bm, bn;
  pick_tile(d, &bm, &bn);
  int splits = pick_splits(d, bm, bn);
  if (splits <= 1) return 0;
  const int KT = d.k_pad / pick_kb(d);
  const int kt_per = cdiv(KT, splits);
  splits = cdiv(KT, kt_per);
  if (splits <= 1) return 0;
  return (long)splits * d.n * d.ho * d.wo * cdiv(d.cout, bn) * bn * (long)sizeof(float);
}

}  // namespace premvos
