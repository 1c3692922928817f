// Dense convolution as implicit GEMM on the gfx950 fp32 matrix pipe.
//
//   M = N*Ho*Wo output pixels, Ncol = Cout, K = KH*KW*cin_pad  (k = tap*cin_pad + c)
//   D[m][j] = act( sum_k A[m][k] * B[j][k] + bias[j] (+ res[m][j]) )
//
// * A (activations) is gathered straight from the NHWC input slice: one 16-byte load per
//   (pixel, tap, 4 channels), zero filled outside the image (any asymmetric padding), staged
//   through LDS; B (packed weights, k contiguous) likewise.  LDS rows are padded to 20 floats so
//   the ds_read_b128 fragment reads of both operands are bank-conflict free.
// * Math: v_mfma_f32_32x32x2_f32 -- exact fp32 (bitwise an fmaf chain in k order) at the
//   157 TFLOP/s fp32 rate, i.e. the PWC-Net fp32 config stays fp32 end to end.
//   One float4 LDS read feeds 4 MFMAs: lane l holds k = 4*(l>>5)+{0..3} of an 8-deep k group for
//   row/col (l&31); MFMA e pairs k = e and k = 4+e of both operands.
// * C/D layout (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)): with M = pixels and
//   N = cout every store instruction writes 32 consecutive output channels of one pixel
//   (128 B runs) into the NHWC destination slice -- which may be a channel window of a DenseNet
//   concat buffer (PWCNet.py:201-205: the 54 torch.cat copies disappear).
// * Double-buffered LDS, one barrier per 16-deep k step, next tile's global loads in flight
//   during the MFMAs.
//
// Reference call sites replaced: see include/premvos_hip.h (premvos_conv2d_f32).
#include "common.h"
#include <type_traits>

namespace premvos {
thread_local char g_err[512] = "";
int conv2d_bf16(const premvos_conv_desc& d, hipStream_t s);               // conv_igemm_bf16.hip
long conv2d_bf16_workspace_bytes(const premvos_conv_desc& d);
}

using f32x16 = __attribute__((ext_vector_type(16))) float;

#ifdef PV_DBG_TIMELINE          // developer build (tools/dev/ab_build.sh tl -DPV_DBG_TIMELINE): per-workgroup phase stamps
__device__ unsigned long long g_tl[1 << 20];
#define PV_TL(slot)                                                                                         \
  do {                                                                                                      \
    if (threadIdx.x == 0) {                                                                                 \
      const unsigned wgl = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                  \
      if (wgl < (1u << 17)) g_tl[wgl * 8 + (slot)] = __builtin_readcyclecounter();                         \
      if ((slot) == 0 && wgl < (1u << 17)) {                                                                \
        g_tl[wgl * 8 + 4] = wall_clock64();                                                                 \
        g_tl[wgl * 8 + 5] = __builtin_amdgcn_s_getreg(4 | (31 << 11));                                      \
        g_tl[wgl * 8 + 6] = __builtin_amdgcn_s_getreg(20 | (31 << 11));                                     \
      }                                                                                                     \
      if ((slot) == 3 && wgl < (1u << 17)) g_tl[wgl * 8 + 7] = wall_clock64();                              \
    }                                                                                                       \
  } while (0)
extern "C" int premvos_dbg_timeline(void* dst, long bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tl), bytes, 0, hipMemcpyDeviceToHost);
}
#else
#define PV_TL(slot) do {} while (0)
#endif

namespace {

constexpr int BK = 16;           // k granularity of the packed weights (k_pad % 16 == 0)

// SPLITK: blockIdx.z owns k-steps [z*kt_per, min(KT,(z+1)*kt_per)) and stores its raw partial tile to
// p.workspace[z][m][ncols]; splitk_reduce_kernel sums the slabs in z order (deterministic) and applies the
// epilogue.  Used when a layer has too few output tiles to fill 256 CUs (coarse PWC levels, batch-1 RoI/feature maps).
// KB: k depth of one LDS stage (16 or 32).  Rows are padded by 4 floats (KB+4): 20 and 36 dwords are both 4 x odd,
// so 16 consecutive rows land on 16 different 16-byte bank slots -> conflict-free ds_read_b128.
// PW: pointwise layers (1x1 taps, no padding; any stride): the gather address of an A row is a fixed pixel base + k, so
// the per-stage tap bookkeeping and 64-bit address arithmetic of the general path drop out (most ResNet / Xception
// layers; the scalar+vector work between the barrier and the first MFMA of a stage was ~15 % of a stage).
#ifndef PV_OCC128
#define PV_OCC128 3      // workgroups per CU the 128x128 tile is compiled for (developer builds: -DPV_OCC128=4)
#endif
// workgroups per CU a tile is compiled for: the 64 x 128 / 128 x 64 wave tiles (128 accumulator registers) run two 4-wave
// workgroups per CU inside 256 registers
constexpr int occ_of(int bm, int bn, int wm, int wn, bool pixshuf) {
  return (bm == 128 && bn == 128 && !pixshuf) ? PV_OCC128 : (bm * bn == 128 * 256 && wm * wn == 4 && !pixshuf) ? 2 : 1;
}
template <int BM, int BN, int WM, int WN, bool PIXSHUF, bool SPLITK, int KB = 16, bool PW = false>
__global__ __launch_bounds__(64 * WM * WN, occ_of(BM, BN, WM, WN, PIXSHUF)) void conv_igemm_f32_kernel(const premvos_conv_desc p, const int kt_per, const int mt0) {
  constexpr int NT = 64 * WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 32, NTL = WTN / 32;
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32x32");
  constexpr int RS = KB + 4;
  constexpr int KU = KB / 4;                   // float4 units per row
  constexpr int A_UNITS = BM * KU, B_UNITS = BN * KU;
  constexpr int A_PER_T = (A_UNITS + NT - 1) / NT, B_PER_T = (B_UNITS + NT - 1) / NT;
  constexpr int BUF = (BM + BN) * RS;

  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  float(*lds)[BUF] = reinterpret_cast<float(*)[BUF]>(lds_dyn);

  PV_TL(0);
#ifdef PV_EDGE_PRIO
  __builtin_amdgcn_s_setprio(3);
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * WTM, wn0 = (wave % WN) * WTN;
  const int M = p.n * p.ho * p.wo;
  // XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8 (each XCD has a private 4 MB L2).  Give every
  // XCD a contiguous chunk of the (m-tile major, n-tile minor) order, so the n-tiles that re-read one A (pixel) tile and
  // the m-tiles that share halo rows run on the SAME L2.  Pure speed: any placement computes the same result.
  int tile_m, tile_n;
  {
    const int n_tiles = gridDim.y, nwg = gridDim.x * gridDim.y;
    const int id = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = id & 7, local = id >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    tile_m = v / n_tiles;
    tile_n = v - tile_m * n_tiles;
    tile_m += mt0;             // tail-split launches start at m-tile row mt0 (0 otherwise)
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-thread gather state -------------------------------------------------------------
  [[maybe_unused]] const bool pw_unit = p.sh == 1 && p.sw == 1 && p.ho == p.h && p.wo == p.w;
  const int j4 = (tid % KU) * 4;  // this thread's float4 column inside the KB-deep stage
  const float* rowbase[A_PER_T];
  int iy0[A_PER_T], ix0[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int row = (tid / KU) + i * (NT / KU);
    const int m = m0 + row;
    const bool ok = (row < BM) && (m < M);
    const int mm = ok ? m : 0;
    if constexpr (PW) {
      // round 5: a unit-stride pointwise layer reads pixel m of the input -- no (image, y, x) split, i.e. none of the two integer
      // divisions per row that made the address arithmetic most of a tile's prologue (profiles/r05_timeline_igemm.txt: 8 us)
      if (pw_unit) {                               // kernel-uniform
        rowbase[i] = ok ? p.in + (long)mm * p.in_ps : nullptr;
        iy0[i] = ix0[i] = 0;
        continue;
      }
    }
    const int hw = p.ho * p.wo;
    const int n = mm / hw, rem = mm - n * hw;
    const int oy = rem / p.wo, ox = rem - oy * p.wo;
    rowbase[i] = p.in + (long)n * p.h * p.w * p.in_ps;
    iy0[i] = ok ? oy * p.sh - p.pt : -(1 << 28);  // invalid rows fail the bounds test below
    ix0[i] = ox * p.sw - p.pl;
    if constexpr (PW) {                            // the one pixel this row reads (nullptr = row out of range)
      rowbase[i] = ok ? rowbase[i] + ((long)(oy * p.sh) * p.w + ox * p.sw) * p.in_ps : nullptr;
    }
  }
  const int KT_all = (p.k_pad + KB - 1) / KB;          // KB = 32 on a k_pad % 32 == 16 matrix: last half stage is zero
  const int kt_begin = SPLITK ? blockIdx.z * kt_per : 0;
  const int kt_end = SPLITK ? (kt_begin + kt_per < KT_all ? kt_begin + kt_per : KT_all) : KT_all;
  int kh = 0, kw = 0, c;
  if constexpr (PW) {
    c = kt_begin * KB + j4;                        // k itself (single tap); k >= cin_pad is zero padding
  } else {
    const int k0 = kt_begin * KB + j4;
    const int tap = k0 / p.cin_pad;
    c = k0 - tap * p.cin_pad;
    kh = tap / p.kw;
    kw = tap - kh * p.kw;
  }
  const float* wrow[B_PER_T];
  bool wok[B_PER_T];
#pragma unroll
  for (int i = 0; i < B_PER_T; ++i) {
    const int row = (tid / KU) + i * (NT / KU);
    wok[i] = (row < BN) && (n0 + row < p.cout_pad);
    wrow[i] = p.wgt + (long)(wok[i] ? n0 + row : 0) * p.k_pad + j4;
  }

  float4 ra[A_PER_T], rb[B_PER_T];
  // Interior stages of interior tiles of a 1x1 layer (all BM rows < M, all BN weight rows < cout_pad, and every stage but the
  // matrix's last lies below cin_pad = k_pad rounded down): nothing to predicate -- plain requests, no zero fill, no branches.
  [[maybe_unused]] auto gload_plain = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) ra[i] = *reinterpret_cast<const float4*>(rowbase[i] + c);
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) rb[i] = *reinterpret_cast<const float4*>(wrow[i] + kt * KB);
    c += KB;
  };
  auto gload = [&](int kt) {
    if constexpr (PW) {
      const bool kok = c < p.cin_pad;
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i)
        ra[i] = (kok && rowbase[i] != nullptr) ? *reinterpret_cast<const float4*>(rowbase[i] + c)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const bool tap_ok = kh < p.kh;
      const int dy = kh * p.dh, dx = kw * p.dw;
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) {
        const int iy = iy0[i] + dy, ix = ix0[i] + dx;
        const bool ok = tap_ok && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w;
        ra[i] = ok ? *reinterpret_cast<const float4*>(rowbase[i] + ((long)iy * p.w + ix) * p.in_ps + c)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i)
      rb[i] = (wok[i] && kt * KB + j4 < p.k_pad) ? *reinterpret_cast<const float4*>(wrow[i] + kt * KB)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
    c += KB;
    if constexpr (!PW) {
      while (c >= p.cin_pad) {
        c -= p.cin_pad;
        if (++kw == p.kw) { kw = 0; ++kh; }
      }
    }
  };
  auto lstore = [&](int buf) {
    float* a = &lds[buf][0];
    float* b = &lds[buf][BM * RS];
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int row = (tid / KU) + i * (NT / KU);
      if (A_UNITS % NT == 0 || row < BM) *reinterpret_cast<float4*>(a + row * RS + j4) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER_T; ++i) {
      const int row = (tid / KU) + i * (NT / KU);
      if (B_UNITS % NT == 0 || row < BN) *reinterpret_cast<float4*>(b + row * RS + j4) = rb[i];
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
  PV_TL(1);
#ifdef PV_EDGE_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif

  // Padding that is never multiplied (round 3).  (1) Columns: a wave whose 32-column blocks lie (partly) beyond cout -- the last
  // column tile of the 728-wide Xception layers holds 88 real columns of 128 -- skips the MFMAs and B-fragment reads of its
  // empty blocks; the matrix pipe is shared by the waves of the 3 workgroups resident on a CU, so the freed slots go to them.
  // (2) K: the last stage of a matrix whose K is not a multiple of the stage depth (728 = 45.5 x 16) only runs the 8-deep groups
  // that hold real k.  Both skip products with an all-zero operand: same sums (up to the sign of an exact zero).
#ifdef PV_DBG_NOSKIP           // developer A/B builds (tools/dev/ab_build.sh): multiply the padding like rounds 1-2 did
  const int nvalid = NTL, h_last = KB / 8;
#else
  const int nvalid = min(NTL, max(0, (p.cout - (n0 + wn0) + 31) / 32));           // wave-uniform
  const int kreal = p.kh * p.kw * p.cin_pad;
  const int h_last = min(KB / 8, max(1, (kreal - (KT_all - 1) * KB + 7) / 8));    // 8-deep groups of the matrix's last stage
#endif

  const int frag_off = (lane & 31) * RS + 4 * (lane >> 5);
  // NARROW (round 3, the 128x128 / 2x2-wave tile only): a column tile with at most 96 real columns -- the sixth tile of the
  // 728-wide Xception layers holds 88 -- is computed with the four waves side by side in M (32 rows x 3 column blocks each)
  // instead of 2 x 2 (64 x 64 each): every wave then multiplies 3/4 (or 1/2, 1/4) of a full tile and the workgroup finishes
  // that much earlier.  Merely skipping the empty blocks in the 2x2 layout did not shorten the tile (its left waves still
  // multiply a full 64x64); the 4x1 layout for ALL tiles reads 5 instead of 4 LDS fragments per 16 MFMAs and was slower
  // overall (round 2) -- here it runs only where it removes work.  Same products, same k order: bit-identical results.
  // Measured (profiles/r03_igemm_ab.txt): +3 % on the 728-wide layers with 32-deep stages (two workgroups per CU: a tile's
  // duration is what the CU waits for); nothing with 16-deep stages (three per CU: the loop is not bound by how long a tile
  // occupies its slot -- occupancy 1 / 2 / 3 / 4 give 0.65 / 0.77 / 0.83 / 0.82 of the matrix peak on an ideal shape).
  constexpr bool NARROW_OK = BM == 128 && BN == 128 && WM == 2 && WN == 2 && !PIXSHUF && !SPLITK;
  bool narrow = false;
  int nvn = 0;
  if constexpr (NARROW_OK) {
#ifndef PV_DBG_NONARROW
    const int real = p.cout - n0;                                   // kernel-uniform per workgroup
    const bool wide_ok = (p.cout & 3) == 0 && (p.out_ps & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15u) == 0 &&
                         (p.res == nullptr || ((p.res_ps & 3) == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15u) == 0)) &&
                         (p.bias == nullptr || (reinterpret_cast<uintptr_t>(p.bias) & 15u) == 0);
    narrow = wide_ok && real > 0 && real <= 96;
    nvn = (real + 31) / 32;
#endif
  }
  // One straight-line copy of the K loop per number of live column blocks (no per-block condition inside it: conditional
  // accumulator updates cost the kernel 45 VGPRs and a wave per SIMD when they were tried); the last stage is peeled so that
  // only it carries the run-time bound on its 8-deep groups.
  auto k_loop = [&](auto nv_tag) {
    constexpr int NV = decltype(nv_tag)::value;
    auto compute = [&](const float* a, const float* b, const int hcnt) {
#pragma unroll
      for (int h = 0; h < KB / 8; ++h) {
        if (h >= hcnt) break;                  // (folds away where hcnt is the constant KB / 8)
        float4 af[MT], bf[NV > 0 ? NV : 1];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) af[mi] = *reinterpret_cast<const float4*>(a + mi * 32 * RS + h * 8);
#pragma unroll
        for (int ni = 0; ni < NV; ++ni) bf[ni] = *reinterpret_cast<const float4*>(b + ni * 32 * RS + h * 8);
        // four k-steps in a row on ONE accumulator (dependent MFMAs issue back to back at the pipe's own 64-cycle pace;
        // measured: walking the accumulators round-robin instead is 1.5 % slower)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NV; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].x, bf[ni].x, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].y, bf[ni].y, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].z, bf[ni].z, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi].w, bf[ni].w, acc[mi][ni], 0, 0, 0);
          }
      }
    };
    const bool ends_matrix = kt_end == KT_all;       // this workgroup's last stage is the matrix's last stage
#ifdef PV_DBG_NOFRAGPF           // developer A/B builds: the loop of rounds 1-3a (fragments read right before their MFMAs)
    constexpr bool FRAG_PF = false;
#else
    constexpr bool FRAG_PF = KB == 16;      // (32-deep stages hold twice the staging registers: the second fragment set spilled)
#endif
    if constexpr (!FRAG_PF) {
    for (int kt = 0; kt + 1 < KT; ++kt) {
      const int buf = kt & 1;
      gload(kt_begin + kt + 1);
      compute(&lds[buf][wm0 * RS + frag_off], &lds[buf][(BM + wn0) * RS + frag_off], KB / 8);
      lstore(buf ^ 1);
      __syncthreads();
    }
    const int buf = (KT - 1) & 1;
    compute(&lds[buf][wm0 * RS + frag_off], &lds[buf][(BM + wn0) * RS + frag_off], ends_matrix ? h_last : KB / 8);
    __syncthreads();
    } else {
    // Fragment double buffering: the LDS reads of the NEXT 8-deep group are in flight while the MFMAs of the current one
    // issue (two register sets), and the first group of the next stage is requested right behind the barrier, under the last
    // row block's MFMAs of this stage -- before, every group began with ds_read + s_waitcnt lgkmcnt(0) (four exposed LDS
    // round trips per stage and wave).  Same products in the same order: bit-identical results.
    constexpr int H = KB / 8, NVV = NV > 0 ? NV : 1;
    float4 af[2][MT], bf[2][NVV];
    auto ldfrag = [&](const int set, const int buf, const int h) {
      const float* a = &lds[buf][wm0 * RS + frag_off] + h * 8;
      const float* b = &lds[buf][(BM + wn0) * RS + frag_off] + h * 8;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) af[set][mi] = *reinterpret_cast<const float4*>(a + mi * 32 * RS);
#pragma unroll
      for (int ni = 0; ni < NV; ++ni) bf[set][ni] = *reinterpret_cast<const float4*>(b + ni * 32 * RS);
    };
    auto mfma_rows = [&](const int set, const int mi0, const int mi1) {
#pragma unroll
      for (int mi = mi0; mi < mi1; ++mi)
#pragma unroll
        for (int ni = 0; ni < NV; ++ni) {
#ifdef PV_ASM_CHAIN
          // developer variant (round 5): the four dependent MFMAs of one accumulator as ONE opaque statement -- the scheduler cannot
          // put a request / fragment read / address instruction between two of them (a break in a dependent chain costs the pipe
          // ~43 cycles, MI355X_MICROARCH.md), only between chains
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %5, %0\n\tv_mfma_f32_32x32x2_f32 %0, %2, %6, %0\n\t"
                       "v_mfma_f32_32x32x2_f32 %0, %3, %7, %0\n\tv_mfma_f32_32x32x2_f32 %0, %4, %8, %0"
                       : "+v"(acc[mi][ni])
                       : "v"(af[set][mi].x), "v"(af[set][mi].y), "v"(af[set][mi].z), "v"(af[set][mi].w), "v"(bf[set][ni].x),
                         "v"(bf[set][ni].y), "v"(bf[set][ni].z), "v"(bf[set][ni].w));
          continue;
#endif
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].x, bf[set][ni].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].y, bf[set][ni].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].z, bf[set][ni].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].w, bf[set][ni].w, acc[mi][ni], 0, 0, 0);
        }
    };
    static_assert(H % 2 == 0, "the two fragment sets alternate per 8-deep group");
#ifdef PV_EARLY_STORE
    // Developer variant (round 5): the staging registers are stored at the START of a stage (their requests went out a whole stage
    // earlier) and refilled at once with the stage after the next -- the ds_write traffic and its lgkmcnt wait move away from
    // the barrier, and a request has a full stage to come back.
    ldfrag(0, 0, 0);
    if (KT > 1) gload(kt_begin + 1);
    auto stage = [&](const int kt, auto mode_tag) {       // 0: nothing left to request, 1: predicated request, 2: plain
      constexpr int MODE = decltype(mode_tag)::value;
      const int buf = kt & 1;
      lstore(buf ^ 1);
      if constexpr (MODE == 2) {
        gload_plain(kt_begin + kt + 2);
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (MODE == 1) gload(kt_begin + kt + 2);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int h = 0; h + 1 < H; ++h) {
        ldfrag((h + 1) & 1, buf, h + 1);
        mfma_rows(h & 1, 0, MT);
      }
      mfma_rows((H - 1) & 1, 0, MT - 1);
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      ldfrag(0, buf ^ 1, 0);
      mfma_rows((H - 1) & 1, MT - 1, MT);
    };
    int kt = 0;
    if constexpr (PW && A_UNITS % NT == 0 && B_UNITS % NT == 0) {
      const bool interior = m0 + BM <= M && n0 + BN <= p.cout_pad && (KT_all - 1) * KB <= p.cin_pad;
      if (interior)
        for (; kt_begin + kt + 2 < KT_all - 1 && kt + 2 < KT; ++kt) stage(kt, std::integral_constant<int, 2>{});
    }
    for (; kt + 2 < KT; ++kt) stage(kt, std::integral_constant<int, 1>{});
    for (; kt + 1 < KT; ++kt) stage(kt, std::integral_constant<int, 0>{});
#else
    ldfrag(0, 0, 0);
    auto stage = [&](const int kt, auto plain_tag) {
      const int buf = kt & 1;
      // (requesting a stage earlier -- right behind the previous barrier -- measured 1.3 % slower with the predicated requests and
      //  4 % slower with the plain ones; sunk to right before the ds_write: 1.7 % slower)
      if constexpr (decltype(plain_tag)::value) {
        gload_plain(kt_begin + kt + 1);
#ifndef PV_DBG_NOPIN_GLOAD
        // a scheduling fence behind the requests: left alone, the scheduler reuses the fragment registers for all four of them and
        // sinks them to right before their ds_write; with the fence two get registers of their own and the next group's fragment
        // reads follow them (+1.7 % in the same-box A/B; other placements of the requests / fences measured equal or slower)
        __builtin_amdgcn_sched_barrier(0);
#endif
      } else gload(kt_begin + kt + 1);
#ifndef PV_DBG_NOSETPRIO
      __builtin_amdgcn_s_setprio(1);       // a wave inside its MFMA run wins the issue arbitration over waves staging / waiting (+0.5 %)
#endif
#pragma unroll
      for (int h = 0; h + 1 < H; ++h) {
        ldfrag((h + 1) & 1, buf, h + 1);
        mfma_rows(h & 1, 0, MT);
      }
      mfma_rows((H - 1) & 1, 0, MT - 1);
#ifndef PV_DBG_NOSETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      lstore(buf ^ 1);
      __syncthreads();
      ldfrag(0, buf ^ 1, 0);
      mfma_rows((H - 1) & 1, MT - 1, MT);
    };
    int kt = 0;
#ifndef PV_DBG_NOPLAIN
    if constexpr (PW && A_UNITS % NT == 0 && B_UNITS % NT == 0) {
      // workgroup-uniform; stages kt + 1 <= KT_all - 2 hold only k < cin_pad (k_pad - cin_pad < KB)
      const bool interior = m0 + BM <= M && n0 + BN <= p.cout_pad && (KT_all - 1) * KB <= p.cin_pad;
      if (interior)
        for (; kt_begin + kt + 1 < KT_all - 1 && kt + 1 < KT; ++kt) stage(kt, std::true_type{});
    }
#endif
    for (; kt + 1 < KT; ++kt) stage(kt, std::false_type{});
#endif
    const int buf = (KT - 1) & 1;
    const int hcnt = ends_matrix ? h_last : H;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      if (h >= hcnt) break;
      if (h + 1 < hcnt) ldfrag((h + 1) & 1, buf, h + 1);
      mfma_rows(h & 1, 0, MT);
    }
    __syncthreads();
    }
  };
  [[maybe_unused]] auto k_loop_narrow = [&](auto nv_tag) {
    constexpr int NV = decltype(nv_tag)::value;                     // live 32-column blocks: 1 ... 3
    auto accn = [&](int j) -> f32x16& { return acc[j >> 1][j & 1]; };
    auto compute = [&](const float* a, const float* b, const int hcnt) {
#pragma unroll
      for (int h = 0; h < KB / 8; ++h) {
        if (h >= hcnt) break;
        const float4 af = *reinterpret_cast<const float4*>(a + h * 8);
        float4 bf[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) bf[j] = *reinterpret_cast<const float4*>(b + j * 32 * RS + h * 8);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          accn(j) = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf[j].x, accn(j), 0, 0, 0);
          accn(j) = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf[j].y, accn(j), 0, 0, 0);
          accn(j) = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf[j].z, accn(j), 0, 0, 0);
          accn(j) = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf[j].w, accn(j), 0, 0, 0);
        }
      }
    };
    const bool ends_matrix = kt_end == KT_all;
    const int arow = 32 * wave * RS + frag_off, brow = BM * RS + frag_off;
    for (int kt = 0; kt + 1 < KT; ++kt) {
      const int buf = kt & 1;
      gload(kt_begin + kt + 1);
      compute(&lds[buf][arow], &lds[buf][brow], KB / 8);
      lstore(buf ^ 1);
      __syncthreads();
    }
    const int buf = (KT - 1) & 1;
    compute(&lds[buf][arow], &lds[buf][brow], ends_matrix ? h_last : KB / 8);
    __syncthreads();
  };
  bool done = false;
  if constexpr (NARROW_OK) {
    if (narrow) {                                                   // workgroup-uniform
      if (nvn == 3) k_loop_narrow(std::integral_constant<int, 3>{});
      else if (nvn == 2) k_loop_narrow(std::integral_constant<int, 2>{});
      else k_loop_narrow(std::integral_constant<int, 1>{});
      done = true;
    }
  }
  if (done) {
  } else if (nvalid == NTL) k_loop(std::integral_constant<int, NTL>{});
  else if (NTL > 3 && nvalid == 3) k_loop(std::integral_constant<int, (NTL > 3 ? 3 : 0)>{});
  else if (NTL > 2 && nvalid == 2) k_loop(std::integral_constant<int, (NTL > 2 ? 2 : 0)>{});
  else if (NTL > 1 && nvalid == 1) k_loop(std::integral_constant<int, (NTL > 1 ? 1 : 0)>{});
  else k_loop(std::integral_constant<int, 0>{});
  PV_TL(2);
#ifdef PV_EDGE_PRIO
  __builtin_amdgcn_s_setprio(3);
#endif

  if constexpr (SPLITK) {   // raw partial slab, ncols = gridDim.y * BN (padded: no column predicate needed)
    const int ncols = gridDim.y * BN;
    const int mb = mt0 * BM;                  // slabs hold rows [mb, M) only
    float* ws = p.workspace + (long)blockIdx.z * (M - mb) * ncols;
#pragma unroll
    for (int ni = 0; ni < NTL; ++ni) {
      const int col = n0 + wn0 + ni * 32 + (lane & 31);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < M) ws[(long)(m - mb) * ncols + col] = acc[mi][ni][r];
        }
    }
    return;
  }
  // ---- epilogue, wide form: the accumulators of one wave row go through LDS (free after the main loop) so that every lane
  // stores 16 bytes of a pixel's contiguous channel run (an MFMA accumulator holds ONE column per lane: straight from
  // registers that is a 4-byte store per element, 64 store instructions per lane for a 128x128 tile -- short-K layers with
  // many output channels were store-ISSUE-bound at ~1.7 TB/s).  Bias, residual (16-byte loads) and activation at read-back.
  if constexpr (!PIXSHUF) {
    const bool wide = (p.cout & 3) == 0 && (p.out_ps & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15u) == 0 &&
                      (p.res == nullptr || ((p.res_ps & 3) == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15u) == 0)) &&
                      (p.bias == nullptr || (reinterpret_cast<uintptr_t>(p.bias) & 15u) == 0);
    constexpr int EP = BN + 4;                                   // row pitch of the staged block (floats)
    // (tiles whose staged wave row outgrows the operand buffers get the difference as extra dynamic LDS: lds_floats())
#ifdef PV_DBG_NOEPI             // developer phase ablation (tools/dev/ab_build.sh): no epilogue at all; the accumulators stay live
    if (wide) {
      float sacc = 0.f;
#pragma unroll
      for (int ni = 0; ni < NTL; ++ni)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc += acc[mi][ni][r];
      if (sacc == 12345.678f) p.out[tid] = sacc;
      return;
    }
#endif
#ifndef PV_DBG_NOWAVEEPI
    // Wave-private epilogue (late round 3, after conv_stream_f32.hip): every wave stages ITS 32-row slices through its own LDS
    // block -- a wave's LDS instructions execute in order, so no barrier is needed between its writes and its reads -- instead of
    // the workgroup staging one wave row at a time behind __syncthreads(): the four waves finish (and free their workgroup slot
    // for the next tile) independently.  Where the blocks fit the operand buffers; not for the narrow layout.
    constexpr int WSC = WTN + 4;                                 // staged row pitch of a wave's block (floats)
    constexpr bool WAVE_EPI = (NT / 64) * 32 * WSC <= 2 * BUF && WTN % 16 == 0 && (WTN / 4) <= 64 && 64 % (WTN / 4) == 0;
    if constexpr (WAVE_EPI) {
      if (wide && !(NARROW_OK && narrow)) {
        float* stg = lds_dyn + wave * (32 * WSC);
        constexpr int WC4 = WTN / 4, RPP = 64 / WC4, UPT = 32 / RPP;      // 16-byte units per staged row; rows per pass of the wave
        const int c4 = lane % WC4, r0 = lane / WC4, col = n0 + wn0 + c4 * 4;
        const bool col_ok = col < p.cout;
        const int colc = col_ok ? col : 0;
        const float4 bv = p.bias != nullptr ? premvos::ld4(p.bias + colc) : make_float4(0.f, 0.f, 0.f, 0.f);
        // (the K loop ends with a barrier: every wave is done reading the operand buffers)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const int mbase = m0 + wm0 + mi * 32;
          float4 rv[UPT];
          if (p.res != nullptr) {
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
              int m = mbase + r0 + i * RPP;
              m = m < M ? m : M - 1;
              rv[i] = premvos::ld4(p.res + (long)m * p.res_ps + colc);
            }
          }
#pragma unroll
          for (int ni = 0; ni < NTL; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              stg[row * WSC + ni * 32 + (lane & 31)] = acc[mi][ni][r];
            }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int i = 0; i < UPT; ++i) {
            const int row = r0 + i * RPP, m = mbase + row;
            float4 v = *reinterpret_cast<const float4*>(&stg[row * WSC + c4 * 4]);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (p.res != nullptr) { v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w; }
            if (p.act == PREMVOS_ACT_RELU) {
              v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            } else if (p.act == PREMVOS_ACT_LEAKY) {
              v.x = v.x > 0.f ? v.x : v.x * p.slope; v.y = v.y > 0.f ? v.y : v.y * p.slope;
              v.z = v.z > 0.f ? v.z : v.z * p.slope; v.w = v.w > 0.f ? v.w : v.w * p.slope;
            } else if (p.act == PREMVOS_ACT_SIGMOID) {
              v.x = 1.f / (1.f + expf(-v.x)); v.y = 1.f / (1.f + expf(-v.y)); v.z = 1.f / (1.f + expf(-v.z)); v.w = 1.f / (1.f + expf(-v.w));
            }
            if (m < M && col_ok) *reinterpret_cast<float4*>(p.out + (long)m * p.out_ps + col) = v;
          }
          __builtin_amdgcn_wave_barrier();
        }
        PV_TL(3);
        return;
      }
    }
#endif
    if (wide) {                                                  // kernel-uniform
      float* stg = lds_dyn;
#pragma unroll 1
      for (int wr = 0; wr < WM; ++wr) {
        if (NARROW_OK && narrow) {                                 // waves 2 wr, 2 wr + 1 hold rows [64 wr, 64 wr + 64), blocks 0 .. nvn-1
          if ((wave >> 1) == wr) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              if (j < nvn) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  const int row = (wave & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                  stg[row * EP + j * 32 + (lane & 31)] = acc[(j >> 1) % MT][(j & 1) % NTL][r];
                }
              }
            }
          }
        } else if (wave / WN == wr) {
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
#ifndef PV_DBG_OLDEPI
        if constexpr (UNITS % NT == 0 && NT % C4 == 0) {
          // Every unit of a thread has the same four columns (NT is a multiple of the units per row): the bias is one request, and
          // ALL the residual requests of the pass go out before anything waits for one -- written per unit under `if (m < M && ...)`
          // hipcc emitted one exec-masked block per unit with `s_waitcnt vmcnt(0)` behind each load (late round 3; rows / columns
          // past the tensor are requested at a clamped address and masked at the store).  Same sums in the same order.
          constexpr int UPT = UNITS / NT, RSTEP = NT / C4;
          const int c4 = tid % C4, row0 = tid / C4, col = n0 + c4 * 4;
          const bool col_ok = col < p.cout;
          const int colc = col_ok ? col : 0;
          const float4 bv = p.bias != nullptr ? premvos::ld4(p.bias + colc) : make_float4(0.f, 0.f, 0.f, 0.f);
          float4 rv[UPT];
          if (p.res != nullptr) {
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
              int m = m0 + wr * WTM + row0 + i * RSTEP;
              m = m < M ? m : M - 1;
              rv[i] = premvos::ld4(p.res + (long)m * p.res_ps + colc);
            }
          }
#pragma unroll
          for (int i = 0; i < UPT; ++i) {
            const int row = row0 + i * RSTEP, m = m0 + wr * WTM + row;
            float4 v = *reinterpret_cast<const float4*>(&stg[row * EP + c4 * 4]);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (p.res != nullptr) { v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w; }
            if (p.act == PREMVOS_ACT_RELU) {
              v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            } else if (p.act == PREMVOS_ACT_LEAKY) {
              v.x = v.x > 0.f ? v.x : v.x * p.slope; v.y = v.y > 0.f ? v.y : v.y * p.slope;
              v.z = v.z > 0.f ? v.z : v.z * p.slope; v.w = v.w > 0.f ? v.w : v.w * p.slope;
            } else if (p.act == PREMVOS_ACT_SIGMOID) {
              v.x = 1.f / (1.f + expf(-v.x)); v.y = 1.f / (1.f + expf(-v.y)); v.z = 1.f / (1.f + expf(-v.z)); v.w = 1.f / (1.f + expf(-v.w));
            }
            if (m < M && col_ok) *reinterpret_cast<float4*>(p.out + (long)m * p.out_ps + col) = v;
          }
          if (wr + 1 < WM) __syncthreads();
          continue;
        }
#endif
#pragma unroll
        for (int u = tid; u < UNITS; u += NT) {
          const int row = u / C4, c4 = u - row * C4;
          const int m = m0 + wr * WTM + row, col = n0 + c4 * 4;
          if (m < M && col < p.cout) {
            float4 v = *reinterpret_cast<const float4*>(&stg[row * EP + c4 * 4]);
            if (p.bias != nullptr) {
              const float4 bv = *reinterpret_cast<const float4*>(p.bias + col);
              v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            if (p.res != nullptr) {
              const float4 rv = *reinterpret_cast<const float4*>(p.res + (long)m * p.res_ps + col);
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
#ifdef PV_DBG_NOSTORE           // developer phase ablation: everything but the global store
            if (v.x == 12345.678f)
#endif
            // (non-temporal stores: +1.5 % on HBM-bound layers alone, nothing in the pipeline -- not used)
            *reinterpret_cast<float4*>(p.out + (long)m * p.out_ps + col) = v;
          }
        }
        if (wr + 1 < WM) __syncthreads();
      }
      PV_TL(3);
      return;
    }
  }
  // ---- epilogue: bias + residual + activation, 128-byte channel runs per pixel -------------
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

// Sum the split-K slabs in fixed order and apply the fused epilogue (bias, residual, activation, layout).
template <bool PIXSHUF>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const premvos_conv_desc p, const int splits, const int ncols, const int m_begin) {
  const int Mt = p.n * p.ho * p.wo - m_begin;      // rows [m_begin, M) were computed as k-slices
  const long total = (long)Mt * p.cout;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int col = idx % p.cout;
    const int ml = idx / p.cout;
    const int m = m_begin + ml;
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += p.workspace[((long)z * Mt + ml) * ncols + col];
    if (p.bias != nullptr) v += p.bias[col];
    if (p.res != nullptr) v += p.res[(long)m * p.res_ps + col];
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
}

}  // namespace

namespace premvos {
bool conv_wino_applicable(const premvos_conv_desc& d);        // conv_wino_f32.hip
long conv_wino_workspace_bytes(const premvos_conv_desc& d);
int conv_wino(const premvos_conv_desc& d, hipStream_t s);
int conv_wino_fused(const premvos_conv_desc& d, hipStream_t s);
bool conv_wino_fused_applicable(const premvos_conv_desc& d);
bool conv_wino4_applicable(const premvos_conv_desc& d);       // conv_wino4_f32.hip
long conv_wino4_workspace_bytes(const premvos_conv_desc& d);
int conv_wino4(const premvos_conv_desc& d, hipStream_t s);
bool conv_stream_applicable(const premvos_conv_desc& d);      // conv_stream_f32.hip
int conv_stream(const premvos_conv_desc& d, hipStream_t s);
bool conv_pwdma_applicable(const premvos_conv_desc& d);       // conv_pwdma_f32.hip
int conv_pwdma(const premvos_conv_desc& d, hipStream_t s);
bool conv_smalln_applicable(const premvos_conv_desc& d);      // conv_smalln_f32.hip
int conv_smalln(const premvos_conv_desc& d, hipStream_t s);
int launch_splitk_reduce(const premvos_conv_desc& d, int splits, int ncols, hipStream_t s, int m_begin) {
  const long total = ((long)d.n * d.ho * d.wo - m_begin) * d.cout;
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  if (d.out_mode == PREMVOS_OUT_PIXSHUF2)
    hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(g), dim3(256), 0, s, d, splits, ncols, m_begin);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(g), dim3(256), 0, s, d, splits, ncols, m_begin);
  return check_launch("splitk_reduce");
}
}  // namespace premvos

namespace {

// how many k-slices a layer is cut into (1 = no split): aim for >= ~3 workgroups per CU
inline int pick_splits(const premvos_conv_desc& d, int bm, int bn, int kb) {
  if (d.split_k > 0) return d.split_k;
  if (d.split_k < 0) return 1;
  const long tiles = (long)premvos::cdiv(d.n * d.ho * d.wo, bm) * premvos::cdiv(d.cout, bn);
  const int KT = premvos::cdiv(d.k_pad, kb);
  if (tiles >= 384 || KT * kb < 256) return 1;
  long s = (768 + tiles - 1) / tiles;
  if (s > KT * kb / 128) s = KT * kb / 128;
  if (s > 32) s = 32;
  return s < 2 ? 1 : (int)s;
}

template <typename K>
inline void allow_lds(K kernel, int bytes) {   // > 64 KB of LDS per workgroup needs the attribute
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// stage depth: 32 halves the barriers per MFMA (helps long-K MFMA-bound layers) at 1.8x the LDS per workgroup
// Measured on MI355X (tools/conv_bench.py): 32-deep stages are +5..28 % on long-K layers whose tile count fills
// whole waves of 2 workgroups/CU (72 KB LDS), and -8..20 % where 16-deep stages fit one more workgroup per CU
// (40 KB LDS -> 3/CU) and thereby avoid a second, mostly empty wave.  Model: time ~ waves(tiles, slots) / f(K).
inline int pick_kb(const premvos_conv_desc& d) {
  if (d.stage_k == 16 || d.stage_k == 32) return d.stage_k;
  const long M = (long)d.n * d.ho * d.wo;
  const long t128 = ((M + 127) / 128) * ((d.cout + 127) / 128);
  if (t128 < 512 || d.cout <= 96) return 16;            // pick_tile would not choose the 128x128 tile
  const int K = d.kh * d.kw * d.cin_pad;
  const double f = K >= 1024 ? 1.10 : K >= 512 ? 1.03 : 0.93;
  auto q = [](long tiles, long slots) { return (double)tiles / (double)(((tiles + slots - 1) / slots) * slots); };
  return f * q(t128, 512) > q(t128, 768) ? 32 : 16;
}

template <int BM, int BN, int WM, int WN, int KB = 16>
int launch_cfg(const premvos_conv_desc& d, hipStream_t s) {
  const int M = d.n * d.ho * d.wo;
  dim3 grid(premvos::cdiv(M, BM), premvos::cdiv(d.cout, BN));
  dim3 block(64 * WM * WN);
#ifndef PV_DBG_LDS_PAD
#define PV_DBG_LDS_PAD 0          // developer builds: extra dynamic LDS per workgroup = fewer workgroups per CU (occupancy experiments)
#endif
  constexpr int OPER = 2 * (BM + BN) * (KB + 4), STAGED = (BM / WM) * (BN + 4);     // floats: operand buffers | one staged wave row
  constexpr int LDS_BYTES = (OPER > STAGED ? OPER : STAGED) * (int)sizeof(float) + PV_DBG_LDS_PAD;
  static const bool attr_done = [] {            // once per instantiation, thread-safe (the file drivers launch from several threads)
    allow_lds(conv_igemm_f32_kernel<BM, BN, WM, WN, false, true, KB>, LDS_BYTES);
    allow_lds(conv_igemm_f32_kernel<BM, BN, WM, WN, true, false, KB>, LDS_BYTES);
    allow_lds(conv_igemm_f32_kernel<BM, BN, WM, WN, false, false, KB>, LDS_BYTES);
    allow_lds(conv_igemm_f32_kernel<BM, BN, WM, WN, false, true, KB, true>, LDS_BYTES);
    allow_lds(conv_igemm_f32_kernel<BM, BN, WM, WN, false, false, KB, true>, LDS_BYTES);
    return true;
  }();
  (void)attr_done;
  const bool pw = d.kh == 1 && d.kw == 1 && d.pt == 0 && d.pl == 0;
  int splits = pick_splits(d, BM, BN, KB);
  const int KT = premvos::cdiv(d.k_pad, KB);
  const int lds_bytes = LDS_BYTES;
  if (splits > 1) {
    const int kt_per = premvos::cdiv(KT, splits);
    splits = premvos::cdiv(KT, kt_per);
    const int ncols = grid.y * BN;
    const long need = (long)splits * M * ncols * sizeof(float);
    if (splits > 1 && d.workspace != nullptr && (long)d.workspace_bytes >= need) {
      grid.z = splits;
      if (pw)
        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, true, KB, true>), grid, block, lds_bytes, s, d, kt_per, 0);
      else
        hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, true, KB>), grid, block, lds_bytes, s, d, kt_per, 0);
      int rc = premvos::check_launch("conv_igemm_f32(split-k)");
      if (rc) return rc;
      return premvos::launch_splitk_reduce(d, splits, ncols, s, 0);
    }
    if (d.split_k > 0) return premvos::fail(PREMVOS_EINVAL, "conv2d: split_k=%d needs %ld workspace bytes", d.split_k, need);
  }
  // Tail split: the last `tail_m_tiles` rows of output tiles run as k-slices (+ fixed-order reduce), so that a partly
  // filled last wave of workgroups (e.g. 588 tiles on 256 CUs: 2 full waves + 76) is spread over the whole chip.
  int tail = 0, tsplits = 1, tkt_per = 0;
  if (d.tail_m_tiles > 0 && d.tail_split_k > 1 && d.tail_m_tiles < (int)grid.x) {
    tkt_per = premvos::cdiv(KT, d.tail_split_k);
    tsplits = premvos::cdiv(KT, tkt_per);
    tail = tsplits > 1 ? d.tail_m_tiles : 0;
    if (tail) {
      const long mt = (long)M - (long)(grid.x - tail) * BM;
      const long need = (long)tsplits * mt * grid.y * BN * (long)sizeof(float);
      if (d.workspace == nullptr || (long)d.workspace_bytes < need)
        return premvos::fail(PREMVOS_EINVAL, "conv2d: tail split %dx%d needs %ld workspace bytes", d.tail_m_tiles,
                             d.tail_split_k, need);
    }
  }
  const dim3 gmain(grid.x - tail, grid.y);
  if (d.out_mode == PREMVOS_OUT_PIXSHUF2)
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, true, false, KB>), gmain, block, lds_bytes, s, d, 0, 0);
  else if (pw)
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, false, KB, true>), gmain, block, lds_bytes, s, d, 0, 0);
  else
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, false, KB>), gmain, block, lds_bytes, s, d, 0, 0);
  int rc = premvos::check_launch("conv_igemm_f32");
  if (rc || !tail) return rc;
  const int mt0 = (int)grid.x - tail;
  if (pw)
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, true, KB, true>), dim3(tail, grid.y, tsplits), block,
                       lds_bytes, s, d, tkt_per, mt0);
  else
    hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false, true, KB>), dim3(tail, grid.y, tsplits), block,
                       lds_bytes, s, d, tkt_per, mt0);
  rc = premvos::check_launch("conv_igemm_f32(tail split-k)");
  if (rc) return rc;
  return premvos::launch_splitk_reduce(d, tsplits, grid.y * BN, s, mt0 * BM);
}

template <int BM, int BN, int WM, int WN>
long ws_cfg(const premvos_conv_desc& d) {
  const int kb = pick_kb(d);
  int splits = pick_splits(d, BM, BN, kb);
  const int KT = premvos::cdiv(d.k_pad, kb);
  const long M = (long)d.n * d.ho * d.wo;
  const long ncols = (long)premvos::cdiv(d.cout, BN) * BN;
  if (splits > 1) {
    const int kt_per = premvos::cdiv(KT, splits);
    splits = premvos::cdiv(KT, kt_per);
    if (splits > 1) return (long)splits * M * ncols * (long)sizeof(float);
  }
  const long mtiles = premvos::cdiv((int)M, BM);
  if (d.tail_m_tiles > 0 && d.tail_split_k > 1 && d.tail_m_tiles < mtiles) {
    const int kt_per = premvos::cdiv(KT, d.tail_split_k);
    const int ts = premvos::cdiv(KT, kt_per);
    if (ts > 1) return (long)ts * (M - (mtiles - d.tail_m_tiles) * BM) * ncols * (long)sizeof(float);
  }
  return 0;
}

inline void pick_tile(const premvos_conv_desc& d, int* bm, int* bn) {
  const int M = d.n * d.ho * d.wo;
  if (d.tile_hint) {
    *bm = d.tile_hint >> 16;
    *bn = d.tile_hint & 0xffff;
    return;
  }
  *bn = d.cout <= 32 ? 32 : d.cout <= 64 ? 64 : d.cout <= 96 ? 96 : 128;
  const long blocks128 = (long)premvos::cdiv(M, 128) * premvos::cdiv(d.cout, *bn);
  *bm = blocks128 >= 512 ? 128 : 64;
  if (*bn == 96 && *bm == 64) *bn = 128;  // (64,96) is not instantiated
}

}  // namespace

// What every dense-conv entry point requires of a descriptor (premvos_conv2d_f32, premvos_conv_wino4_slab_f32).
int premvos::conv_desc_check(const premvos_conv_desc& d) {
  PV_REQUIRE(d.in && d.wgt && d.out, "conv2d: null tensor pointer");
  PV_REQUIRE(d.n > 0 && d.h > 0 && d.w > 0 && d.cin > 0 && d.ho > 0 && d.wo > 0 && d.cout > 0,
             "conv2d: non-positive dimension");
  PV_REQUIRE(d.kh > 0 && d.kw > 0 && d.sh > 0 && d.sw > 0 && d.dh > 0 && d.dw > 0, "conv2d: bad kernel geometry");
  PV_REQUIRE(d.cin_pad == (d.cin + 3) / 4 * 4, "conv2d: cin_pad must be roundup(cin,4)");
  PV_REQUIRE(d.k_pad % BK == 0 && d.k_pad >= d.kh * d.kw * d.cin_pad, "conv2d: bad k_pad");
  PV_REQUIRE(d.precision == PREMVOS_PREC_F32 || d.precision == PREMVOS_PREC_BF16 || d.precision == PREMVOS_PREC_BF16X3,
             "conv2d: bad precision");
  PV_REQUIRE(d.cout_pad % 32 == 0 && d.cout_pad >= d.cout, "conv2d: bad cout_pad");
  PV_REQUIRE(d.in_ps % 4 == 0 && d.in_ps >= d.cin_pad, "conv2d: input pixel stride must be a multiple of 4 and >= cin_pad");
  PV_REQUIRE(premvos::aligned16(d.in) && premvos::aligned16(d.wgt), "conv2d: in/wgt must be 16-byte aligned");
  PV_REQUIRE(d.out_ps >= (d.out_mode == PREMVOS_OUT_PIXSHUF2 ? d.cout_ps : d.cout), "conv2d: out_ps < cout");
  PV_REQUIRE(d.res == nullptr || d.res_ps >= d.cout, "conv2d: res_ps < cout");
  PV_REQUIRE(d.res == nullptr || d.out_mode == PREMVOS_OUT_NHWC, "conv2d: residual needs NHWC output");
  PV_REQUIRE(d.act >= PREMVOS_ACT_NONE && d.act <= PREMVOS_ACT_SIGMOID,
             "conv2d: bad activation");
  if (d.out_mode == PREMVOS_OUT_PIXSHUF2)
    PV_REQUIRE(d.cout_ps > 0 && d.cout == 4 * d.cout_ps, "conv2d: PIXSHUF2 needs cout == 4*cout_ps");
  else
    PV_REQUIRE(d.out_mode == PREMVOS_OUT_NHWC, "conv2d: bad out_mode");
  PV_REQUIRE((long)d.n * d.ho * d.wo < (1L << 31), "conv2d: too many output pixels");
  return PREMVOS_OK;
}

extern "C" int premvos_conv2d_f32(const premvos_conv_desc* dp, void* stream) {
  PV_REQUIRE(dp != nullptr, "conv2d: null descriptor");
  const premvos_conv_desc& d = *dp;
  if (const int rc = premvos::conv_desc_check(d)) return rc;

  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d.precision != PREMVOS_PREC_F32) return premvos::conv2d_bf16(d, s);
  if (d.tile_hint == 2) {      // Winograd F(2x2,3x3): chosen per layer by the host's plan-time autotuner
    PV_REQUIRE(premvos::conv_wino_applicable(d), "conv2d: Winograd needs a 3x3 / stride 1 / dilation 1 fp32 layer with cout %% 4 == 0, "
               "symmetric padding and packed filter transforms (wgt_wino)");
    return premvos::conv_wino(d, s);
  }
  if (d.tile_hint == 4) {      // Winograd F(4x4,3x3) for K-rich layers (input transform, 36 batched GEMMs, output transform)
    PV_REQUIRE(premvos::conv_wino4_applicable(d), "conv2d: Winograd F(4x4,3x3) needs a 3x3 / stride 1 fp32 layer (dilated: equal in both directions) with "
               "cout %% 4 == 0, symmetric padding and packed filter transforms (wgt_wino4)");
    return premvos::conv_wino4(d, s);
  }
  if (d.tile_hint == 3) {      // ... the slab-free variant of it (no workspace, one kernel); stage_k = block id
    PV_REQUIRE(premvos::conv_wino_fused_applicable(d), "conv2d: Winograd needs a 3x3 / stride 1 fp32 layer with cout %% 4 == 0, symmetric "
               "padding (= the dilation for atrous layers) and packed filter transforms (wgt_wino)");
    return premvos::conv_wino_fused(d, s);
  }
  if (d.tile_hint == 5) {      // short-K pointwise layers: persistent workgroups, weights resident in LDS (conv_stream_f32.hip); same sums
    PV_REQUIRE(premvos::conv_stream_applicable(d), "conv2d: the streaming pointwise kernel needs a 1x1 / stride 1 fp32 layer with cin = 64 or 128 "
               "(= k_pad), cout %% 128 == 0 (<= 512) and 16-byte aligned pixels");
    return premvos::conv_stream(d, s);
  }
  if (d.tile_hint == 6) {      // pointwise layers, operands staged by LDS-DMA (conv_pwdma_f32.hip); same sums as the implicit GEMM
    PV_REQUIRE(premvos::conv_pwdma_applicable(d), "conv2d: the LDS-DMA pointwise kernel needs a 1x1 fp32 layer without padding, k_pad >= 32, "
               "cout %% 4 == 0 and 16-byte aligned pixels");
    return premvos::conv_pwdma(d, s);
  }
  // 1- and 2-channel heads: per-pixel dot products, not GEMM tiles (tile_hint 0 = auto, 1 = forced; any other hint
  // keeps them on the MFMA kernel, which is what the autotuner compares against)
  if ((d.tile_hint == 0 || d.tile_hint == 1) && premvos::conv_smalln_applicable(d)) return premvos::conv_smalln(d, s);
  PV_REQUIRE(d.tile_hint != 1, "conv2d: the direct small-N kernel does not apply to this layer");
  int bm, bn;
  pick_tile(d, &bm, &bn);
  if (pick_kb(d) == 32) {
    switch ((bm << 16) | bn) {
      case (256 << 16) | 128: return launch_cfg<256, 128, 4, 2, 32>(d, s);
      case (128 << 16) | 128: return launch_cfg<128, 128, 2, 2, 32>(d, s);
      case (128 << 16) | 64: return launch_cfg<128, 64, 2, 2, 32>(d, s);
      case (64 << 16) | 128: return launch_cfg<64, 128, 2, 2, 32>(d, s);
      default: break;   // other tiles only exist with 16-deep stages
    }
  }
  switch ((bm << 16) | bn) {
    case (256 << 16) | 128: return launch_cfg<256, 128, 4, 2>(d, s);      // 8 waves of 64x64: half the B staging per MFMA
    case (128 << 16) | 128: return launch_cfg<128, 128, 2, 2>(d, s);
    // four waves of 128 x 64 (hint 256x129; round 5): half the A fragment reads per MFMA of the 2 x 2 layout's B side, two workgroups
    // per CU -- wins short-K layers with few column tiles (profiles/r05_wave_tile_ab.txt; the 128 x 256 / 256 x 256 developer tiles of
    // that A/B lost on every layer of the nets and were removed)
    case (256 << 16) | 129: return launch_cfg<256, 128, 2, 2>(d, s);
    case (128 << 16) | 96: return launch_cfg<128, 96, 4, 1>(d, s);
    case (128 << 16) | 64: return launch_cfg<128, 64, 2, 2>(d, s);
    case (128 << 16) | 32: return launch_cfg<128, 32, 4, 1>(d, s);
    case (64 << 16) | 128: return launch_cfg<64, 128, 2, 2>(d, s);
    case (64 << 16) | 64: return launch_cfg<64, 64, 2, 2>(d, s);
    case (64 << 16) | 32: return launch_cfg<64, 32, 2, 1>(d, s);
    default: return premvos::fail(PREMVOS_EINVAL, "conv2d: no tile config %dx%d", bm, bn);
  }
}

extern "C" int64_t premvos_conv2d_workspace_bytes(const premvos_conv_desc* dp) {
  if (dp == nullptr || dp->k_pad <= 0 || dp->n <= 0 || dp->ho <= 0 || dp->wo <= 0 || dp->cout <= 0) return 0;
  if (dp->precision != PREMVOS_PREC_F32) return premvos::conv2d_bf16_workspace_bytes(*dp);
  if (dp->tile_hint == 2) return premvos::conv_wino_applicable(*dp) ? premvos::conv_wino_workspace_bytes(*dp) : 0;
  if (dp->tile_hint == 3) return 0;
  if (dp->tile_hint == 4) return premvos::conv_wino4_applicable(*dp) ? premvos::conv_wino4_workspace_bytes(*dp) : 0;
  if (dp->tile_hint == 5 || dp->tile_hint == 6) return 0;
  if ((dp->tile_hint == 0 || dp->tile_hint == 1) && premvos::conv_smalln_applicable(*dp)) return 0;
  int bm, bn;
  pick_tile(*dp, &bm, &bn);
  switch ((bm << 16) | bn) {
    case (256 << 16) | 128: return ws_cfg<256, 128, 4, 2>(*dp);
    case (128 << 16) | 128: return ws_cfg<128, 128, 2, 2>(*dp);
    case (256 << 16) | 129: return ws_cfg<256, 128, 2, 2>(*dp);
    case (128 << 16) | 96: return ws_cfg<128, 96, 4, 1>(*dp);
    case (128 << 16) | 64: return ws_cfg<128, 64, 2, 2>(*dp);
    case (128 << 16) | 32: return ws_cfg<128, 32, 4, 1>(*dp);
    case (64 << 16) | 128: return ws_cfg<64, 128, 2, 2>(*dp);
    case (64 << 16) | 64: return ws_cfg<64, 64, 2, 2>(*dp);
    case (64 << 16) | 32: return ws_cfg<64, 32, 2, 1>(*dp);
    default: return 0;
  }
}

extern "C" const char* premvos_last_error(void) { return premvos::g_err; }
namespace {
// Calibration: nothing but independent fp32 MFMAs (4 accumulators per wave, operands in registers) -- the issue-rate
// ceiling of v_mfma_f32_32x32x2_f32 on this GPU under its own power/clock management.  bench.py reports it next to the
// datasheet peak so that roofline.frac can be read against what the silicon sustains.
__global__ __launch_bounds__(256) void mfma_f32_calibrate_kernel(long iters, float* sink) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = 1.0f + (float)threadIdx.x * 1e-6f, b = 1.0f - (float)threadIdx.x * 1e-6f;
  for (long it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    a = -a;
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[i][r];
  if (t == 123.456f) sink[0] = t;          // keeps the loop alive
}
}  // namespace

namespace {
// The same loop on CHANGING operands (round 5): eight pseudo-random A and B registers per lane, a different pair for every MFMA.  The
// fp32 matrix pipe's power depends on the switching activity of its operands (profiles/r05_power_data.txt: the implicit GEMM draws
// 0.83 kW on all-zero operands and 1.35 ... 1.40 kW -- the socket cap -- on random ones, where the shader clock gives way by 3 ... 5 %);
// the constant operands of the kernel above never meet the cap.  Timed over a second or more this is the fp32 MFMA rate a box
// sustains ON REAL DATA -- the ceiling a GEMM's loop can be priced against beside the nominal 157.3.
__global__ __launch_bounds__(256) void mfma_f32_calibrate_random_kernel(long iters, float* sink) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
  for (int j = 0; j < 8; ++j) {                 // values in (-1, 1) with full mantissas, ~N(0, 0.33)-like spread
    h = h * 1664525u + 1013904223u; a[j] = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
    h = h * 1664525u + 1013904223u; b[j] = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23));
  }
  for (long it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(4 * u + i) & 7], b[(4 * u + i + u) & 7], acc[i], 0, 0, 0);
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[i][r];
  if (t == 123.456f) sink[0] = t;
}
}  // namespace

extern "C" int premvos_mfma_f32_calibrate_random(int64_t iters, int32_t blocks, float* sink, void* stream) {
  PV_REQUIRE(iters > 0 && blocks > 0 && sink != nullptr, "mfma_calibrate_random: bad arguments");
  hipLaunchKernelGGL(mfma_f32_calibrate_random_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), (long)iters, sink);
  return premvos::check_launch("mfma_f32_calibrate_random");
}

extern "C" int premvos_mfma_f32_calibrate(int64_t iters, int32_t blocks, float* sink, void* stream) {
  PV_REQUIRE(iters > 0 && blocks > 0 && sink != nullptr, "mfma_calibrate: bad arguments");
  hipLaunchKernelGGL(mfma_f32_calibrate_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), (long)iters,
                     sink);
  return premvos::check_launch("mfma_f32_calibrate");
}

namespace {
// Calibration: a float4 copy, ONE 16-byte word per thread on a flat grid of n / 256 workgroups -- the HBM rate this GPU sustains on
// a streaming read + write (6.1 TB/s; the same copy as a grid-stride loop of 4096 persistent workgroups with four loads in flight
// reaches only 4.3, non-temporal accesses 6.5: tools/dev/hbm_copy_variants.hip).  bench.py reports it beside the MFMA calibration so
// that a line carries its own box factor for the memory system too.
__global__ __launch_bounds__(256) void hbm_copy_calibrate_kernel(const float4* __restrict__ src, float4* __restrict__ dst, const long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
}  // namespace

extern "C" int premvos_hbm_copy_calibrate(const void* src, void* dst, int64_t n_float4, void* stream) {
  PV_REQUIRE(src != nullptr && dst != nullptr && n_float4 > 0 && n_float4 < (1LL << 38), "hbm_copy_calibrate: bad arguments");
  hipLaunchKernelGGL(hbm_copy_calibrate_kernel, dim3((unsigned)((n_float4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float4*>(src), static_cast<float4*>(dst), (long)n_float4);
  return premvos::check_launch("hbm_copy_calibrate");
}

namespace {
// order-independent 64-bit digest of a strided pixel-major window: sum over (pixel, channel) of word * (2 * position + 1)
__global__ __launch_bounds__(256) void digest_kernel(const unsigned* __restrict__ p, const long pixels, const int c, const int ps,
                                                     unsigned long long* out) {
  unsigned long long acc = 0;
  const long total = pixels * c;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long pix = i / c;
    const int ch = (int)(i - pix * c);
    acc += (unsigned long long)p[pix * ps + ch] * (unsigned long long)(2 * i + 1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
}  // namespace

extern "C" int premvos_digest_u64(const void* buf, int64_t pixels, int32_t c, int32_t ps, void* out_u64, void* stream) {
  PV_REQUIRE(buf != nullptr && out_u64 != nullptr && pixels > 0 && c > 0 && ps >= c, "digest: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(out_u64, 0, 8, s) != hipSuccess) return premvos::fail(PREMVOS_ELAUNCH, "digest: memset failed");
  const long total = pixels * c;
  const int blocks = (int)(total / 256 / 8 + 1 < 2048 ? total / 256 / 8 + 1 : 2048);
  hipLaunchKernelGGL(digest_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const unsigned*>(buf), (long)pixels, c, ps,
                     static_cast<unsigned long long*>(out_u64));
  return premvos::check_launch("digest");
}

extern "C" int premvos_abi_version(void) { return 18; }   // bump with every change of include/premvos_hip.h
