// Pointwise (1x1, any stride, no padding) fp32 conv as a 128 x 128-tile GEMM whose operands are staged by LDS-DMA
// (`global_load_lds_dwordx4`: L2 -> LDS, no staging registers, no ds_write) -- round 5, premvos_conv2d_f32 with tile_hint 6.
//
// Why: the implicit GEMM (conv_igemm_f32.hip) stages a 16-deep K step through registers: four 16-byte requests per thread, a
// `s_waitcnt vmcnt(0)`, four `ds_write_b128` (padded rows) and the workgroup barrier.  Round 1's ablation ladder priced exactly that
// step: pure MFMAs 154 TFLOP/s -> + fragment reads 150 -> + barrier 147 -> + the ds_write staging 137; and every reordering of it
// tried in rounds 3 and 5 lost (profiles/r05_igemm_investigation.md).  Here the step does not exist: a wave's DMA instruction drops
// 64 x 16 bytes lane-linearly into LDS = 16 rows x 64 bytes of a 16-deep stage, THREE stage buffers, the DMA of stage kt + 2 issued
// while stage kt is multiplied -- a request has two whole stages to land and the only wait in the loop is a counted `s_waitcnt
// vmcnt` that is normally already satisfied.  One raw `s_barrier` per stage (never __syncthreads: its fence would drain the DMA queue).
//
// Bank conflicts: rows are 64 bytes, unpadded (a DMA lands lane-linearly), so the XOR swizzle of conv_bf16x3_s8.hip is applied on the
// SOURCE side -- the lane that lands at physical chunk c of row r fetches logical chunk c ^ ((r >> 2) & 3) -- and again on the
// ds_read_b128 fragment reads: the 16 rows of a 16-lane group hit 16 different 16-byte slots.
//
// Arithmetic: v_mfma_f32_32x32x2_f32; an output element's products are summed exactly as in conv_igemm_f32.hip (8-deep groups in
// ascending k, lane half l >> 5 holds k = 4 (l >> 5) + {0..3}, MFMA e pairs k = e and 4 + e; the all-zero trailing group of the last
// stage is skipped; bias, then residual, then activation): BIT-IDENTICAL results, so the choice between the kernels is an
// order-neutral knob of the tune table (premvos_amd/ops.py: numerics_key).
#include "common.h"
#include <type_traits>

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int BM = 128, BN = 128, NT = 256, NW = 4;
constexpr int RB = 64;                         // bytes per row per stage (16 floats)
constexpr int STAGE = (BM + BN) * RB;          // 16 KB
constexpr int NSTAGE = 3;
constexpr int WSC = 64 + 4;                    // staged row pitch of a wave's epilogue block (floats)
constexpr int LDS_BYTES = NSTAGE * STAGE;      // 48 KB: three workgroups per CU
static_assert(NW * 32 * WSC * 4 <= LDS_BYTES, "the wave-private epilogue blocks live in the operand buffers");

__device__ __attribute__((aligned(16))) unsigned int g_pw_zero_page[64];      // the source of chunks beyond the layer's channels

__device__ __forceinline__ void dma16(const char* g, char* l) {
#if defined(__HIP_DEVICE_COMPILE__)          // (hipcc parses kernels for the host too, where the gfx950 builtin does not exist)
  __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
#endif
}

__device__ __forceinline__ int SW(const int r) { return (r >> 2) & 3; }

__global__ __launch_bounds__(NT, 3) void conv_pwdma_f32_kernel(const premvos_conv_desc p, const int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: DMA bases (M0) and wave offsets stay in SGPRs
  const int wm = wave >> 1, wn = wave & 1;
  const int M = p.n * p.ho * p.wo;
  const int v = premvos::xcd_contiguous(blockIdx.x, gridDim.x);       // m-tile major: the column tiles of one A tile share an L2
  const int tile_m = v / n_tiles, tile_n = v - tile_m * n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- DMA source maps: piece q = rows [16 q, 16 q + 16) of an operand; wave w issues pieces w and w + 4 of A and of B.
  // Lane l lands at row 16 q + (l >> 2), physical chunk l & 3, and therefore FETCHES logical chunk (l & 3) ^ SW(row).
  const bool pw_unit = p.sh == 1 && p.sw == 1 && p.ho == p.h && p.wo == p.w;      // kernel-uniform
  const char* a_src[2];
  const char* b_src[2];
  int a_k[2];                      // first k (floats) of the lane's chunk inside a stage
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave + i * NW) * 16 + (lane >> 2);
    const int lc = (lane & 3) ^ SW(row);
    int m = m0 + row;
    m = m < M ? m : M - 1;                                 // rows past M: clamped, multiplied, never stored
    if (pw_unit) {                                         // unit stride: pixel m of the input, no (image, y, x) split
      a_src[i] = reinterpret_cast<const char*>(p.in + (long)m * p.in_ps + lc * 4);
    } else {
      const int hw = p.ho * p.wo;
      const int n = m / hw, rem = m - n * hw, oy = rem / p.wo, ox = rem - oy * p.wo;
      a_src[i] = reinterpret_cast<const char*>(p.in + ((long)(n * p.h + oy * p.sh) * p.w + ox * p.sw) * p.in_ps + lc * 4);
    }
    a_k[i] = lc * 4;
    int c = n0 + row;
    c = c < p.cout_pad ? c : p.cout_pad - 1;               // columns past cout_pad: clamped, multiplied, never stored
    b_src[i] = reinterpret_cast<const char*>(p.wgt + (long)c * p.k_pad + lc * 4);
  }
  const char* zero = reinterpret_cast<const char*>(g_pw_zero_page);
  const int KT = p.k_pad / 16;
  // every stage but the matrix's last holds only k < cin_pad (k_pad - cin_pad < 16): plain requests; the last one selects the zero
  // page for chunks beyond the layer's channels (the weights are zero there, but the bytes behind a pixel's channels need not be finite)
  auto issue = [&](const int kt, const int buf) {
    char* sb = lds + buf * STAGE;
    const long koff = (long)kt * RB;
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(a_src[i] + koff, sb + (wave + i * NW) * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(b_src[i] + koff, sb + BM * RB + (wave + i * NW) * 1024);
  };
  auto issue_last = [&](const int kt, const int buf) {
    char* sb = lds + buf * STAGE;
    const long koff = (long)kt * RB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const char* src = a_src[i] + koff;
      src = kt * 16 + a_k[i] < p.cin_pad ? src : zero;
      dma16(src, sb + (wave + i * NW) * 1024);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) dma16(b_src[i] + koff, sb + BM * RB + (wave + i * NW) * 1024);
  };
  auto issue_any = [&](const int kt, const int buf) {
    if (kt + 1 < KT) issue(kt, buf);
    else issue_last(kt, buf);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---- fragment reads.  Row (lane & 31) of a 32-row block, logical chunk 2 h + (lane >> 5) of the 8-deep group h.  They are issued
  // as inline `ds_read_b128` with hand-counted `s_waitcnt lgkmcnt`: a compiler-visible LDS read behind an LDS-DMA instruction makes
  // hipcc insert `s_waitcnt vmcnt(0)` in front of it (it cannot prove that the DMA in flight writes ANOTHER stage buffer), which
  // drained the DMA queue in the middle of every stage (first version of this kernel: 136.7 instead of 140.8 TFLOP/s at K = 3072).
  // The waits carry the fragment registers as in / out operands, so no MFMA that reads them can be scheduled above its wait.
  using v4f = __attribute__((ext_vector_type(4))) float;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const int frow = lane & 31, fsw = SW(frow), fg = lane >> 5;
  const unsigned fo[2] = {(unsigned)(frow * RB + (((0 + fg) ^ fsw) << 4)), (unsigned)(frow * RB + (((2 + fg) ^ fsw) << 4))};
  const unsigned fa = lds0 + wm * 64 * RB, fb = lds0 + BM * RB + wn * 64 * RB;
  v4f af[2][2], bf[2][2];                                  // [set][block]
  auto ldfrag = [&](const int set, const int buf, const int h) {
    const unsigned a = fa + buf * STAGE + fo[h], b = fb + buf * STAGE + fo[h];
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:2048"
                 : "=&v"(af[set][0]), "=&v"(af[set][1]), "=&v"(bf[set][0]), "=&v"(bf[set][1])
                 : "v"(a), "v"(b)
                 : "memory");
  };
  // lgkmcnt(n): the reads of `set` (and everything issued before them) have returned
  auto wait0 = [&](const int set) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[set][0]), "+v"(af[set][1]), "+v"(bf[set][0]), "+v"(bf[set][1])::"memory");
  };
  auto wait4 = [&](const int set) {
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[set][0]), "+v"(af[set][1]), "+v"(bf[set][0]), "+v"(bf[set][1])::"memory");
  };
  auto mfma_rows = [&](const int set, const int mi0, const int mi1) {
#pragma unroll
    for (int mi = mi0; mi < mi1; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].x, bf[set][ni].x, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].y, bf[set][ni].y, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].z, bf[set][ni].z, acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].w, bf[set][ni].w, acc[mi][ni], 0, 0, 0);
      }
  };

  // ---- the K loop.  Top of iteration kt: this wave's pieces of stage kt have landed (the four of stage kt + 1 may stay in flight),
  // then ONE barrier: every wave's pieces are in, and every wave is done reading buffer (kt - 1) % 3 -- the one the DMA of stage
  // kt + 2 overwrites.  The last row block of a stage's second group is multiplied BEHIND the next barrier, under the first
  // fragment reads of the next stage (as in conv_igemm_f32.hip).
  issue_any(0, 0);
  if (KT > 1) issue_any(1, 1);
  const int kreal = p.cin_pad;
  const int h_last = (kreal - (KT - 1) * 16 + 7) / 8 >= 2 ? 2 : 1;        // 8-deep groups of the last stage that hold real k
  int buf = 0;
  for (int kt = 0; kt + 1 < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ldfrag(0, buf, 0);
    if (kt > 0) mfma_rows(1, 1, 2);                        // (the previous stage's last row block)
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < KT) {
      int nb = buf + 2;
      nb = nb >= NSTAGE ? nb - NSTAGE : nb;
      issue_any(kt + 2, nb);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    wait0(0);
    mfma_rows(0, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    ldfrag(1, buf, 1);                                     // (the second group's reads go out under the first group's MFMAs)
    mfma_rows(0, 1, 2);
    __builtin_amdgcn_sched_barrier(0);                     // (else the wait is hoisted above those MFMAs)
    wait0(1);
    mfma_rows(1, 0, 1);
    __builtin_amdgcn_s_setprio(0);
    buf = buf + 1 == NSTAGE ? 0 : buf + 1;
  }
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ldfrag(0, buf, 0);
    if (KT > 1) mfma_rows(1, 1, 2);
    if (h_last == 2) {
      ldfrag(1, buf, 1);
      wait4(0);
      mfma_rows(0, 0, 2);
      __builtin_amdgcn_sched_barrier(0);
      wait0(1);
      mfma_rows(1, 0, 2);
    } else {
      wait0(0);
      mfma_rows(0, 0, 2);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // the operand buffers become the epilogue's staging blocks

  // ---- epilogue: conv_igemm_f32.hip's wave-private wide form (32-row slices of the wave's 64 x 64 block through its own LDS block;
  // 16 bytes of a pixel's channel run per lane; bias, residual, activation at read-back)
  float* stg = reinterpret_cast<float*>(lds) + wave * (32 * WSC);
  constexpr int WC4 = 16, RPP = 4, UPT = 8;
  const int wm0 = wm * 64, wn0 = wn * 64;
  const int c4 = lane % WC4, r0 = lane / WC4, col = n0 + wn0 + c4 * 4;
  const bool col_ok = col < p.cout;
  const int colc = col_ok ? col : 0;
  const float4 bv = p.bias != nullptr ? premvos::ld4(p.bias + colc) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
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
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        stg[row * WSC + ni * 32 + (lane & 31)] = acc[mi][ni][r];
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
      const int row = r0 + i * RPP, m = mbase + row;
      float4 v4 = *reinterpret_cast<const float4*>(&stg[row * WSC + c4 * 4]);
      v4.x += bv.x; v4.y += bv.y; v4.z += bv.z; v4.w += bv.w;
      if (p.res != nullptr) { v4.x += rv[i].x; v4.y += rv[i].y; v4.z += rv[i].z; v4.w += rv[i].w; }
      if (p.act == PREMVOS_ACT_RELU) {
        v4.x = v4.x > 0.f ? v4.x : 0.f; v4.y = v4.y > 0.f ? v4.y : 0.f; v4.z = v4.z > 0.f ? v4.z : 0.f; v4.w = v4.w > 0.f ? v4.w : 0.f;
      } else if (p.act == PREMVOS_ACT_LEAKY) {
        v4.x = v4.x > 0.f ? v4.x : v4.x * p.slope; v4.y = v4.y > 0.f ? v4.y : v4.y * p.slope;
        v4.z = v4.z > 0.f ? v4.z : v4.z * p.slope; v4.w = v4.w > 0.f ? v4.w : v4.w * p.slope;
      } else if (p.act == PREMVOS_ACT_SIGMOID) {
        v4.x = 1.f / (1.f + expf(-v4.x)); v4.y = 1.f / (1.f + expf(-v4.y)); v4.z = 1.f / (1.f + expf(-v4.z)); v4.w = 1.f / (1.f + expf(-v4.w));
      }
      if (m < M && col_ok) *reinterpret_cast<float4*>(p.out + (long)m * p.out_ps + col) = v4;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

namespace premvos {

// 1x1 taps, no padding, NHWC output, the wide epilogue's alignment conditions, a K of at least two stages
bool conv_pwdma_applicable(const premvos_conv_desc& d) {
  return d.precision == PREMVOS_PREC_F32 && d.kh == 1 && d.kw == 1 && d.pt == 0 && d.pl == 0 && d.out_mode == PREMVOS_OUT_NHWC &&
         d.ho == (d.h - 1) / d.sh + 1 && d.wo == (d.w - 1) / d.sw + 1 && d.k_pad >= 32 && d.k_pad % 16 == 0 && d.cin_pad % 4 == 0 &&
         d.cin_pad <= d.k_pad && d.k_pad - d.cin_pad < 16 && d.in_ps >= d.cin_pad && (d.cout & 3) == 0 && (d.out_ps & 3) == 0 &&
         (d.in_ps & 3) == 0 && aligned16(d.in) && aligned16(d.out) && aligned16(d.wgt) &&
         (d.res == nullptr || ((d.res_ps & 3) == 0 && aligned16(d.res))) && (d.bias == nullptr || aligned16(d.bias)) &&
         (long)d.n * d.h * d.w * d.in_ps < (1L << 40);
}

int conv_pwdma(const premvos_conv_desc& d, hipStream_t s) {
  static const bool attr_done = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pwdma_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    return true;
  }();
  (void)attr_done;
  const long M = (long)d.n * d.ho * d.wo;
  const long m_tiles = (M + BM - 1) / BM;
  const int n_tiles = cdiv(d.cout, BN);
  if (m_tiles * n_tiles >= (1L << 31)) return fail(PREMVOS_EINVAL, "conv2d(pwdma): too many tiles");
  hipLaunchKernelGGL(conv_pwdma_f32_kernel, dim3((unsigned)(m_tiles * n_tiles)), dim3(NT), LDS_BYTES, s, d, n_tiles);
  return check_launch("conv_pwdma_f32");
}

}  // namespace premvos
