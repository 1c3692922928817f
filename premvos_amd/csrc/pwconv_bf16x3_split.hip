// Pointwise (1x1 / stride 1) convolution on the gfx950 bf16 matrix pipe with fp32-class accuracy ("bf16x3":
// a*b ~= hi(a)hi(b) + hi(a)lo(b) + lo(a)hi(b), fp32 accumulate) for an input that its PRODUCER already split.
//
// The bf16x3 mode of conv_igemm_bf16.hip keeps fp32 activations in HBM and splits them into bf16 hi / lo while it stages
// them -- once per column tile (6 times for a 728-wide layer), with ~10 VALU instructions per staged float4, behind a
// one-stage register prefetch: L2- and VALU-bound at ~1.2x the fp32 kernel.  Here the producer (premvos_dwconv3x3_f32 with
// PREMVOS_ACT_SPLIT_BF16: the depthwise half of every separable conv of the refinement net) stores, IN PLACE of each group of
// four floats, the 16 bytes {hi(4 x bf16), lo(4 x bf16)}: same buffer, same pixel stride, same bytes per element, split once.
// This kernel then stages pure bf16: a thread's 32-byte request is 8 channels' hi AND lo parts (two 16-byte units, re-paired by
// register renaming), a row's 32-deep stage is one 128-byte line, and nothing but loads, LDS stores and MFMAs is left in the loop.
//
//   M = pixels, N = cout, K = cin padded to 32;  tile 128 x 128, four waves (2 x 2) of 64 x 64, 32-deep stages,
//   LDS: {A_hi, A_lo, B_hi, B_lo} x 128 rows x (64 + 16) B, double-buffered = 80 KB -> two workgroups per CU;
//   v_mfma_f32_32x32x16_bf16: lane l holds the 8 consecutive k [8 (l >> 5), +8) of row l & 31 (one ds_read_b128).
//
// Weights: the bf16 hi / lo matrices ops.pack_conv(precision="bf16x3") already makes ([cout_pad][k_pad], k_pad % 32 == 0, zero
// padded).  Epilogue: fp32 out = act(acc + bias (+ residual)), 16 bytes per lane through LDS like conv_igemm_f32.hip.
// Reference call sites: the pointwise halves of slim.separable_conv2d, refinement_net/network/deeplab/core/xception.py:154-178.
#include "common.h"
#include <type_traits>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

namespace {

constexpr int BM = 128, BN = 128, KS = 32, RSB = KS * 2 + 16, NT = 256;
constexpr int PLANE = BM * RSB;                       // bytes of one operand part of one stage (BM == BN)
constexpr int BUF = 4 * PLANE;                        // A_hi | A_lo | B_hi | B_lo
constexpr int LDS_BYTES = 2 * BUF;

struct PwArgs {
  const char* in;        // split activations, pixel stride in_ps floats
  const char* wbase;        // the lower of the two weight matrices; hi_off / lo_off = their byte offsets from it
  const float* bias;
  const float* res;
  float* out;
  float* out_split;      // optional second copy of the output as {hi, lo} bf16 groups (the next pointwise conv's input)
  long m;
  unsigned hi_off, lo_off;
  int in_ps, cin4, k_pad, cout, cout_pad, res_ps, out_ps, out_split_ps, act;
  float slope;
};

__global__ __launch_bounds__(NT) void pwconv_bf16x3_split_kernel(const PwArgs p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  // XCD-contiguous (m-tile major, n-tile minor) order: the column tiles that re-read one A tile share an L2
  const int n_tiles = gridDim.y, nwg = gridDim.x * gridDim.y;
  const int v = premvos::xcd_contiguous(blockIdx.y * gridDim.x + blockIdx.x, nwg);
  const int tile_m = v / n_tiles, tile_n = v - tile_m * n_tiles;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- staging maps: A = 2 requests of 32 B (8 channels: hi + lo), B = 4 requests of 16 B (8 k of one part) per thread.
  // 32-bit byte offsets from kernel-uniform bases (the launcher checks the operands are below 4 GB): scalar base + vector offset
  // addressing instead of a 64-bit add per request
  unsigned aoff[2], ahome[2];
  int aj[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = tid + i * NT, row = q >> 2;
    aj[i] = q & 3;
    const long m = m0 + row < p.m ? m0 + row : p.m - 1;                  // rows past M: clamped, computed, never stored
    ahome[i] = (unsigned)(m * p.in_ps * 4);
    aoff[i] = ahome[i] + aj[i] * 32;
  }
  unsigned boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + i * NT, part = q >> 9, rem = q & 511, row = rem >> 2, ck = rem & 3;
    const int c = n0 + row < p.cout_pad ? n0 + row : p.cout_pad - 1;
    boff[i] = (unsigned)(((long)c * p.k_pad + ck * 8) * 2) + (part ? p.lo_off : p.hi_off);
  }
  // (register arrays are filled component by component: an aggregate copy into an array element goes through scratch,
  //  common.h ld4)
  auto ldu = [](const void* q) {
    const uint4 t = *reinterpret_cast<const uint4*>(q);
    return make_uint4(t.x, t.y, t.z, t.w);
  };
  // TWO register sets: the requests of stage kt + 2 are issued while stage kt is multiplied and go to LDS during stage kt + 1.
  // A stage is only 24 MFMAs x 32 cycles per wave -- shorter than an L2 round trip -- so with one set (requests of the next
  // stage, waited for at the end of this one) the loop was latency-bound at 0.29 of the bf16 pipe (239 TFLOP/s-equivalent).
  uint4 ra0[2][2], ra1[2][2], rb[2][4];
  // Requests are UNCONDITIONAL (a conditional 16-byte load compiles to an exec-masked block with `s_waitcnt vmcnt(0)` behind
  // it: every request of the first version of this kernel was waited for on the spot): a 4-channel unit past the row's real
  // channels -- only the matrix's last stage can hold one -- is requested at the row's first unit instead and zeroed when the
  // stage goes to LDS.
  auto gload = [&](int kt, uint4 (&q0)[2], uint4 (&q1)[2], uint4 (&qb)[4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c4 = kt * (KS / 4) + aj[i] * 2;                          // first of the two 4-channel units of this request
      const unsigned src = aoff[i] + (unsigned)kt * (KS * 4);
      q0[i] = ldu(p.in + (c4 < p.cin4 ? src : ahome[i]));                // (ahome: unit 0 of the row, always inside the tensor)
      q1[i] = ldu(p.in + (c4 + 1 < p.cin4 ? src + 16 : ahome[i]));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) qb[i] = ldu(p.wbase + (boff[i] + (unsigned)kt * (KS * 2)));
  };
  auto lstore = [&](int buf, const int kt, const uint4 (&q0)[2], const uint4 (&q1)[2], const uint4 (&qb)[4]) {
    char* base = lds + buf * BUF;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + i * NT, row = q >> 2;
      uint4 u0 = q0[i], u1 = q1[i];
      if (kt * (KS / 4) + 8 > p.cin4) {                                  // kernel-uniform: the stage that holds the K padding
        const int c4 = kt * (KS / 4) + aj[i] * 2;
        if (c4 >= p.cin4) u0 = make_uint4(0, 0, 0, 0);
        if (c4 + 1 >= p.cin4) u1 = make_uint4(0, 0, 0, 0);
      }
      // {h0-3, l0-3} + {h4-7, l4-7} -> hi of 8 channels, lo of 8 channels
      *reinterpret_cast<uint4*>(base + row * RSB + aj[i] * 16) = make_uint4(u0.x, u0.y, u1.x, u1.y);
      *reinterpret_cast<uint4*>(base + PLANE + row * RSB + aj[i] * 16) = make_uint4(u0.z, u0.w, u1.z, u1.w);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + i * NT, part = q >> 9, rem = q & 511, row = rem >> 2, ck = rem & 3;
      *reinterpret_cast<uint4*>(base + (2 + part) * PLANE + row * RSB + ck * 16) = qb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int KT = p.k_pad / KS;
  gload(0, ra0[0], ra1[0], rb[0]);
  if (KT > 1) gload(1, ra0[1], ra1[1], rb[1]);
  lstore(0, 0, ra0[0], ra1[0], rb[0]);
  __syncthreads();
  const int frag_off = (lane & 31) * RSB + (lane >> 5) * 16;
  auto compute = [&](const int buf) {
    const char* a = lds + buf * BUF + wm0 * RSB + frag_off;
    const char* b = lds + buf * BUF + 2 * PLANE + wn0 * RSB + frag_off;
#pragma unroll
    for (int s = 0; s < KS / 16; ++s) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        ah[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a + mi * 32 * RSB + s * 32));
        al[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a + PLANE + mi * 32 * RSB + s * 32));
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        bh[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(b + ni * 32 * RSB + s * 32));
        bl[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(b + PLANE + ni * 32 * RSB + s * 32));
      }
      // term-major: the four accumulators take the lo.hi products, then the hi.lo ones, then hi.hi (small terms first) -- an
      // accumulator is touched again four MFMAs later, not in the next instruction (8-pass MFMAs: a dependent issue stalls)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
    }
  };
#ifdef PV_DBG_SPLIT_OLDLOOP      // developer A/B builds: the first loop of this kernel (fragments read right before their MFMAs)
  // stage kt is multiplied out of LDS buffer kt & 1; register set kt & 1 (stored one stage ago) takes stage kt + 2, set
  // (kt + 1) & 1 -- requested one stage ago -- goes to LDS for stage kt + 1
  auto step = [&](auto par_tag, const int kt) {
    constexpr int P = decltype(par_tag)::value;
    if (kt + 2 < KT) gload(kt + 2, ra0[P], ra1[P], rb[P]);
    compute(P);
    if (kt + 1 < KT) lstore(P ^ 1, kt + 1, ra0[P ^ 1], ra1[P ^ 1], rb[P ^ 1]);
    __syncthreads();
  };
  int kt = 0;
  for (; kt + 1 < KT; kt += 2) {
    step(std::integral_constant<int, 0>{}, kt);
    step(std::integral_constant<int, 1>{}, kt + 1);
  }
  if (kt < KT) step(std::integral_constant<int, 0>{}, kt);

#else
  // Late round 3 (the fp32 kernel's lesson, conv_igemm_f32.hip): TWO FRAGMENT sets -- the LDS reads of the next 16-deep group are
  // in flight while the 12 MFMAs of the current one issue, and the first group of the next stage is requested right behind the
  // barrier, under the last hi.hi MFMAs of this stage -- plus a scheduling fence behind the global requests and a raised issue
  // priority over the MFMA run.  A group is only 12 x 32 cycles: before, its eight ds_read_b128 were waited for in full, twice
  // per stage, with two waves per SIMD to cover them.  Same products in the same order.
  static_assert(KS == 32, "two 16-deep groups per stage");
  bf16x8 fah[2][2], fal[2][2], fbh[2][2], fbl[2][2];                   // [set][block]
  auto ldfrag = [&](const int set, const int buf, const int sgrp) {
    const char* a = lds + buf * BUF + wm0 * RSB + frag_off + sgrp * 32;
    const char* b = lds + buf * BUF + 2 * PLANE + wn0 * RSB + frag_off + sgrp * 32;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      fah[set][mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a + mi * 32 * RSB));
      fal[set][mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a + PLANE + mi * 32 * RSB));
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      fbh[set][ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(b + ni * 32 * RSB));
      fbl[set][ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(b + PLANE + ni * 32 * RSB));
    }
  };
  auto mfma_small = [&](const int set) {                                // lo.hi then hi.lo (small terms first)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[set][mi], fbh[set][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[set][mi], fbl[set][ni], acc[mi][ni], 0, 0, 0);
  };
  auto mfma_big = [&](const int set) {                                  // hi.hi
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[set][mi], fbh[set][ni], acc[mi][ni], 0, 0, 0);
  };
  // one register set of global requests here (ra0[0] ...): stage kt + 1 is requested at the top of stage kt
  ldfrag(0, 0, 0);
  for (int kt = 0; kt + 1 < KT; ++kt) {
    const int buf = kt & 1;
    if (kt > 0 || KT <= 1) gload(kt + 1, ra0[0], ra1[0], rb[0]);      // (stage 1 was requested before the loop)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    ldfrag(1, buf, 1);
    mfma_small(0);
    mfma_big(0);
    mfma_small(1);
    __builtin_amdgcn_s_setprio(0);
    if (kt == 0) lstore(buf ^ 1, kt + 1, ra0[1], ra1[1], rb[1]);
    else lstore(buf ^ 1, kt + 1, ra0[0], ra1[0], rb[0]);
    __syncthreads();
    ldfrag(0, buf ^ 1, 0);
    mfma_big(1);
  }
  {
    const int buf = (KT - 1) & 1;
    ldfrag(1, buf, 1);
    mfma_small(0);
    mfma_big(0);
    mfma_small(1);
    mfma_big(1);
    __syncthreads();
  }
#endif

  // ---- epilogue: one wave row (64 rows) at a time through LDS, 16 bytes per lane out
  constexpr int EP = BN + 4;
  float* stg = reinterpret_cast<float*>(lds);
  const bool wide = (p.cout & 3) == 0 && (p.out_ps & 3) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15u) == 0 &&
                    (p.res == nullptr || ((p.res_ps & 3) == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15u) == 0)) &&
                    (p.bias == nullptr || (reinterpret_cast<uintptr_t>(p.bias) & 15u) == 0);
#pragma unroll 1
  for (int wr = 0; wr < 2; ++wr) {
    if ((wave >> 1) == wr) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            stg[row * EP + wn0 + ni * 32 + (lane & 31)] = acc[mi][ni][r];
          }
    }
    __syncthreads();
    if (wide) {
      // one bias request per thread (its units share their four columns) and all residual requests of the pass in flight before
      // anything waits for one (conv_igemm_f32.hip, late round 3); rows / columns past the tensor: clamped address, masked store
      constexpr int C4 = BN / 4, UPT = 64 * C4 / NT, RSTEP = NT / C4;
      const int c4 = tid % C4, row0 = tid / C4, col = n0 + c4 * 4;
      const bool col_ok = col < p.cout;
      const int colc = col_ok ? col : 0;
      const float4 bv = p.bias != nullptr ? premvos::ld4(p.bias + colc) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 rv[UPT];
      if (p.res != nullptr) {
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
          long m = m0 + wr * 64 + row0 + i * RSTEP;
          m = m < p.m ? m : p.m - 1;
          rv[i] = premvos::ld4(p.res + m * p.res_ps + colc);
        }
      }
#pragma unroll
      for (int i = 0; i < UPT; ++i) {
        const int row = row0 + i * RSTEP;
        const long m = m0 + wr * 64 + row;
        float4 v = *reinterpret_cast<const float4*>(&stg[row * EP + c4 * 4]);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (p.res != nullptr) { v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w; }
        if (p.act == PREMVOS_ACT_RELU) {
          v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        } else if (p.act == PREMVOS_ACT_LEAKY) {
          v.x = v.x > 0.f ? v.x : v.x * p.slope; v.y = v.y > 0.f ? v.y : v.y * p.slope;
          v.z = v.z > 0.f ? v.z : v.z * p.slope; v.w = v.w > 0.f ? v.w : v.w * p.slope;
        }
        if (m < p.m && col_ok) {
          *reinterpret_cast<float4*>(p.out + m * p.out_ps + col) = v;
          if (p.out_split != nullptr) *reinterpret_cast<float4*>(p.out_split + m * p.out_split_ps + col) = premvos::split_bf16_group(v);
        }
      }
    } else {
      for (int u = tid; u < 64 * BN; u += NT) {
        const int row = u / BN, c = u - row * BN;
        const long m = m0 + wr * 64 + row;
        const int col = n0 + c;
        if (m < p.m && col < p.cout) {
          float v = stg[row * EP + c];
          if (p.bias != nullptr) v += p.bias[col];
          if (p.res != nullptr) v += p.res[m * p.res_ps + col];
          if (p.act == PREMVOS_ACT_RELU) v = v > 0.f ? v : 0.f;
          else if (p.act == PREMVOS_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
          p.out[m * p.out_ps + col] = v;
        }
      }
    }
    if (wr == 0) __syncthreads();
  }
}

}  // namespace

extern "C" int premvos_pwconv_bf16x3_split_f32(const void* in_split, int32_t in_ps, int64_t m, int32_t cin, const void* wgt_hi,
                                               const void* wgt_lo, int32_t k_pad, int32_t cout, int32_t cout_pad, const float* bias,
                                               const float* res, int32_t res_ps, float* out, int32_t out_ps, void* out_split,
                                               int32_t out_split_ps, int32_t act, float slope, void* stream) {
  PV_REQUIRE(in_split && wgt_hi && wgt_lo && out, "pwconv_bf16x3_split: null pointer");
  PV_REQUIRE(m > 0 && m < (1L << 31) && cin > 0 && cout > 0, "pwconv_bf16x3_split: bad dims");
  PV_REQUIRE(in_ps % 4 == 0 && in_ps >= (cin + 3) / 4 * 4, "pwconv_bf16x3_split: in_ps must be a multiple of 4 and >= roundup(cin, 4)");
  PV_REQUIRE(k_pad % KS == 0 && k_pad >= cin, "pwconv_bf16x3_split: k_pad must be a multiple of 32 and >= cin");
  PV_REQUIRE(cout_pad % 32 == 0 && cout_pad >= cout, "pwconv_bf16x3_split: bad cout_pad");
  PV_REQUIRE(premvos::aligned16(in_split) && premvos::aligned16(wgt_hi) && premvos::aligned16(wgt_lo),
             "pwconv_bf16x3_split: in / wgt must be 16-byte aligned");
  PV_REQUIRE(out_ps >= cout && (res == nullptr || res_ps >= cout), "pwconv_bf16x3_split: pixel strides < cout");
  PV_REQUIRE(act == PREMVOS_ACT_NONE || act == PREMVOS_ACT_RELU || act == PREMVOS_ACT_LEAKY, "pwconv_bf16x3_split: bad activation");
  PV_REQUIRE(out_split == nullptr || (cout % 4 == 0 && out_ps % 4 == 0 && out_split_ps % 4 == 0 && out_split_ps >= cout &&
                                      premvos::aligned16(out) && premvos::aligned16(out_split) &&
                                      (res == nullptr || (res_ps % 4 == 0 && premvos::aligned16(res))) &&
                                      (bias == nullptr || premvos::aligned16(bias))),
             "pwconv_bf16x3_split: a split second output needs cout %% 4 == 0 and 16-byte aligned pixels (out, out_split, res, bias)");
  static const bool attr_done = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pwconv_bf16x3_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS_BYTES);
    return true;
  }();
  (void)attr_done;
  PwArgs a;
  a.in = static_cast<const char*>(in_split);
  a.bias = bias; a.res = res; a.out = out; a.out_split = static_cast<float*>(out_split); a.m = m;
  a.in_ps = in_ps; a.cin4 = (cin + 3) / 4; a.k_pad = k_pad; a.cout = cout; a.cout_pad = cout_pad;
  a.res_ps = res_ps; a.out_ps = out_ps; a.out_split_ps = out_split_ps; a.act = act; a.slope = slope;
  // 32-bit operand offsets: the activations, and the two weight matrices together, must each span less than 4 GB
  const char* hi = static_cast<const char*>(wgt_hi);
  const char* lo = static_cast<const char*>(wgt_lo);
  a.wbase = hi < lo ? hi : lo;
  const long wbytes = (long)cout_pad * k_pad * 2, span = (hi < lo ? lo - hi : hi - lo) + wbytes;
  PV_REQUIRE(m * in_ps * 4 < (1L << 32) && span < (1L << 32), "pwconv_bf16x3_split: operands must lie within 4 GB windows");
  a.hi_off = (unsigned)(hi - a.wbase);
  a.lo_off = (unsigned)(lo - a.wbase);
  dim3 grid((unsigned)premvos::cdiv((int)m, BM), (unsigned)premvos::cdiv(cout, BN));
  hipLaunchKernelGGL(pwconv_bf16x3_split_kernel, grid, dim3(NT), LDS_BYTES, static_cast<hipStream_t>(stream), a);
  return premvos::check_launch("pwconv_bf16x3_split");
}
