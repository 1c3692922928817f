// Convolution (1x1 and k x k, any stride / dilation / padding) as an implicit GEMM on the gfx950 bf16 matrix pipe with fp32-class
// accuracy ("bf16x3": a*b ~= hi(a)hi(b) + hi(a)lo(b) + lo(a)hi(b), fp32 accumulate) on activations that are RESIDENT in HBM in the
// split layout "S8": per pixel, every group of 8 channels is the 32 bytes {hi(8 x bf16), lo(8 x bf16)} -- 4 bytes per element like
// fp32, same pixel stride, written once by the producing kernel (this kernel's own epilogue, premvos_dwconv3x3_f32 with
// PREMVOS_ACT_SPLIT8_BF16, premvos_split8_f32).  Round 4: the round-3 split kernel
// (128 x 128 tile, register staging + ds_write_b128, {hi4, lo4} groups re-paired in registers) sat at 0.29 of the bf16 pipe because
// a stage's LDS traffic -- 16 KB of ds_write_b128 at ~79 B/clk plus the fragment reads -- took as many cycles as its MFMAs.  Here:
//
//   * staging by LDS-DMA (`global_load_lds_dwordx4`: L2 -> LDS, no VGPRs, no ds_write): a wave instruction lands 64 x 16 B
//     lane-linearly = 8 rows x 128 B of a stage; the bank swizzle is applied on the SOURCE side (each lane picks WHICH 16-byte
//     chunk of its row it fetches) and again on the fragment reads, so the image in LDS is conflict-free for ds_read_b128:
//     physical chunk = logical chunk ^ ((row >> 1) & 7)  (rows are 128 B = half a bank line);
//   * S8 makes a 16-byte chunk exactly one MFMA operand: lane l of v_mfma_f32_32x32x16_bf16 holds the 8 consecutive k of row l & 31
//     that start at 8 (l >> 5) -- one ds_read_b128 of the hi (or lo) half of a channel group, no re-pairing;
//   * tiles of 256 x 256 (eight waves, 2 x 4, 128 x 64 per wave) or 256 x 128 (four waves): 12 fragment reads per 24 MFMAs,
//     32 channels (128 B per row) per stage, NSTAGE LDS buffers with the DMA of stage kt + NSTAGE - 1 in flight behind ONE raw
//     s_barrier per stage and a COUNTED s_waitcnt vmcnt (never __syncthreads in the loop: its fence would drain the DMA queue);
//   * implicit GEMM: the K loop walks (tap, 32-channel block); a lane's source address is the input pixel of its row under the
//     tap, or a 16-byte zero page when the tap falls into the padding or the chunk lies beyond the layer's channels -- so 3x3 /
//     strided / dilated layers run on the same loop as pointwise ones (ResNet conv2, RPN 3x3, PWC-Net estimators).
//
//   M = output pixels, N = cout, K = taps x roundup(cin, 32).  Weights: packed by ops.pack_conv_s8 as
//   [cout_pad][tap][cin block of 32][group of 8][hi 8 | lo 8] bf16 -- row n of the B operand is read exactly like a pixel of A.
//   Epilogue: acc + bias (+ fp32 residual), activation; written as fp32 NHWC and / or as S8 (the next conv's operand), 32 bytes of a
//   pixel per lane through wave-private LDS staging.
// Reference call sites: every Conv2D / slim.conv2d / separable_conv2d pointwise half of the three nets (proposal_net/basemodel.py:
// 29-99, refinement_net/network/deeplab/core/xception.py:154-178, optical_flow_net-PWC-Net/models/PWCNet.py:24-34).
#include "common.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

struct PremvosS8Args {
  const char* in;          // S8 activations [n][h][w][in_ps floats]
  const char* wgt;         // packed weights: row stride = taps * kc32 * 128 bytes
  const float* bias;
  const float* res;        // fp32 residual (optional)
  const char* res_s8;      // ... or the residual in S8 (hi + lo: x to ~2^-17 relative; saves the producer an fp32 copy of its output)
  float* out;              // fp32 output (optional)
  char* out_s8;            // S8 output (optional)
  long M;
  int h, w, ho, wo, in_ps, cin_groups, kc32, taps, kw, sh, sw, dh, dw, pt, pl;
  int cout, cout_pad, res_ps, out_ps, out_s8_ps, act, n_tiles;
  float slope;
};

namespace {

using S8Args = PremvosS8Args;

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[64];      // 256 bytes of zeros: the source of padded / absent chunks

// One LDS-DMA instruction of a wave: lane l's 16 bytes at `g` (per lane) land at `l` (wave-uniform) + 16 l.  (The guard is not a second
// code path: hipcc also parses kernel templates for the host, where the gfx950 builtin does not exist -- unguarded, the host side
// silently drops the kernel's launch stub and the library fails to load with an undefined symbol.)
__device__ __forceinline__ void dma16(const char* g, char* l) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
#endif
}

__device__ __forceinline__ void wait_vm(const int n) {
  switch (n) {        // (immediate operand: the few counts the pipelines below need)
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// The epilogue of a wave: its MI x NI accumulator blocks (rows m_base + 32 mi + ..., columns n_base + 32 ni + ...) -> bias, residual
// (fp32 or S8), activation -> fp32 NHWC and / or S8, wave-private: MI passes of 32 rows x (32 NI) columns through this wave's LDS
// block; a lane owns 8 consecutive columns (one S8 group) of a row per unit.
template <int MI, int NI>
__device__ __forceinline__ void epilogue(const S8Args& p, f32x16 (&acc)[MI][NI], char* lds, const int wave, const int lane, const long m_base,
                                         const int n_base) {
  constexpr int WC = 32 * NI, SC = WC + 4, UNITS = 32 * (WC / 8) / 64, RSTEP = 64 / (WC / 8);
  float* stg = reinterpret_cast<float*>(lds) + wave * (32 * SC);
  const int c8 = lane % (WC / 8), urow0 = lane / (WC / 8);
  const int col = n_base + c8 * 8;
  const bool col_ok = col < p.cout;                       // (cout % 8 == 0 is required by the launcher)
  const int colc = col_ok ? col : 0;
  float4 bv0 = make_float4(0.f, 0.f, 0.f, 0.f), bv1 = bv0;
  if (p.bias != nullptr) {
    bv0 = premvos::ld4(p.bias + colc);
    bv1 = premvos::ld4(p.bias + colc + 4);
  }
  const int act = p.act;
  auto activate = [&](float x) {
    if (act == PREMVOS_ACT_RELU) return x > 0.f ? x : 0.f;
    if (act == PREMVOS_ACT_LEAKY) return x > 0.f ? x : x * p.slope;
    return x;
  };
#pragma unroll               // (fully unrolled: a runtime index into acc[][] would put the accumulators into scratch memory)
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        stg[row * SC + ni * 32 + (lane & 31)] = acc[mi][ni][r];
      }
    __builtin_amdgcn_wave_barrier();                       // (a wave's LDS instructions execute in order: no fence needed)
    const long mrow0 = m_base + mi * 32;
    float4 r0[UNITS], r1[UNITS];
    const bool has_res = p.res != nullptr || p.res_s8 != nullptr;
    if (p.res != nullptr) {                                // all residual requests of the pass go out before anything waits for one
#pragma unroll
      for (int i = 0; i < UNITS; ++i) {
        long m = mrow0 + urow0 + i * RSTEP;
        m = m < p.M ? m : p.M - 1;
        r0[i] = premvos::ld4(p.res + m * p.res_ps + colc);
        r1[i] = premvos::ld4(p.res + m * p.res_ps + colc + 4);
      }
    } else if (p.res_s8 != nullptr) {                      // {hi8, lo8}: the raw 32 bytes now, hi + lo when they are used
#pragma unroll
      for (int i = 0; i < UNITS; ++i) {
        long m = mrow0 + urow0 + i * RSTEP;
        m = m < p.M ? m : p.M - 1;
        const float* g = reinterpret_cast<const float*>(p.res_s8 + (m * p.res_ps + colc) * 4);
        r0[i] = premvos::ld4(g);
        r1[i] = premvos::ld4(g + 4);
      }
    }
#pragma unroll
    for (int i = 0; i < UNITS; ++i) {
      const int row = urow0 + i * RSTEP;
      const long m = mrow0 + row;
      float4 v0 = *reinterpret_cast<const float4*>(stg + row * SC + c8 * 8);
      float4 v1 = *reinterpret_cast<const float4*>(stg + row * SC + c8 * 8 + 4);
      v0.x += bv0.x; v0.y += bv0.y; v0.z += bv0.z; v0.w += bv0.w;
      v1.x += bv1.x; v1.y += bv1.y; v1.z += bv1.z; v1.w += bv1.w;
      if (p.res_s8 != nullptr) premvos::join_split8(r0[i], r1[i]);       // -> the eight floats hi + lo
      if (has_res) {
        v0.x += r0[i].x; v0.y += r0[i].y; v0.z += r0[i].z; v0.w += r0[i].w;
        v1.x += r1[i].x; v1.y += r1[i].y; v1.z += r1[i].z; v1.w += r1[i].w;
      }
      v0.x = activate(v0.x); v0.y = activate(v0.y); v0.z = activate(v0.z); v0.w = activate(v0.w);
      v1.x = activate(v1.x); v1.y = activate(v1.y); v1.z = activate(v1.z); v1.w = activate(v1.w);
#ifdef PV_DBG_S8_NOSTORE            // developer A/B builds (tools/dev/ab_build.sh): the whole epilogue but no global store
      if (m < p.M && col_ok && p.slope == 12345.f) {
#else
      if (m < p.M && col_ok) {
#endif
        if (p.out != nullptr) {
          float* o = p.out + m * p.out_ps + col;
          *reinterpret_cast<float4*>(o) = v0;
          *reinterpret_cast<float4*>(o + 4) = v1;
        }
        if (p.out_s8 != nullptr) premvos::store_split8(p.out_s8 + (m * p.out_s8_ps + col) * 4, v0, v1);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// BM x BN tile, WM x WN waves (each (BM / WM) x (BN / WN), in 32 x 32 MFMA blocks), NSTAGE LDS buffers of (BM + BN) rows x KC channels
// (KC = 32: 128-byte rows, two 16-deep MFMA steps per stage; KC = 16: 64-byte rows, one step -- half the LDS, so that a 256 x 128
// tile fits a CU twice and the two workgroups cover each other's barriers, prologues and epilogues).
template <int BM, int BN, int WM, int WN, int NSTAGE, int KC, int OCC, bool PW>      // OCC: waves per SIMD the register allocation must allow; PW: 1x1 / stride 1 / no padding
__global__ __launch_bounds__(64 * WM * WN, OCC) void conv_bf16x3_s8_kernel(const S8Args p) {
  constexpr int NW = WM * WN;
  constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  constexpr int RB = KC * 4, CPR = RB / 16, RPP = 1024 / RB;      // row bytes, 16-byte chunks per row, rows per 1-KiB DMA piece
  constexpr int GPT = KC / 8, KSTEPS = KC / 16;                   // channel groups per stage, MFMA steps per stage
  constexpr int STAGE = (BM + BN) * RB;                           // bytes
  constexpr int APW = BM / RPP / NW, BPW = BN / RPP / NW;         // DMA pieces per wave and stage
  constexpr int PPW = APW + BPW;
  static_assert(KC == 32 || KC == 16, "stage depth");
  static_assert(BM % (RPP * NW) == 0 && BN % (RPP * NW) == 0, "pieces must divide over the waves");
  // bank swizzle of row r: physical chunk = logical chunk ^ SW(r).  128-byte rows: two rows per 256-byte bank line -> (r >> 1) & 7;
  // 64-byte rows: four rows per line -> (r >> 2) & 3.  Either way the 16 rows of a ds_read_b128 lane group hit 16 distinct slots.
  auto SW = [](const int r) { return KC == 32 ? (r >> 1) & 7 : (r >> 2) & 3; };
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: LDS-DMA bases (M0) and wave offsets stay in SGPRs
  const int wm = wave / WN, wn = wave - wm * WN;
  // XCD-contiguous (row-tile major) order: the column tiles that re-read one A tile share an L2
  const int nwg = gridDim.x;
  const int v = premvos::xcd_contiguous(blockIdx.x, nwg);
  const int tile_m = v / p.n_tiles, tile_n = v - tile_m * p.n_tiles;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- DMA source maps.  Piece q of a stage = rows [RPP q, RPP q + RPP); lane l lands at row RPP q + l / CPR, physical chunk l % CPR,
  // and therefore FETCHES logical chunk (l % CPR) ^ SW(row) of that row.
  const int prow = lane / CPR, pchunk = lane % CPR;
  long a_base[APW];                 // byte offset of input pixel (img, oy * sh - pt, ox * sw - pl) + the lane's chunk: tap (0, 0), block 0
  int a_iy[APW], a_ix[APW], a_grp[APW];
#pragma unroll
  for (int i = 0; i < APW; ++i) {
    const int row = (wave + i * NW) * RPP + prow;
    long m = m0 + row;
    m = m < p.M ? m : p.M - 1;                           // rows past M: clamped, multiplied, never stored
    const int lc = pchunk ^ SW(row);
    a_grp[i] = lc >> 1;
    const long img = m / ((long)p.ho * p.wo);
    const int rem = (int)(m - img * p.ho * p.wo), oy = rem / p.wo, ox = rem - oy * p.wo;
    a_iy[i] = oy * p.sh - p.pt;
    a_ix[i] = ox * p.sw - p.pl;
    a_base[i] = ((img * p.h + a_iy[i]) * p.w + a_ix[i]) * (long)p.in_ps * 4 + lc * 16;      // (may point outside: only used when in range)
  }
  long b_off[BPW];
  const int kct = p.kc32 * (32 / KC);                    // stages per tap (the weights pad every tap to whole 32-channel blocks)
  const long wrow = (long)p.taps * p.kc32 * 128;
#pragma unroll
  for (int i = 0; i < BPW; ++i) {
    const int row = (wave + i * NW) * RPP + prow;
    const int c = n0 + row < p.cout_pad ? n0 + row : p.cout_pad - 1;
    b_off[i] = c * wrow + (pchunk ^ SW(row)) * 16;
  }
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  // The DMA of a stage is issued in two halves (A pieces, B pieces) BETWEEN the MFMA groups of the stage before it: behind a barrier
  // all waves of a workgroup reach the same point together, and ~20 instructions per piece issued in one block left the matrix pipe
  // idle for a fifth of every stage (first version: 285 instead of ~330 TFLOP/s-equivalent on the 728-wide layers).
  // Stages are issued in order kt = 0, 1, 2, ...: (tap row, tap column, channel block) advance as scalars, no division in the loop.
  int i_ct = 0, i_ky = 0, i_kx = 0;
  auto issue_a = [&](const int buf) {
    char* sb = lds + buf * STAGE;
    const int dy = i_ky * p.dh, dx = i_kx * p.dw;
    const long tap_off = PW ? (long)(i_ct * RB) : ((long)dy * p.w + dx) * p.in_ps * 4 + i_ct * RB;
    const int lim = p.cin_groups - i_ct * GPT;           // channel groups of the layer left at this stage (>= GPT: the stage is whole)
#pragma unroll
    for (int i = 0; i < APW; ++i) {
      // in range: the pixel under the tap; else (padding, or a chunk beyond the layer's channels): 16 bytes of zeros
      bool ok = a_grp[i] < lim;
      if (!PW) ok = ok & ((unsigned)(a_iy[i] + dy) < (unsigned)p.h) & ((unsigned)(a_ix[i] + dx) < (unsigned)p.w);   // (& not &&: no branches)
      const char* src = p.in + (a_base[i] + tap_off);
      src = ok ? src : zero;
      dma16(src, sb + (wave + i * NW) * 1024);
    }
  };
  auto issue_b = [&](const int kt, const int buf) {
    char* sb = lds + buf * STAGE;
    const char* wb = p.wgt + (long)kt * RB;
#pragma unroll
    for (int i = 0; i < BPW; ++i)
      dma16(wb + b_off[i], sb + BM * RB + (wave + i * NW) * 1024);
    if (++i_ct == kct) {
      i_ct = 0;
      if (++i_kx == p.kw) {
        i_kx = 0;
        ++i_ky;
      }
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---- fragment addresses: row (lane & 31) of a 32-row block, logical chunk 2 (2 s + (lane >> 5)) + part; the swizzle term depends
  // on lane & 31 only (block origins are multiples of 32 rows)
  const int frow = lane & 31, fsw = SW(frow), fg = lane >> 5;
  int foff[KSTEPS][2];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
    for (int part = 0; part < 2; ++part) foff[s][part] = frow * RB + (((2 * (2 * s + fg) + part) ^ fsw) << 4);
  const int fa_base = wm * (BM / WM) * RB, fb_base = BM * RB + wn * (BN / WN) * RB;

  bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
  auto frags = [&](const char* sb, const int s) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      bh[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + fb_base + ni * 32 * RB + foff[s][0]));
      bl[ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + fb_base + ni * 32 * RB + foff[s][1]));
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      ah[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + fa_base + mi * 32 * RB + foff[s][0]));
      al[mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + fa_base + mi * 32 * RB + foff[s][1]));
    }
  };
  // one product term of a 16-deep step over the wave's MI x NI blocks: an accumulator is touched again MI * NI MFMAs later
  auto term = [&](const bf16x8 (&a)[MI], const bf16x8 (&b)[NI]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
  };

  // ---- the K loop: stage kt is multiplied out of buffer kt % NSTAGE while the DMA of stages kt + 1 .. kt + NSTAGE - 1 is in flight.
  // Top of iteration kt: wait until this wave's pieces of stage kt have landed (PPW x (stages issued behind it) may stay in
  // flight), then ONE barrier: every wave's pieces are in, and every wave is done reading buffer (kt - 1) % NSTAGE -- which is the
  // buffer the next DMA (stage kt + NSTAGE - 1) overwrites.  Per 16-deep step: lo.hi, hi.lo, hi.hi (small terms first).
  const int KT = p.taps * kct;
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < KT) {
      issue_a(s);
      issue_b(s, s);
    }
  int buf = 0;
  for (int kt = 0; kt < KT; ++kt) {
    const int behind = KT - 1 - kt < NSTAGE - 2 ? KT - 1 - kt : NSTAGE - 2;      // stages issued after stage kt so far
    if (behind == 0) wait_vm(0);
    else if (behind == 1) wait_vm(PPW);
    else wait_vm(2 * PPW);
    // (lgkmcnt(0): this wave's fragment reads of the buffer the next DMA overwrites have returned -- they fed MFMAs that were
    //  issued already, so the wait is free; it makes the write-after-read order architectural instead of a matter of latencies)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const bool more = kt + NSTAGE - 1 < KT;
    int nb = buf + NSTAGE - 1;
    nb = nb >= NSTAGE ? nb - NSTAGE : nb;
    const char* sb = lds + buf * STAGE;
    frags(sb, 0);
    term(al, bh);
    __builtin_amdgcn_sched_barrier(0);
    if (more) issue_a(nb);
    __builtin_amdgcn_sched_barrier(0);
    term(ah, bl);
    __builtin_amdgcn_sched_barrier(0);
    if (more) issue_b(kt + NSTAGE - 1, nb);
    __builtin_amdgcn_sched_barrier(0);
    term(ah, bh);
    if (KSTEPS == 2) {
      frags(sb, KSTEPS - 1);
      term(al, bh);
      term(ah, bl);
      term(ah, bh);
    }
    buf = buf + 1 == NSTAGE ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");        // the operand buffers become the epilogue's staging blocks

#ifdef PV_DBG_S8_NOEPI               // developer A/B builds: no epilogue at all (the accumulators stay live through a never-true store)
  float sacc = 0.f;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc += acc[mi][ni][r];
  if (sacc == 12345.f && p.out != nullptr) p.out[tid] = sacc;
#else
  epilogue<MI, NI>(p, acc, lds, wave, lane, m0 + wm * (BM / WM), n0 + wn * (32 * NI));
#endif
}

// ---- The PING-PONG form for pointwise layers (1x1, stride 1): 256 x 256 tile, eight waves as two GROUPS of four (one wave of each
// group per SIMD), 32-channel stages in two LDS buffers.  In the kernel above all eight waves reach every barrier together: they
// read fragments together, issue DMA together and then queue for the matrix pipe together -- PMC on the 728-wide layer: 32 % of
// the wave cycles parked on s_waitcnt / s_barrier, the pipe 38 % busy.  Here the groups run the SAME program one phase apart:
//
//     group 0:  | L0 + DMA | M0 | L1 + DMA | M1 | ...          L = all fragment reads of a stage into registers (24 x ds_read_b128),
//     group 1:  |    -     | L0 | M0       | L1 | M1 | ...     M = its 48 MFMAs, back to back;  | = s_barrier (whole workgroup)
//
// so on every SIMD one wave multiplies while its partner loads -- the pipe only waits at the barriers themselves.  Group 0 also
// issues the stage DMA (16 pieces per wave, behind its fragment reads): stage k + 1 goes into the buffer both groups finished
// reading one phase ago and is waited for (vmcnt(0)) at the end of group 0's MFMA phase, a full phase before anyone reads it.
// Same products in the same order as the kernel above: bit-identical results (tests/test_gpu_conv_s8.py).
__global__ __launch_bounds__(512, 2) void conv_bf16x3_s8_pp_kernel(const S8Args p) {
  constexpr int BM = 256, BN = 256, MI = 4, NI = 2, RB = 128, STAGE = (BM + BN) * RB;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;              // group = wave row: rows [128 grp, +128), columns [64 wn, +64)
  const int nwg = gridDim.x;
  const int v = premvos::xcd_contiguous(blockIdx.x, nwg);
  const int tile_m = v / p.n_tiles, tile_n = v - tile_m * p.n_tiles;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;
  auto SW = [](const int r) { return (r >> 1) & 7; };

  // DMA maps of group 0: piece q = wn + 4 i, i < 16: q < 32 -> A rows [8 q, +8), else B rows [8 (q - 32), +8); 32-bit byte offsets
  // from the kernel-uniform bases (the launcher checks both operands span < 4 GB)
  const int prow = lane >> 3, pchunk = lane & 7;
  // (pieces of one wave are 32 rows apart: the swizzle term, hence the logical chunk a lane fetches, is the same for all of them;
  //  the weight rows are a fixed stride apart -- the packer pads the matrix to whole 256-row tiles -- so one register addresses all
  //  eight B pieces; the pixel rows keep an offset each, because the last row tile clamps them to the tensor)
  const int lc = pchunk ^ SW(wn * 8 + prow);
  const int a_grp = lc >> 1;
  unsigned a_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    long m = m0 + (wn + 4 * i) * 8 + prow;
    m = m < p.M ? m : p.M - 1;
    a_off[i] = (unsigned)(m * p.in_ps * 4 + lc * 16);
  }
  const long wrow = (long)p.kc32 * 128;
  const unsigned b_off0 = (unsigned)((n0 + wn * 8 + prow) * wrow + lc * 16);
  const unsigned b_step = (unsigned)(32 * wrow);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  auto issue = [&](const int kt, const int buf) {
    char* sb = lds + buf * STAGE;
    const bool ok = a_grp < p.cin_groups - kt * 4;       // the layer's last 32-channel block may be partial: absent chunks are zeros
    const char* ab = p.in + kt * RB;
    const char* wb = p.wgt + kt * RB + b_off0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const char* src = ab + a_off[i];
      src = ok ? src : zero;
      dma16(src, sb + (wn + 4 * i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) dma16(wb + i * b_step, sb + BM * RB + (wn + 4 * i) * 1024);
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  const int frow = lane & 31, fsw = SW(frow), fg = lane >> 5;
  int foff[2][2];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
    for (int part = 0; part < 2; ++part) foff[s_][part] = frow * RB + (((2 * (2 * s_ + fg) + part) ^ fsw) << 4);
  const int fa_base = grp * 128 * RB, fb_base = BM * RB + wn * 64 * RB;
  bf16x8 ah[2][MI], al[2][MI], bh[2][NI], bl[2][NI];                   // [16-deep step][block]
  auto frags = [&](const char* sb) {
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        bh[s_][ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + fb_base + ni * 32 * RB + foff[s_][0]));
        bl[s_][ni] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + fb_base + ni * 32 * RB + foff[s_][1]));
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        ah[s_][mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + fa_base + mi * 32 * RB + foff[s_][0]));
        al[s_][mi] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + fa_base + mi * 32 * RB + foff[s_][1]));
      }
    }
  };
  auto term = [&](const bf16x8 (&a)[MI], const bf16x8 (&b)[NI]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
  };

  const int KT = p.kc32;
  if (grp == 0) {
    issue(0, 0);
    if (KT > 1) {
      issue(1, 1);
      wait_vm(16);                                        // stage 0 has landed (stage 1 may still be in flight)
    } else {
      wait_vm(0);
    }
  } else {
    asm volatile("s_barrier" ::: "memory");               // group 1 runs one phase behind
  }
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_barrier" ::: "memory");               // ---- load phase of this group (the other group multiplies)
    frags(lds + (kt & 1) * STAGE);
    if (grp == 0 && kt >= 1 && kt + 1 < KT) issue(kt + 1, (kt + 1) & 1);     // (buffer (kt + 1) & 1: group 1 finished reading it a phase ago)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // ---- MFMA phase of this group (the other group loads)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      term(al[s_], bh[s_]);
      term(ah[s_], bl[s_]);
      term(ah[s_], bh[s_]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 0) wait_vm(0);                             // the next stage is in LDS before the barrier that opens its first reader's phase
  }
  if (grp == 0) asm volatile("s_barrier" ::: "memory");   // (the barrier group 1 spent ahead of the loop)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // both groups are done with the operand buffers
  epilogue<MI, NI>(p, acc, lds, wave, lane, m0 + grp * 128, n0 + wn * 64);
}

int launch_pp(const S8Args& a0, hipStream_t s) {
  constexpr int LDS_BYTES = 2 * 512 * 128;
  static const bool attr_done = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16x3_s8_pp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    return true;
  }();
  (void)attr_done;
  S8Args a = a0;
  a.n_tiles = premvos::cdiv(a.cout, 256);
  const long m_tiles = (a.M + 255) / 256;
  hipLaunchKernelGGL(conv_bf16x3_s8_pp_kernel, dim3((unsigned)(m_tiles * a.n_tiles)), dim3(512), LDS_BYTES, s, a);
  return premvos::check_launch("conv_bf16x3_s8 (ping-pong)");
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int KC = 32, int OCC = 1>
int launch(const S8Args& a0, hipStream_t s) {
  constexpr int LDS_BYTES = NSTAGE * (BM + BN) * KC * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(WM * WN * 32 * (32 * (BN / WN / 32) + 4) * 4 <= LDS_BYTES, "epilogue staging must fit the operand buffers");
  static const bool attr_done = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16x3_s8_kernel<BM, BN, WM, WN, NSTAGE, KC, OCC, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16x3_s8_kernel<BM, BN, WM, WN, NSTAGE, KC, OCC, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    return true;
  }();
  (void)attr_done;
  S8Args a = a0;
  a.n_tiles = premvos::cdiv(a.cout, BN);
  const long m_tiles = (a.M + BM - 1) / BM;
  const bool pw = a.taps == 1 && a.sh == 1 && a.sw == 1 && a.pt == 0 && a.pl == 0 && a.h == a.ho && a.w == a.wo;
  if (pw)
    hipLaunchKernelGGL((conv_bf16x3_s8_kernel<BM, BN, WM, WN, NSTAGE, KC, OCC, true>), dim3((unsigned)(m_tiles * a.n_tiles)), dim3(64 * WM * WN),
                       LDS_BYTES, s, a);
  else
    hipLaunchKernelGGL((conv_bf16x3_s8_kernel<BM, BN, WM, WN, NSTAGE, KC, OCC, false>), dim3((unsigned)(m_tiles * a.n_tiles)), dim3(64 * WM * WN),
                       LDS_BYTES, s, a);
  return premvos::check_launch("conv_bf16x3_s8");
}

// fp32 NHWC -> S8 (the entry of a chain whose first layers run on the fp32 kernels): one thread per (pixel, group of 8 channels)
__global__ void split8_kernel(const float* __restrict__ in, const int in_ps, char* __restrict__ out, const int out_ps, const long pixels,
                              const int groups, const int c) {
  const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= pixels * groups) return;
  const long pix = u / groups;
  const int g = (int)(u - pix * groups);
  const float* src = in + pix * in_ps + g * 8;
  float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
  if (g * 8 + 8 <= c) {
    v0 = premvos::ld4(src);
    v1 = premvos::ld4(src + 4);
  } else {                                                 // a partial last group: absent channels are stored as zeros
    float t[8];
    for (int j = 0; j < 8; ++j) t[j] = g * 8 + j < c ? src[j] : 0.f;
    v0 = make_float4(t[0], t[1], t[2], t[3]);
    v1 = make_float4(t[4], t[5], t[6], t[7]);
  }
  premvos::store_split8(out + (pix * out_ps + g * 8) * 4, v0, v1);
}

}  // namespace

extern "C" int premvos_split8_f32(const float* in, int32_t in_ps, void* out_s8, int32_t out_ps, int64_t pixels, int32_t c, void* stream) {
  PV_REQUIRE(in && out_s8 && pixels > 0 && c > 0, "split8: bad arguments");
  PV_REQUIRE(in_ps % 4 == 0 && out_ps % 8 == 0 && out_ps >= (c + 7) / 8 * 8 && in_ps >= c && premvos::aligned16(in) && premvos::aligned16(out_s8),
             "split8: in_ps %% 4, out_ps %% 8 == 0, out_ps >= roundup(c, 8), 16-byte aligned pixels");
  const int groups = (c + 7) / 8;
  const long units = pixels * groups;
  hipLaunchKernelGGL(split8_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), in, in_ps,
                     static_cast<char*>(out_s8), out_ps, (long)pixels, groups, c);
  return premvos::check_launch("split8");
}

extern "C" int premvos_conv_bf16x3_s8_f32(const premvos_conv_desc* dp, const void* in_s8, const void* wgt_s8, void* out_s8,
                                          int32_t out_s8_ps, const void* res_s8, int32_t res_s8_ps, int32_t tile, void* stream) {
  PV_REQUIRE(dp != nullptr && in_s8 != nullptr && wgt_s8 != nullptr, "conv_bf16x3_s8: null pointer");
  const premvos_conv_desc& d = *dp;
  PV_REQUIRE(d.out != nullptr || out_s8 != nullptr, "conv_bf16x3_s8: no output");
  PV_REQUIRE(d.n > 0 && d.h > 0 && d.w > 0 && d.ho > 0 && d.wo > 0 && d.cin > 0 && d.cout > 0, "conv_bf16x3_s8: bad dims");
  PV_REQUIRE(d.kh >= 1 && d.kw >= 1 && d.kh * d.kw <= 49 && d.sh >= 1 && d.sw >= 1 && d.dh >= 1 && d.dw >= 1, "conv_bf16x3_s8: bad geometry");
  PV_REQUIRE(d.cin % 8 == 0 && d.in_ps % 8 == 0 && d.in_ps >= d.cin && premvos::aligned16(in_s8) && (reinterpret_cast<uintptr_t>(in_s8) & 31u) == 0,
             "conv_bf16x3_s8: S8 input needs cin %% 8 == 0, in_ps %% 8 == 0 and a 32-byte aligned channel window");
  PV_REQUIRE(d.cout % 8 == 0 && d.cout_pad % 32 == 0 && d.cout_pad >= d.cout, "conv_bf16x3_s8: cout %% 8 == 0, cout_pad %% 32 == 0");
  PV_REQUIRE(d.out == nullptr || (d.out_ps % 4 == 0 && d.out_ps >= d.cout && premvos::aligned16(d.out)), "conv_bf16x3_s8: fp32 output: out_ps %% 4 == 0, 16-byte aligned");
  PV_REQUIRE(out_s8 == nullptr || (out_s8_ps % 8 == 0 && out_s8_ps >= d.cout && (reinterpret_cast<uintptr_t>(out_s8) & 31u) == 0),
             "conv_bf16x3_s8: S8 output: out_s8_ps %% 8 == 0, 32-byte aligned channel window");
  PV_REQUIRE(d.res == nullptr || (d.res_ps % 4 == 0 && d.res_ps >= d.cout && premvos::aligned16(d.res)), "conv_bf16x3_s8: residual: res_ps %% 4 == 0, 16-byte aligned");
  PV_REQUIRE(d.bias == nullptr || premvos::aligned16(d.bias), "conv_bf16x3_s8: bias must be 16-byte aligned");
  PV_REQUIRE(res_s8 == nullptr || (d.res == nullptr && res_s8_ps % 8 == 0 && res_s8_ps >= d.cout && (reinterpret_cast<uintptr_t>(res_s8) & 31u) == 0),
             "conv_bf16x3_s8: an S8 residual excludes an fp32 one; res_s8_ps %% 8 == 0, 32-byte aligned channel window");
  PV_REQUIRE(d.act == PREMVOS_ACT_NONE || d.act == PREMVOS_ACT_RELU || d.act == PREMVOS_ACT_LEAKY, "conv_bf16x3_s8: bad activation");
  PV_REQUIRE(d.out_mode == PREMVOS_OUT_NHWC, "conv_bf16x3_s8: NHWC output only");
  S8Args a;
  a.in = static_cast<const char*>(in_s8);
  a.wgt = static_cast<const char*>(wgt_s8);
  a.bias = d.bias; a.res = d.res; a.out = d.out; a.out_s8 = static_cast<char*>(out_s8);
  a.res_s8 = static_cast<const char*>(res_s8);
  a.M = (long)d.n * d.ho * d.wo;
  a.h = d.h; a.w = d.w; a.ho = d.ho; a.wo = d.wo; a.in_ps = d.in_ps; a.cin_groups = d.cin / 8; a.kc32 = (d.cin + 31) / 32;
  a.taps = d.kh * d.kw; a.kw = d.kw; a.sh = d.sh; a.sw = d.sw; a.dh = d.dh; a.dw = d.dw; a.pt = d.pt; a.pl = d.pl;
  a.cout = d.cout; a.cout_pad = d.cout_pad; a.res_ps = res_s8 != nullptr ? res_s8_ps : d.res_ps; a.out_ps = d.out_ps; a.out_s8_ps = out_s8_ps; a.act = d.act;
  a.n_tiles = 0; a.slope = d.slope;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (tile) {
    case 0: return launch<256, 256, 2, 4, 2>(a, s);       // eight waves, two 128 KB ... 64 KB buffers
    case 1: return launch<256, 128, 2, 2, 3>(a, s);       // four waves, three 48 KB buffers
    case 2: return launch<256, 128, 2, 2, 2>(a, s);
    case 3: return launch<256, 128, 4, 2, 3>(a, s);       // eight waves of 64 x 64
    case 4: return launch<128, 128, 2, 2, 3>(a, s);       // four waves of 64 x 64, 96 KB: small layers
    case 5: return launch<128, 128, 2, 2, 2>(a, s);       // 64 KB: two workgroups per CU
    case 6: return launch<256, 128, 2, 2, 3, 16, 2>(a, s);   // 64-byte rows: 3 x 24 KB, two workgroups of four waves per CU
    case 7: return launch<256, 128, 2, 2, 2, 16, 2>(a, s);   // 2 x 24 KB
    case 8: return launch<128, 256, 2, 2, 3, 16, 2>(a, s);   // the transposed wave tile (64 x 128 per wave)
    case 9: return launch<256, 64, 4, 1, 3, 32>(a, s);    // narrow layers (cout <= 64): four waves of 64 x 64, 3 x 40 KB
    case 10: {                                            // ping-pong 256 x 256 (pointwise layers whose operands span < 4 GB each)
      const bool pw = a.taps == 1 && a.sh == 1 && a.sw == 1 && a.pt == 0 && a.pl == 0 && a.h == a.ho && a.w == a.wo;
      const bool small = a.M * (long)a.in_ps * 4 < (1L << 32) && (long)a.cout_pad * a.kc32 * 128 < (1L << 32);
      const bool padded = a.cout_pad % 256 == 0;         // (ops.pack_conv_s8 pads the weight rows to whole 256-row tiles)
      return pw && small && padded ? launch_pp(a, s) : launch<256, 256, 2, 4, 2>(a, s);
    }
    case 11: return launch<256, 256, 2, 2, 2>(a, s);      // four waves of 128 x 128 (one per SIMD, 256 accumulator registers): 16 fragment reads per 48 MFMAs
    default: return premvos::fail(PREMVOS_EINVAL, "conv_bf16x3_s8: unknown tile %d", tile);
  }
}
