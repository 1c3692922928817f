// PWC-Net cost volume for the instantiation the network uses (md = 4 -> 9x9 = 81 displacements, kernel 1,
// strides 1; PWCNet.py:69), LDS-tiled for gfx950.
//
// What it replaces: corr_cuda_kernel.cu:59-127 (one 32-thread block per output pixel, the f1 pixel in shared
// memory, a serial loop over the 81 displacements with a per-thread partial sum and a serial final reduce) plus the
// two blob_rearrange passes and the two zero fills of corr_cuda.c:52-63.
//
// Design (HBM-bound: f1 and f2 are read once, the 81 (+C copied) floats of a pixel are written once):
//   * a workgroup owns an 8 x 32 pixel tile of one image; the (8+8) x (32+8) halo of f2 is staged in LDS 16 channels
//     at a time with a 20-float pixel stride (5 x 16 B, odd: ds_read_b128 of 16 consecutive pixels is conflict-free);
//   * TWO threads own one pixel: its f1 channels sit in the registers of both, thread 0 keeps the sums of displacement rows
//     dy = -4..0 (45 accumulators), thread 1 those of dy = 1..4 (36): per chunk a thread reads 45 (36) x 4 float4 from LDS for
//     as many x 16 FMAs (one LDS read per 4 FMAs -- the LDS pipe, not HBM, would bound a naive "one lane per displacement"
//     mapping).  Halving the accumulators per thread is what lets FOUR waves share a SIMD (<= 128 VGPRs): the kernel is
//     VALU-issue- and latency-bound (a wave64 v_fma_f32 holds its SIMD for 4 cycles; with two or three waves per SIMD the
//     waves were parked on s_waitcnt / barriers for half of their cycles and nothing else was ready);
//   * the sums leave through LDS so that the 81 floats of a pixel go out as one contiguous run in 16-byte pieces (two
//     passes of 44 / 37 floats reuse the f2 tile's LDS); LeakyReLU(0.1) (PWCNet.py:198) is fused, and the torch.cat copy
//     of c1 (PWCNet.py:213) is written from the registers that hold the f1 pixel.
//   * WARP: f2 is not read but produced on the fly by the backward bilinear warp of image-2 features by the
//     up-sampled flow (PWCDCNet.warp, PWCNet.py:140-176: same float sequence as warp_kernel in flow_ops.hip), so the
//     warped feature map never exists in HBM.
// hipcc-flags: -fno-slp-vectorize   (the SLP pass packs neighbouring displacements into v_pk_fma_f32 at the price of
// ~1100 register moves and odd-width LDS reads per chunk: 3x slower)
#include <type_traits>

#include <stdlib.h>

#include "common.h"
#include "warp_math.h"



namespace {

constexpr int MD = 4, D = 2 * MD + 1;
constexpr int D0 = 5;               // displacement rows of a pixel's first thread (dy = -4..0); the second has D - D0 = 4

struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };     // 16 bytes at 4-byte alignment

// T_H x T_W pixel tile, two threads per pixel, CCH channels per LDS chunk, OCC = waves per SIMD the register allocation must allow.
// PREFETCH (round 6): the global loads of chunk k + 1 are issued BEFORE chunk k is multiplied (the staging registers are dead once
// their values are in LDS; only the f1 registers double) -- the committed kernel requested a chunk, waited, multiplied, and so paid
// the memory latency of every chunk pass in full: its load phase ALONE took 101 of 150 us at the 128x224 level (section 7.3).
// PPT = 2 (round 6 experiment): a thread owns TWO vertically adjacent pixels -- displacement row dy of pixel (x, y + 1) multiplies the
// same f2 row as row dy + 1 of pixel (x, y), so a thread reads nrows + 1 halo rows from LDS for 2 x nrows rows of sums (0.6x the LDS
// reads per FMA; twice the accumulators).
template <bool WARP, int T_H, int T_W, int CCH, int OCC, bool PREFETCH, int PPT = 1>
__global__ __launch_bounds__(2 * T_H * T_W / PPT, OCC) void corr81_tile_kernel(const float* __restrict__ f1, int f1_ps,
                                                                        const float* __restrict__ f2, int f2_ps,
                                                                        const float* __restrict__ flow, int flow_ps, float fscale,
                                                                        float* __restrict__ out, int out_ps, int h, int w, int c,
                                                                        float slope, int copy_f1, int tiles_x, int tiles_y) {
  constexpr int NPIX = T_H * T_W, NTH = NPIX / PPT, NT = 2 * NTH;
  constexpr int HALO_H = T_H + 2 * MD, HALO_W = T_W + 2 * MD;
  constexpr int PSTR = CCH + 4;      // padded pixel stride (floats): 5 (9) x 16 B, odd -> ds_read_b128 of consecutive pixels is conflict-free
  constexpr int NA = 44, NB = D * D - NA, PB = 40;
  constexpr int LDS_HALO = HALO_H * HALO_W * PSTR, LDS_OUT = NPIX * NA;
  constexpr int LDS_FLOATS = LDS_HALO > LDS_OUT ? LDS_HALO : LDS_OUT;
  constexpr int Q = CCH / 4;
  constexpr int NUNITS = HALO_H * HALO_W * Q, NSTG = (NUNITS + NT - 1) / NT;
  static_assert(NTH % 64 == 0 && T_H % PPT == 0, "a wave is one half of 64 pixels (or pixel pairs)");
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  const int tid = threadIdx.x;
  const int tix = tid % NTH, half = tid / NTH;          // pixel (pair) of the tile, which half of the displacement rows
  const int tx = tix % T_W, ty = (tix / T_W) * PPT;     // (the thread's first pixel; its second one is the pixel below)
  const int dy0 = half ? D0 : 0;                        // this thread's displacement rows are [dy0, dy0 + (half ? D - D0 : D0))
  // Tile order: the dispatcher places workgroup b on XCD b % 8 (private 4 MB L2 each).  Give every XCD a contiguous run of
  // the (image, tile row, tile column) raster, so that the tiles sharing f2 halo rows / columns read them through ONE L2
  // (with the plain order 55 % of the f2 requests missed L2: 2.2x the algorithmic fetch).  Pure speed, any order is correct.
  const int tile = premvos::xcd_contiguous(blockIdx.x, gridDim.x);
  const int n = tile / (tiles_x * tiles_y), trem = tile - n * (tiles_x * tiles_y);
  const int x0 = (trem % tiles_x) * T_W, y0 = (trem / tiles_x) * T_H;
  const int x = x0 + tx;
  const long img = (long)n * h * w;
  bool valid[PPT];
  const float* a_ptr[PPT];
  int pidx[PPT];                                        // index of the pixel in the tile's row-major order (output staging)
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int y = y0 + ty + p;
    valid[p] = x < w && y < h;
    a_ptr[p] = f1 + (img + (long)(valid[p] ? y : 0) * w + (valid[p] ? x : 0)) * f1_ps;
    pidx[p] = (ty + p) * T_W + tx;
  }

  float acc[PPT][D0 * D];                               // (half 1 uses the first (D - D0) * D of them)
#pragma unroll
  for (int p = 0; p < PPT; ++p)
#pragma unroll
    for (int i = 0; i < D0 * D; ++i) acc[p][i] = 0.f;

  // ---- staging of the f2 halo tile: HALO_H x HALO_W pixels x Q float4 per chunk, NSTG per thread.  Out-of-range lanes load from a
  // clamped address and are zeroed when written to LDS: a select on a register with a load in flight would force the wave to wait
  // for that load before issuing the next one.
  auto halo = [&](int j, int k0, int& slot, long& pix, int& ch, int& gy, int& gx) {
    int i = tid + j * NT;
    const bool in = NUNITS % NT == 0 || i < NUNITS;
    i = in ? i : 0;
    const int px = i / Q, q = i - px * Q;
    const int hy = px / HALO_W, hx = px - hy * HALO_W;
    gy = y0 + hy - MD, gx = x0 + hx - MD;
    ch = k0 + q * 4;
    slot = in ? px * PSTR + q * 4 : -1;
    const bool ok = in && (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w && ch < c;
    pix = img + (long)(ok ? gy : 0) * w + (ok ? gx : 0);
    if (!ok) ch = 0;
    return ok;
  };
  float4 v[NSTG];
  float4 a[PPT][Q], an[PPT][Q];
  auto request = [&](int k0, float4 (&av)[PPT][Q]) {     // all global loads of one chunk, nothing waits
#pragma unroll
    for (int j = 0; j < NSTG; ++j) {
      int slot, ch, gy, gx;
      long pix;
      halo(j, k0, slot, pix, ch, gy, gx);
      if constexpr (WARP) {
        const float* fl = flow + pix * flow_ps;
        v[j] = premvos::warp_sample4(f2 + img * f2_ps + ch, f2_ps, fl[0] * fscale, fl[1] * fscale, gx, gy, h, w);
      } else {
        v[j] = *reinterpret_cast<const float4*>(f2 + pix * f2_ps + ch);
      }
    }
#pragma unroll
    for (int p = 0; p < PPT; ++p)
#pragma unroll
      for (int q = 0; q < Q; ++q) av[p][q] = *reinterpret_cast<const float4*>(a_ptr[p] + (valid[p] && k0 + q * 4 < c ? k0 + q * 4 : 0));
  };

  if (PREFETCH) request(0, a);
  for (int k0 = 0; k0 < c; k0 += CCH) {
    if (k0) __syncthreads();       // every wave is done reading the previous chunk
    if (!PREFETCH) request(k0, a);
#pragma unroll
    for (int j = 0; j < NSTG; ++j) {
      int slot, ch, gy, gx;
      long pix;
      const bool ok = halo(j, k0, slot, pix, ch, gy, gx);
      if (slot >= 0) *reinterpret_cast<float4*>(&lds[slot]) = ok ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < PPT; ++p)
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const bool ok = valid[p] && k0 + q * 4 < c;
        if (!ok) a[p][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        // torch.cat((corr, c1, ...)) (PWCNet.py:213): c1 goes out straight from the registers (the window starts 81 floats
        // into the pixel, so only 4-byte alignment is known)
        if (copy_f1 && ok && half == (q & 1))
          *reinterpret_cast<f4u*>(out + (img + (long)(y0 + ty + p) * w + x) * out_ps + D * D + k0 + q * 4) = f4u{a[p][q].x, a[p][q].y, a[p][q].z, a[p][q].w};
      }
    __syncthreads();
    const bool more = k0 + CCH < c;
    if (PREFETCH && more) request(k0 + CCH, an);         // in flight under the multiplications below
    // ---- 81 displacements x CCH channels ------------------------------------------------------
    const float* base = &lds[((ty + dy0) * HALO_W + tx) * PSTR];       // (ty = the thread's FIRST pixel)
    // per (displacement row, 4-channel group): 9 independent LDS reads in flight, then 9 independent FMA chains
    auto rows = [&](auto nrows_) {
      constexpr int nrows = decltype(nrows_)::value;
#pragma unroll
      for (int rr = 0; rr < nrows + PPT - 1; ++rr)       // halo row rr below the first pixel's first displacement row
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          float4 b[D];
#pragma unroll
          for (int dx = 0; dx < D; ++dx) b[dx] = *reinterpret_cast<const float4*>(base + (rr * HALO_W + dx) * PSTR + q * 4);
#pragma unroll
          for (int p = 0; p < PPT; ++p) {
            const int dy = rr - p;                       // pixel p sits p rows lower: the same halo row is its displacement row rr - p
            if (dy < 0 || dy >= nrows) continue;
#pragma unroll
            for (int dx = 0; dx < D; ++dx) {
              float s = acc[p][dy * D + dx];
              s = fmaf(a[p][q].x, b[dx].x, s);
              s = fmaf(a[p][q].y, b[dx].y, s);
              s = fmaf(a[p][q].z, b[dx].z, s);
              s = fmaf(a[p][q].w, b[dx].w, s);
              acc[p][dy * D + dx] = s;
            }
          }
          __builtin_amdgcn_sched_barrier(0);        // one round of reads ahead at most
        }
    };
    if (half == 0) rows(std::integral_constant<int, D0>{});          // wave-uniform: a wave is one half of 64 pixels
    else rows(std::integral_constant<int, D - D0>{});
    if (PREFETCH && more) {
#pragma unroll
      for (int p = 0; p < PPT; ++p)
#pragma unroll
        for (int q = 0; q < Q; ++q) a[p][q] = an[p][q];
    }
  }

  // ---- mean over C (sum / (float)sumelems, corr_cuda_kernel.cu:124-126), LeakyReLU, coalesced runs ----------------
  // A power-of-two C divides exactly by multiplying with 1/C (same bits as the IEEE division, a tenth of the instructions).
  const float fc = (float)c;
  const bool pow2 = (c & (c - 1)) == 0;
  const float rc = 1.0f / fc;
  const int wave = tid >> 6, lane = tid & 63;
  const bool wide = ((reinterpret_cast<uintptr_t>(out) & 15u) == 0) && (out_ps & 3) == 0;
  auto mean_act = [&](float sum) {
    const float v_ = pow2 ? sum * rc : sum / fc;
    return v_ < 0.f ? v_ * slope : v_;
  };
  // The 81 sums of a pixel leave through LDS so that the pixel's run goes out in 16-byte pieces (the run starts 16-byte
  // aligned when `wide`): pass A = elements [0,44) = 11 float4 per pixel, all from the pixel's thread 0, LDS row pitch 44 (the
  // rows tile LDS linearly: conflict-free b128 writes and reads); pass B = elements [44,81) = 9 float4 + 1 float, row pitch 40:
  // element 44 (dy = 0, dx = +4) from thread 0, elements [45,81) = the 36 sums of thread 1.
  static_assert(NA % 4 == 0 && NB == 37 && D0 * D == NA + 1 && NPIX * PB <= LDS_FLOATS, "output staging layout");
  auto stage_a = [&]() {
    if (half == 0) {
#pragma unroll
      for (int p = 0; p < PPT; ++p)
#pragma unroll
        for (int j = 0; j < NA / 4; ++j)
          *reinterpret_cast<float4*>(&lds[pidx[p] * NA + 4 * j]) =
              make_float4(mean_act(acc[p][4 * j]), mean_act(acc[p][4 * j + 1]), mean_act(acc[p][4 * j + 2]), mean_act(acc[p][4 * j + 3]));
    }
  };
  auto stage_b = [&]() {
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      if (half == 0) {
        lds[pidx[p] * PB] = mean_act(acc[p][NA]);
      } else {
#pragma unroll
        for (int e = 0; e < (D - D0) * D; ++e) lds[pidx[p] * PB + 1 + e] = mean_act(acc[p][e]);
      }
    }
  };
  auto pix_of = [&](int p, bool& ok) {
    const int yy = y0 + p / T_W, xx = x0 + p % T_W;
    ok = yy < h && xx < w;
    return (img + (long)yy * w + xx) * out_ps;
  };
  auto flush = [&](int e0, int n4, int pitch, int tail) {     // n4 float4 (+ `tail` single floats) per pixel row of LDS
    __syncthreads();
    if (wide) {
      for (int u = tid; u < NPIX * n4; u += NT) {
        const int p = u / n4, j = u - p * n4;
        bool ok;
        const long o = pix_of(p, ok);
        if (ok) *reinterpret_cast<float4*>(out + o + e0 + 4 * j) = *reinterpret_cast<const float4*>(&lds[p * pitch + 4 * j]);
      }
      if (tail && half == 0) {
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
          bool ok;
          const long o = pix_of(pidx[p], ok);
          if (ok) out[o + e0 + 4 * n4] = lds[pidx[p] * pitch + 4 * n4];
        }
      }
    } else {                                          // unaligned destination: one wave per pixel run, 4-byte stores
      const int ne = 4 * n4 + tail;
      for (int p = wave; p < NPIX; p += NT / 64) {
        bool ok;
        const long o = pix_of(p, ok);
        if (ok && lane < ne) out[o + e0 + lane] = lds[p * pitch + lane];
      }
    }
  };
  __syncthreads();                                   // f2 tile no longer needed
  stage_a();
  flush(0, NA / 4, NA, 0);
  __syncthreads();
  stage_b();
  flush(NA, NB / 4, PB, 1);
}

template <bool WARP, int T_H, int T_W, int CCH, int OCC, bool PREFETCH, int PPT = 1>
int launch_variant(const float* f1, int f1_ps, const float* f2, int f2_ps, const float* flow, int flow_ps, float fscale, float* out,
                   int out_ps, int n, int h, int w, int c, float slope, int copy_f1, hipStream_t s) {
  const int tx = premvos::cdiv(w, T_W), ty = premvos::cdiv(h, T_H);
  hipLaunchKernelGGL((corr81_tile_kernel<WARP, T_H, T_W, CCH, OCC, PREFETCH, PPT>), dim3(tx * ty * n), dim3(2 * T_H * T_W / PPT), 0, s, f1, f1_ps, f2,
                     f2_ps, flow, flow_ps, fscale, out, out_ps, h, w, c, slope, copy_f1, tx, ty);
  return premvos::check_launch("corr81_tile");
}

}  // namespace

namespace premvos {
// PREMVOS_CORR_VARIANT (developer A/B, tools/time_corr.py): 0 = the round-2 form (8 x 32 tile, 16-channel chunks, four waves per
// SIMD, load - wait - multiply); the others differ in tile, chunk depth, occupancy and whether the next chunk's loads are in flight
// under the multiplications.  Every variant adds the same products in the same order: bit-identical outputs.
int corr81_tile(const float* f1, int f1_ps, const float* f2, int f2_ps, const float* flow, int flow_ps, float fscale,
                float* out, int out_ps, int n, int h, int w, int c, float slope, int copy_f1, hipStream_t s) {
  static const int env_variant = [] {
    const char* e = getenv("PREMVOS_CORR_VARIANT");
    return e ? atoi(e) : -1;
  }();
  // Default (profiles/r06_corr_variants.txt, 16 pairs): whole 128-byte pixel rows per chunk on an 8 x 16 tile (variant 5: 256-thread
  // workgroups, two per CU) where the map has <= 32 channels -- the 128 x 224 level: 152 -> 126 us, its rows used to be requested as
  // two 64-byte halves 20 ... 30 us apart -- or too few 8 x 32 tiles to give every CU two (the 32 x 56 / 16 x 28 levels: 40 -> 35 us);
  // the 64-channel 64 x 112 level stays on the round-2 form (56 vs 64 us).  The fused warp form keeps the round-2 tile (its
  // bilinear gathers need the registers).
  const long tiles_8x32 = (long)cdiv(w, 32) * cdiv(h, 8) * n;
  // ... and there two pixels per thread (variant 11: 0.6x the LDS reads per FMA) are worth 55 -> 51 us; on small maps that form starves
  // (half the threads per tile) and at C = 32 the level is bound by its memory phases, not by LDS (135 vs 134 us).
  const int rule = flow != nullptr ? 0 : (c <= 32 || tiles_8x32 < 512) ? 5 : c <= 64 ? 11 : 0;
  const int variant = env_variant >= 0 ? env_variant : rule;
#define PV_CORR(TH, TW, CC, OC, PF)                                                                                                   \
  return flow != nullptr ? launch_variant<true, TH, TW, CC, OC, PF>(f1, f1_ps, f2, f2_ps, flow, flow_ps, fscale, out, out_ps, n, h, w, c, \
                                                                     slope, copy_f1, s)                                                 \
                         : launch_variant<false, TH, TW, CC, OC, PF>(f1, f1_ps, f2, f2_ps, flow, flow_ps, fscale, out, out_ps, n, h, w, c, \
                                                                      slope, copy_f1, s)
  switch (variant) {
    case 1: PV_CORR(8, 32, 16, 2, true);
    case 2: PV_CORR(8, 16, 16, 4, false);
    case 3: PV_CORR(8, 16, 16, 3, true);
    case 4: PV_CORR(4, 32, 16, 3, true);
    case 5: PV_CORR(8, 16, 32, 3, false);
    case 6: PV_CORR(8, 16, 32, 2, true);
    case 7: PV_CORR(16, 16, 16, 4, false);
    case 8: PV_CORR(16, 16, 16, 2, true);
    case 9: PV_CORR(8, 32, 32, 2, false);
    case 10: PV_CORR(8, 16, 16, 2, true);
    case 11: return launch_variant<false, 8, 32, 16, 2, false, 2>(f1, f1_ps, f2, f2_ps, flow, flow_ps, fscale, out, out_ps, n, h, w, c, slope, copy_f1, s);
    case 12: return launch_variant<false, 8, 32, 16, 2, true, 2>(f1, f1_ps, f2, f2_ps, flow, flow_ps, fscale, out, out_ps, n, h, w, c, slope, copy_f1, s);
    case 13: return launch_variant<false, 16, 16, 16, 2, false, 2>(f1, f1_ps, f2, f2_ps, flow, flow_ps, fscale, out, out_ps, n, h, w, c, slope, copy_f1, s);
    case 14: return launch_variant<false, 8, 32, 16, 3, false, 2>(f1, f1_ps, f2, f2_ps, flow, flow_ps, fscale, out, out_ps, n, h, w, c, slope, copy_f1, s);
    default: PV_CORR(8, 32, 16, 4, false);
  }
#undef PV_CORR
}
}  // namespace premvos

#ifdef CORR_DBG_ENTRY      // stand-alone timing builds of tools/dev/corr_variants.sh (phases compiled out compute garbage)
namespace premvos { thread_local char g_err[512] = ""; }
extern "C" int corr_dbg(const float* f1, int f1_ps, const float* f2, int f2_ps, float* out, int out_ps, int n, int h, int w, int c,
                        void* stream) {
  return premvos::corr81_tile(f1, f1_ps, f2, f2_ps, nullptr, 0, 0.f, out, out_ps, n, h, w, c, 0.1f, 1, static_cast<hipStream_t>(stream));
}
#endif
