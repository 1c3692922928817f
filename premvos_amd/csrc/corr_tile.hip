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

#include "common.h"
#include "warp_math.h"

namespace {

constexpr int T_H = 8, T_W = 32, MD = 4, D = 2 * MD + 1;
constexpr int HALO_H = T_H + 2 * MD, HALO_W = T_W + 2 * MD;
constexpr int NT = 512;             // two threads per pixel of the tile
constexpr int D0 = 5;               // displacement rows of a pixel's first thread (dy = -4..0); the second has D - D0 = 4
constexpr int CCH = 16;            // channels per LDS chunk
constexpr int PSTR = CCH + 4;      // padded pixel stride (floats)
constexpr int LDS_FLOATS = HALO_H * HALO_W * PSTR;   // 12 800 floats = 51 200 B; two 8-wave workgroups per CU (128 VGPRs)

struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };     // 16 bytes at 4-byte alignment

template <bool WARP>
__global__ __launch_bounds__(NT, 4) void corr81_tile_kernel(const float* __restrict__ f1, int f1_ps,
                                                          const float* __restrict__ f2, int f2_ps,
                                                          const float* __restrict__ flow, int flow_ps, float fscale,
                                                          float* __restrict__ out, int out_ps, int h, int w, int c,
                                                          float slope, int copy_f1, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  const int tid = threadIdx.x;
  const int pid = tid & 255, half = tid >> 8;           // pixel of the tile, which half of the displacement rows
  const int tx = pid & 31, ty = pid >> 5;
  const int dy0 = half ? D0 : 0;                        // this thread's displacement rows are [dy0, dy0 + (half ? D - D0 : D0))
  // Tile order: the dispatcher places workgroup b on XCD b % 8 (private 4 MB L2 each).  Give every XCD a contiguous run of
  // the (image, tile row, tile column) raster, so that the tiles sharing f2 halo rows / columns read them through ONE L2
  // (with the plain order 55 % of the f2 requests missed L2: 2.2x the algorithmic fetch).  Pure speed, any order is correct.
  int tile;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int xcd = id & 7, local = id >> 3, q8 = nwg >> 3, r8 = nwg & 7;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;
  }
  const int n = tile / (tiles_x * tiles_y), trem = tile - n * (tiles_x * tiles_y);
  const int x0 = (trem % tiles_x) * T_W, y0 = (trem / tiles_x) * T_H;
  const int x = x0 + tx, y = y0 + ty;
  const bool valid = x < w && y < h;
  const long img = (long)n * h * w;
  const float* a_ptr = f1 + (img + (long)(valid ? y : 0) * w + (valid ? x : 0)) * f1_ps;

  float acc[D0 * D];                                    // (half 1 uses the first (D - D0) * D of them)
#pragma unroll
  for (int i = 0; i < D0 * D; ++i) acc[i] = 0.f;

  for (int k0 = 0; k0 < c; k0 += CCH) {
    if (k0) __syncthreads();       // every wave is done reading the previous chunk
    // ---- stage the f2 halo tile: 640 pixels x 4 float4 = 10 per thread, all loads issued before the first LDS write
    constexpr int NSTG = HALO_H * HALO_W * (CCH / 4) / NT;
    static_assert(NSTG * NT == HALO_H * HALO_W * (CCH / 4), "staging loop must divide evenly");
    // (out-of-range lanes load from a clamped address and are zeroed when written to LDS: a select on a register with
    //  a load in flight would force the wave to wait for that load before issuing the next one)
    float4 v[NSTG];
    auto halo = [&](int j, int& slot, long& pix, int& ch, int& gy, int& gx) {
      const int i = tid + j * NT;
      const int px = i >> 2, q = i & 3;
      const int hy = px / HALO_W, hx = px - hy * HALO_W;
      gy = y0 + hy - MD, gx = x0 + hx - MD;
      ch = k0 + q * 4;
      slot = px * PSTR + q * 4;
      const bool ok = (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w && ch < c;
      pix = img + (long)(ok ? gy : 0) * w + (ok ? gx : 0);
      if (!ok) ch = 0;
      return ok;
    };
#pragma unroll
    for (int j = 0; j < NSTG; ++j) {
      int slot, ch, gy, gx;
      long pix;
      halo(j, slot, pix, ch, gy, gx);
      if constexpr (WARP) {
        const float* fl = flow + pix * flow_ps;
        v[j] = premvos::warp_sample4(f2 + img * f2_ps + ch, f2_ps, fl[0] * fscale, fl[1] * fscale, gx, gy, h, w);
      } else {
#ifdef CORR_DBG_NO_LOAD
        v[j] = make_float4((float)pix, (float)ch, 0.f, 1.f);
#else
        v[j] = *reinterpret_cast<const float4*>(f2 + pix * f2_ps + ch);
#endif
      }
    }
    float4 a[CCH / 4];
#pragma unroll
    for (int q = 0; q < CCH / 4; ++q) a[q] = *reinterpret_cast<const float4*>(a_ptr + (valid && k0 + q * 4 < c ? k0 + q * 4 : 0));
#pragma unroll
    for (int j = 0; j < NSTG; ++j) {
      int slot, ch, gy, gx;
      long pix;
      const bool ok = halo(j, slot, pix, ch, gy, gx);
      *reinterpret_cast<float4*>(&lds[slot]) = ok ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < CCH / 4; ++q) {
      const bool ok = valid && k0 + q * 4 < c;
      if (!ok) a[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      // torch.cat((corr, c1, ...)) (PWCNet.py:213): c1 goes out straight from the registers (the window starts 81 floats
      // into the pixel, so only 4-byte alignment is known)
      if (copy_f1 && ok && half == (q & 1)) *reinterpret_cast<f4u*>(out + (img + (long)y * w + x) * out_ps + D * D + k0 + q * 4) = f4u{a[q].x, a[q].y, a[q].z, a[q].w};
    }
    __syncthreads();
#ifdef CORR_DBG_NO_COMPUTE
    acc[0] += lds[pid] + a[0].x;
    continue;
#endif
    // ---- 81 displacements x 16 channels ------------------------------------------------------
    const float* base = &lds[((ty + dy0) * HALO_W + tx) * PSTR];
    // per (displacement row, 4-channel group): 9 independent LDS reads in flight, then 9 independent FMA chains
    auto rows = [&](auto nrows_) {
      constexpr int nrows = decltype(nrows_)::value;
#pragma unroll
      for (int dy = 0; dy < nrows; ++dy)
#pragma unroll
        for (int q = 0; q < CCH / 4; ++q) {
          float4 b[D];
#pragma unroll
          for (int dx = 0; dx < D; ++dx) b[dx] = *reinterpret_cast<const float4*>(base + (dy * HALO_W + dx) * PSTR + q * 4);
#pragma unroll
          for (int dx = 0; dx < D; ++dx) {
            float s = acc[dy * D + dx];
            s = fmaf(a[q].x, b[dx].x, s);
            s = fmaf(a[q].y, b[dx].y, s);
            s = fmaf(a[q].z, b[dx].z, s);
            s = fmaf(a[q].w, b[dx].w, s);
            acc[dy * D + dx] = s;
          }
          __builtin_amdgcn_sched_barrier(0);        // one round of reads ahead at most: 128 VGPRs have no room for more
        }
    };
    if (half == 0) rows(std::integral_constant<int, D0>{});          // wave-uniform: a wave is one half of 64 pixels
    else rows(std::integral_constant<int, D - D0>{});
  }

  // ---- mean over C (sum / (float)sumelems, corr_cuda_kernel.cu:124-126), LeakyReLU, coalesced runs ----------------
  // A power-of-two C divides exactly by multiplying with 1/C (same bits as the IEEE division, a tenth of the instructions).
  const float fc = (float)c;
  const bool pow2 = (c & (c - 1)) == 0;
  const float rc = 1.0f / fc;
  const int wave = tid >> 6, lane = tid & 63;
  const bool wide = ((reinterpret_cast<uintptr_t>(out) & 15u) == 0) && (out_ps & 3) == 0;
  auto mean_act = [&](float sum) {
    const float v = pow2 ? sum * rc : sum / fc;
    return v < 0.f ? v * slope : v;
  };
  // The 81 sums of a pixel leave through LDS so that the pixel's run goes out in 16-byte pieces (the run starts 16-byte
  // aligned when `wide`): pass A = elements [0,44) = 11 float4 per pixel, all from the pixel's thread 0, LDS row pitch 44 (the
  // rows tile LDS linearly: conflict-free b128 writes and reads); pass B = elements [44,81) = 9 float4 + 1 float, row pitch 40:
  // element 44 (dy = 0, dx = +4) from thread 0, elements [45,81) = the 36 sums of thread 1.
  constexpr int NA = 44, NB = D * D - NA, PB = 40;
  static_assert(NA % 4 == 0 && NB == 37 && D0 * D == NA + 1 && 256 * NA <= LDS_FLOATS && 256 * PB <= LDS_FLOATS, "output staging layout");
  auto stage_a = [&]() {
    if (half == 0) {
#pragma unroll
      for (int j = 0; j < NA / 4; ++j)
        *reinterpret_cast<float4*>(&lds[pid * NA + 4 * j]) = make_float4(mean_act(acc[4 * j]), mean_act(acc[4 * j + 1]), mean_act(acc[4 * j + 2]), mean_act(acc[4 * j + 3]));
    }
  };
  auto stage_b = [&]() {
    if (half == 0) {
      lds[pid * PB] = mean_act(acc[NA]);
    } else {
#pragma unroll
      for (int e = 0; e < (D - D0) * D; ++e) lds[pid * PB + 1 + e] = mean_act(acc[e]);
    }
  };
  auto pix_of = [&](int p, bool& ok) {
    const int yy = y0 + (p >> 5), xx = x0 + (p & 31);
    ok = yy < h && xx < w;
    return (img + (long)yy * w + xx) * out_ps;
  };
  auto flush = [&](int e0, int n4, int pitch, int tail) {     // n4 float4 (+ `tail` single floats) per pixel row of LDS
    __syncthreads();
    if (wide) {
      for (int u = tid; u < 256 * n4; u += NT) {
        const int p = u / n4, j = u - p * n4;
        bool ok;
        const long o = pix_of(p, ok);
        if (ok) *reinterpret_cast<float4*>(out + o + e0 + 4 * j) = *reinterpret_cast<const float4*>(&lds[p * pitch + 4 * j]);
      }
      if (tail && half == 0) {
        bool ok;
        const long o = pix_of(pid, ok);
        if (ok) out[o + e0 + 4 * n4] = lds[pid * pitch + 4 * n4];
      }
    } else {                                          // unaligned destination: one wave per pixel run, 4-byte stores
      const int ne = 4 * n4 + tail;
      for (int p = wave; p < T_H * T_W; p += NT / 64) {
        bool ok;
        const long o = pix_of(p, ok);
        if (ok && lane < ne) out[o + e0 + lane] = lds[p * pitch + lane];
      }
    }
  };
#ifdef CORR_DBG_NO_OUTPUT
  {
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < D0 * D; ++e) t += acc[e];
    if (t == 1234.5f) out[0] = t;
    return;
  }
#endif
  __syncthreads();                                   // f2 tile no longer needed
  stage_a();
  flush(0, NA / 4, NA, 0);
  __syncthreads();
  stage_b();
  flush(NA, NB / 4, PB, 1);
}

}  // namespace

namespace premvos {
int corr81_tile(const float* f1, int f1_ps, const float* f2, int f2_ps, const float* flow, int flow_ps, float fscale,
                float* out, int out_ps, int n, int h, int w, int c, float slope, int copy_f1, hipStream_t s) {
  const int tx = cdiv(w, T_W), ty = cdiv(h, T_H);
  const dim3 grid(tx * ty * n);
  if (flow != nullptr)
    hipLaunchKernelGGL(corr81_tile_kernel<true>, grid, dim3(NT), 0, s, f1, f1_ps, f2, f2_ps, flow, flow_ps, fscale, out,
                       out_ps, h, w, c, slope, copy_f1, tx, ty);
  else
    hipLaunchKernelGGL(corr81_tile_kernel<false>, grid, dim3(NT), 0, s, f1, f1_ps, f2, f2_ps, flow, flow_ps, fscale, out,
                       out_ps, h, w, c, slope, copy_f1, tx, ty);
  return check_launch("corr81_tile");
}
}  // namespace premvos

#ifdef CORR_DBG_ENTRY      // stand-alone timing builds of tools/dev/corr_variants.sh (phases compiled out compute garbage)
namespace premvos { thread_local char g_err[512] = ""; }
extern "C" int corr_dbg(const float* f1, int f1_ps, const float* f2, int f2_ps, float* out, int out_ps, int n, int h, int w, int c,
                        void* stream) {
  return premvos::corr81_tile(f1, f1_ps, f2, f2_ps, nullptr, 0, 0.f, out, out_ps, n, h, w, c, 0.1f, 1, static_cast<hipStream_t>(stream));
}
#endif
