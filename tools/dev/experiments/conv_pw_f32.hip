// Pointwise (1x1) fp32 conv as a PERSISTENT implicit GEMM on the gfx950 fp32 matrix pipe (premvos_conv2d_f32, tile_hint 6).
//
// Same tile, same LDS layout, same MFMA order as the 128x128 / 16-deep-stage instance of conv_igemm_f32.hip -- every output element
// is the same fmaf chain in k order, so the two kernels are BIT-IDENTICAL and the choice between them is an order-neutral knob
// (premvos_amd/ops.py: numerics_key).  What differs is who runs a tile and what happens between two tiles:
//
// * Round 5 measured (tools/dev/r05_timeline.py, profiles/r05_timeline.txt: per-workgroup phase stamps of the one-tile-per-workgroup
//   kernel on the 728 -> 728 middle-flow layer): a tile lives 124 us, of which 8 us are its prologue (kernel arguments, address
//   arithmetic and the first HBM round trip, issued in competition with two workgroups that sit in their MFMA loops), 10.8 us its
//   epilogue, and the slot it frees stays EMPTY for another 10 us on average (p90: 28 us) until the dispatcher refills it -- only
//   37 % of the CU time had all three resident workgroups inside their main loops, 6 % none (start + ragged end).
// * Here 3 workgroups per CU are launched ONCE and walk a static tile list (XCD-contiguous, rotated so that every workgroup meets
//   every column tile -- the 728-wide layers' sixth tile is a quarter shorter); between two tiles a workgroup requests the next
//   tile's first stage BEFORE it runs the epilogue of the current one, stages its output through the second operand buffer (16 rows
//   per wave and pass, wave-private: no barrier) and stores the first stage into the first operand buffer meanwhile: one barrier
//   per tile beyond the K loop's own.  The leftover rows of tiles (a partial round) go to conv_igemm_f32.hip's k-sliced tail
//   launch + fixed-order reduce -- the same arithmetic as that kernel's own `tail_m_tiles` / `tail_split_k` configuration.
//
// Reference call sites replaced: the 1x1 convolutions of refinement_net/network/deeplab/core/xception.py:154-178,508-550 (pointwise
// halves of the separable convs, shortcuts) and proposal_net/basemodel.py:49-89 (bottleneck conv1 / conv3 / shortcuts).
#include "common.h"
#include <atomic>
#include <type_traits>

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace premvos {
int launch_igemm_tail128(const premvos_conv_desc& d, int mt0, int tail_split_k, hipStream_t s);      // conv_igemm_f32.hip
long igemm_tail128_ws_bytes(const premvos_conv_desc& d, int tail_rows, int tail_split_k);
int igemm_tail128_splits(const premvos_conv_desc& d, int tail_split_k);
}

#ifdef PV_DBG_TIMELINE          // developer build: per-workgroup, per-tile phase stamps (tools/dev/r05_timeline_pw.py)
__device__ unsigned long long g_tlp[1 << 20];     // [workgroup][tile 0..15][4]: wall at first barrier passed, loop done, epilogue done (100 MHz), shader cycles at loop done
#define PW_TL(tile, slot, val)                                                                      \
  do {                                                                                              \
    if (threadIdx.x == 0 && (tile) < 16) g_tlp[(blockIdx.x * 16 + (tile)) * 4 + (slot)] = (val);    \
  } while (0)
extern "C" int premvos_dbg_timeline_pw(void* dst, long bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tlp), bytes, 0, hipMemcpyDeviceToHost);
}
#else
#define PW_TL(tile, slot, val) do {} while (0)
#endif

namespace {

constexpr int BM = 128, BN = 128, KB = 16, RS = KB + 4, NT = 256;
constexpr int BUF = (BM + BN) * RS;                 // floats of one operand buffer (A rows, then B rows)
constexpr int WSC = 64 + 4;                         // staged row pitch of a wave's epilogue block (floats)
constexpr int LDS_BYTES = 2 * BUF * (int)sizeof(float) + 16;     // + the two queue words
static_assert(4 * 16 * WSC <= BUF, "the four waves' 16-row staging blocks live in the second operand buffer");

template <int ACT, bool HAS_RES>     // activation (NONE | RELU) and residual compiled in: no branches, no second copy of the epilogue's addresses
__global__ __launch_bounds__(NT, 3) void conv_pw_f32_kernel(const premvos_conv_desc p, const int mt_rows, unsigned* __restrict__ ctr) {
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  float(*lds)[BUF] = reinterpret_cast<float(*)[BUF]>(lds_dyn);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: so are wm0 / wn0 and what hangs on them)
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int M = p.n * p.ho * p.wo;
  const int n_nt = (p.cout + BN - 1) / BN;
  const int T = mt_rows * n_nt;
  const int KT = p.k_pad / KB;

  // ---- the tile queue.  Every XCD owns one contiguous run of the (m-tile major, n-tile minor) order (the column tiles that
  // re-read one A tile are taken one after the other and run on the same L2 at about the same time) behind a counter of its own;
  // a workgroup takes its XCD's next tile (XCC_ID: where it really runs) and, when that run is used up, the next XCD's.
  // Dynamic, because residency is not fair: with three resident workgroups per CU the arbiter's age order let the youngest one
  // finish a static list 10 % after the oldest (profiles/r05_timeline.txt).  One lane asks; the answer for tile i + 2 is requested
  // while tile i's epilogue runs and crosses to the other waves through LDS behind tile i + 1's barriers.
  const int xcd = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 7;
  const int q = T >> 3, r8 = T & 7;
  const int len_own = q + (xcd < r8 ? 1 : 0), c0_own = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
  auto ask = [&]() -> unsigned {                   // lane 0 of wave 0: take a number from this XCD's counter (not waited for here)
    return __hip_atomic_fetch_add(&ctr[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto resolve = [&](const unsigned ticket) -> int {   // ... and turn it into a tile; an exhausted run: the other XCDs' counters, in turn
    if (ticket < (unsigned)len_own) return c0_own + (int)ticket;
#pragma unroll 1
    for (int k = 1; k < 8; ++k) {
      const int x = (xcd + k) & 7;
      const int len = q + (x < r8 ? 1 : 0);
      if (__hip_atomic_load(&ctr[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)len) continue;
      const unsigned local = __hip_atomic_fetch_add(&ctr[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (local < (unsigned)len) return (x < r8 ? x * (q + 1) : r8 * (q + 1) + (x - r8) * q) + (int)local;
    }
    return -1;
  };
  int* tq = reinterpret_cast<int*>(lds_dyn + 2 * BUF);     // [2]: the tiles after the current one, by parity

  // ---- per-thread gather state of the tile whose operands are being requested
  const int j4 = (tid & 3) * 4;          // this thread's float4 column inside a 16-deep stage
  const int row_t = tid >> 2;            // its rows: row_t and row_t + 64 of the A tile and of the B tile
  // (32-bit element offsets from uniform bases: the requests take the scalar-base + vector-offset form, no 64-bit vector arithmetic)
  unsigned aoff[2], woff[2];             // A: pixel offset + j4 (0xffffffff = row past M); B: row * k_pad + j4 below the tile's first filter row
  bool wok[2];
  const float* wbase = p.wgt;            // first filter row of the tile (uniform)
  bool interior = false;
  auto setup = [&](const int tm, const int tn) {
    const int m0 = tm * BM, n0 = tn * BN;
    const int hw = p.ho * p.wo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = row_t + i * 64;
      const int m = m0 + row;
      const bool ok = m < M;
      const int mm = ok ? m : 0;
      const int n = mm / hw, rem = mm - n * hw;
      const int oy = rem / p.wo, ox = rem - oy * p.wo;
      aoff[i] = ok ? (unsigned)(((n * p.h + oy * p.sh) * p.w + ox * p.sw) * p.in_ps + j4) : 0xffffffffu;
      wok[i] = n0 + row < p.cout_pad;
      woff[i] = (unsigned)((wok[i] ? row : 0) * p.k_pad + j4);
    }
    wbase = p.wgt + (long)n0 * p.k_pad;
    interior = m0 + BM <= M && n0 + BN <= p.cout_pad && (KT - 1) * KB <= p.cin_pad;     // workgroup-uniform
  };
  float4 ra[2], rb[2];
  auto gload_plain = [&](const int kt) {           // interior stages of interior tiles: nothing to predicate
    const float* ab = p.in + kt * KB;
    const float* bb = wbase + kt * KB;
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const float4*>(ab + aoff[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const float4*>(bb + woff[i]);
  };
  auto gload = [&](const int kt) {
    const float* ab = p.in + kt * KB;
    const float* bb = wbase + kt * KB;
    const bool kok = kt * KB + j4 < p.cin_pad;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      ra[i] = (kok && aoff[i] != 0xffffffffu) ? *reinterpret_cast<const float4*>(ab + aoff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      rb[i] = (wok[i] && kt * KB + j4 < p.k_pad) ? *reinterpret_cast<const float4*>(bb + woff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto lstore = [&](const int buf) {
    float* a = &lds[buf][0];
    float* b = &lds[buf][BM * RS];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(a + (row_t + i * 64) * RS + j4) = ra[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(b + (row_t + i * 64) * RS + j4) = rb[i];
  };

  const int frag_off = (lane & 31) * RS + 4 * (lane >> 5);
  const int kreal = p.cin_pad;
  const int h_last = min(KB / 8, max(1, (kreal - (KT - 1) * KB + 7) / 8));     // 8-deep groups of the matrix's last stage that hold real k

#ifdef PW_STAGGER
  {   // developer experiment: the three workgroups of a CU start a fraction of a stage apart (slope = cycles / 64 per slot)
    const int slotj = blockIdx.x / (gridDim.x / 3);
    for (int w = 0; w < slotj * (int)p.slope; ++w) __builtin_amdgcn_s_sleep(1);
  }
#endif
  if (tid == 0) {
    const unsigned t0 = ask(), t1 = ask();
    tq[0] = resolve(t0);
    tq[1] = resolve(t1);
  }
  __syncthreads();
  int cur = tq[0], tl_par = 1;                   // tq[tl_par]: the tile after `cur`
  if (cur < 0) {
    if (tid == 0) {                              // nothing to do (more workgroups than tiles): still counted as finished
      __threadfence();
      if (__hip_atomic_fetch_add(&ctr[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1)
        for (int k = 0; k < 9; ++k) __hip_atomic_store(&ctr[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  int tm = cur / n_nt, tn = cur - tm * n_nt;
  setup(tm, tn);
  gload(0);
  [[maybe_unused]] int tl_i = 0;
  PW_TL(15, 3, ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11)));
  for (;;) {
    const int m0 = tm * BM, n0 = tn * BN;
    const bool tile_interior = interior;
    lstore(0);                           // (the second operand buffer may still hold other waves' epilogue blocks: not touched here)
    __syncthreads();
    PW_TL(tl_i, 0, wall_clock64());

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nvalid = min(2, max(0, (p.cout - (n0 + wn0) + 31) / 32));       // live 32-column blocks of this wave (wave-uniform)
    // The K loop of conv_igemm_f32.hip (fragment double buffering, plain requests on interior tiles, a scheduling fence behind
    // them, the first group of the next stage requested right behind the barrier); one straight-line copy per count of live blocks.
    auto k_loop = [&](auto nv_tag) {
      constexpr int NV = decltype(nv_tag)::value;
      constexpr int H = KB / 8, NVV = NV > 0 ? NV : 1;
      float4 af[2][2], bf[2][NVV];
      auto ldfrag = [&](const int set, const int buf, const int h) {
        const float* a = &lds[buf][wm0 * RS + frag_off] + h * 8;
        const float* b = &lds[buf][(BM + wn0) * RS + frag_off] + h * 8;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[set][mi] = *reinterpret_cast<const float4*>(a + mi * 32 * RS);
#pragma unroll
        for (int ni = 0; ni < NV; ++ni) bf[set][ni] = *reinterpret_cast<const float4*>(b + ni * 32 * RS);
      };
      auto mfma_rows = [&](const int set, const int mi0, const int mi1) {
#pragma unroll
        for (int mi = mi0; mi < mi1; ++mi)
#pragma unroll
          for (int ni = 0; ni < NV; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].x, bf[set][ni].x, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].y, bf[set][ni].y, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].z, bf[set][ni].z, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mi].w, bf[set][ni].w, acc[mi][ni], 0, 0, 0);
          }
      };
      ldfrag(0, 0, 0);
      auto stage = [&](const int kt, auto plain_tag) {
        const int buf = kt & 1;
        if constexpr (decltype(plain_tag)::value) {
          gload_plain(kt + 1);
          __builtin_amdgcn_sched_barrier(0);
        } else gload(kt + 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int h = 0; h + 1 < H; ++h) {
          ldfrag((h + 1) & 1, buf, h + 1);
          mfma_rows(h & 1, 0, 2);
        }
        mfma_rows((H - 1) & 1, 0, 1);
        __builtin_amdgcn_s_setprio(0);
        lstore(buf ^ 1);
        __syncthreads();
        ldfrag(0, buf ^ 1, 0);
        mfma_rows((H - 1) & 1, 1, 2);
      };
      int kt = 0;
      if (tile_interior)
        for (; kt + 1 < KT - 1; ++kt) stage(kt, std::true_type{});
      for (; kt + 1 < KT; ++kt) stage(kt, std::false_type{});
      const int buf = (KT - 1) & 1;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        if (h >= h_last) break;
        if (h + 1 < h_last) ldfrag((h + 1) & 1, buf, h + 1);
        mfma_rows(h & 1, 0, 2);
      }
      __syncthreads();                   // every wave is done reading the operand buffers
    };
    if (nvalid == 2) k_loop(std::integral_constant<int, 2>{});
    else if (nvalid == 1) k_loop(std::integral_constant<int, 1>{});
    else k_loop(std::integral_constant<int, 0>{});

    PW_TL(tl_i, 1, wall_clock64());
    PW_TL(tl_i, 3, __builtin_readcyclecounter());
    // ---- the next tile's first stage is requested now: its HBM round trip runs under this tile's epilogue
    const int nxt = tq[tl_par];                  // (written before this tile's barriers)
    const bool more = nxt >= 0;
    unsigned ticket = 0;
    if (tid == 0 && more) ticket = ask();        // the tile after the next: a number is taken now, looked at behind the epilogue
    const int tm2 = more ? nxt / n_nt : 0, tn2 = more ? nxt - (nxt / n_nt) * n_nt : 0;
    if (more) {
      setup(tm2, tn2);
      gload(0);
    }

    // ---- epilogue (bias, residual, activation), wave-private: 16 rows x 64 columns per pass through this wave's block of the
    // SECOND operand buffer (a wave's LDS instructions execute in order: no barrier between its writes and its reads)
    {
      float* stg = lds_dyn + BUF + wave * (16 * WSC);
      const int c4 = lane & 15, r0 = lane >> 4, col = n0 + wn0 + c4 * 4;      // 16-byte units: 16 per staged row, 4 rows per pass of the wave
      const bool col_ok = col < p.cout;
      const int colc = col_ok ? col : 0;
      const float4 bv = p.bias != nullptr ? premvos::ld4(p.bias + colc) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int mbase = m0 + wm0 + mi * 32 + hh * 16;
          float4 rv[4];
          if constexpr (HAS_RES) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              int m = mbase + r0 + i * 4;
              m = m < M ? m : M - 1;
              rv[i] = premvos::ld4(p.res + (long)m * p.res_ps + colc);
            }
          }
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              const int r = hh * 8 + rr;
              const int row = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
              stg[row * WSC + ni * 32 + (lane & 31)] = acc[mi][ni][r];
            }
          __builtin_amdgcn_wave_barrier();
          if (tl_i == 2) PW_TL(8 + mi * 2 + hh, 0, wall_clock64());
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = r0 + i * 4, m = mbase + row;
            float4 v = *reinterpret_cast<const float4*>(&stg[row * WSC + c4 * 4]);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if constexpr (HAS_RES) { v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w; }
            if constexpr (ACT == PREMVOS_ACT_RELU) {
              v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            }
            if (m < M && col_ok) *reinterpret_cast<float4*>(p.out + (long)m * p.out_ps + col) = v;
          }
          __builtin_amdgcn_wave_barrier();
          if (tl_i == 2) PW_TL(8 + mi * 2 + hh, 1, wall_clock64());
        }
    }
    PW_TL(tl_i, 2, wall_clock64());
    ++tl_i;
    if (!more) break;
    if (tid == 0) tq[tl_par ^ 1] = resolve(ticket);        // (read behind the next tile's barriers)
    tl_par ^= 1;
    tm = tm2;
    tn = tn2;
  }
  // the queue cleans up after itself: the last workgroup to leave zeroes the counters for the next launch that uses this slot
  if (tid == 0) {
    __threadfence();
    if (__hip_atomic_fetch_add(&ctr[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1)
      for (int k = 0; k < 9; ++k) __hip_atomic_store(&ctr[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Queue counters: a ring of slots, one per launch in flight (launches of different streams / host threads never share one; a
// replayed graph node keeps its slot, and replays of one graph are ordered).  Zero at load, and every launch leaves its slot zero.
constexpr int RING = 4096;
__device__ unsigned g_pw_ctr[RING][16];
unsigned* next_ctr_slot() {
  static unsigned* base = [] {
    void* ptr = nullptr;
    return hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_pw_ctr)) == hipSuccess ? static_cast<unsigned*>(ptr) : nullptr;
  }();
  static std::atomic<unsigned> next{0};
  return base == nullptr ? nullptr : base + (size_t)(next.fetch_add(1) % RING) * 16;
}

int cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}

}  // namespace

namespace premvos {

bool conv_pw_applicable(const premvos_conv_desc& d) {
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
  return d.precision == PREMVOS_PREC_F32 && (d.act == PREMVOS_ACT_NONE || d.act == PREMVOS_ACT_RELU) && d.kh == 1 && d.kw == 1 && d.pt == 0 && d.pl == 0 && d.out_mode == PREMVOS_OUT_NHWC &&
         (d.cout & 3) == 0 && (d.out_ps & 3) == 0 && a16(d.out) && (d.res == nullptr || ((d.res_ps & 3) == 0 && a16(d.res))) &&
         (d.bias == nullptr || a16(d.bias)) && d.k_pad >= 2 * KB && (long)d.n * d.h * d.w * d.in_ps < (1L << 31) &&
         (long)d.cout_pad * d.k_pad < (1L << 31) && (long)d.ho * d.sh <= d.h + d.sh - 1 && (long)d.wo * d.sw <= d.w + d.sw - 1;
}

// rows of 128-row tiles the persistent launch covers; the rest (d.tail_m_tiles of them, when d.tail_split_k > 1) are k-sliced
static int pw_tail_rows(const premvos_conv_desc& d) {
  const int mt = cdiv(d.n * d.ho * d.wo, BM);
  return (d.tail_m_tiles > 0 && d.tail_m_tiles < mt && igemm_tail128_splits(d, d.tail_split_k) > 1) ? d.tail_m_tiles : 0;
}

long conv_pw_workspace_bytes(const premvos_conv_desc& d) {
  const int tail = pw_tail_rows(d);
  return tail ? igemm_tail128_ws_bytes(d, tail, d.tail_split_k) : 0;
}

int conv_pw(const premvos_conv_desc& d, hipStream_t s) {
  const int M = d.n * d.ho * d.wo, mt = cdiv(M, BM), n_nt = cdiv(d.cout, BN);
  const int tail = pw_tail_rows(d);
  const int rows = mt - tail;
  const long T = (long)rows * n_nt;
  int G = 3 * cu_count();                              // three workgroups per CU (40 KB of LDS, <= 168 registers each)
  if (T < G) G = (int)T;
  unsigned* ctr = next_ctr_slot();
  if (ctr == nullptr) return fail(PREMVOS_ELAUNCH, "conv_pw: no queue counters");
  const bool relu = d.act == PREMVOS_ACT_RELU, res = d.res != nullptr;
  if (relu && res) hipLaunchKernelGGL((conv_pw_f32_kernel<PREMVOS_ACT_RELU, true>), dim3(G), dim3(NT), LDS_BYTES, s, d, rows, ctr);
  else if (relu) hipLaunchKernelGGL((conv_pw_f32_kernel<PREMVOS_ACT_RELU, false>), dim3(G), dim3(NT), LDS_BYTES, s, d, rows, ctr);
  else if (res) hipLaunchKernelGGL((conv_pw_f32_kernel<PREMVOS_ACT_NONE, true>), dim3(G), dim3(NT), LDS_BYTES, s, d, rows, ctr);
  else hipLaunchKernelGGL((conv_pw_f32_kernel<PREMVOS_ACT_NONE, false>), dim3(G), dim3(NT), LDS_BYTES, s, d, rows, ctr);
  int rc = check_launch("conv_pw_f32");
  if (rc || !tail) return rc;
  return launch_igemm_tail128(d, rows, d.tail_split_k, s);
}

}  // namespace premvos
