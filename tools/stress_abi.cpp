// Multi-threaded launch stress test of the C-ABI: N host threads, each with its own HIP stream and
// buffers, issue eager launches through libpremvos_hip.so and -- every G iterations -- capture a launch list into a HIP graph
// (thread-local capture mode) and replay it, while the other threads keep launching.  No Python, no torch: this isolates the
// library (and the HIP runtime under it) from the host code of the drivers.  Every thread checks its outputs against the first
// iteration's (the inputs never change, every kernel is deterministic); a watchdog reports the thread that stopped making
// progress instead of hanging.
//
//   hipcc -O2 -std=c++17 tools/stress_abi.cpp -Iinclude -Lpremvos_amd/csrc -lpremvos_hip -Wl,-rpath,$PWD/premvos_amd/csrc -o /tmp/stress_abi
//   /tmp/stress_abi <threads> <launches per thread> <graph every> [<mode: 0 eager + thread-local captures (the supported pattern),
//        1 eager only, 2 global-mode captures (expected to disturb the other threads), 3 = mode 0 + synchronous hipMemcpy (the hazard)>]
//
// Exit code 0 = all launches done and verified, 2 = wrong result, 3 = a thread made no progress for 60 s, 4 = API error.
#include <hip/hip_runtime.h>

#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "premvos_hip.h"

#define HIPCHECK(x)                                                                              \
  do {                                                                                           \
    hipError_t e_ = (x);                                                                         \
    if (e_ != hipSuccess) {                                                                      \
      fprintf(stderr, "thread %d: %s -> %s\n", tid, #x, hipGetErrorString(e_));                  \
      failed.store(4);                                                                           \
      return;                                                                                    \
    }                                                                                            \
  } while (0)
#define PVCHECK(x)                                                                               \
  do {                                                                                           \
    int r_ = (x);                                                                                \
    if (r_ != 0) {                                                                               \
      fprintf(stderr, "thread %d: %s -> %d (%s)\n", tid, #x, r_, premvos_last_error());          \
      failed.store(4);                                                                           \
      return;                                                                                    \
    }                                                                                            \
  } while (0)

static std::atomic<int> failed{0}, ready{0};
static int nthreads = 0;
static std::vector<std::atomic<long>*> progress;

static float frand(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
}

static void worker(int tid, long launches, int graph_every, int mode) {
  HIPCHECK(hipSetDevice(0));
  hipStream_t st;
  HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  // a small 3x3 conv 64 -> 64 on a 2 x 24 x 40 map, a 1x1 conv 64 -> 96 behind it, a warp of the result, and a mask pack
  const int n = 2, h = 24, w = 40, c = 64, c2 = 96;
  const long px = (long)n * h * w;
  std::vector<float> hx(px * c), hw1(64 * 576), hw2(96 * 64), hb(96), hflow(px * 4);
  unsigned seed = 1234u + 77u * tid;
  for (auto& v : hx) v = frand(seed);
  for (auto& v : hw1) v = frand(seed) * 0.1f;
  for (auto& v : hw2) v = frand(seed) * 0.2f;
  for (auto& v : hb) v = frand(seed);
  for (long i = 0; i < px; ++i) {
    hflow[4 * i] = 3.f * frand(seed);
    hflow[4 * i + 1] = 3.f * frand(seed);
    hflow[4 * i + 2] = hflow[4 * i + 3] = 0.f;
  }
  float *x, *w1, *w2, *b, *y1, *y2, *flow, *y3, *ws;
  uint8_t *mask, *bits;
  HIPCHECK(hipMalloc(&x, hx.size() * 4));
  HIPCHECK(hipMalloc(&w1, hw1.size() * 4));
  HIPCHECK(hipMalloc(&w2, hw2.size() * 4));
  HIPCHECK(hipMalloc(&b, 128 * 4));
  HIPCHECK(hipMalloc(&y1, px * c * 4));
  HIPCHECK(hipMalloc(&y2, px * c2 * 4));
  HIPCHECK(hipMalloc(&y3, px * c2 * 4));
  HIPCHECK(hipMalloc(&flow, px * 4 * 4));
  HIPCHECK(hipMalloc(&ws, 8 << 20));
  HIPCHECK(hipMalloc(&mask, px * c2));
  HIPCHECK(hipMalloc(&bits, px * c2 / 8 + 16));
  HIPCHECK(hipMemset(b, 0, 128 * 4));
  HIPCHECK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(w1, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(flow, hflow.data(), hflow.size() * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemset(mask, 0, px * c2));

  premvos_conv_desc d1;
  memset(&d1, 0, sizeof(d1));
  d1.in = x; d1.wgt = w1; d1.bias = b; d1.out = y1;
  d1.n = n; d1.h = h; d1.w = w; d1.cin = c; d1.in_ps = c; d1.ho = h; d1.wo = w; d1.cout = c; d1.out_ps = c;
  d1.kh = d1.kw = 3; d1.sh = d1.sw = d1.dh = d1.dw = 1; d1.pt = d1.pl = 1;
  d1.cin_pad = c; d1.k_pad = 576; d1.cout_pad = 64; d1.act = PREMVOS_ACT_RELU; d1.slope = 0.1f;
  d1.out_mode = PREMVOS_OUT_NHWC; d1.tile_hint = (64 << 16) | 64; d1.split_k = (tid & 1) ? 2 : -1;   // odd threads: k-slices + reduce
  d1.workspace = ws; d1.workspace_bytes = 8 << 20; d1.precision = PREMVOS_PREC_F32; d1.stage_k = 16;
  premvos_conv_desc d2 = d1;
  d2.in = y1; d2.wgt = w2; d2.out = y2; d2.cout = c2; d2.out_ps = c2; d2.kh = d2.kw = 1; d2.pt = d2.pl = 0; d2.k_pad = 64;
  d2.cout_pad = 96; d2.tile_hint = (128 << 16) | 96; d2.split_k = -1; d2.act = PREMVOS_ACT_LEAKY;

  auto sequence = [&]() -> int {                       // 4 launches (5 with the k-slice reduce)
    int r = premvos_conv2d_f32(&d1, st);
    if (r) return r;
    if ((r = premvos_conv2d_f32(&d2, st))) return r;
    if ((r = premvos_warp_fwd_f32(y2, c2, flow, 4, 0.5f, y3, c2, n, h, w, c2, st))) return r;
    return premvos_mask_pack_bits_u8(reinterpret_cast<const uint8_t*>(y3), px * c2, bits, st);
  };
  std::vector<float> ref(px * c2), got(px * c2);
  PVCHECK(sequence());
  HIPCHECK(hipStreamSynchronize(st));
  HIPCHECK(hipMemcpy(ref.data(), y3, ref.size() * 4, hipMemcpyDeviceToHost));
  // Results come back through THIS thread's stream (hipMemcpyAsync + hipStreamSynchronize): calls that are legal on any thread
  // while another thread captures.  mode 3 uses the synchronous hipMemcpy instead -- a legacy-default-stream operation, which
  // the runtime refuses ("operation not permitted when stream is capturing") and which invalidates the OTHER thread's capture
  // even in thread-local capture mode: the hazard behind the "no capture while other host threads use the GPU" rule of the
  // drivers (every .cpu() of a torch tensor is such a call).  1 = ok, 0 = wrong bytes, -1 = API error.
  float* pinned = nullptr;
  HIPCHECK(hipHostMalloc(&pinned, got.size() * 4));
  auto verify = [&]() -> int {
    hipError_t e;
    if (mode == 3) {
      if ((e = hipStreamSynchronize(st)) != hipSuccess || (e = hipMemcpy(pinned, y3, got.size() * 4, hipMemcpyDeviceToHost)) != hipSuccess) {
        fprintf(stderr, "thread %d: synchronous copy refused: %s\n", tid, hipGetErrorString(e));
        return -1;
      }
    } else {
      if ((e = hipMemcpyAsync(pinned, y3, got.size() * 4, hipMemcpyDeviceToHost, st)) != hipSuccess ||
          (e = hipStreamSynchronize(st)) != hipSuccess) {
        fprintf(stderr, "thread %d: async copy / sync failed: %s\n", tid, hipGetErrorString(e));
        return -1;
      }
    }
    return memcmp(pinned, ref.data(), got.size() * 4) == 0 ? 1 : 0;
  };
  // every thread finishes its set-up (hipMalloc, synchronous copies: legacy-stream operations) before anyone captures
  ready.fetch_add(1);
  while (ready.load() < nthreads && !failed.load()) std::this_thread::yield();
  long done = 0, it = 0;
  while (done < launches && !failed.load()) {
    ++it;
    if (mode != 1 && graph_every > 0 && it % graph_every == 0) {
      hipGraph_t g;
      hipGraphExec_t ge;
      HIPCHECK(hipStreamBeginCapture(st, mode == 2 ? hipStreamCaptureModeGlobal : hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < 4; ++k) PVCHECK(sequence());
      HIPCHECK(hipStreamEndCapture(st, &g));
      HIPCHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int k = 0; k < 8; ++k) HIPCHECK(hipGraphLaunch(ge, st));
      HIPCHECK(hipStreamSynchronize(st));
      HIPCHECK(hipGraphExecDestroy(ge));
      HIPCHECK(hipGraphDestroy(g));
      done += 8 * 16;
    } else {
      PVCHECK(sequence());
      done += 4;
    }
    if (it % 500 == 0) {
      const int v = verify();
      if (v != 1) {
        if (v == 0) fprintf(stderr, "thread %d: WRONG RESULT after %ld launches\n", tid, done);
        failed.store(v == 0 ? 2 : 4);
        return;
      }
    }
    progress[tid]->store(done);
  }
  if (!failed.load()) {
    const int v = verify();
    if (v == 0) fprintf(stderr, "thread %d: WRONG final result\n", tid);
    if (v != 1) failed.store(v == 0 ? 2 : 4);
  }
  progress[tid]->store(launches + 1);
}

int main(int argc, char** argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : 6;
  const long launches = argc > 2 ? atol(argv[2]) : 200000;
  const int graph_every = argc > 3 ? atoi(argv[3]) : 50;
  const int mode = argc > 4 ? atoi(argv[4]) : 0;
  nthreads = threads;
  for (int i = 0; i < threads; ++i) progress.push_back(new std::atomic<long>(0));
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int i = 0; i < threads; ++i) th.emplace_back(worker, i, launches, graph_every, mode);
  std::vector<long> last(threads, -1);
  int stale = 0;
  for (;;) {
    std::this_thread::sleep_for(std::chrono::seconds(2));
    bool all = true, moved = false;
    for (int i = 0; i < threads; ++i) {
      const long p = progress[i]->load();
      all = all && p > launches;
      moved = moved || p != last[i];
      last[i] = p;
    }
    if (all || failed.load()) break;
    stale = moved ? 0 : stale + 1;
    if (stale >= 30) {
      fprintf(stderr, "NO PROGRESS for 60 s:");
      for (int i = 0; i < threads; ++i) fprintf(stderr, " t%d=%ld", i, last[i]);
      fprintf(stderr, "\n");
      _exit(3);
    }
  }
  for (auto& t : th) t.join();
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  long total = 0;
  for (int i = 0; i < threads; ++i) total += std::min(progress[i]->load(), launches);
  printf("{\"threads\": %d, \"launches\": %ld, \"graph_every\": %d, \"mode\": %d, \"seconds\": %.1f, \"launches_per_s\": %.0f, \"rc\": %d}\n",
         threads, total, graph_every, mode, s, total / s, failed.load());
  return failed.load();
}
