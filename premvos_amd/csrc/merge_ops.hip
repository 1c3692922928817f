// MergeTrack-side device helpers: the per-frame mask work the merge loop does on the CPU through
// OpenCV / pycocotools, as integer-exact HIP kernels on masks that are already in HBM.
//   * mask warp by optical flow      MergeTrack/merge_functions.py:209-217 (cv2.remap INTER_LINEAR on uint8 + "== 1")
//   * mask-vs-mask intersection/area merge_functions.py:38-45          (pycocotools iou on the RLEs of the same masks)
//   * run boundaries for COCO RLE    merge_functions.py:224-226         (pycocotools encode(np.asfortranarray(mask)))
// All three are byte/integer work bounded by HBM (a 480x854 mask is 410 KB): coalesced row-major reads, no GEMM shapes.
#include "common.h"

namespace {

// cv2.remap(img_u8, map_f32x2, None, INTER_LINEAR), border constant 0, restated from OpenCV's fixed-point path:
// map -> 1/32-pixel fixed point with round-half-even (cvRound), integer cell (arithmetic >> 5, saturated to int16) and
// 5-bit fractions; tap weights = (32-ay|ay)*(32-ax|ax)*32 (sum 32768); value = (sum w*v + 2^14) >> 15.
// The map is built as merge_functions.py:211-214 does: -flow (float32), then "+= arange" which numpy evaluates in
// float64 and stores back as float32.
__global__ __launch_bounds__(256) void mask_warp_kernel(const uint8_t* __restrict__ masks, int n, int h, int w,
                                                        const float* __restrict__ flow, uint8_t* __restrict__ out,
                                                        int binarize) {
  const long hw = (long)h * w;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < hw; p += (long)gridDim.x * 256) {
    const int y = (int)(p / w), x = (int)(p - (long)y * w);
    const float mx = (float)((double)(-flow[2 * p]) + (double)x);
    const float my = (float)((double)(-flow[2 * p + 1]) + (double)y);
    const int sx = __float2int_rn(mx * 32.f), sy = __float2int_rn(my * 32.f);
    int ix = sx >> 5, iy = sy >> 5;
    ix = ix < -32768 ? -32768 : (ix > 32767 ? 32767 : ix);
    iy = iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy);
    const int ax = sx & 31, ay = sy & 31;
    const int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
    const bool x0 = (unsigned)ix < (unsigned)w, x1 = (unsigned)(ix + 1) < (unsigned)w;
    const bool y0 = (unsigned)iy < (unsigned)h, y1 = (unsigned)(iy + 1) < (unsigned)h;
    const long o00 = (long)iy * w + ix;
    for (int i = 0; i < n; ++i) {
      const uint8_t* m = masks + (long)i * hw;
      int acc = 1 << 14;
      if (y0 && x0) acc += w00 * m[o00];
      if (y0 && x1) acc += w01 * m[o00 + 1];
      if (y1 && x0) acc += w10 * m[o00 + w];
      if (y1 && x1) acc += w11 * m[o00 + w + 1];
      int v = acc >> 15;
      v = v > 255 ? 255 : v;
      out[(long)i * hw + p] = binarize ? (uint8_t)(v == 1) : (uint8_t)v;
    }
  }
}

// counts[ib][ia] += |a_ia & b_ib| over a pixel chunk; area_a / area_b likewise (nonzero = foreground).  Integer atomics:
// the result does not depend on the order of arrival.
__global__ __launch_bounds__(256) void mask_overlap_kernel(const uint8_t* __restrict__ a, int na,
                                                           const uint8_t* __restrict__ b, int nb, long hw,
                                                           unsigned long long* __restrict__ inter,
                                                           unsigned long long* __restrict__ area_a,
                                                           unsigned long long* __restrict__ area_b) {
  const int ia = blockIdx.y, ib = blockIdx.z;
  const uint8_t* pa = a + (long)ia * hw;
  const uint8_t* pb = b + (long)ib * hw;
  unsigned ci = 0, ca = 0, cb = 0;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < hw; p += (long)gridDim.x * 256) {
    const bool va = pa[p] != 0, vb = pb[p] != 0;
    ci += va && vb;
    ca += va;
    cb += vb;
  }
  __shared__ unsigned red[3][256];
  red[0][threadIdx.x] = ci; red[1][threadIdx.x] = ca; red[2][threadIdx.x] = cb;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
      red[2][threadIdx.x] += red[2][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(&inter[(long)ib * na + ia], (unsigned long long)red[0][0]);
    if (ib == 0) atomicAdd(&area_a[ia], (unsigned long long)red[1][0]);
    if (ia == 0) atomicAdd(&area_b[ib], (unsigned long long)red[2][0]);
  }
}

// Column-major run boundaries: position q = x*h + y is a boundary when value(q) != value(q-1), value(-1) = 0.
// Pass 1 counts boundaries per chunk of RLE_CHUNK positions, pass 2 writes them in ascending order.
constexpr int RLE_CHUNK = 2048;   // 8 positions per thread

__device__ inline int rle_val(const uint8_t* m, int h, int rs, long q) {       // rs: bytes between rows (>= the mask's width)
  const int x = (int)(q / h), y = (int)(q - (long)x * h);
  return m[(long)y * rs + x] != 0;
}

__global__ __launch_bounds__(256) void rle_count_kernel(const uint8_t* __restrict__ masks, int h, int w, long ms, int rs, int nchunks,
                                                        int* __restrict__ chunk_counts) {
  const long hw = (long)h * w;
  const uint8_t* m = masks + (long)blockIdx.y * ms;
  const long q0 = (long)blockIdx.x * RLE_CHUNK + (long)threadIdx.x * 8;
  int c = 0;
  if (q0 < hw) {
    int prev = q0 == 0 ? 0 : rle_val(m, h, rs, q0 - 1);
    for (int k = 0; k < 8 && q0 + k < hw; ++k) {
      const int v = rle_val(m, h, rs, q0 + k);
      c += v != prev;
      prev = v;
    }
  }
  __shared__ int red[256];
  red[threadIdx.x] = c;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) chunk_counts[(long)blockIdx.y * nchunks + blockIdx.x] = red[0];
}

// POOLED = false: mask i writes positions[i * cap ...] and nruns[i] (the per-mask form).
// POOLED = true : mask i writes pool[offsets[i] ...] (offsets = exclusive prefix of the masks' totals, rle_offsets_kernel);
//                 cap = capacity of the whole pool.
template <bool POOLED>
__global__ __launch_bounds__(256) void rle_write_kernel(const uint8_t* __restrict__ masks, int h, int w, long ms, int rs, int nchunks,
                                                        const int* __restrict__ chunk_counts, int cap,
                                                        int* __restrict__ positions, int* __restrict__ nruns) {
  const long hw = (long)h * w;
  const uint8_t* m = masks + (long)blockIdx.y * ms;
  const int* cc = chunk_counts + (long)blockIdx.y * nchunks;
  __shared__ int scan[256];
  __shared__ int base;
  if (threadIdx.x == 0) {
    int b = 0;
    for (int i = 0; i < (int)blockIdx.x; ++i) b += cc[i];
    if (!POOLED && blockIdx.x == (unsigned)nchunks - 1) nruns[blockIdx.y] = b + cc[nchunks - 1];
    base = POOLED ? b + nruns[blockIdx.y] : b;        // (pooled: `nruns` holds the offsets)
  }
  const long q0 = (long)blockIdx.x * RLE_CHUNK + (long)threadIdx.x * 8;
  int flags = 0, c = 0;
  if (q0 < hw) {
    int prev = q0 == 0 ? 0 : rle_val(m, h, rs, q0 - 1);
    for (int k = 0; k < 8 && q0 + k < hw; ++k) {
      const int v = rle_val(m, h, rs, q0 + k);
      if (v != prev) { flags |= 1 << k; ++c; }
      prev = v;
    }
  }
  scan[threadIdx.x] = c;
  __syncthreads();
  for (int s = 1; s < 256; s <<= 1) {      // inclusive Hillis-Steele scan
    const int t = (int)threadIdx.x >= s ? scan[threadIdx.x - s] : 0;
    __syncthreads();
    scan[threadIdx.x] += t;
    __syncthreads();
  }
  int o = base + scan[threadIdx.x] - c;
  int* dst = POOLED ? positions : positions + (long)blockIdx.y * cap;
  for (int k = 0; k < 8; ++k)
    if (flags & (1 << k)) {
      if (o < cap) dst[o] = (int)(q0 + k);
      ++o;
    }
}

// offsets[0] = 0, offsets[i + 1] = offsets[i] + (boundaries of mask i): one workgroup; n is a few hundred masks of a chunk of frames
__global__ __launch_bounds__(256) void rle_offsets_kernel(const int* __restrict__ chunk_counts, int n, int nchunks, int* __restrict__ offsets) {
  for (int i = threadIdx.x; i < n; i += 256) {
    const int* cc = chunk_counts + (long)i * nchunks;
    int t = 0;
    for (int k = 0; k < nchunks; ++k) t += cc[k];
    offsets[i + 1] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    offsets[0] = 0;
    for (int i = 0; i < n; ++i) {
      run += offsets[i + 1];
      offsets[i + 1] = run;
    }
  }
}

// 8 mask bytes -> 1 byte (bit k = mask[8i + k] != 0) and back: the masks a rank hands to the merge rank travel bit-packed
// (480x854 x 20 boxes: 8.2 MB -> 1.0 MB per frame).  One thread per 16 output bytes / 16 input bytes.
__global__ __launch_bounds__(256) void mask_pack_bits_kernel(const uint8_t* __restrict__ m, long n_bits, uint8_t* __restrict__ out) {
  const long nbytes = (n_bits + 7) / 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i * 4 < nbytes; i += (long)gridDim.x * 256) {
    uint32_t word = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const long o = i * 4 + b, base = o * 8;
      if (o >= nbytes) break;
      uint32_t byte = 0;
      if (base + 8 <= n_bits && (reinterpret_cast<uintptr_t>(m + base) & 7u) == 0) {
        const uint64_t q = *reinterpret_cast<const uint64_t*>(m + base);
#pragma unroll
        for (int k = 0; k < 8; ++k) byte |= (uint32_t)(((q >> (8 * k)) & 0xffu) != 0) << k;
      } else {
        for (int k = 0; k < 8 && base + k < n_bits; ++k) byte |= (uint32_t)(m[base + k] != 0) << k;
      }
      word |= byte << (8 * b);
    }
    if (i * 4 + 4 <= nbytes && (reinterpret_cast<uintptr_t>(out) & 3u) == 0) {
      reinterpret_cast<uint32_t*>(out)[i] = word;
    } else {
      for (int b = 0; b < 4 && i * 4 + b < nbytes; ++b) out[i * 4 + b] = (uint8_t)(word >> (8 * b));
    }
  }
}

__global__ __launch_bounds__(256) void mask_unpack_bits_kernel(const uint8_t* __restrict__ bits, long n_bits, uint8_t* __restrict__ m) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_bits; i += (long)gridDim.x * 256)
    m[i] = (bits[i >> 3] >> (i & 7)) & 1u;
}

}  // namespace

extern "C" int premvos_mask_pack_bits_u8(const uint8_t* masks, int64_t n, uint8_t* bits, void* stream) {
  PV_REQUIRE(masks && bits && n > 0, "mask_pack_bits: bad arguments");
  long g = ((n + 7) / 8 + 4 * 256 - 1) / (4 * 256);
  if (g > 65535) g = 65535;
  hipLaunchKernelGGL(mask_pack_bits_kernel, dim3((int)g), dim3(256), 0, static_cast<hipStream_t>(stream), masks, (long)n, bits);
  return premvos::check_launch("mask_pack_bits");
}

extern "C" int premvos_mask_unpack_bits_u8(const uint8_t* bits, int64_t n, uint8_t* masks, void* stream) {
  PV_REQUIRE(masks && bits && n > 0, "mask_unpack_bits: bad arguments");
  long g = (n + 255) / 256;
  if (g > 65535) g = 65535;
  hipLaunchKernelGGL(mask_unpack_bits_kernel, dim3((int)g), dim3(256), 0, static_cast<hipStream_t>(stream), bits, (long)n, masks);
  return premvos::check_launch("mask_unpack_bits");
}

extern "C" int premvos_mask_warp_u8(const uint8_t* masks, int32_t n, int32_t h, int32_t w, const float* flow,
                                    uint8_t* out, int32_t binarize, void* stream) {
  PV_REQUIRE(masks && flow && out, "mask_warp: null pointer");
  PV_REQUIRE(n > 0 && h > 0 && w > 0, "mask_warp: bad dims");
  PV_REQUIRE(masks != out, "mask_warp: in-place warp is not supported");
  const long hw = (long)h * w;
  int g = (int)((hw + 255) / 256);
  hipLaunchKernelGGL(mask_warp_kernel, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream), masks, n, h, w, flow, out,
                     binarize);
  return premvos::check_launch("mask_warp");
}

extern "C" int premvos_mask_overlap_u8(const uint8_t* a, int32_t na, const uint8_t* b, int32_t nb, int64_t hw,
                                       int64_t* inter, int64_t* area_a, int64_t* area_b, void* stream) {
  PV_REQUIRE(a && b && inter && area_a && area_b, "mask_overlap: null pointer");
  PV_REQUIRE(na > 0 && nb > 0 && hw > 0 && na <= 65535 && nb <= 65535, "mask_overlap: bad dims");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(inter, 0, sizeof(int64_t) * (size_t)na * nb, s) != hipSuccess ||
      hipMemsetAsync(area_a, 0, sizeof(int64_t) * (size_t)na, s) != hipSuccess ||
      hipMemsetAsync(area_b, 0, sizeof(int64_t) * (size_t)nb, s) != hipSuccess)
    return premvos::fail(PREMVOS_ELAUNCH, "mask_overlap: memset failed");
  int gx = (int)((hw + 256L * 16 - 1) / (256L * 16));
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(mask_overlap_kernel, dim3(gx, na, nb), dim3(256), 0, s, a, na, b, nb, (long)hw,
                     reinterpret_cast<unsigned long long*>(inter), reinterpret_cast<unsigned long long*>(area_a),
                     reinterpret_cast<unsigned long long*>(area_b));
  return premvos::check_launch("mask_overlap");
}

extern "C" int64_t premvos_rle_workspace_bytes(int32_t n, int32_t h, int32_t w) {
  const long nchunks = ((long)h * w + RLE_CHUNK - 1) / RLE_CHUNK;
  return (int64_t)n * nchunks * (int64_t)sizeof(int) + 256;
}

extern "C" int premvos_rle_boundaries_u8(const uint8_t* masks, int32_t n, int32_t h, int32_t w, int32_t* positions,
                                         int32_t capacity, int32_t* nruns, void* workspace, void* stream) {
  PV_REQUIRE(masks && positions && nruns && workspace, "rle_boundaries: null pointer");
  PV_REQUIRE(n > 0 && n <= 65535 && h > 0 && w > 0 && capacity > 0, "rle_boundaries: bad dims");
  PV_REQUIRE((long)h * w < (1L << 31), "rle_boundaries: mask too large");
  const int nchunks = (int)(((long)h * w + RLE_CHUNK - 1) / RLE_CHUNK);
  int* cc = static_cast<int*>(workspace);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(rle_count_kernel, dim3(nchunks, n), dim3(256), 0, s, masks, h, w, (long)h * w, w, nchunks, cc);
  int rc = premvos::check_launch("rle_count");
  if (rc) return rc;
  hipLaunchKernelGGL(rle_write_kernel<false>, dim3(nchunks, n), dim3(256), 0, s, masks, h, w, (long)h * w, w, nchunks, cc, capacity,
                     positions, nruns);
  return premvos::check_launch("rle_write");
}

extern "C" int premvos_rle_boundaries_pooled_u8(const uint8_t* masks, int32_t n, int32_t h, int32_t w, int64_t mask_stride,
                                                int32_t row_stride, int32_t* pool, int32_t pool_capacity, int32_t* offsets,
                                                void* workspace, void* stream) {
  PV_REQUIRE(masks && pool && offsets && workspace, "rle_boundaries_pooled: null pointer");
  PV_REQUIRE(n > 0 && n <= 65535 && h > 0 && w > 0 && pool_capacity > 0, "rle_boundaries_pooled: bad dims");
  PV_REQUIRE(row_stride >= w && mask_stride >= (int64_t)(h - 1) * row_stride + w, "rle_boundaries_pooled: strides smaller than the mask");
  PV_REQUIRE((long)h * w < (1L << 31), "rle_boundaries_pooled: mask too large");
  const int nchunks = (int)(((long)h * w + RLE_CHUNK - 1) / RLE_CHUNK);
  int* cc = static_cast<int*>(workspace);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(rle_count_kernel, dim3(nchunks, n), dim3(256), 0, s, masks, h, w, (long)mask_stride, row_stride, nchunks, cc);
  int rc = premvos::check_launch("rle_count");
  if (rc) return rc;
  hipLaunchKernelGGL(rle_offsets_kernel, dim3(1), dim3(256), 0, s, cc, n, nchunks, offsets);
  rc = premvos::check_launch("rle_offsets");
  if (rc) return rc;
  hipLaunchKernelGGL(rle_write_kernel<true>, dim3(nchunks, n), dim3(256), 0, s, masks, h, w, (long)mask_stride, row_stride, nchunks, cc,
                     pool_capacity, pool, offsets);
  return premvos::check_launch("rle_write");
}

// Host utility (no GPU): COCO maskApi rleToString -- run lengths (delta-coded against the run two back from the 4th on) in
// 5-bit groups, LSB first, 0x20 = continuation, +48.  Returns the string length (without terminator), or -1 if `cap` is
// too small.  The Python twin (premvos_amd/rle.py:counts_to_string) is the reference for tests; this one is ~100x faster.
extern "C" int64_t premvos_rle_counts_to_string_host(const int64_t* counts, int64_t n, char* out, int64_t cap) {
  int64_t p = 0;
  for (int64_t i = 0; i < n; ++i) {
    long long x = counts[i];
    if (i > 2) x -= counts[i - 2];
    bool more = true;
    while (more) {
      char c = (char)(x & 0x1f);
      x >>= 5;                                  // arithmetic shift
      more = (c & 0x10) ? x != -1 : x != 0;
      if (more) c |= 0x20;
      if (p >= cap) return -1;
      out[p++] = (char)(c + 48);
    }
  }
  return p;
}

// Host utility (no GPU): the "counts" strings of n masks from their pooled run boundaries (premvos_rle_boundaries_pooled_u8 after a
// copy to the host): mask i's runs = successive differences of [0, pool[offsets[i]] ... pool[offsets[i+1] - 1], hw].  The strings are
// written back to back into `out` (no terminators), str_offsets[i] .. str_offsets[i + 1] delimit string i.  Returns the total
// length, or -1 if `cap` is too small.  One call per chunk of frames, the interpreter lock released for its duration.
extern "C" int64_t premvos_rle_strings_host(const int32_t* pool, const int32_t* offsets, int32_t n, int64_t hw, char* out, int64_t cap,
                                            int64_t* str_offsets) {
  int64_t p = 0;
  for (int32_t i = 0; i < n; ++i) {
    str_offsets[i] = p;
    const int32_t* e = pool + offsets[i];
    const int64_t m = (int64_t)offsets[i + 1] - offsets[i];
    long long prev_edge = 0, c1 = 0, c2 = 0;      // c1 / c2: the run one / two back
    for (int64_t k = 0; k <= m; ++k) {
      const long long edge = k < m ? (long long)e[k] : (long long)hw;
      const long long cnt = edge - prev_edge;
      prev_edge = edge;
      long long x = k > 2 ? cnt - c2 : cnt;
      c2 = c1;
      c1 = cnt;
      bool more = true;
      while (more) {
        char c = (char)(x & 0x1f);
        x >>= 5;
        more = (c & 0x10) ? x != -1 : x != 0;
        if (more) c |= 0x20;
        if (p >= cap) return -1;
        out[p++] = (char)(c + 48);
      }
    }
  }
  str_offsets[n] = p;
  return p;
}
