// Non-GEMM kernels of the proposal_net forward (gfx950): image pre-processing, max-pool, RPN proposal
// generation (top-k select + sort + decode + clip + filter + greedy NMS, one workgroup per image),
// RoIAlign (TF crop_and_resize semantics + 2x2 average fused), global average pool and the
// Fast R-CNN inference tail.  Reference call sites: see include/premvos_hip.h.
//
// fp contraction is OFF in this file: the box arithmetic must round like the TF graph (separate
// mul/add) so that thresholded decisions (IoU > t, w > 0, p > 0.5) pick the same indices.
#include "common.h"
#include "resize_cv.h"

#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------------------------------
// eval.py:76-77 CustomResize (cv2 INTER_LINEAR on uint8 BGR) + basemodel.py:12-26 normalisation,
// written as NHWC fp32 with a zero 4th channel.
__global__ __launch_bounds__(256) void proposal_preprocess_kernel(const uint8_t* __restrict__ img, int batch, int h,
                                                                  int w, float* __restrict__ out, int nh, int nw,
                                                                  int src_is_rgb) {
  const long total = (long)batch * nh * nw;
  const double sx = 1.0 / ((double)nw / (double)w), sy = 1.0 / ((double)nh / (double)h);
  const float mean[3] = {0.406f, 0.456f, 0.485f};   // BGR order (basemodel.py:20-22)
  const float stdv[3] = {0.225f, 0.224f, 0.229f};
  const float inv255 = (float)(1.0 / 255);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = idx % nw, y = (idx / nw) % nh;
    const uint8_t* im = img + (idx / ((long)nw * nh)) * h * w * 3;
    int x0, x1, y0, y1;
    short a0, a1, b0, b1;
    premvos::cv_lin_coef(x, sx, w, &x0, &x1, &a0, &a1);
    premvos::cv_lin_coef(y, sy, h, &y0, &y1, &b0, &b1);
    float px[4];
    for (int ch = 0; ch < 3; ++ch) {   // ch indexes the BGR order the net sees
      const int sc = src_is_rgb ? 2 - ch : ch;
      const int v = (nh == h && nw == w) ? im[((long)y * w + x) * 3 + sc]
                                         : premvos::cv_resize_u8_px(im, w, 3, sc, x0, x1, y0, y1, a0, a1, b0, b1);
      px[ch] = ((float)v * inv255 - mean[ch]) / stdv[ch];
    }
    px[3] = 0.f;
    *reinterpret_cast<float4*>(out + idx * 4) = make_float4(px[0], px[1], px[2], px[3]);
  }
}

// ------------------------------------------------------------------------------------------
// Max pooling, NHWC, explicit (top,left) padding with a pad VALUE (the reference pads with zeros via
// tf.pad before a VALID pool, basemodel.py:81-82).  One thread per (pixel, 4 channels).
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ in, int in_ps, int n, int h, int w,
                                                      int c4, float* __restrict__ out, int out_ps, int ho, int wo,
                                                      int k, int stride, int pt, int pl, float padv) {
  const long total = (long)n * ho * wo * c4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = idx % c4;
    const long pix = idx / c4;
    const int ox = pix % wo, oy = (pix / wo) % ho, b = pix / ((long)wo * ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int j = 0; j < k; ++j)
      for (int i = 0; i < k; ++i) {
        const int iy = oy * stride - pt + j, ix = ox * stride - pl + i;
        float4 v = make_float4(padv, padv, padv, padv);
        if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w)
          v = *reinterpret_cast<const float4*>(in + (((long)b * h + iy) * w + ix) * in_ps + cg * 4);
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
        m.z = fmaxf(m.z, v.z);
        m.w = fmaxf(m.w, v.w);
      }
    *reinterpret_cast<float4*>(out + pix * out_ps + cg * 4) = m;
  }
}

// ------------------------------------------------------------------------------------------
// shared box helpers (TF non_max_suppression_op IoU: corners via min/max, 0 if an area <= 0)
__device__ inline float iou_tf(const float4 a, const float4 b) {   // (x1,y1,x2,y2); symmetric in x/y
  const float ya0 = fminf(a.y, a.w), ya1 = fmaxf(a.y, a.w), xa0 = fminf(a.x, a.z), xa1 = fmaxf(a.x, a.z);
  const float yb0 = fminf(b.y, b.w), yb1 = fmaxf(b.y, b.w), xb0 = fminf(b.x, b.z), xb1 = fmaxf(b.x, b.z);
  const float area_a = (ya1 - ya0) * (xa1 - xa0), area_b = (yb1 - yb0) * (xb1 - xb0);
  if (area_a <= 0.f || area_b <= 0.f) return 0.f;
  const float ih = fmaxf(fminf(ya1, yb1) - fmaxf(ya0, yb0), 0.f);
  const float iw = fmaxf(fminf(xa1, xb1) - fmaxf(xa0, xb0), 0.f);
  const float inter = ih * iw;
  return inter / (area_a + area_b - inter);
}

// model.py:113-139 decode_bbox_target
__device__ inline float4 decode_box(float tx, float ty, float tw, float th, const float4 a, float clipv) {
  const float wa = a.z - a.x, ha = a.w - a.y;
  const float xa = (a.z + a.x) * 0.5f, ya = (a.w + a.y) * 0.5f;
  const float wb = expf(fminf(tw, clipv)) * wa, hb = expf(fminf(th, clipv)) * ha;
  const float xb = tx * wa + xa, yb = ty * ha + ya;
  return make_float4(xb - wb * 0.5f, yb - hb * 0.5f, xb + wb * 0.5f, yb + hb * 0.5f);
}

__device__ inline float4 clip_box(float4 b, float imh, float imw) {   // model.py:17-27
  b.x = fminf(fmaxf(b.x, 0.f), imw);
  b.y = fminf(fmaxf(b.y, 0.f), imh);
  b.z = fminf(fmaxf(b.z, 0.f), imw);
  b.w = fminf(fmaxf(b.w, 0.f), imh);
  return b;
}

__device__ inline unsigned ordered_key(float f) {   // monotone float -> uint
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// descending bitonic sort of NT 64-bit keys held in LDS (NT threads, NT power of two)
template <int NT>
__device__ inline void bitonic_desc(unsigned long long* key, int tid) {
  for (int k = 2; k <= NT; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      const int ixj = tid ^ j;
      if (ixj > tid) {
        const unsigned long long a = key[tid], b = key[ixj];
        const bool desc = (tid & k) == 0;
        if (desc ? (a < b) : (a > b)) {
          key[tid] = b;
          key[ixj] = a;
        }
      }
    }
  __syncthreads();
}

// Greedy NMS run by ONE wave: candidates visited in order, the kept boxes live in registers
// (lane l holds kept boxes l, l+64, ...), decision by wave vote -- exactly TF's "compare with every
// already selected box" loop.  Returns the number kept (wave-uniform).  `emit(rank, cand)` is
// called by lane 0 for each kept candidate.
template <int SLOTS, typename Emit>
__device__ inline int nms_wave(const float4* __restrict__ boxes, int n, int max_out, float thresh, int lane,
                               Emit emit) {
  float4 kb[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) kb[s] = make_float4(0.f, 0.f, 0.f, 0.f);
  int kept = 0;
  for (int i = 0; i < n && kept < max_out; ++i) {
    const float4 bi = boxes[i];
    bool sup = false;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
      if (s * 64 + lane < kept && iou_tf(bi, kb[s]) > thresh) sup = true;
    if (!__any(sup)) {
#pragma unroll
      for (int s = 0; s < SLOTS; ++s)
        if (kept == s * 64 + lane) kb[s] = bi;
      if (lane == 0) emit(kept, i);
      ++kept;
    }
  }
  return kept;
}

// ------------------------------------------------------------------------------------------
// generate_rpn_proposals (model.py:169-217) + anchors (data.py:34-74) + decode, one 1024-thread
// workgroup per image:  radix-select the pre_k largest logits of the fh*fw*na anchors (ties -> lower
// index, the set tf.nn.top_k returns), bitonic sort by (logit desc, index asc), decode + clip + drop
// w/h <= min_size (order preserving compaction), greedy NMS (IoU > thresh suppressed) keep <= post_k.
constexpr int RPN_NT = 1024;

struct RpnArgs {
  const float* rpn;   // NHWC [n][fh][fw][ps]: logits at [logit_off, +na), deltas at box_off + a*4 + {tx,ty,tw,th}
  int ps, fh, fw, na, logit_off, box_off;
  const float* cell_anchors;   // [na][4] anchors of cell (0,0), x2/y2 already +1
  float stride, img_h, img_w, nms_thresh, min_size, decode_clip;
  int pre_k, post_k;
  float* out_boxes;    // [n][post_k][4]
  float* out_scores;   // [n][post_k]
  int* out_idx;        // [n][post_k]  flat anchor index ((y*fw+x)*na + a)
  int* out_count;      // [n]
};

__global__ __launch_bounds__(RPN_NT) void rpn_proposals_kernel(const RpnArgs p) {
  __shared__ unsigned long long key[RPN_NT];
  __shared__ float4 vbox[RPN_NT];
  __shared__ float vscore[RPN_NT];
  __shared__ int vidx[RPN_NT];
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_remaining;
  __shared__ int wtot[RPN_NT / 64];
  __shared__ int s_ncand, s_nvalid;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int img = blockIdx.x;
  const int N = p.fh * p.fw * p.na;
  const float* base = p.rpn + (long)img * p.fh * p.fw * p.ps;
  auto logit = [&](int i) { return base[(long)(i / p.na) * p.ps + p.logit_off + (i % p.na)]; };

  const int K = p.pre_k < N ? p.pre_k : N;
  // ---- radix select: key value T of the K-th largest, `remaining` = how many == T to take ----
  unsigned prefix = 0, mask = 0, remaining = (unsigned)K;
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += RPN_NT) {
      const unsigned k = ordered_key(logit(i));
      if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned cum = 0;
      int b = 255;
      for (; b > 0; --b) {
        if (cum + hist[b] >= remaining) break;
        cum += hist[b];
      }
      s_prefix = prefix | ((unsigned)b << shift);
      s_remaining = remaining - cum;
    }
    __syncthreads();
    prefix = s_prefix;
    remaining = s_remaining;
    mask |= 0xFFu << shift;
  }
  const unsigned T = prefix;

  // ---- gather the K selected (all > T, and the first `remaining` == T in index order) ----
  key[tid] = 0ull;
  if (tid == 0) s_ncand = 0;
  __syncthreads();
  int eq_before = 0;
  for (int b0 = 0; b0 < N; b0 += RPN_NT) {
    const int i = b0 + tid;
    const unsigned k = i < N ? ordered_key(logit(i)) : 0u;
    const bool gt = i < N && k > T, eq = i < N && k == T;
    const unsigned long long m = __ballot(eq);
    const int rank_in_wave = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[wave] = __popcll(m);
    __syncthreads();
    int off = eq_before, tot = 0;
    for (int wv = 0; wv < RPN_NT / 64; ++wv) {
      if (wv < wave) off += wtot[wv];
      tot += wtot[wv];
    }
    const bool take = gt || (eq && (unsigned)(off + rank_in_wave) < remaining);
    if (take) {
      const int pos = atomicAdd(&s_ncand, 1);
      key[pos] = ((unsigned long long)k << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    }
    eq_before += tot;
    __syncthreads();
  }
  bitonic_desc<RPN_NT>(key, tid);
  const int ncand = s_ncand;   // == K

  // ---- decode + clip + validity, order-preserving compaction ----
  bool valid = false;
  float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
  float sc = 0.f;
  int ai = 0;
  if (tid < ncand) {
    ai = (int)(0xFFFFFFFFu - (unsigned)(key[tid] & 0xFFFFFFFFull));
    const int a = ai % p.na, cell = ai / p.na;
    const int cx = cell % p.fw, cy = cell / p.fw;
    const float* ca = p.cell_anchors + a * 4;
    const float sxs = (float)cx * p.stride, sys = (float)cy * p.stride;
    const float4 anc = make_float4(ca[0] + sxs, ca[1] + sys, ca[2] + sxs, ca[3] + sys);
    const float* d = base + (long)cell * p.ps + p.box_off + a * 4;
    sc = base[(long)cell * p.ps + p.logit_off + a];
    bx = clip_box(decode_box(d[0], d[1], d[2], d[3], anc, p.decode_clip), p.img_h, p.img_w);
    valid = (bx.z - bx.x > p.min_size) && (bx.w - bx.y > p.min_size);
  }
  {
    const unsigned long long m = __ballot(valid);
    if (lane == 0) wtot[wave] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
    for (int wv = 0; wv < RPN_NT / 64; ++wv) {
      if (wv < wave) off += wtot[wv];
      tot += wtot[wv];
    }
    if (valid) {
      const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
      vbox[pos] = bx;
      vscore[pos] = sc;
      vidx[pos] = ai;
    }
    if (tid == 0) s_nvalid = tot;
    __syncthreads();
  }

  // ---- greedy NMS by wave 0 ----
  if (wave == 0) {
    float* ob = p.out_boxes + (long)img * p.post_k * 4;
    float* os = p.out_scores + (long)img * p.post_k;
    int* oi = p.out_idx + (long)img * p.post_k;
    const int kept = nms_wave<2>(vbox, s_nvalid, p.post_k < 128 ? p.post_k : 128, p.nms_thresh, lane,
                                 [&](int r, int i) {
                                   const float4 b = vbox[i];
                                   ob[r * 4 + 0] = b.x;
                                   ob[r * 4 + 1] = b.y;
                                   ob[r * 4 + 2] = b.z;
                                   ob[r * 4 + 3] = b.w;
                                   os[r] = vscore[i];
                                   oi[r] = vidx[i];
                                 });
    for (int r = kept + lane; r < p.post_k; r += 64) {   // deterministic tail for fixed-size consumers
      ob[r * 4 + 0] = ob[r * 4 + 1] = ob[r * 4 + 2] = ob[r * 4 + 3] = 0.f;
      os[r] = 0.f;
      oi[r] = -1;
    }
    if (lane == 0) p.out_count[img] = kept;
  }
}

// ------------------------------------------------------------------------------------------
// RoIAlign (model.py:300-374): the fpcoor box -> normalised box remap, tf.image.crop_and_resize to
// (2*out)^2 with extrapolation 0 and the 2x2 average pool, fused; NHWC, one thread per
// (roi, bin, 4 channels).  RoI slots >= count[img] are written as zeros (fixed-size graph).
__global__ __launch_bounds__(256) void roi_align_kernel(const float* __restrict__ fm, int ps, int H, int W, int c4,
                                                        const float* __restrict__ rois, const int* __restrict__ count,
                                                        int rois_per_img, int n_img, float scale, int outsz,
                                                        float* __restrict__ out, int out_ps) {
  const long total = (long)n_img * rois_per_img * outsz * outsz * c4;
  const int crop = 2 * outsz;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = idx % c4;
    long t = idx / c4;
    const int bx_ = t % outsz;
    t /= outsz;
    const int by_ = t % outsz;
    const long r = t / outsz;
    const int img = r / rois_per_img, rl = r % rois_per_img;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < count[img]) {
      const float* rb = rois + r * 4;
      const float x0 = rb[0] * scale, y0 = rb[1] * scale, x1 = rb[2] * scale, y1 = rb[3] * scale;
      const float sw = (x1 - x0) / (float)crop, sh = (y1 - y0) / (float)crop;
      const float nx0 = (x0 + sw / 2.f - 0.5f) / (float)(W - 1), ny0 = (y0 + sh / 2.f - 0.5f) / (float)(H - 1);
      const float nw = sw * (float)(crop - 1) / (float)(W - 1), nh = sh * (float)(crop - 1) / (float)(H - 1);
      const float by1 = ny0, bx1 = nx0, by2 = ny0 + nh, bx2 = nx0 + nw;
      const float hs = (by2 - by1) * (float)(H - 1) / (float)(crop - 1);
      const float ws = (bx2 - bx1) * (float)(W - 1) / (float)(crop - 1);
      const float* fb = fm + (long)img * H * W * ps + cg * 4;
      // the 2x2 samples of the bin: coordinates first, then all 16 corner loads (samples outside the map are not loaded:
      // extrapolation value 0), then the bilinear sums in sample order -- no load waits for arithmetic on an earlier one
      bool oky[2], okx[2];
      int yt[2], yb[2], xl[2], xr[2];
      float yl[2], xlp[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float in_y = by1 * (float)(H - 1) + (float)(2 * by_ + j) * hs;
        oky[j] = !(in_y < 0.f || in_y > (float)(H - 1));
        yt[j] = (int)floorf(in_y), yb[j] = (int)ceilf(in_y);
        yl[j] = in_y - floorf(in_y);
        const float in_x = bx1 * (float)(W - 1) + (float)(2 * bx_ + j) * ws;
        okx[j] = !(in_x < 0.f || in_x > (float)(W - 1));
        xl[j] = (int)floorf(in_x), xr[j] = (int)ceilf(in_x);
        xlp[j] = in_x - floorf(in_x);
      }
      float4 tl[2][2], tr[2][2], bl[2][2], br[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (oky[j] && okx[i]) {
            tl[j][i] = *reinterpret_cast<const float4*>(fb + ((long)yt[j] * W + xl[i]) * ps);
            tr[j][i] = *reinterpret_cast<const float4*>(fb + ((long)yt[j] * W + xr[i]) * ps);
            bl[j][i] = *reinterpret_cast<const float4*>(fb + ((long)yb[j] * W + xl[i]) * ps);
            br[j][i] = *reinterpret_cast<const float4*>(fb + ((long)yb[j] * W + xr[i]) * ps);
          }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (oky[j] && okx[i]) {
            const float4 a = tl[j][i], b = tr[j][i], c = bl[j][i], d = br[j][i];
            float top, bot;
            top = a.x + (b.x - a.x) * xlp[i]; bot = c.x + (d.x - c.x) * xlp[i]; acc.x += top + (bot - top) * yl[j];
            top = a.y + (b.y - a.y) * xlp[i]; bot = c.y + (d.y - c.y) * xlp[i]; acc.y += top + (bot - top) * yl[j];
            top = a.z + (b.z - a.z) * xlp[i]; bot = c.z + (d.z - c.z) * xlp[i]; acc.z += top + (bot - top) * yl[j];
            top = a.w + (b.w - a.w) * xlp[i]; bot = c.w + (d.w - c.w) * xlp[i]; acc.w += top + (bot - top) * yl[j];
          }
      acc.x *= 0.25f; acc.y *= 0.25f; acc.z *= 0.25f; acc.w *= 0.25f;
    }
    *reinterpret_cast<float4*>(out + ((r * outsz + by_) * outsz + bx_) * out_ps + cg * 4) = acc;
  }
}

// The same RoIAlign with one thread per (RoI, output ROW, 4 channels) that walks the row's 2 x 28 samples from left to right and
// keeps the corner pixels of its four feature-map rows (top / bottom row of the two sample rows) in registers: a sample loads only
// when its column pair moved on -- an object-sized box (100 px = 6 feature-map pixels wide) takes ~2 loads per output pixel instead
// of 16, a frame-wide one ~8; the per-bin form above is bound by those loads (80 M 16-byte requests out of L2 / Infinity Cache for
// 80 MB of output: 1.2 TB/s).  A wave = 64 channel groups of ONE (RoI, row): every branch below is wave-uniform.  Same sample
// values (identical expressions) added in the same order: bit-identical to roi_align_kernel (tests/test_gpu_proposal.py).
__global__ __launch_bounds__(256) void roi_align_row_kernel(const float* __restrict__ fm, int ps, int H, int W, int c4,
                                                            const float* __restrict__ rois, const int* __restrict__ count,
                                                            int rois_per_img, int n_img, float scale, int outsz,
                                                            float* __restrict__ out, int out_ps) {
  const long total = (long)n_img * rois_per_img * outsz * c4;
  const int crop = 2 * outsz;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = idx % c4;
    long t = idx / c4;
    const int by_ = t % outsz;
    const long r = t / outsz;
    const int img = r / rois_per_img, rl = r % rois_per_img;
    float* orow = out + ((r * outsz + by_) * outsz) * out_ps + cg * 4;
    if (rl >= count[img]) {
      for (int bx_ = 0; bx_ < outsz; ++bx_) *reinterpret_cast<float4*>(orow + (long)bx_ * out_ps) = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    const float* rb = rois + r * 4;
    const float x0 = rb[0] * scale, y0 = rb[1] * scale, x1 = rb[2] * scale, y1 = rb[3] * scale;
    const float sw = (x1 - x0) / (float)crop, sh = (y1 - y0) / (float)crop;
    const float nx0 = (x0 + sw / 2.f - 0.5f) / (float)(W - 1), ny0 = (y0 + sh / 2.f - 0.5f) / (float)(H - 1);
    const float nw = sw * (float)(crop - 1) / (float)(W - 1), nh = sh * (float)(crop - 1) / (float)(H - 1);
    const float by1 = ny0, bx1 = nx0, by2 = ny0 + nh, bx2 = nx0 + nw;
    const float hs = (by2 - by1) * (float)(H - 1) / (float)(crop - 1);
    const float ws = (bx2 - bx1) * (float)(W - 1) / (float)(crop - 1);
    const float* fb = fm + (long)img * H * W * ps + cg * 4;
    bool oky[2];
    float yl[2];
    const float* rt[2];
    const float* rbt[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float in_y = by1 * (float)(H - 1) + (float)(2 * by_ + j) * hs;
      oky[j] = !(in_y < 0.f || in_y > (float)(H - 1));
      const int yt = oky[j] ? (int)floorf(in_y) : 0, yb = oky[j] ? (int)ceilf(in_y) : 0;
      yl[j] = in_y - floorf(in_y);
      rt[j] = fb + (long)yt * W * ps;
      rbt[j] = fb + (long)yb * W * ps;
    }
    // corner cache: [j][0] = top row, [j][1] = bottom row of sample row j; L = column cxl, R = column cxr
    float4 L[2][2], R[2][2];
    int cxl = -2, cxr = -2;
    float4 sv[2][2];                       // sample values of the current bin: [j][i]
    bool sok[2];
    for (int bx_ = 0; bx_ < outsz; ++bx_) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {          // (static indices: sv / sok stay in registers)
        const int sx = 2 * bx_ + i;
        const float in_x = bx1 * (float)(W - 1) + (float)sx * ws;
        const bool okx = !(in_x < 0.f || in_x > (float)(W - 1));
        sok[i] = okx;
        if (okx) {
          const int xl = (int)floorf(in_x), xr = (int)ceilf(in_x);
          const float xlp = in_x - floorf(in_x);
          if (xl != cxl || xr != cxr) {
            const bool shift = xl == cxr;                  // the old right column is the new left one
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (oky[j]) {
                if (shift) { L[j][0] = R[j][0]; L[j][1] = R[j][1]; }
                else {
                  L[j][0] = *reinterpret_cast<const float4*>(rt[j] + (long)xl * ps);
                  L[j][1] = *reinterpret_cast<const float4*>(rbt[j] + (long)xl * ps);
                }
                if (xr == xl) { R[j][0] = L[j][0]; R[j][1] = L[j][1]; }
                else {
                  R[j][0] = *reinterpret_cast<const float4*>(rt[j] + (long)xr * ps);
                  R[j][1] = *reinterpret_cast<const float4*>(rbt[j] + (long)xr * ps);
                }
              }
            cxl = xl; cxr = xr;
          }
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (oky[j]) {
              const float4 a = L[j][0], b = R[j][0], c = L[j][1], d = R[j][1];
              float top, bot;
              top = a.x + (b.x - a.x) * xlp; bot = c.x + (d.x - c.x) * xlp; sv[j][i].x = top + (bot - top) * yl[j];
              top = a.y + (b.y - a.y) * xlp; bot = c.y + (d.y - c.y) * xlp; sv[j][i].y = top + (bot - top) * yl[j];
              top = a.z + (b.z - a.z) * xlp; bot = c.z + (d.z - c.z) * xlp; sv[j][i].z = top + (bot - top) * yl[j];
              top = a.w + (b.w - a.w) * xlp; bot = c.w + (d.w - c.w) * xlp; sv[j][i].w = top + (bot - top) * yl[j];
            }
        }
      }
      // the bin is complete: add its samples in the per-bin kernel's order (j, then i)
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
          if (oky[j] && sok[ii]) { acc.x += sv[j][ii].x; acc.y += sv[j][ii].y; acc.z += sv[j][ii].z; acc.w += sv[j][ii].w; }
      acc.x *= 0.25f; acc.y *= 0.25f; acc.z *= 0.25f; acc.w *= 0.25f;
      *reinterpret_cast<float4*>(orow + (long)bx_ * out_ps) = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Global average pool NHWC [n][hw][c] -> [n][c]  (GlobalAvgPooling, model.py:387, 561)
__global__ __launch_bounds__(256) void gap_kernel(const float* __restrict__ in, int in_ps, int n, int hw, int c4,
                                                  float* __restrict__ out, int out_ps) {
  const long total = (long)n * c4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = idx % c4;
    const long b = idx / c4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < hw; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(in + (b * hw + i) * in_ps + cg * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float inv = (float)hw;
    *reinterpret_cast<float4*>(out + b * out_ps + cg * 4) = make_float4(s.x / inv, s.y / inv, s.z / inv, s.w / inv);
  }
}

// Large-hw variant (ASPP image pooling: hw = 625, n = boxes): one block = 32 channel quads x 8 pixel slices, partial sums
// combined through LDS in slice order (deterministic).
__global__ __launch_bounds__(256) void gap_sliced_kernel(const float* __restrict__ in, int in_ps, int hw, int c4,
                                                         float* __restrict__ out, int out_ps) {
  __shared__ float4 part[8][32];
  const int lane = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int cg = blockIdx.x * 32 + lane;
  const long b = blockIdx.y;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cg < c4) {
    const float* base = in + b * hw * (long)in_ps + cg * 4;
    for (int i = slice; i < hw; i += 8) {
      const float4 v = *reinterpret_cast<const float4*>(base + (long)i * in_ps);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[slice][lane] = s;
  __syncthreads();
  if (slice == 0 && cg < c4) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const float4 v = part[k][lane];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float inv = (float)hw;
    *reinterpret_cast<float4*>(out + b * out_ps + cg * 4) = make_float4(s.x / inv, s.y / inv, s.z / inv, s.w / inv);
  }
}

// ------------------------------------------------------------------------------------------
// Fast R-CNN inference tail for the class-agnostic net (train.py:275-295, model.py:438-491):
// softmax over {bg,fg}, decode deltas/[10,10,5,5] on the proposals, clip, keep p > thresh, greedy NMS
// (IoU > nms_thresh) keep <= max_out, results ordered by descending p (ties lower proposal index).
constexpr int TAIL_NT = 128;

struct TailArgs {
  const float* head;   // [n][rois_per_img][ps]: cls logits at [0,2), box deltas at [2,6)
  int ps, rois_per_img;
  const float* rois;   // [n][rois_per_img][4]
  const int* count;    // [n]
  float img_h, img_w, score_thresh, nms_thresh, decode_clip;
  float rw0, rw1, rw2, rw3;   // FASTRCNN_BBOX_REG_WEIGHTS
  int max_out;
  float* out_boxes;    // [n][max_out][4]
  float* out_probs;    // [n][max_out]
  int* out_idx;        // [n][max_out] proposal index
  int* out_count;      // [n]
};

__global__ __launch_bounds__(TAIL_NT) void frcnn_tail_kernel(const TailArgs p) {
  __shared__ unsigned long long key[TAIL_NT];
  __shared__ float4 dbox[TAIL_NT];
  __shared__ float dprob[TAIL_NT];
  __shared__ float4 sbox[TAIL_NT];
  __shared__ int sidx[TAIL_NT];
  __shared__ int s_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, img = blockIdx.x;
  const int cnt = p.count[img] < p.rois_per_img ? p.count[img] : p.rois_per_img;
  key[tid] = 0ull;
  if (tid < cnt) {
    const float* hd = p.head + ((long)img * p.rois_per_img + tid) * p.ps;
    const float l0 = hd[0], l1 = hd[1], m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float pr = e1 / (e0 + e1);
    const float* rb = p.rois + ((long)img * p.rois_per_img + tid) * 4;
    const float4 anc = make_float4(rb[0], rb[1], rb[2], rb[3]);
    const float4 b = clip_box(decode_box(hd[2] / p.rw0, hd[3] / p.rw1, hd[4] / p.rw2, hd[5] / p.rw3, anc, p.decode_clip),
                              p.img_h, p.img_w);
    dbox[tid] = b;
    dprob[tid] = pr;
    if (pr > p.score_thresh)
      key[tid] = ((unsigned long long)ordered_key(pr) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)tid);
  }
  bitonic_desc<TAIL_NT>(key, tid);
  {
    const bool ok = key[tid] != 0ull;
    const int i = (int)(0xFFFFFFFFu - (unsigned)(key[tid] & 0xFFFFFFFFull));
    if (ok) {
      sbox[tid] = dbox[i];
      sidx[tid] = i;
    }
    const unsigned long long m = __ballot(ok);
    if (tid == 0) s_n = 0;
    __syncthreads();
    if (lane == 0) atomicAdd(&s_n, __popcll(m));
    __syncthreads();
  }
  if (wave == 0) {
    float* ob = p.out_boxes + (long)img * p.max_out * 4;
    float* op = p.out_probs + (long)img * p.max_out;
    int* oi = p.out_idx + (long)img * p.max_out;
    const int kept = nms_wave<1>(sbox, s_n, p.max_out < 64 ? p.max_out : 64, p.nms_thresh, lane, [&](int r, int i) {
      const float4 b = sbox[i];
      ob[r * 4 + 0] = b.x;
      ob[r * 4 + 1] = b.y;
      ob[r * 4 + 2] = b.z;
      ob[r * 4 + 3] = b.w;
      op[r] = dprob[sidx[i]];
      oi[r] = sidx[i];
    });
    for (int r = kept + lane; r < p.max_out; r += 64) {
      ob[r * 4 + 0] = ob[r * 4 + 1] = ob[r * 4 + 2] = ob[r * 4 + 3] = 0.f;
      op[r] = 0.f;
      oi[r] = -1;
    }
    if (lane == 0) p.out_count[img] = kept;
  }
}

// (round 5: a FLAT grid -- one item per thread -- moves 6.1 TB/s where a grid-stride loop over 4 ... 8 k resident workgroups moves 4.3 ... 4.7,
// tools/dev/hbm_copy_variants.hip; the cap only bounds the grid dimension)
inline int grid_for(long total, int per_block = 256, int cap = 1 << 22) {
  long g = (total + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : g > cap ? cap : g);
}

}  // namespace

extern "C" int premvos_proposal_preprocess_u8(const uint8_t* img_bgr, int32_t batch, int32_t h, int32_t w, float* out,
                                              int32_t nh, int32_t nw, int32_t src_is_rgb, void* stream) {
  PV_REQUIRE(img_bgr && out, "proposal_preprocess: null pointer");
  PV_REQUIRE(batch > 0 && h > 0 && w > 0 && nh > 0 && nw > 0, "proposal_preprocess: bad dims");
  PV_REQUIRE(premvos::aligned16(out), "proposal_preprocess: out must be 16-byte aligned");
  hipLaunchKernelGGL(proposal_preprocess_kernel, dim3(grid_for((long)batch * nh * nw)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), img_bgr, batch, h, w, out, nh, nw, src_is_rgb);
  return premvos::check_launch("proposal_preprocess");
}

extern "C" int premvos_maxpool_f32(const float* in, int32_t in_ps, int32_t n, int32_t h, int32_t w, int32_t c,
                                   float* out, int32_t out_ps, int32_t ho, int32_t wo, int32_t k, int32_t stride,
                                   int32_t pt, int32_t pl, float pad_value, void* stream) {
  PV_REQUIRE(in && out, "maxpool: null pointer");
  PV_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && ho > 0 && wo > 0 && k > 0 && stride > 0, "maxpool: bad dims");
  PV_REQUIRE(c % 4 == 0 && in_ps % 4 == 0 && out_ps % 4 == 0 && in_ps >= c && out_ps >= c,
             "maxpool: C and pixel strides must be multiples of 4");
  PV_REQUIRE(premvos::aligned16(in) && premvos::aligned16(out), "maxpool: in/out must be 16-byte aligned");
  hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for((long)n * ho * wo * (c / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, in_ps, n, h, w, c / 4, out, out_ps, ho, wo, k, stride, pt,
                     pl, pad_value);
  return premvos::check_launch("maxpool");
}

extern "C" int premvos_rpn_proposals_f32(const float* rpn, int32_t ps, int32_t n, int32_t fh, int32_t fw, int32_t na,
                                         int32_t logit_off, int32_t box_off, const float* cell_anchors, float stride,
                                         float img_h, float img_w, int32_t pre_nms_topk, int32_t post_nms_topk,
                                         float nms_thresh, float min_size, float decode_clip, float* out_boxes,
                                         float* out_scores, int32_t* out_idx, int32_t* out_count, void* stream) {
  PV_REQUIRE(rpn && cell_anchors && out_boxes && out_scores && out_idx && out_count, "rpn_proposals: null pointer");
  PV_REQUIRE(n > 0 && fh > 0 && fw > 0 && na > 0, "rpn_proposals: bad dims");
  PV_REQUIRE(ps >= logit_off + na && ps >= box_off + 4 * na && logit_off >= 0 && box_off >= 0,
             "rpn_proposals: channel window exceeds pixel stride");
  PV_REQUIRE(pre_nms_topk > 0 && pre_nms_topk <= RPN_NT, "rpn_proposals: pre_nms_topk must be in [1,%d]", RPN_NT);
  PV_REQUIRE(post_nms_topk > 0 && post_nms_topk <= 128, "rpn_proposals: post_nms_topk must be in [1,128]");
  PV_REQUIRE((long)fh * fw * na < (1L << 31), "rpn_proposals: too many anchors");
  RpnArgs a;
  a.rpn = rpn; a.ps = ps; a.fh = fh; a.fw = fw; a.na = na; a.logit_off = logit_off; a.box_off = box_off;
  a.cell_anchors = cell_anchors; a.stride = stride; a.img_h = img_h; a.img_w = img_w; a.nms_thresh = nms_thresh;
  a.min_size = min_size; a.decode_clip = decode_clip; a.pre_k = pre_nms_topk; a.post_k = post_nms_topk;
  a.out_boxes = out_boxes; a.out_scores = out_scores; a.out_idx = out_idx; a.out_count = out_count;
  hipLaunchKernelGGL(rpn_proposals_kernel, dim3(n), dim3(RPN_NT), 0, static_cast<hipStream_t>(stream), a);
  return premvos::check_launch("rpn_proposals");
}

extern "C" int premvos_roi_align_f32(const float* fmap, int32_t fmap_ps, int32_t n_img, int32_t h, int32_t w,
                                     int32_t c, const float* rois, const int32_t* count, int32_t rois_per_img,
                                     float spatial_scale, int32_t out_size, float* out, int32_t out_ps, void* stream) {
  PV_REQUIRE(fmap && rois && count && out, "roi_align: null pointer");
  PV_REQUIRE(n_img > 0 && h > 1 && w > 1 && c > 0 && rois_per_img > 0 && out_size > 0, "roi_align: bad dims");
  PV_REQUIRE(c % 4 == 0 && fmap_ps % 4 == 0 && out_ps % 4 == 0 && fmap_ps >= c && out_ps >= c,
             "roi_align: C and pixel strides must be multiples of 4");
  PV_REQUIRE(premvos::aligned16(fmap) && premvos::aligned16(out), "roi_align: fmap/out must be 16-byte aligned");
  // PREMVOS_ROI_ALIGN=bin: the per-bin kernel (developer A/B; both give the same bits)
  static const bool per_bin = [] { const char* e = getenv("PREMVOS_ROI_ALIGN"); return e != nullptr && e[0] == 'b'; }();
  if (per_bin) {
    const long total = (long)n_img * rois_per_img * out_size * out_size * (c / 4);
    hipLaunchKernelGGL(roi_align_kernel, dim3(grid_for(total)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), fmap, fmap_ps, h, w, c / 4, rois, count, rois_per_img, n_img,
                       spatial_scale, out_size, out, out_ps);
  } else {
    const long total = (long)n_img * rois_per_img * out_size * (c / 4);
    hipLaunchKernelGGL(roi_align_row_kernel, dim3(grid_for(total)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), fmap, fmap_ps, h, w, c / 4, rois, count, rois_per_img, n_img,
                       spatial_scale, out_size, out, out_ps);
  }
  return premvos::check_launch("roi_align");
}

extern "C" int premvos_global_avgpool_f32(const float* in, int32_t in_ps, int32_t n, int32_t hw, int32_t c, float* out,
                                          int32_t out_ps, void* stream) {
  PV_REQUIRE(in && out, "global_avgpool: null pointer");
  PV_REQUIRE(n > 0 && hw > 0 && c > 0, "global_avgpool: bad dims");
  PV_REQUIRE(c % 4 == 0 && in_ps % 4 == 0 && out_ps % 4 == 0 && in_ps >= c && out_ps >= c,
             "global_avgpool: C and strides must be multiples of 4");
  PV_REQUIRE(premvos::aligned16(in) && premvos::aligned16(out), "global_avgpool: in/out must be 16-byte aligned");
  if (hw >= 128) {
    hipLaunchKernelGGL(gap_sliced_kernel, dim3((c / 4 + 31) / 32, n), dim3(256), 0, static_cast<hipStream_t>(stream), in,
                       in_ps, hw, c / 4, out, out_ps);
  } else {
    hipLaunchKernelGGL(gap_kernel, dim3(grid_for((long)n * (c / 4))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       in, in_ps, n, hw, c / 4, out, out_ps);
  }
  return premvos::check_launch("global_avgpool");
}

extern "C" int premvos_frcnn_tail_f32(const float* head, int32_t head_ps, const float* rois, const int32_t* count,
                                      int32_t n_img, int32_t rois_per_img, float img_h, float img_w,
                                      float score_thresh, float nms_thresh, int32_t max_out, float decode_clip,
                                      float rw_x, float rw_y, float rw_w, float rw_h, float* out_boxes,
                                      float* out_probs, int32_t* out_idx, int32_t* out_count, void* stream) {
  PV_REQUIRE(head && rois && count && out_boxes && out_probs && out_idx && out_count, "frcnn_tail: null pointer");
  PV_REQUIRE(rw_x > 0.f && rw_y > 0.f && rw_w > 0.f && rw_h > 0.f, "frcnn_tail: bad regression weights");
  PV_REQUIRE(n_img > 0 && rois_per_img > 0 && rois_per_img <= TAIL_NT, "frcnn_tail: rois_per_img must be in [1,%d]",
             TAIL_NT);
  PV_REQUIRE(head_ps >= 6 && max_out > 0 && max_out <= 64, "frcnn_tail: bad head_ps/max_out");
  TailArgs a;
  a.head = head; a.ps = head_ps; a.rois_per_img = rois_per_img; a.rois = rois; a.count = count;
  a.img_h = img_h; a.img_w = img_w; a.score_thresh = score_thresh; a.nms_thresh = nms_thresh;
  a.decode_clip = decode_clip;
  a.rw0 = rw_x; a.rw1 = rw_y; a.rw2 = rw_w; a.rw3 = rw_h;
  a.max_out = max_out; a.out_boxes = out_boxes; a.out_probs = out_probs; a.out_idx = out_idx; a.out_count = out_count;
  hipLaunchKernelGGL(frcnn_tail_kernel, dim3(n_img), dim3(TAIL_NT), 0, static_cast<hipStream_t>(stream), a);
  return premvos::check_launch("frcnn_tail");
}
