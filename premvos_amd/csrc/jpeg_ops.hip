// Optional GPU JPEG decode: baseline Huffman JPEG -> uint8 RGB / BGR frame in HBM.
//
// Frames enter the reference through cv2.imread (proposal_net/train.py:500), scipy.ndimage.imread = PIL
// (optical_flow_net-PWC-Net/script_pwc_multi.py:34) and PIL (ReID_net/prepare_input.py:38): libjpeg(-turbo) with its defaults
// (JDCT_ISLOW, fancy up-sampling, YCbCr -> RGB).  The bit stream is inherently serial, so the split is
//   host  : marker parsing + Huffman decoding into quantised coefficients (premvos_jpeg_entropy_decode_host, plain C++, no GPU),
//   device: de-quantisation + jidctint.c's 8x8 inverse DCT (one thread per block), then triangle-filter chroma up-sampling fused
//           with the fixed-point colour conversion (one thread per pixel) -- 6 bytes of HBM traffic per output pixel.
// Results are the library's bytes (tests/test_gpu_jpeg.py compares with PIL and with oracle/jpeg_oracle.py).
#include "common.h"

#include <string.h>

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// host side: T.81 baseline sequential decoding
// ---------------------------------------------------------------------------------------------------------------------------
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int64_t kMaxPixels = 64LL << 20;

struct Huff {
  bool present = false;
  int nsyms = 0;
  uint8_t syms[256];
  int32_t maxcode[18];       // largest code of each length (-1: none), T.81 F.2.2.3
  int32_t valptr[17];        // index of the first symbol of each length
  int32_t mincode[17];
  uint16_t fast[512];        // 9-bit prefix -> (length << 8) | symbol, 0 = longer code
};

// false: the code-length counts do not describe a prefix code (over-subscribed: more codes of some length than the code
// space has left -- libjpeg's jdhuff.c rejects the same tables with JERR_BAD_HUFF_TABLE).  Without the check `code` outgrows
// 1 << len and the fast-table fill below writes past its 512 entries.
bool build_huff(Huff& h, const uint8_t* counts, const uint8_t* syms, int n) {
  int code = 0;
  for (int len = 1; len <= 16; ++len) {
    if (code + counts[len - 1] > (1 << len)) return false;
    code = (code + counts[len - 1]) << 1;
  }
  h.present = true;
  h.nsyms = n;
  memcpy(h.syms, syms, n);
  memset(h.fast, 0, sizeof(h.fast));
  code = 0;
  int k = 0;
  for (int len = 1; len <= 16; ++len) {
    h.valptr[len] = k;
    h.mincode[len] = code;
    for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
      if (len <= 9) {
        const int first = code << (9 - len);
        for (int f = 0; f < (1 << (9 - len)); ++f) h.fast[first + f] = (uint16_t)((len << 8) | syms[k]);
      }
    }
    h.maxcode[len] = counts[len - 1] ? code - 1 : -1;
    code <<= 1;
  }
  h.maxcode[17] = 0x7fffffff;
  return true;
}

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t acc = 0;
  int n = 0;
  bool hit_marker = false;
  int fed = 0;                                    // zero bytes fed behind the end of the segment
  bool overran() const { return n < 8 * fed; }    // some of them were consumed: the data ended early
  void fill() {                                   // top the accumulator up (>= 57 bits, or zeros behind the end of the segment)
    while (n <= 56) {
      if (n <= 32 && !hit_marker && p + 4 <= end) {               // four stuffing-free bytes at once (the common case)
        const uint32_t w = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
        const uint32_t inv = ~w;
        if (((inv - 0x01010101u) & ~inv & 0x80808080u) == 0) {   // no 0xFF byte among them
          acc |= (uint64_t)w << (32 - n);
          n += 32;
          p += 4;
          continue;
        }
      }
      uint32_t b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0) {
            p += 2;
          } else {
            hit_marker = true;
            b = 0;
          }
        } else {
          ++p;
        }
      } else {
        hit_marker = true;
      }
      fed += hit_marker;
      acc |= (uint64_t)b << (56 - n);
      n += 8;
    }
  }
  uint32_t peek(int k) { return (uint32_t)(acc >> (64 - k)); }
  void skip(int k) {
    acc <<= k;
    n -= k;
  }
  bool restart() {                                // byte-align, step over RSTn
    acc = 0;
    n = 0;
    fed = 0;
    hit_marker = false;
    while (p + 1 < end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;
    if (p + 1 >= end) return false;
    p += 2;
    return true;
  }
};

// Both helpers expect the caller to have topped the reader up to >= 32 bits: a Huffman code (<= 16 bits) followed by its value
// bits (<= 15) never needs more.
inline int decode_symbol(BitReader& br, const Huff& h) {
  const uint16_t f = h.fast[br.peek(9)];
  if (f) {
    br.skip(f >> 8);
    return f & 0xFF;
  }
  int code = (int)br.peek(10), len = 10;
  while (len <= 16 && code > h.maxcode[len]) {
    ++len;
    code = (int)br.peek(len);
  }
  if (len > 16) return -1;
  br.skip(len);
  const int idx = h.valptr[len] + code - h.mincode[len];
  return (unsigned)idx < (unsigned)h.nsyms ? h.syms[idx] : -1;     // (a bit pattern below the first code of its length)
}

inline int receive_extend(BitReader& br, int t) {   // T.81 F.2.2.1
  const int v = (int)br.peek(t);
  br.skip(t);
  return v >= (1 << (t - 1)) ? v : v - (1 << t) + 1;
}

struct Parsed {
  premvos_jpeg_info info;
  Huff huff[2][4];
  int dc_sel[3], ac_sel[3];
  int restart_interval = 0;
  const uint8_t* scan = nullptr;
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// Marker segments up to SOS.  PREMVOS_EUNSUPPORTED: a valid file this decoder does not cover (the caller falls back).
int parse_header(const uint8_t* d, int64_t n, Parsed& ps) {
  PV_REQUIRE(n >= 4 && d[0] == 0xFF && d[1] == 0xD8, "jpeg: no SOI marker");
  uint16_t qt[4][64];
  bool qt_present[4] = {false, false, false, false};
  int comp_id[3] = {0, 0, 0}, comp_h[3] = {1, 1, 1}, comp_v[3] = {1, 1, 1}, comp_tq[3] = {0, 0, 0};
  int ncomp = 0, adobe = -1;
  bool sof = false;
  premvos_jpeg_info& I = ps.info;
  memset(&I, 0, sizeof(I));
  int64_t pos = 2;
  for (;;) {
    while (pos < n && d[pos] != 0xFF) ++pos;
    while (pos < n && d[pos] == 0xFF) ++pos;
    PV_REQUIRE(pos < n, "jpeg: truncated before SOS");
    const int m = d[pos++];
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    PV_REQUIRE(m != 0xD9, "jpeg: EOI before SOS");
    PV_REQUIRE(pos + 2 <= n, "jpeg: truncated marker segment");
    const int len = be16(d + pos);
    PV_REQUIRE(len >= 2 && pos + len <= n, "jpeg: marker segment runs past the end of the file");
    const uint8_t* seg = d + pos + 2;
    const int sl = len - 2;
    pos += len;
    if (m == 0xDB) {
      int i = 0;
      while (i < sl) {
        const int pq = seg[i] >> 4, tq = seg[i] & 15;
        ++i;
        PV_REQUIRE(tq < 4 && pq < 2 && i + (pq ? 128 : 64) <= sl, "jpeg: bad DQT segment");
        for (int k = 0; k < 64; ++k) {
          qt[tq][kZigzag[k]] = pq ? (uint16_t)be16(seg + i + 2 * k) : seg[i + k];
        }
        i += pq ? 128 : 64;
        qt_present[tq] = true;
      }
    } else if (m == 0xC4) {
      int i = 0;
      while (i < sl) {
        PV_REQUIRE(i + 17 <= sl, "jpeg: bad DHT segment");
        const int tc = seg[i] >> 4, th = seg[i] & 15;
        int cnt = 0;
        for (int k = 0; k < 16; ++k) cnt += seg[i + 1 + k];
        PV_REQUIRE(tc < 2 && th < 4 && cnt <= 256 && i + 17 + cnt <= sl, "jpeg: bad DHT segment");
        PV_REQUIRE(build_huff(ps.huff[tc][th], seg + i + 1, seg + i + 17, cnt),
                   "jpeg: DHT code lengths do not form a prefix code (over-subscribed table)");
        i += 17 + cnt;
      }
    } else if (m == 0xC0 || m == 0xC1) {
      PV_REQUIRE(sl >= 6, "jpeg: bad SOF segment");
      if (seg[0] != 8) return premvos::fail(PREMVOS_EUNSUPPORTED, "jpeg: %d-bit samples", seg[0]);
      I.height = be16(seg + 1);
      I.width = be16(seg + 3);
      ncomp = seg[5];
      PV_REQUIRE(I.height > 0 && I.width > 0, "jpeg: empty frame");
      // a header may claim 65535 x 65535 pixels (8 GB of pinned coefficients) in front of no data at all: bound what a caller
      // will be asked to allocate (64 Mpixel = 8192 x 8192; the frames of this pipeline are 0.4 ... 2 Mpixel)
      PV_REQUIRE((int64_t)I.height * I.width <= kMaxPixels, "jpeg: frame larger than 64 Mpixel");
      if (ncomp != 1 && ncomp != 3) return premvos::fail(PREMVOS_EUNSUPPORTED, "jpeg: %d components", ncomp);
      PV_REQUIRE(sl >= 6 + 3 * ncomp, "jpeg: bad SOF segment");
      for (int c = 0; c < ncomp; ++c) {
        comp_id[c] = seg[6 + 3 * c];
        comp_h[c] = seg[7 + 3 * c] >> 4;
        comp_v[c] = seg[7 + 3 * c] & 15;
        comp_tq[c] = seg[8 + 3 * c];
        PV_REQUIRE(comp_tq[c] < 4, "jpeg: bad quantisation table selector");
      }
      sof = true;
    } else if (m >= 0xC2 && m <= 0xCF && m != 0xC8 && m != 0xCC) {
      return premvos::fail(PREMVOS_EUNSUPPORTED, "jpeg: SOF marker 0x%02x (progressive / lossless / arithmetic coding)", m);
    } else if (m == 0xDD) {
      PV_REQUIRE(sl >= 2, "jpeg: bad DRI segment");
      ps.restart_interval = be16(seg);
    } else if (m == 0xEE && sl >= 12 && memcmp(seg, "Adobe", 5) == 0) {
      adobe = seg[11];
    } else if (m == 0xDA) {
      PV_REQUIRE(sof, "jpeg: SOS before SOF");
      PV_REQUIRE(sl >= 1 + 2 * ncomp + 3, "jpeg: bad SOS segment");
      if (seg[0] != ncomp) return premvos::fail(PREMVOS_EUNSUPPORTED, "jpeg: multi-scan file");
      for (int c = 0; c < ncomp; ++c) {
        if (seg[1 + 2 * c] != comp_id[c]) return premvos::fail(PREMVOS_EUNSUPPORTED, "jpeg: scan component order");
        ps.dc_sel[c] = seg[2 + 2 * c] >> 4;
        ps.ac_sel[c] = seg[2 + 2 * c] & 15;
        PV_REQUIRE(ps.dc_sel[c] < 4 && ps.ac_sel[c] < 4 && ps.huff[0][ps.dc_sel[c]].present && ps.huff[1][ps.ac_sel[c]].present,
                   "jpeg: scan refers to a Huffman table the file does not define");
        PV_REQUIRE(qt_present[comp_tq[c]], "jpeg: component refers to a quantisation table the file does not define");
      }
      ps.scan = d + pos;
      break;
    }
  }
  if (ncomp == 3) {
    if (adobe >= 0 && adobe != 1) return premvos::fail(PREMVOS_EUNSUPPORTED, "jpeg: Adobe transform %d (RGB / CMYK data)", adobe);
    if (adobe < 0 && comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B')
      return premvos::fail(PREMVOS_EUNSUPPORTED, "jpeg: RGB component ids");
    const bool ok = comp_h[1] == 1 && comp_v[1] == 1 && comp_h[2] == 1 && comp_v[2] == 1 &&
                    ((comp_h[0] == 1 && comp_v[0] == 1) || (comp_h[0] == 2 && comp_v[0] == 1) || (comp_h[0] == 2 && comp_v[0] == 2));
    if (!ok)
      return premvos::fail(PREMVOS_EUNSUPPORTED, "jpeg: sampling factors %dx%d %dx%d %dx%d", comp_h[0], comp_v[0], comp_h[1],
                           comp_v[1], comp_h[2], comp_v[2]);
    I.hs = comp_h[0];
    I.vs = comp_v[0];
    // libjpeg replicates instead of filtering when the chroma plane has fewer than three columns (jdsample.c)
    if (I.hs == 2 && (I.width + 1) / 2 <= 2) return premvos::fail(PREMVOS_EUNSUPPORTED, "jpeg: frame narrower than 5 pixels");
  } else {
    I.hs = I.vs = 1;                                // a one-component scan is not interleaved: MCU = one block
  }
  I.ncomp = ncomp;
  I.mcux = premvos::cdiv(I.width, 8 * I.hs);
  I.mcuy = premvos::cdiv(I.height, 8 * I.vs);
  int64_t off = 0;
  for (int c = 0; c < ncomp; ++c) {
    I.blocks_w[c] = I.mcux * (c == 0 ? I.hs : 1);
    I.blocks_h[c] = I.mcuy * (c == 0 ? I.vs : 1);
    I.coef_offset[c] = off;
    off += (int64_t)I.blocks_w[c] * I.blocks_h[c] * 64;
    memcpy(I.quant[c], qt[comp_tq[c]], sizeof(qt[0]));
  }
  I.coef_count = off;
  return PREMVOS_OK;
}

inline bool decode_block(BitReader& br, const Huff& dc, const Huff& ac, int& pred, int16_t* blk) {
  if (br.n < 32) br.fill();
  int t = decode_symbol(br, dc);
  if (t < 0 || t > 11) return false;
  if (t) pred += receive_extend(br, t);
  blk[0] = (int16_t)pred;
  for (int k = 1; k < 64;) {
    if (br.n < 32) br.fill();
    const int rs = decode_symbol(br, ac);
    if (rs < 0) return false;
    const int r = rs >> 4, s = rs & 15;
    if (s == 0) {
      if (r != 15) break;                           // EOB
      k += 16;                                      // ZRL
      continue;
    }
    k += r;
    if (k > 63) return false;
    blk[kZigzag[k]] = (int16_t)receive_extend(br, s);
    ++k;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------------------------------------
struct Quant {
  uint16_t q[3][64];
};

struct PlaneDesc {
  int32_t blocks_w[3], blocks_h[3];
  int64_t coef_offset[3];      // int16 elements
  int64_t plane_offset[3];     // bytes into the plane workspace; plane c is [blocks_h*8][blocks_w*8]
  int32_t first_block[4];      // prefix sums of the block counts
};

constexpr int FIX_0_298631336 = 2446, FIX_0_390180644 = 3196, FIX_0_541196100 = 4433, FIX_0_765366865 = 6270,
              FIX_0_899976223 = 7373, FIX_1_175875602 = 9633, FIX_1_501321110 = 12299, FIX_1_847759065 = 15137,
              FIX_1_961570560 = 16069, FIX_2_053119869 = 16819, FIX_2_562915447 = 20995, FIX_3_072711026 = 25172;

// One pass of jidctint.c (jpeg_idct_islow) over eight values; SHIFT = 11 for the column pass, 18 for the row pass.
template <int SHIFT>
__device__ inline void idct8(const int (&d)[8], int (&o)[8]) {
  int z2 = d[2], z3 = d[6];
  int z1 = (z2 + z3) * FIX_0_541196100;
  int tmp2 = z1 + z3 * (-FIX_1_847759065);
  int tmp3 = z1 + z2 * FIX_0_765366865;
  int tmp0 = (int)((unsigned)(d[0] + d[4]) << 13), tmp1 = (int)((unsigned)(d[0] - d[4]) << 13);
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = d[7];
  tmp1 = d[5];
  tmp2 = d[3];
  tmp3 = d[1];
  z1 = tmp0 + tmp3;
  z2 = tmp1 + tmp2;
  z3 = tmp0 + tmp2;
  int z4 = tmp1 + tmp3;
  const int z5 = (z3 + z4) * FIX_1_175875602;
  tmp0 *= FIX_0_298631336;
  tmp1 *= FIX_2_053119869;
  tmp2 *= FIX_3_072711026;
  tmp3 *= FIX_1_501321110;
  z1 *= -FIX_0_899976223;
  z2 *= -FIX_2_562915447;
  z3 = z3 * (-FIX_1_961570560) + z5;
  z4 = z4 * (-FIX_0_390180644) + z5;
  tmp0 += z1 + z3;
  tmp1 += z2 + z4;
  tmp2 += z2 + z3;
  tmp3 += z1 + z4;
  constexpr int R = 1 << (SHIFT - 1);
  o[0] = (tmp10 + tmp3 + R) >> SHIFT;
  o[7] = (tmp10 - tmp3 + R) >> SHIFT;
  o[1] = (tmp11 + tmp2 + R) >> SHIFT;
  o[6] = (tmp11 - tmp2 + R) >> SHIFT;
  o[2] = (tmp12 + tmp1 + R) >> SHIFT;
  o[5] = (tmp12 - tmp1 + R) >> SHIFT;
  o[3] = (tmp13 + tmp0 + R) >> SHIFT;
  o[4] = (tmp13 - tmp0 + R) >> SHIFT;
}

// One thread per 8x8 block of any component: 128 B of coefficients in, 64 B of samples out.
__global__ void __launch_bounds__(128) jpeg_idct_kernel(const int16_t* __restrict__ coef, Quant qt, PlaneDesc pd,
                                                        uint8_t* __restrict__ planes) {
  const int b = blockIdx.x * 128 + threadIdx.x;
  if (b >= pd.first_block[3]) return;
  const int c = b >= pd.first_block[2] ? 2 : b >= pd.first_block[1] ? 1 : 0;
  const int lb = b - pd.first_block[c];
  const int by = lb / pd.blocks_w[c], bx = lb - by * pd.blocks_w[c];
  const int4* src = reinterpret_cast<const int4*>(coef + pd.coef_offset[c] + (long)lb * 64);
  int ws[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int4 v = src[r];
    const int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ws[r][2 * k] = (int)(int16_t)(w4[k] & 0xFFFF) * (int)qt.q[c][r * 8 + 2 * k];
      ws[r][2 * k + 1] = (w4[k] >> 16) * (int)qt.q[c][r * 8 + 2 * k + 1];
    }
  }
#pragma unroll
  for (int col = 0; col < 8; ++col) {              // pass 1: columns
    int d[8], o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) d[r] = ws[r][col];
    idct8<11>(d, o);
#pragma unroll
    for (int r = 0; r < 8; ++r) ws[r][col] = o[r];
  }
  uint8_t* dst = planes + pd.plane_offset[c] + ((long)by * 8) * (pd.blocks_w[c] * 8) + bx * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {                    // pass 2: rows, saturate, level shift
    int o[8];
    idct8<18>(ws[r], o);
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      lo |= (uint32_t)(min(max(o[k], -128), 127) + 128) << (8 * k);
      hi |= (uint32_t)(min(max(o[4 + k], -128), 127) + 128) << (8 * k);
    }
    *reinterpret_cast<uint2*>(dst + (long)r * (pd.blocks_w[c] * 8)) = make_uint2(lo, hi);
  }
}

// jdsample.c fancy up-sampling of one chroma sample at output pixel (y, x), HS x VS luma samples per chroma sample.
template <int HS, int VS>
__device__ inline int chroma_at(const uint8_t* __restrict__ p, int pitch, int dw, int dh, int y, int x) {
  if constexpr (HS == 1) {
    return p[(long)y * pitch + x];
  } else {
    const int cx = x >> 1;
    const bool odd = x & 1;
    const int nx = odd ? cx + 1 : cx - 1;
    const bool edge = nx < 0 || nx >= dw;          // first / last column: the sample itself stands in for the missing one
    if constexpr (VS == 1) {
      const uint8_t* row = p + (long)y * pitch;
      const int t = row[cx];
      if (edge) return t;
      return (3 * t + row[nx] + (odd ? 2 : 1)) >> 2;
    } else {
      const int cy = y >> 1;
      const int oy = min(max((y & 1) ? cy + 1 : cy - 1, 0), dh - 1);       // above the first / below the last row: that row again
      const uint8_t* r0 = p + (long)cy * pitch;
      const uint8_t* r1 = p + (long)oy * pitch;
      const int t = 3 * r0[cx] + r1[cx];
      const int bias = odd ? 7 : 8;
      if (edge) return (4 * t + bias) >> 4;
      return (3 * t + 3 * r0[nx] + r1[nx] + bias) >> 4;
    }
  }
}

template <int HS, int VS, bool GREY>
__global__ void __launch_bounds__(256) jpeg_color_kernel(const uint8_t* __restrict__ planes, PlaneDesc pd, int h, int w, int bgr,
                                                         uint8_t* __restrict__ out) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)h * w) return;
  const int y = (int)(idx / w), x = (int)(idx - (long)y * w);
  const int yv = planes[pd.plane_offset[0] + (long)y * (pd.blocks_w[0] * 8) + x];
  int r, g, b;
  if constexpr (GREY) {
    r = g = b = yv;
  } else {
    const int dw = (w + HS - 1) / HS, dh = (h + VS - 1) / VS, pitch = pd.blocks_w[1] * 8;
    const int cb = chroma_at<HS, VS>(planes + pd.plane_offset[1], pitch, dw, dh, y, x) - 128;
    const int cr = chroma_at<HS, VS>(planes + pd.plane_offset[2], pitch, dw, dh, y, x) - 128;
    // jdcolor.c build_ycc_rgb_table: FIX(x) = x * 65536 + 0.5; ONE_HALF = 32768
    r = yv + ((91881 * cr + 32768) >> 16);
    b = yv + ((116130 * cb + 32768) >> 16);
    g = yv + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    r = min(max(r, 0), 255);
    g = min(max(g, 0), 255);
    b = min(max(b, 0), 255);
  }
  uint8_t* o = out + idx * 3;
  o[0] = (uint8_t)(bgr ? b : r);
  o[1] = (uint8_t)g;
  o[2] = (uint8_t)(bgr ? r : b);
}

int plane_desc(const premvos_jpeg_info* I, PlaneDesc& pd, Quant& qt) {
  PV_REQUIRE(I && (I->ncomp == 1 || I->ncomp == 3) && I->width > 0 && I->height > 0, "jpeg: bad info block");
  PV_REQUIRE((I->hs == 1 || I->hs == 2) && (I->vs == 1 || I->vs == 2) && !(I->hs == 1 && I->vs == 2), "jpeg: bad sampling factors");
  int64_t poff = 0;
  int first = 0;
  for (int c = 0; c < 3; ++c) {
    const bool on = c < I->ncomp;
    pd.blocks_w[c] = on ? I->blocks_w[c] : 0;
    pd.blocks_h[c] = on ? I->blocks_h[c] : 0;
    if (on) {
      PV_REQUIRE(I->blocks_w[c] == I->mcux * (c == 0 ? I->hs : 1) && I->blocks_h[c] == I->mcuy * (c == 0 ? I->vs : 1) &&
                     I->mcux == premvos::cdiv(I->width, 8 * I->hs) && I->mcuy == premvos::cdiv(I->height, 8 * I->vs),
                 "jpeg: info block geometry is inconsistent");
    }
    pd.coef_offset[c] = on ? I->coef_offset[c] : 0;
    pd.plane_offset[c] = poff;
    pd.first_block[c] = first;
    poff += (int64_t)pd.blocks_w[c] * pd.blocks_h[c] * 64;
    first += pd.blocks_w[c] * pd.blocks_h[c];
    memcpy(qt.q[c], I->quant[c], sizeof(qt.q[c]));
  }
  pd.first_block[3] = first;
  return PREMVOS_OK;
}

}  // namespace

extern "C" int premvos_jpeg_entropy_decode_host(const uint8_t* data, int64_t n, premvos_jpeg_info* info, int16_t* coef,
                                                int64_t coef_capacity) {
  PV_REQUIRE(data && info && n > 0, "jpeg: null argument");
  Parsed ps;
  const int rc = parse_header(data, n, ps);
  if (rc != PREMVOS_OK) return rc;
  *info = ps.info;
  if (!coef) return PREMVOS_OK;                    // header only
  const premvos_jpeg_info& I = ps.info;
  PV_REQUIRE(coef_capacity >= I.coef_count, "jpeg: coefficient buffer holds %lld values, the frame needs %lld",
             (long long)coef_capacity, (long long)I.coef_count);
  memset(coef, 0, sizeof(int16_t) * I.coef_count);
  BitReader br;
  br.p = ps.scan;
  br.end = data + n;
  int pred[3] = {0, 0, 0};
  const int nmcu = I.mcux * I.mcuy;
  for (int mcu = 0; mcu < nmcu; ++mcu) {
    if (ps.restart_interval && mcu && mcu % ps.restart_interval == 0) {
      PV_REQUIRE(!br.overran(), "jpeg: entropy-coded data ends early (before MCU %d)", mcu);
      PV_REQUIRE(br.restart(), "jpeg: restart marker missing at MCU %d", mcu);
      pred[0] = pred[1] = pred[2] = 0;
    }
    const int my = mcu / I.mcux, mx = mcu - my * I.mcux;
    for (int c = 0; c < I.ncomp; ++c) {
      const int ch = c == 0 ? I.hs : 1, cv = c == 0 ? I.vs : 1;
      const Huff& dc = ps.huff[0][ps.dc_sel[c]];
      const Huff& ac = ps.huff[1][ps.ac_sel[c]];
      for (int by = 0; by < cv; ++by)
        for (int bx = 0; bx < ch; ++bx) {
          int16_t* blk = coef + I.coef_offset[c] + ((int64_t)(my * cv + by) * I.blocks_w[c] + mx * ch + bx) * 64;
          PV_REQUIRE(decode_block(br, dc, ac, pred[c], blk), "jpeg: corrupt entropy-coded data at MCU %d", mcu);
        }
    }
  }
  PV_REQUIRE(!br.overran(), "jpeg: entropy-coded data ends early (truncated file?)");
  return PREMVOS_OK;
}

extern "C" int64_t premvos_jpeg_workspace_bytes(const premvos_jpeg_info* info) {
  if (!info) return 0;
  int64_t b = 0;
  for (int c = 0; c < info->ncomp && c < 3; ++c) b += (int64_t)info->blocks_w[c] * info->blocks_h[c] * 64;
  return b;
}

extern "C" int premvos_jpeg_reconstruct_u8(const int16_t* coef, const premvos_jpeg_info* info, void* workspace, uint8_t* out,
                                           int32_t bgr, void* stream) {
  PV_REQUIRE(coef && info && workspace && out, "jpeg: null argument");
  PV_REQUIRE(premvos::aligned16(coef) && (reinterpret_cast<uintptr_t>(workspace) & 7u) == 0,
             "jpeg: coefficient buffer must be 16-byte aligned, the workspace 8-byte aligned");
  PlaneDesc pd;
  Quant qt;
  const int rc = plane_desc(info, pd, qt);
  if (rc != PREMVOS_OK) return rc;
  for (int c = 0; c < info->ncomp; ++c) PV_REQUIRE(pd.coef_offset[c] % 8 == 0, "jpeg: coefficient offsets must be block aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  uint8_t* planes = static_cast<uint8_t*>(workspace);
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3(premvos::cdiv(pd.first_block[3], 128)), dim3(128), 0, s, coef, qt, pd, planes);
  const long npix = (long)info->height * info->width;
  const dim3 grid((unsigned)((npix + 255) / 256)), block(256);
  if (info->ncomp == 1) {
    hipLaunchKernelGGL((jpeg_color_kernel<1, 1, true>), grid, block, 0, s, planes, pd, info->height, info->width, bgr, out);
  } else if (info->hs == 1) {
    hipLaunchKernelGGL((jpeg_color_kernel<1, 1, false>), grid, block, 0, s, planes, pd, info->height, info->width, bgr, out);
  } else if (info->vs == 1) {
    hipLaunchKernelGGL((jpeg_color_kernel<2, 1, false>), grid, block, 0, s, planes, pd, info->height, info->width, bgr, out);
  } else {
    hipLaunchKernelGGL((jpeg_color_kernel<2, 2, false>), grid, block, 0, s, planes, pd, info->height, info->width, bgr, out);
  }
  return premvos::check_launch("jpeg_reconstruct");
}
