// Host-only translation unit (no kernels): the files of ONE frame -- <frame>.flo, general / specific / combined / refined proposal
// JSON -- written from the arrays a rank's results consist of, in one call that never touches the Python interpreter.
//
// Why it exists: the merge rank of a gathered 8-GPU job turns ~430 frames/s of gathered buffers into files
// (premvos_amd/stream.py: DeviceGather) while its own stage threads drive ~1500 kernel launches per chunk from Python.  Per frame
// the host work is small (3 ms) but two thirds of it held the interpreter lock (numpy scalar arithmetic of the box conversion, the
// JSON encoder, string slicing): four writer threads scaled NEGATIVELY (tools/dev/ingest_host_bench.py: 3.3 ms of writer time per
// frame with one thread, 15.9 ms with four) and slowed the rank's own launches by 15 %.  ctypes releases the lock for the duration
// of a foreign call, so this function runs on N threads at once.
//
// The bytes are the ones the Python writers produce -- and those are the reference's:
//   .flo           script_pwc_multi.py:16-31                      (tag 202021.25f, int32 W, int32 H, H*W*2 float32)
//   proposal JSON  proposal_net/eval.py:93-94 + train.py:388-428  boxes / scale and clip in float32, xywh, round(x, 1) / round(s, 2)
//                  on numpy float32 scalars (= multiply, rint, divide in float32), float(...), json.dump: repr of the double
//   combined       combine_general_and_specific.py:33             (general + specific)
//   refined        FewShotSegmentationForwarder.py:137-155        + "segmentation": {"size": [h, w], "counts": str}, "conf_score": str(float32)
// Number formatting: Python's float repr = the shortest decimal string that reads back as the same double (exponent form below 1e-4
// and from 1e16); numpy's str(float32) = the shortest that reads back as the same float (same thresholds, "1.0" / "1e-05" shapes).
// Both are produced here by searching the digit count with the C library's correctly rounded printf / strtod; tests/test_cpu_host_files.py
// compares a few hundred thousand values and whole frames against the interpreter.
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/uio.h>
#include <unistd.h>

#include <string>

#include "common.h"

namespace {

// Shortest digits d1 d2 ... dn (no trailing zeros) and decimal point position `decpt` (value = 0.d1d2...dn x 10^decpt) of a finite,
// non-zero |x|; SINGLE: shortest that identifies the float, else the double.
template <bool SINGLE>
int shortest_digits(double x, char* digits, int* decpt) {
  const int maxp = SINGLE ? 9 : 17;
  char buf[40];
  auto ok = [&](int p) {
    snprintf(buf, sizeof(buf), "%.*e", p - 1, x);
    return SINGLE ? strtof(buf, nullptr) == (float)x : strtod(buf, nullptr) == x;
  };
  int lo = 1, hi = maxp;                     // smallest p that round-trips (round-tripping is monotone in p)
  // (the doubles of this file are float32 values widened: nearly all need 16 or 17 digits -- probe the top first)
  if (!SINGLE) {
    if (!ok(15)) lo = 16; else hi = 15;
  }
  while (lo < hi) {
    const int mid = (lo + hi) / 2;
    if (ok(mid)) hi = mid; else lo = mid + 1;
  }
  snprintf(buf, sizeof(buf), "%.*e", lo - 1, x);        // d.ddddde[+-]XX
  int n = 0;
  const char* p = buf;
  if (*p == '-') ++p;
  for (; *p && *p != 'e'; ++p)
    if (*p >= '0' && *p <= '9') digits[n++] = *p;
  const int e10 = atoi(p + 1);
  while (n > 1 && digits[n - 1] == '0') --n;
  digits[n] = 0;
  *decpt = e10 + 1;
  return n;
}

// Python: repr(float) -- what json.dumps writes for a float
void py_repr_double(double x, std::string& out) {
  if (isnan(x)) { out += "NaN"; return; }
  if (isinf(x)) { out += x < 0 ? "-Infinity" : "Infinity"; return; }
  if (x == 0.0) { out += signbit(x) ? "-0.0" : "0.0"; return; }
  if (x < 0) { out += '-'; x = -x; }
  char d[24];
  int decpt;
  const int n = shortest_digits<false>(x, d, &decpt);
  if (decpt <= -4 || decpt > 16) {             // exponent form: d[.ddd]e[+-]XX (at least two exponent digits)
    out += d[0];
    if (n > 1) { out += '.'; out.append(d + 1, n - 1); }
    char e[8];
    snprintf(e, sizeof(e), "e%c%02d", decpt - 1 < 0 ? '-' : '+', abs(decpt - 1));
    out += e;
  } else if (decpt <= 0) {
    out += "0.";
    out.append(-decpt, '0');
    out.append(d, n);
  } else if (decpt >= n) {
    out.append(d, n);
    out.append(decpt - n, '0');
    out += ".0";
  } else {
    out.append(d, decpt);
    out += '.';
    out.append(d + decpt, n - decpt);
  }
}

// numpy: str(np.float32(x)) (Dragon4, unique digits; positional for 1e-4 <= |x| < 1e16 with at least one fractional digit, else
// scientific with the mantissa's trailing zeros and point trimmed and at least two exponent digits)
void np_str_float32(float xf, std::string& out) {
  double x = xf;
  if (isnan(x)) { out += "nan"; return; }
  if (isinf(x)) { out += x < 0 ? "-inf" : "inf"; return; }
  if (x == 0.0) { out += signbit(x) ? "-0.0" : "0.0"; return; }
  if (x < 0) { out += '-'; x = -x; }
  char d[16];
  int decpt;
  const int n = shortest_digits<true>(x, d, &decpt);
  if (x >= 1e16 || x < 1e-4) {
    out += d[0];
    if (n > 1) { out += '.'; out.append(d + 1, n - 1); }
    char e[8];
    snprintf(e, sizeof(e), "e%c%02d", decpt - 1 < 0 ? '-' : '+', abs(decpt - 1));
    out += e;
  } else if (decpt <= 0) {
    out += "0.";
    out.append(-decpt, '0');
    out.append(d, n);
  } else if (decpt >= n) {
    out.append(d, n);
    out.append(decpt - n, '0');
    out += ".0";
  } else {
    out.append(d, decpt);
    out += '.';
    out.append(d + decpt, n - decpt);
  }
}

// one detection -> `"bbox": [x, y, w, h], "score": s` (without the braces: the refined file appends to it)
void item_text(const float* box_resized, float prob, float scale, int h, int w, std::string& out) {
  // eval.py:93-94: boxes / scale, clip (float32 array arithmetic); train.py:404-406: xywh; round(., 1), round(., 2) on float32 scalars
  volatile float x0 = box_resized[0] / scale, y0 = box_resized[1] / scale, x1 = box_resized[2] / scale, y1 = box_resized[3] / scale;
  float b[4];
  b[0] = x0 <= 0.f ? 0.f : x0;                 // proposal/driver.py clip_boxes: values <= 0 (a negative zero included) -> +0; NaN stays
  b[1] = y0 <= 0.f ? 0.f : y0;
  b[2] = x1 > (float)w ? (float)w : x1;
  b[3] = y1 > (float)h ? (float)h : y1;
  volatile float bw = b[2] - b[0], bh = b[3] - b[1];
  b[2] = bw;
  b[3] = bh;
  out += "\"bbox\": [";
  for (int k = 0; k < 4; ++k) {
    volatile float t = b[k] * 10.0f;           // (volatile: one rounding per operation, no contraction into an fma)
    volatile float r = rintf(t);
    volatile float q = r / 10.0f;
    if (k) out += ", ";
    py_repr_double((double)q, out);
  }
  out += "], \"score\": ";
  volatile float t = prob * 100.0f;
  volatile float r = rintf(t);
  volatile float q = r / 100.0f;
  py_repr_double((double)q, out);
}

int write_file(const char* path, const void* head, size_t head_n, const void* body, size_t body_n) {
  const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) return errno == ENOENT ? 1 : premvos::fail(PREMVOS_EINVAL, "write_frame_files: open %s: %s", path, strerror(errno));
  struct iovec iov[2] = {{const_cast<void*>(head), head_n}, {const_cast<void*>(body), body_n}};
  int first = head_n ? 0 : 1;
  size_t left = head_n + body_n;
  while (left > 0) {
    const ssize_t n = writev(fd, iov + first, 2 - first);
    if (n < 0) {
      if (errno == EINTR) continue;
      const int e = errno;
      close(fd);
      return premvos::fail(PREMVOS_EINVAL, "write_frame_files: write %s: %s", path, strerror(e));
    }
    left -= (size_t)n;
    size_t adv = (size_t)n;
    for (int k = first; k < 2 && adv > 0; ++k) {
      const size_t take = adv < iov[k].iov_len ? adv : iov[k].iov_len;
      iov[k].iov_base = static_cast<char*>(iov[k].iov_base) + take;
      iov[k].iov_len -= take;
      adv -= take;
      if (iov[k].iov_len == 0 && k == first) first = k + 1;
    }
  }
  if (close(fd) != 0) return premvos::fail(PREMVOS_EINVAL, "write_frame_files: close %s: %s", path, strerror(errno));
  return 0;
}

}  // namespace

extern "C" int premvos_format_floats_host(const double* values, int64_t n, int32_t as_float32_str, char* out, int64_t cap) {
  std::string s;
  for (int64_t i = 0; i < n; ++i) {
    if (as_float32_str) np_str_float32((float)values[i], s); else py_repr_double(values[i], s);
    s += '\n';
  }
  if ((int64_t)s.size() > cap) return premvos::fail(PREMVOS_EINVAL, "format_floats: output needs %ld bytes", (long)s.size());
  memcpy(out, s.data(), s.size());
  return (int)s.size();
}

extern "C" int premvos_write_frame_files_host(const premvos_frame_files* f) {
  PV_REQUIRE(f != nullptr && f->h > 0 && f->w > 0, "write_frame_files: bad arguments");
  PV_REQUIRE(f->count[0] >= 0 && f->count[1] >= 0 && f->count[0] <= 4096 && f->count[1] <= 4096, "write_frame_files: bad detection counts");
  int missing_dir = 0;
  // ---- <frame>.flo
  if (f->flo_path != nullptr) {
    PV_REQUIRE(f->flow != nullptr && f->flow_row_stride >= 2 * (int64_t)f->w, "write_frame_files: flow");
    struct { float tag; int32_t w, h; } head = {202021.25f, f->w, f->h};
    int rc;
    if (f->flow_row_stride == 2 * (int64_t)f->w) {
      rc = write_file(f->flo_path, &head, 12, f->flow, (size_t)f->h * f->w * 8);
    } else {                                 // a window of a wider block: gather the rows first
      std::string rows;
      rows.resize((size_t)f->h * f->w * 8);
      for (int y = 0; y < f->h; ++y) memcpy(&rows[(size_t)y * f->w * 8], f->flow + (int64_t)y * f->flow_row_stride, (size_t)f->w * 8);
      rc = write_file(f->flo_path, &head, 12, rows.data(), rows.size());
    }
    if (rc < 0) return rc;
    missing_dir |= rc;
  }
  // ---- the detections' texts, once
  const int n_all = f->count[0] + f->count[1];
  std::string items[2], one;
  std::string refined = "[";
  for (int which = 0; which < 2; ++which) {
    PV_REQUIRE(f->count[which] == 0 || (f->boxes[which] != nullptr && f->probs[which] != nullptr), "write_frame_files: detections");
    for (int i = 0; i < f->count[which]; ++i) {
      one.clear();
      item_text(f->boxes[which] + 4 * i, f->probs[which][i], f->scale, f->h, f->w, one);
      if (i) items[which] += ", ";
      items[which] += '{';
      items[which] += one;
      items[which] += '}';
      if (f->json_path[3] != nullptr) {
        const int slot = (which ? f->count[0] : 0) + i;
        if (slot) refined += ", ";
        refined += '{';
        refined += one;
        // "segmentation": {"size": [h, w], "counts": "<COCO string>"}, "conf_score": "<str(float32)>"
        PV_REQUIRE(f->rle_pool != nullptr && f->rle_offsets != nullptr && f->conf != nullptr, "write_frame_files: refined file without masks / conf");
        char sz[64];
        snprintf(sz, sizeof(sz), ", \"segmentation\": {\"size\": [%d, %d], \"counts\": \"", f->h, f->w);
        refined += sz;
        const int32_t* e = f->rle_pool + f->rle_offsets[slot];
        const int64_t m = (int64_t)f->rle_offsets[slot + 1] - f->rle_offsets[slot];
        const long long hw = (long long)f->h * f->w;
        long long prev_edge = 0, c1 = 0, c2 = 0;
        for (int64_t k = 0; k <= m; ++k) {           // maskApi.c rleToString on the successive differences of [0, edges..., hw]
          const long long edge = k < m ? (long long)e[k] : hw;
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
            c = (char)(c + 48);
            if (c == '\\') refined += '\\';          // the one character of the alphabet (48 .. 111) JSON escapes
            refined += c;
          }
        }
        refined += "\"}, \"conf_score\": \"";
        np_str_float32(f->conf[slot], refined);
        refined += "\"}";
      }
    }
  }
  refined += ']';
  for (int which = 0; which < 2; ++which)
    if (f->json_path[which] != nullptr) {
      const std::string text = "[" + items[which] + "]";
      const int rc = write_file(f->json_path[which], nullptr, 0, text.data(), text.size());
      if (rc < 0) return rc;
      missing_dir |= rc;
    }
  if (f->json_path[2] != nullptr) {
    std::string text = "[" + items[0];
    if (f->count[0] && f->count[1]) text += ", ";
    text += items[1];
    text += ']';
    const int rc = write_file(f->json_path[2], nullptr, 0, text.data(), text.size());
    if (rc < 0) return rc;
    missing_dir |= rc;
  }
  if (f->json_path[3] != nullptr) {
    (void)n_all;
    const int rc = write_file(f->json_path[3], nullptr, 0, refined.data(), refined.size());
    if (rc < 0) return rc;
    missing_dir |= rc;
  }
  return missing_dir;       // 1: a directory does not exist (nothing else failed): the caller creates it and calls again
}
