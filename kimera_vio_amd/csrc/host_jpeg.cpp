// Baseline JPEG -> 8-bit grey for the input side, host code behind include/kvfe.h:
//   kvfe_jpeg_info / kvfe_jpeg_decode_gray == UtilsOpenCV::ReadAndConvertToGrayScale (src/utils/UtilsOpenCV.cpp:390-399)
//   for JPEG files: cv::imread(IMREAD_ANYCOLOR) [libjpeg with its defaults: JDCT_ISLOW, fancy upsampling,
//   YCbCr -> RGB] + cv::cvtColor(BGR2GRAY) when the file has three components.
// The reference's own front-end test frames (tests/data/ForStereoTracker/*.jpg, testStereoVisionImuFrontend.cpp:
// 674-926) are baseline 4:2:0 JFIF files.  libjpeg is not in this image, so its published decoding pipeline is
// restated: Huffman entropy decoding (ITU T.81 F.2), the slow-but-accurate integer IDCT (jidctint.c: Loeffler-
// Ligtenberg-Moschytz, 13-bit constants, two passes), "fancy" triangle-filter upsampling of the chroma planes for
// h2v1 and h2v2 (jdsample.c), the fixed-point YCbCr -> RGB tables (jdcolor.c), range limiting -- integer arithmetic
// throughout, so the result is bit-identical to libjpeg / libjpeg-turbo (pinned against PIL, which decodes with them).
// Supported: 8-bit baseline / extended-sequential Huffman (SOF0 / SOF1), 1 or 3 components, luma sampling 1x1, 2x1,
// 2x2 with 1x1 chroma, restart intervals.  Progressive, arithmetic-coded, 12-bit, CMYK and other sampling layouts:
// KVFE_ERR_UNSUPPORTED.  EXIF orientation is not applied.
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/kvfe.h"

namespace {

struct Huff {
  // canonical code tables (T.81 annex C / F.2.2.3): for code length l, mincode / maxcode / valptr
  int mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  int nvals = 0;
  bool present = false;
  uint8_t look_len[512];   // 9-bit look-ahead: code length (0 = longer than 9 bits)
  uint8_t look_val[512];
};

struct Comp {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int wblocks = 0, hblocks = 0;          // allocated blocks (MCU padded)
  int dw = 0, dh = 0;                    // downsampled_width / height (real samples)
  std::vector<uint8_t> plane;            // wblocks*8 x hblocks*8 samples after the IDCT
  int pred = 0;
};

struct Reader {
  const uint8_t* d;
  size_t n, pos;
  uint32_t bits = 0;
  int nbits = 0;
  bool hit_marker = false;
  int marker = 0;
  // entropy-coded segment bit reader with FF00 unstuffing; a marker ends the data (zeros are fed, like libjpeg)
  inline void fill() {
    while (nbits <= 24) {
      int c = 0;
      if (!hit_marker) {
        if (pos < n) {
          c = d[pos++];
          if (c == 0xFF) {
            int c2;
            do {
              c2 = pos < n ? d[pos++] : 0xD9;
            } while (c2 == 0xFF);
            if (c2 == 0) {
              c = 0xFF;
            } else {
              hit_marker = true;
              marker = c2;
              c = 0;
            }
          }
        } else {
          hit_marker = true;
          marker = 0xD9;
        }
      }
      bits |= (uint32_t)c << (24 - nbits);
      nbits += 8;
    }
  }
  inline int peek(int k) {
    if (nbits < k) fill();
    return (int)(bits >> (32 - k));
  }
  inline void skip(int k) {
    bits <<= k;
    nbits -= k;
  }
  inline int get(int k) {
    if (k == 0) return 0;
    const int v = peek(k);
    skip(k);
    return v;
  }
  void reset_bits() {
    bits = 0;
    nbits = 0;
  }
};

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

bool build_huff(Huff& H, const uint8_t counts[16]) {
  int code = 0, k = 0;
  for (int l = 1; l <= 16; l++) {
    H.valptr[l] = k;
    H.mincode[l] = code;
    k += counts[l - 1];
    code += counts[l - 1];
    H.maxcode[l] = counts[l - 1] ? code - 1 : -1;
    if (code > (1 << l)) return false;   // over-subscribed
    code <<= 1;
  }
  H.maxcode[17] = 0x7fffffff;
  std::memset(H.look_len, 0, sizeof(H.look_len));
  int c = 0, p = 0;
  for (int l = 1; l <= 9; l++) {
    for (int i = 0; i < counts[l - 1]; i++, p++) {
      const int first = c << (9 - l);
      for (int j = 0; j < (1 << (9 - l)); j++) {
        H.look_len[first + j] = (uint8_t)l;
        H.look_val[first + j] = H.vals[p];
      }
      c++;
    }
    c <<= 1;
  }
  H.present = true;
  return true;
}

inline int decode_sym(Reader& R, const Huff& H) {
  const int look = R.peek(9);
  const int l9 = H.look_len[look];
  if (l9) {
    R.skip(l9);
    return H.look_val[look];
  }
  int code = R.peek(16), l = 10;
  for (; l <= 16; l++) {
    const int c = code >> (16 - l);
    if (H.maxcode[l] >= 0 && c <= H.maxcode[l] && c >= H.mincode[l]) {
      R.skip(l);
      const int idx = H.valptr[l] + c - H.mincode[l];
      return idx < H.nvals ? H.vals[idx] : 0;
    }
  }
  R.skip(16);   // corrupt data: libjpeg warns and returns 0
  return 0;
}

const int ZIGZAG[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                             63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};   // (run overflow guard)

// post-IDCT range limit: sample_range_limit + CENTERJSAMPLE indexed with x & 1023 (jdmaster.c prepare_range_limit_table)
inline uint8_t idct_limit(int x) {
  const int i = x & 1023;
  if (i < 128) return (uint8_t)(i + 128);
  if (i < 512) return 255;
  if (i < 896) return 0;
  return (uint8_t)(i - 896);
}
inline uint8_t clamp255(int x) { return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); }

// jpeg_idct_islow (jidctint.c), coefficients already dequantised; out: 8 rows of 8 samples at `stride`
void idct_islow(const int* in, uint8_t* out, size_t stride) {
  constexpr int CONST_BITS = 13, PASS1_BITS = 2;
  constexpr int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
                F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
  auto descale = [](long x, int n) { return (int)((x + (1L << (n - 1))) >> n); };
  int ws[64];
  for (int c = 0; c < 8; c++) {
    const int* ip = in + c;
    int* wp = ws + c;
    if (ip[8] == 0 && ip[16] == 0 && ip[24] == 0 && ip[32] == 0 && ip[40] == 0 && ip[48] == 0 && ip[56] == 0) {
      const int dc = ip[0] * (1 << PASS1_BITS);
      for (int r = 0; r < 8; r++) wp[8 * r] = dc;
      continue;
    }
    long z2 = ip[16], z3 = ip[48];
    long z1 = (z2 + z3) * F_0_541;
    long tmp2 = z1 + z3 * (-F_1_847);
    long tmp3 = z1 + z2 * F_0_765;
    z2 = ip[0];
    z3 = ip[32];
    long tmp0 = (z2 + z3) * (1L << CONST_BITS);
    long tmp1 = (z2 - z3) * (1L << CONST_BITS);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = ip[56];
    tmp1 = ip[40];
    tmp2 = ip[24];
    tmp3 = ip[8];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F_1_175;
    tmp0 *= F_0_298;
    tmp1 *= F_2_053;
    tmp2 *= F_3_072;
    tmp3 *= F_1_501;
    z1 *= -F_0_899;
    z2 *= -F_2_562;
    z3 *= -F_1_961;
    z4 *= -F_0_390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    wp[0] = descale(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
    wp[56] = descale(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
    wp[8] = descale(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
    wp[48] = descale(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
    wp[16] = descale(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
    wp[40] = descale(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
    wp[24] = descale(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
    wp[32] = descale(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
  }
  for (int r = 0; r < 8; r++) {
    const int* wp = ws + 8 * r;
    uint8_t* o = out + (size_t)r * stride;
    if (wp[1] == 0 && wp[2] == 0 && wp[3] == 0 && wp[4] == 0 && wp[5] == 0 && wp[6] == 0 && wp[7] == 0) {
      const uint8_t dc = idct_limit(descale(wp[0], PASS1_BITS + 3));
      for (int c = 0; c < 8; c++) o[c] = dc;
      continue;
    }
    long z2 = wp[2], z3 = wp[6];
    long z1 = (z2 + z3) * F_0_541;
    long tmp2 = z1 + z3 * (-F_1_847);
    long tmp3 = z1 + z2 * F_0_765;
    long tmp0 = ((long)wp[0] + (long)wp[4]) * (1L << CONST_BITS);
    long tmp1 = ((long)wp[0] - (long)wp[4]) * (1L << CONST_BITS);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = wp[7];
    tmp1 = wp[5];
    tmp2 = wp[3];
    tmp3 = wp[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F_1_175;
    tmp0 *= F_0_298;
    tmp1 *= F_2_053;
    tmp2 *= F_3_072;
    tmp3 *= F_1_501;
    z1 *= -F_0_899;
    z2 *= -F_2_562;
    z3 *= -F_1_961;
    z4 *= -F_0_390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    constexpr int S = CONST_BITS + PASS1_BITS + 3;
    o[0] = idct_limit(descale(tmp10 + tmp3, S));
    o[7] = idct_limit(descale(tmp10 - tmp3, S));
    o[1] = idct_limit(descale(tmp11 + tmp2, S));
    o[6] = idct_limit(descale(tmp11 - tmp2, S));
    o[2] = idct_limit(descale(tmp12 + tmp1, S));
    o[5] = idct_limit(descale(tmp12 - tmp1, S));
    o[3] = idct_limit(descale(tmp13 + tmp0, S));
    o[4] = idct_limit(descale(tmp13 - tmp0, S));
  }
}

struct Jpeg {
  int w = 0, h = 0, ncomp = 0;
  Comp comp[3];
  uint16_t qt[4][64];
  bool qt_present[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  int restart_interval = 0;
  int hmax = 1, vmax = 1;
  int adobe_transform = -1;
  bool jfif = false;
  bool sof_seen = false;
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// header scan up to (and including) the first SOF; `full` continues to SOS and returns the offset of the scan data
kvfe_status parse(const uint8_t* d, size_t n, Jpeg& J, bool full, size_t* scan_at, int scan_comp[3], int* ns) {
  if (!d || n < 4 || d[0] != 0xFF || d[1] != 0xD8) return KVFE_ERR_INVALID_ARG;
  size_t pos = 2;
  for (;;) {
    if (pos + 4 > n) return KVFE_ERR_INVALID_ARG;
    if (d[pos] != 0xFF) return KVFE_ERR_INVALID_ARG;
    while (pos < n && d[pos] == 0xFF) pos++;
    if (pos >= n) return KVFE_ERR_INVALID_ARG;
    const int m = d[pos++];
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (m == 0xD9) return KVFE_ERR_INVALID_ARG;
    if (pos + 2 > n) return KVFE_ERR_INVALID_ARG;
    const int len = be16(d + pos);
    if (len < 2 || pos + (size_t)len > n) return KVFE_ERR_INVALID_ARG;
    const uint8_t* p = d + pos + 2;
    const int body = len - 2;
    if (m == 0xC0 || m == 0xC1) {
      if (J.sof_seen || body < 6) return KVFE_ERR_INVALID_ARG;
      if (p[0] != 8) return KVFE_ERR_UNSUPPORTED;
      J.h = be16(p + 1);
      J.w = be16(p + 3);
      J.ncomp = p[5];
      if (J.w <= 0 || J.h <= 0) return KVFE_ERR_INVALID_ARG;
      if (J.ncomp != 1 && J.ncomp != 3) return KVFE_ERR_UNSUPPORTED;
      if (body < 6 + 3 * J.ncomp) return KVFE_ERR_INVALID_ARG;
      for (int i = 0; i < J.ncomp; i++) {
        Comp& c = J.comp[i];
        c.id = p[6 + 3 * i];
        c.h = p[7 + 3 * i] >> 4;
        c.v = p[7 + 3 * i] & 15;
        c.tq = p[8 + 3 * i];
        if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return KVFE_ERR_UNSUPPORTED;
        J.hmax = c.h > J.hmax ? c.h : J.hmax;
        J.vmax = c.v > J.vmax ? c.v : J.vmax;
      }
      if (J.ncomp == 1) {   // a single component is never interleaved: its sampling factors do not matter
        J.comp[0].h = J.comp[0].v = 1;
        J.hmax = J.vmax = 1;
      } else {
        if (J.comp[1].h != 1 || J.comp[1].v != 1 || J.comp[2].h != 1 || J.comp[2].v != 1) return KVFE_ERR_UNSUPPORTED;
        if (J.comp[0].h == 1 && J.comp[0].v == 2) return KVFE_ERR_UNSUPPORTED;   // h1v2
      }
      J.sof_seen = true;
      if (!full) return KVFE_OK;
    } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      return KVFE_ERR_UNSUPPORTED;   // progressive, lossless, arithmetic, hierarchical
    } else if (m == 0xCC) {
      return KVFE_ERR_UNSUPPORTED;
    } else if (m == 0xDB) {
      int o = 0;
      while (o < body) {
        const int pq = p[o] >> 4, tq = p[o] & 15;
        if (tq > 3 || pq > 1) return KVFE_ERR_INVALID_ARG;
        if (o + 1 + (pq ? 128 : 64) > body) return KVFE_ERR_INVALID_ARG;
        for (int i = 0; i < 64; i++) J.qt[tq][ZIGZAG[i]] = pq ? (uint16_t)be16(p + o + 1 + 2 * i) : p[o + 1 + i];
        J.qt_present[tq] = true;
        o += 1 + (pq ? 128 : 64);
      }
    } else if (m == 0xC4) {
      int o = 0;
      while (o < body) {
        if (o + 17 > body) return KVFE_ERR_INVALID_ARG;
        const int tc = p[o] >> 4, th = p[o] & 15;
        if (tc > 1 || th > 3) return KVFE_ERR_INVALID_ARG;
        int total = 0;
        for (int i = 0; i < 16; i++) total += p[o + 1 + i];
        if (total > 256 || o + 17 + total > body) return KVFE_ERR_INVALID_ARG;
        Huff& H = tc ? J.ac[th] : J.dc[th];
        std::memcpy(H.vals, p + o + 17, total);
        H.nvals = total;
        if (!build_huff(H, p + o + 1)) return KVFE_ERR_INVALID_ARG;
        o += 17 + total;
      }
    } else if (m == 0xDD) {
      if (body < 2) return KVFE_ERR_INVALID_ARG;
      J.restart_interval = be16(p);
    } else if (m == 0xE0) {
      if (body >= 5 && !std::memcmp(p, "JFIF", 5)) J.jfif = true;
    } else if (m == 0xEE) {
      if (body >= 12 && !std::memcmp(p, "Adobe", 5)) J.adobe_transform = p[11];
    } else if (m == 0xDA) {
      if (!J.sof_seen || body < 1) return KVFE_ERR_INVALID_ARG;
      const int k = p[0];
      if (k < 1 || k > J.ncomp || body < 1 + 2 * k + 3) return KVFE_ERR_INVALID_ARG;
      for (int i = 0; i < k; i++) {
        int ci = -1;
        for (int j = 0; j < J.ncomp; j++)
          if (J.comp[j].id == p[1 + 2 * i]) ci = j;
        if (ci < 0) return KVFE_ERR_INVALID_ARG;
        J.comp[ci].td = p[2 + 2 * i] >> 4;
        J.comp[ci].ta = p[2 + 2 * i] & 15;
        if (J.comp[ci].td > 3 || J.comp[ci].ta > 3) return KVFE_ERR_INVALID_ARG;
        scan_comp[i] = ci;
      }
      *ns = k;
      *scan_at = pos + (size_t)len;
      return KVFE_OK;
    }
    pos += (size_t)len;
  }
}

// one 8x8 block: Huffman decode, dequantise, IDCT into the component plane
bool decode_block(Reader& R, const Jpeg& J, Comp& c, int bx, int by) {
  const Huff& HD = J.dc[c.td];
  const Huff& HA = J.ac[c.ta];
  if (!HD.present || !HA.present || !J.qt_present[c.tq]) return false;
  int coef[64];
  std::memset(coef, 0, sizeof(coef));
  const int s = decode_sym(R, HD);
  int diff = 0;
  if (s) {
    if (s > 15) return false;
    diff = extend(R.get(s), s);
  }
  // (crafted streams: keep the DC prediction and the dequantised coefficient inside int without overflow; a
  // conforming stream never leaves 16 bits here)
  const long long pred = (long long)c.pred + diff;
  c.pred = (int)std::max<long long>(-(1LL << 20), std::min<long long>(1LL << 20, pred));
  const uint16_t* q = J.qt[c.tq];
  coef[0] = (int)std::max<long long>(-(1LL << 30), std::min<long long>(1LL << 30, (long long)c.pred * q[0]));
  for (int k = 1; k < 64;) {
    const int rs = decode_sym(R, HA);
    const int r = rs >> 4, sz = rs & 15;
    if (sz) {
      k += r;
      const int v = extend(R.get(sz), sz);
      const int z = ZIGZAG[k < 80 ? k : 79];   // (a run past the block lands on coefficient 63, as in libjpeg)
      coef[z] = v * q[z];
      k++;
    } else {
      if (r != 15) break;
      k += 16;
    }
  }
  if (bx < c.wblocks && by < c.hblocks)
    idct_islow(coef, c.plane.data() + ((size_t)by * 8) * ((size_t)c.wblocks * 8) + (size_t)bx * 8, (size_t)c.wblocks * 8);
  return true;
}

kvfe_status decode(const uint8_t* d, size_t n, uint8_t* dst, size_t dst_stride, int32_t width, int32_t height) {
  Jpeg J;
  size_t scan_at = 0;
  int scan_comp[3] = {0, 1, 2}, ns = 0;
  kvfe_status st = parse(d, n, J, true, &scan_at, scan_comp, &ns);
  if (st != KVFE_OK) return st;
  if (!dst || J.w != width || J.h != height || dst_stride < (size_t)width) return KVFE_ERR_INVALID_ARG;
  if (ns != J.ncomp) return KVFE_ERR_UNSUPPORTED;   // one interleaved scan (what baseline encoders write)
  const int mcu_w = 8 * J.hmax, mcu_h = 8 * J.vmax;
  const int mcux = (J.w + mcu_w - 1) / mcu_w, mcuy = (J.h + mcu_h - 1) / mcu_h;
  for (int i = 0; i < J.ncomp; i++) {
    Comp& c = J.comp[i];
    c.wblocks = mcux * c.h;
    c.hblocks = mcuy * c.v;
    c.dw = (J.w * c.h + J.hmax - 1) / J.hmax;
    c.dh = (J.h * c.v + J.vmax - 1) / J.vmax;
    c.plane.assign((size_t)c.wblocks * 8 * c.hblocks * 8, 0);
    c.pred = 0;
  }
  Reader R{d, n, scan_at};
  int restarts_left = J.restart_interval, next_rst = 0;
  for (int my = 0; my < mcuy; my++)
    for (int mx = 0; mx < mcux; mx++) {
      if (J.restart_interval && restarts_left == 0) {
        // process_restart (jdhuff.c): drop the padding bits, take the RSTn marker, reset the DC predictions
        R.reset_bits();
        if (!R.hit_marker && R.pos + 1 < R.n && R.d[R.pos] == 0xFF && R.d[R.pos + 1] >= 0xD0 && R.d[R.pos + 1] <= 0xD7) {
          R.pos += 2;
        } else if (R.hit_marker && R.marker >= 0xD0 && R.marker <= 0xD7) {
          R.hit_marker = false;
        }   // anything else: damaged data -- the reader keeps feeding zeros from the marker it stopped at
        next_rst = (next_rst + 1) & 7;
        restarts_left = J.restart_interval;
        for (int i = 0; i < J.ncomp; i++) J.comp[i].pred = 0;
      }
      for (int s = 0; s < ns; s++) {
        Comp& c = J.comp[scan_comp[s]];
        for (int by = 0; by < c.v; by++)
          for (int bx = 0; bx < c.h; bx++)
            if (!decode_block(R, J, c, mx * c.h + bx, my * c.v + by)) return KVFE_ERR_INVALID_ARG;
      }
      if (J.restart_interval) restarts_left--;
    }
  // ---- to grey ---------------------------------------------------------------------------------------------
  const Comp& Y = J.comp[0];
  const size_t ys = (size_t)Y.wblocks * 8;
  if (J.ncomp == 1) {
    for (int y = 0; y < J.h; y++) std::memcpy(dst + (size_t)y * dst_stride, Y.plane.data() + (size_t)y * ys, (size_t)J.w);
    return KVFE_OK;
  }
  // chroma upsampled to full resolution row by row (jdsample.c: fullsize / h2v1_fancy / h2v2_fancy)
  const bool rgb_direct = J.adobe_transform == 0 && !J.jfif;   // Adobe marker with transform 0: the components ARE R, G, B
  const int hs = J.hmax, vs = J.vmax;
  std::vector<uint8_t> up[2];
  up[0].resize((size_t)2 * J.comp[1].dw + 8 + (size_t)J.w);
  up[1].resize((size_t)2 * J.comp[2].dw + 8 + (size_t)J.w);
  // jdcolor.c build_ycc_rgb_table
  struct YccTables {
    int Cr_r[256], Cb_b[256];
    long Cr_g[256], Cb_g[256];
    YccTables() {
      for (int i = 0; i < 256; i++) {
        const long x = i - 128;
        Cr_r[i] = (int)((91881L * x + 32768L) >> 16);
        Cb_b[i] = (int)((116130L * x + 32768L) >> 16);
        Cr_g[i] = -46802L * x;
        Cb_g[i] = -22554L * x + 32768L;
      }
    }
  };
  static const YccTables ycc;   // (function-local static: initialised once, thread-safe)
  const int* Cr_r = ycc.Cr_r;
  const int* Cb_b = ycc.Cb_b;
  const long* Cr_g = ycc.Cr_g;
  const long* Cb_g = ycc.Cb_g;
  for (int y = 0; y < J.h; y++) {
    for (int k = 0; k < 2; k++) {
      const Comp& C = J.comp[1 + k];
      const size_t cs = (size_t)C.wblocks * 8;
      uint8_t* o = up[k].data();
      if (hs == 1 && vs == 1) {
        std::memcpy(o, C.plane.data() + (size_t)y * cs, (size_t)J.w);
      } else if (C.dw <= 2) {
        // jinit_upsampler picks the fancy routines only for downsampled_width > 2; narrower components are
        // replicated (h2v1_upsample / h2v2_upsample)
        const uint8_t* in = C.plane.data() + (size_t)(vs == 2 ? (y >> 1) : y) * cs;
        for (int x = 0; x < C.dw; x++) o[2 * x] = o[2 * x + 1] = in[x];
      } else if (hs == 2 && vs == 1) {   // h2v1_fancy_upsample
        const uint8_t* in = C.plane.data() + (size_t)y * cs;
        const int n = C.dw;
        const uint8_t* ip = in;
        uint8_t* op = o;
        int v = *ip++;
        *op++ = (uint8_t)v;
        *op++ = (uint8_t)((v * 3 + ip[0] + 2) >> 2);
        for (int c = n - 2; c > 0; c--) {
          v = (*ip++) * 3;
          *op++ = (uint8_t)((v + ip[-2] + 1) >> 2);
          *op++ = (uint8_t)((v + ip[0] + 2) >> 2);
        }
        v = *ip;
        *op++ = (uint8_t)((v * 3 + ip[-1] + 1) >> 2);
        *op++ = (uint8_t)v;
      } else {   // h2v2_fancy_upsample: nearer row weighs 3, the other (above for even, below for odd output rows) 1
        const int r = y >> 1;
        int r1 = (y & 1) ? r + 1 : r - 1;
        r1 = r1 < 0 ? 0 : (r1 > C.dh - 1 ? C.dh - 1 : r1);
        const uint8_t* in0 = C.plane.data() + (size_t)r * cs;
        const uint8_t* in1 = C.plane.data() + (size_t)r1 * cs;
        const int n = C.dw;
        const uint8_t *p0 = in0, *p1 = in1;
        uint8_t* op = o;
        int thiscol = (*p0++) * 3 + (*p1++);
        int nextcol = (*p0++) * 3 + (*p1++);
        *op++ = (uint8_t)((thiscol * 4 + 8) >> 4);
        *op++ = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
        int lastcol = thiscol;
        thiscol = nextcol;
        for (int c = n - 2; c > 0; c--) {
          nextcol = (*p0++) * 3 + (*p1++);
          *op++ = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
          *op++ = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
          lastcol = thiscol;
          thiscol = nextcol;
        }
        *op++ = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
        *op++ = (uint8_t)((thiscol * 4 + 7) >> 4);
      }
    }
    const uint8_t* yr = Y.plane.data() + (size_t)y * ys;
    uint8_t* o = dst + (size_t)y * dst_stride;
    for (int x = 0; x < J.w; x++) {
      int R8, G8, B8;
      if (rgb_direct) {
        R8 = yr[x];
        G8 = up[0][x];
        B8 = up[1][x];
      } else {   // ycc_rgb_convert
        const int yy = yr[x], cb = up[0][x], cr = up[1][x];
        R8 = clamp255(yy + Cr_r[cr]);
        G8 = clamp255(yy + (int)((Cb_g[cb] + Cr_g[cr]) >> 16));
        B8 = clamp255(yy + Cb_b[cb]);
      }
      // cv::cvtColor(BGR2GRAY), 8U (OpenCV 4: 15-bit weights)
      o[x] = (uint8_t)((B8 * 3735 + G8 * 19235 + R8 * 9798 + (1 << 14)) >> 15);
    }
  }
  return KVFE_OK;
}

}  // namespace

extern "C" {

kvfe_status kvfe_jpeg_info(const uint8_t* data, size_t size, int32_t* width, int32_t* height, int32_t* channels) {
  Jpeg J;
  size_t at = 0;
  int sc[3], ns = 0;
  const kvfe_status st = parse(data, size, J, false, &at, sc, &ns);
  if (st != KVFE_OK) return st;
  if (width) *width = J.w;
  if (height) *height = J.h;
  if (channels) *channels = J.ncomp;
  return KVFE_OK;
}

kvfe_status kvfe_jpeg_decode_gray(const uint8_t* data, size_t size, uint8_t* dst, size_t dst_stride, int32_t width,
                                  int32_t height) {
  try {
    return decode(data, size, dst, dst_stride, width, height);
  } catch (const std::bad_alloc&) {
    return KVFE_ERR_CAPACITY;
  } catch (...) {
    return KVFE_ERR_INVALID_ARG;
  }
}

}  // extern "C"
