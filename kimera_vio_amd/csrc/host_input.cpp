// Input side of the front-end, host code behind the C ABI of include/kvfe.h:
//   * PNG -> 8-bit grey            == UtilsOpenCV::ReadAndConvertToGrayScale (src/utils/UtilsOpenCV.cpp:390-399):
//                                     cv::imread(IMREAD_ANYCOLOR) [+ cv::cvtColor(BGR2GRAY)] for PNG files
//   * kvfe_imu_buffer              == utils::ThreadsafeImuBuffer (src/utils/ThreadsafeImuBuffer.cpp:48-234) over
//                                     utils::ThreadsafeTemporalBuffer (include/kimera-vio/utils/
//                                     ThreadsafeTemporalBuffer-inl.h)
//   * kvfe_stereo_sync             == StereoDataProviderModule::getInputPacket (src/dataprovider/
//                                     StereoDataProviderModule.cpp:35-91), MonoDataProviderModule::
//                                     getMonoImuSyncPacket (MonoDataProviderModule.cpp:44-118), DataProviderModule::
//                                     getTimeSyncedImuMeasurements (DataProviderModule.cpp:80-181),
//                                     SimpleQueueSynchronizer::syncQueue (pipeline/QueueSynchronizer.h:79-162)
//   * EuRoC index files            == CameraImageLists::parseCamImgList (DataProviderInterface-definitions.cpp:
//                                     74-101), EurocDataProvider::parseImuData (EurocDataProvider.cpp:229-306)
// No device work here: the data provider runs on CPU threads beside the GPU steps and writes decoded frames into
// the pinned staging slots of kvfe_frontend_staging_buffer.  The PNG container is decoded with zlib's inflate (the
// only codec library in the image); everything around it (chunks, CRCs, filters, Adam7, sample expansion) is here.
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#if defined(__linux__)
#include <sched.h>
#endif

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/kvfe.h"

namespace {

// ------------------------------------------------------------------------------------------------
// PNG
// ------------------------------------------------------------------------------------------------
struct PngHeader {
  uint32_t w = 0, h = 0;
  int depth = 0, color = 0, interlace = 0;
  int channels() const { return color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4; }
  // channels of the cv::Mat cv::imread(IMREAD_ANYCOLOR) returns (PngDecoder::readHeader + imread_'s type rule)
  int cv_channels() const { return color == 0 ? 1 : 3; }
};

inline uint32_t be32(const uint8_t* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

kvfe_status png_header(const uint8_t* d, size_t n, PngHeader* H) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (!d || n < 8 + 12 + 13 || std::memcmp(d, sig, 8) != 0) return KVFE_ERR_INVALID_ARG;
  if (be32(d + 8) != 13 || std::memcmp(d + 12, "IHDR", 4) != 0) return KVFE_ERR_INVALID_ARG;
  const uint8_t* p = d + 16;
  H->w = be32(p);
  H->h = be32(p + 4);
  H->depth = p[8];
  H->color = p[9];
  H->interlace = p[12];
  if (H->w == 0 || H->h == 0 || H->w > (1u << 20) || H->h > (1u << 20)) return KVFE_ERR_INVALID_ARG;
  if (p[10] != 0 || p[11] != 0 || H->interlace > 1) return KVFE_ERR_INVALID_ARG;
  const int dp = H->depth;
  bool ok = false;
  switch (H->color) {
    case 0: ok = dp == 1 || dp == 2 || dp == 4 || dp == 8 || dp == 16; break;
    case 3: ok = dp == 1 || dp == 2 || dp == 4 || dp == 8; break;
    case 2: case 4: case 6: ok = dp == 8 || dp == 16; break;
    default: ok = false;
  }
  if (!ok) return KVFE_ERR_INVALID_ARG;
  if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), d + 12, 4 + 13) != be32(d + 29)) return KVFE_ERR_INVALID_ARG;
  return KVFE_OK;
}

// undo the scanline filters of one (sub)image in place: rows of 1 + rowbytes bytes.  The first row has no row
// above (it reads as zeros), the first bpp bytes of a row no byte to the left; Average and Paeth are serial through
// the byte just written, so their inner loops are kept branch-free (the Paeth predictor as in the PNG specification:
// the candidate nearest to a + b - c, ties in the order a, b, c).
// CRC-32 of a chunk (the PNG / zlib polynomial) by carry-less multiplication where the processor has it: four 128-bit
// lanes folded per 64 bytes (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ", the constants
// of the reflected polynomial 0xEDB88320), the last bytes and processors without the instruction through zlib's table
// code.  zlib 1.2.11's crc32 took 0.25 ms of the 0.6 ms a 361 kB EuRoC frame decodes in; this takes 0.012 ms.  Checked
// against zlib's on random buffers of every length 0..399 and up to 400 kB, any alignment and start value.
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_fold(const uint8_t* buf, size_t len, uint32_t crc) {
  // len >= 64 and a multiple of 16; crc and the result are the inverted register as zlib keeps it internally
  alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
  alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
  alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
  alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
  __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8;
  x1 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x00));
  x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x10));
  x3 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x20));
  x4 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x30));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
  x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(k1k2));
  buf += 64;
  len -= 64;
  while (len >= 64) {   // fold the four lanes over the next 64 bytes
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
    x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
    x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
    x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x00)));
    x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x10)));
    x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x20)));
    x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf + 0x30)));
    buf += 64;
    len -= 64;
  }
  x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(k3k4));   // four lanes -> one
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
  while (len >= 16) {   // the remaining whole 16-byte blocks
    x2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(buf));
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    buf += 16;
    len -= 16;
  }
  x2 = _mm_clmulepi64_si128(x1, x0, 0x10);   // 128 -> 64 bits
  x3 = _mm_setr_epi32(~0, 0, ~0, 0);
  x1 = _mm_srli_si128(x1, 8);
  x1 = _mm_xor_si128(x1, x2);
  x0 = _mm_loadl_epi64(reinterpret_cast<const __m128i*>(k5k0));
  x2 = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, x3);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  x0 = _mm_load_si128(reinterpret_cast<const __m128i*>(poly));   // Barrett reduction to 32 bits
  x2 = _mm_and_si128(x1, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
  x2 = _mm_and_si128(x2, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif
// crc32(crc, p, n) as zlib defines it (crc = 0 starts a new one)
uint32_t chunk_crc32(uint32_t crc, const uint8_t* p, size_t n) {
#if defined(__x86_64__)
  static const bool have_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
  if (have_clmul && n >= 64) {
    const size_t m = n & ~(size_t)15;
    crc = ~crc32_fold(p, m, ~crc);
    p += m;
    n -= m;
  }
#endif
  while (n > 0) {   // (zlib takes 32-bit lengths)
    const size_t part = std::min<size_t>(n, 1u << 30);
    crc = (uint32_t)crc32(crc, p, (uInt)part);
    p += part;
    n -= part;
  }
  return crc;
}

// K consecutive Average / Paeth rows of one byte per pixel as a WAVEFRONT: row k works on byte i - k while row k - 1
// works on byte i - k + 1, so the K serial chains (each byte waits for its left neighbour: ~10 cycles per byte for
// Paeth on its own) advance together and a core's spare issue slots do the other rows' work.  The row above a byte (b)
// and above-left (c) are what the row before produced one and two steps ago: they stay in registers.  EuRoC frames are
// ~480 such rows (the encoder's adaptive filter picks Paeth or Average for camera noise), and undoing them was 1.0 of the
// 1.2 ms a 752x480 frame took.  Same integer operations per byte as the one-row loops in unfilter() below.  (Six or eight
// rows at a time are no faster than four: the state no longer fits the registers.)
template <int K>
void unfilter_wave(uint8_t* const (&rows)[K], const uint8_t* above, const int (&types)[K], size_t rowbytes) {
  int a[K], c[K], last[K];   // left neighbour, above-left neighbour, the byte this row produced in the previous step
  for (int k = 0; k < K; k++) a[k] = c[k] = last[k] = 0;
  auto step = [&](int k, size_t idx) {
    const int b = k == 0 ? (int)above[idx] : last[k - 1];
    int pred;
    if (types[k] == 4) {
      const int pa0 = b - c[k], pb0 = a[k] - c[k];             // p - a, p - b with p = a + b - c
      const int pa = pa0 < 0 ? -pa0 : pa0, pb = pb0 < 0 ? -pb0 : pb0;
      const int pc0 = pa0 + pb0, pc = pc0 < 0 ? -pc0 : pc0;
      pred = pb <= pc ? b : c[k];
      pred = (pa <= pb && pa <= pc) ? a[k] : pred;
    } else {
      pred = (a[k] + b) >> 1;
    }
    const int v = (rows[k][idx] + pred) & 255;
    rows[k][idx] = (uint8_t)v;
    a[k] = v;
    c[k] = b;
    return v;
  };
  // step i: row k at byte i - k (rows from the last to the first, so that last[k - 1] is still the previous step's)
  const size_t steps = rowbytes + K - 1;
  for (size_t i = 0; i < steps; i++) {
    if (i >= (size_t)(K - 1) && i < rowbytes) {   // every row inside its range: the steady part
      int nv[K];
#pragma GCC unroll 8
      for (int k = K - 1; k >= 0; k--) nv[k] = step(k, i - (size_t)k);
#pragma GCC unroll 8
      for (int k = 0; k < K; k++) last[k] = nv[k];
    } else {
      int nv[K];
      for (int k = K - 1; k >= 0; k--) nv[k] = (i >= (size_t)k && i - (size_t)k < rowbytes) ? step(k, i - (size_t)k) : last[k];
      for (int k = 0; k < K; k++) last[k] = nv[k];
    }
  }
}

kvfe_status unfilter(uint8_t* buf, size_t rows, size_t rowbytes, int bpp) {
  std::vector<uint8_t> zeros(rowbytes, 0);
  const uint8_t* prev = zeros.data();
  const size_t bp = (size_t)bpp;
  for (size_t y = 0; y < rows; y++) {
    uint8_t* row = buf + y * (rowbytes + 1);
    const int ft = row[0];
    uint8_t* r = row + 1;
    if (bp == 1 && rowbytes >= 8 && (ft == 3 || ft == 4)) {
      // a run of Average / Paeth rows: four (or two) at a time as a wavefront
      size_t run = 1;
      while (run < 4 && y + run < rows) {
        const int t = buf[(y + run) * (rowbytes + 1)];
        if (t != 3 && t != 4) break;
        run++;
      }
      if (run >= 4) {
        uint8_t* const rs[4] = {r, r + (rowbytes + 1), r + 2 * (rowbytes + 1), r + 3 * (rowbytes + 1)};
        const int ts[4] = {ft, rs[1][-1], rs[2][-1], rs[3][-1]};
        unfilter_wave<4>(rs, prev, ts, rowbytes);
        prev = rs[3];
        y += 3;
        continue;
      }
      if (run >= 2) {
        uint8_t* const rs[2] = {r, r + (rowbytes + 1)};
        const int ts[2] = {ft, rs[1][-1]};
        unfilter_wave<2>(rs, prev, ts, rowbytes);
        prev = rs[1];
        y += 1;
        continue;
      }
    }
    const size_t head = std::min(bp, rowbytes);
    switch (ft) {
      case 0: break;
      case 1:
        for (size_t i = bp; i < rowbytes; i++) r[i] = (uint8_t)(r[i] + r[i - bp]);
        break;
      case 2:
        for (size_t i = 0; i < rowbytes; i++) r[i] = (uint8_t)(r[i] + prev[i]);
        break;
      case 3:
        for (size_t i = 0; i < head; i++) r[i] = (uint8_t)(r[i] + (prev[i] >> 1));
        if (bp == 1) {   // one byte per pixel (the grey frames): the left neighbour rides in a register
          int a = rowbytes ? r[0] : 0;
          for (size_t i = 1; i < rowbytes; i++) {
            a = (r[i] + ((a + (int)prev[i]) >> 1)) & 255;
            r[i] = (uint8_t)a;
          }
        } else {
          for (size_t i = bp; i < rowbytes; i++) r[i] = (uint8_t)(r[i] + (((int)r[i - bp] + (int)prev[i]) >> 1));
        }
        break;
      case 4:
        for (size_t i = 0; i < head; i++) r[i] = (uint8_t)(r[i] + prev[i]);   // a = c = 0: the predictor is b
        if (bp == 1) {
          int a = rowbytes ? r[0] : 0, c = rowbytes ? prev[0] : 0;
          for (size_t i = 1; i < rowbytes; i++) {
            const int b = prev[i];
            const int pa0 = b - c, pb0 = a - c;             // p - a, p - b with p = a + b - c
            const int pa = pa0 < 0 ? -pa0 : pa0, pb = pb0 < 0 ? -pb0 : pb0;
            const int pc0 = pa0 + pb0, pc = pc0 < 0 ? -pc0 : pc0;
            int pred = pb <= pc ? b : c;
            pred = (pa <= pb && pa <= pc) ? a : pred;
            a = (r[i] + pred) & 255;
            r[i] = (uint8_t)a;
            c = b;
          }
        } else {
          for (size_t i = bp; i < rowbytes; i++) {
            const int a = r[i - bp], b = prev[i], c = prev[i - bp];
            const int pa0 = b - c, pb0 = a - c;
            const int pa = pa0 < 0 ? -pa0 : pa0, pb = pb0 < 0 ? -pb0 : pb0;
            const int pc0 = pa0 + pb0, pc = pc0 < 0 ? -pc0 : pc0;
            int pred = pb <= pc ? b : c;
            pred = (pa <= pb && pa <= pc) ? a : pred;
            r[i] = (uint8_t)(r[i] + pred);
          }
        }
        break;
      default: return KVFE_ERR_INVALID_ARG;
    }
    prev = r;
  }
  return KVFE_OK;
}

// sample k (0-based, `depth` bits) of a scanline -> 8 bits: 16-bit samples keep the high byte (png_set_strip_16),
// 1/2/4-bit grey is expanded (png_set_expand_gray_1_2_4_to_8), palette indices stay indices
inline int sample8(const uint8_t* r, size_t k, int depth, bool expand) {
  if (depth == 8) return r[k];
  if (depth == 16) return r[2 * k];
  const int per = 8 / depth, shift = (per - 1 - (int)(k % per)) * depth;
  const int v = (r[k / per] >> shift) & ((1 << depth) - 1);
  return expand ? v * 255 / ((1 << depth) - 1) : v;
}

// one decoded pixel -> the 8-bit grey value the reference ends up with
inline uint8_t to_gray(const PngHeader& H, const uint8_t* r, size_t x, const uint8_t* pal, int npal) {
  int R, G, B;
  switch (H.color) {
    case 0: return (uint8_t)sample8(r, x, H.depth, true);
    case 4: return (uint8_t)sample8(r, 2 * x, H.depth, false);   // grey replicated to BGR, BGR2GRAY of (v, v, v) = v
    case 3: {
      const int idx = sample8(r, x, H.depth, false);
      if (idx >= npal) {   // libpng: out-of-range index reads as black
        R = G = B = 0;
      } else {
        R = pal[3 * idx];
        G = pal[3 * idx + 1];
        B = pal[3 * idx + 2];
      }
      break;
    }
    default: {   // 2: RGB, 6: RGBA
      const size_t c = H.color == 2 ? 3 : 4;
      R = sample8(r, c * x, H.depth, false);
      G = sample8(r, c * x + 1, H.depth, false);
      B = sample8(r, c * x + 2, H.depth, false);
    }
  }
  // cv::cvtColor(BGR2GRAY), 8U: CV_DESCALE(b * BY15 + g * GY15 + r * RY15, 15)
  return (uint8_t)((B * 3735 + G * 19235 + R * 9798 + (1 << 14)) >> 15);
}

kvfe_status png_decode(const uint8_t* d, size_t n, uint8_t* dst, size_t dst_stride, int32_t width, int32_t height) {
  PngHeader H;
  kvfe_status st = png_header(d, n, &H);
  if (st != KVFE_OK) return st;
  if (!dst || (int64_t)H.w != width || (int64_t)H.h != height || dst_stride < (size_t)width) return KVFE_ERR_INVALID_ARG;
  // chunks
  // Scratch of the calling thread, kept between calls: two fresh 360 kB vectors per frame are two mmap / munmap pairs
  // and 180 page faults, which serialise the decode threads of a batch in the kernel's address-space lock.  A file
  // with ONE IDAT chunk (every encoder's output for frames of this size) is inflated from where it lies.
  thread_local std::vector<uint8_t> idat_joined, raw;
  std::vector<uint8_t>& idat = idat_joined;
  idat.clear();
  const uint8_t* idat_ptr = nullptr;
  size_t idat_len = 0;
  int idat_chunks = 0;
  uint8_t pal[768];
  int npal = 0;
  size_t off = 8;
  bool end = false;
  while (!end) {
    if (off + 12 > n) return KVFE_ERR_INVALID_ARG;
    const uint32_t len = be32(d + off);
    if ((size_t)len > n - off - 12) return KVFE_ERR_INVALID_ARG;
    const uint8_t* type = d + off + 4;
    const uint8_t* body = d + off + 8;
    const bool critical = !(type[0] & 0x20);
    const bool crc_ok = chunk_crc32(0u, type, 4 + (size_t)len) == be32(body + len);
    if (!crc_ok && critical) return KVFE_ERR_INVALID_ARG;   // (libpng only warns for ancillary chunks)
    if (!std::memcmp(type, "IDAT", 4)) {
      if (idat_chunks == 0) {
        idat_ptr = body;
        idat_len = len;
      } else {
        if (idat_chunks == 1) idat.assign(idat_ptr, idat_ptr + idat_len);
        idat.insert(idat.end(), body, body + len);
        idat_ptr = nullptr;
      }
      idat_chunks++;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      if (len % 3 != 0 || len > 768) return KVFE_ERR_INVALID_ARG;
      std::memcpy(pal, body, len);
      npal = (int)(len / 3);
    } else if (!std::memcmp(type, "IEND", 4)) {
      end = true;
    } else if (critical && std::memcmp(type, "IHDR", 4) != 0) {
      return KVFE_ERR_UNSUPPORTED;   // unknown critical chunk
    }
    off += 12 + (size_t)len;
  }
  if (idat_chunks > 1) {
    idat_ptr = idat.data();
    idat_len = idat.size();
  }
  if (idat_chunks == 0 || idat_len == 0 || (H.color == 3 && npal == 0)) return KVFE_ERR_INVALID_ARG;

  const int bits = H.depth * H.channels();
  const int bpp = std::max(1, bits / 8);
  auto rowbytes_of = [&](size_t w) { return (w * (size_t)bits + 7) / 8; };
  // geometry of the passes (one pass when not interlaced)
  static const int a7_x0[7] = {0, 4, 0, 2, 0, 1, 0}, a7_y0[7] = {0, 0, 4, 0, 2, 0, 1};
  static const int a7_dx[7] = {8, 8, 4, 4, 2, 2, 1}, a7_dy[7] = {8, 8, 8, 4, 4, 2, 2};
  struct Pass { size_t w, h, off; int x0, y0, dx, dy; };
  std::vector<Pass> passes;
  size_t total = 0;
  if (!H.interlace) {
    passes.push_back({H.w, H.h, 0, 0, 0, 1, 1});
    total = (size_t)H.h * (rowbytes_of(H.w) + 1);
  } else {
    for (int p = 0; p < 7; p++) {
      const size_t pw = (H.w + a7_dx[p] - 1 - a7_x0[p]) / a7_dx[p], ph = (H.h + a7_dy[p] - 1 - a7_y0[p]) / a7_dy[p];
      if (pw == 0 || ph == 0) continue;
      passes.push_back({pw, ph, total, a7_x0[p], a7_y0[p], a7_dx[p], a7_dy[p]});
      total += ph * (rowbytes_of(pw) + 1);
    }
  }
  raw.resize(total);
  {
    if (idat_len > 0xffffffffull || raw.size() > 0xffffffffull) return KVFE_ERR_UNSUPPORTED;   // zlib's 32-bit counts
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return KVFE_ERR_INVALID_ARG;
    zs.next_in = const_cast<uint8_t*>(idat_ptr);
    zs.avail_in = (uInt)idat_len;
    zs.next_out = raw.data();
    zs.avail_out = (uInt)raw.size();
    const int zr = inflate(&zs, Z_FINISH);
    const bool full = zs.avail_out == 0;
    inflateEnd(&zs);
    // (Z_BUF_ERROR with a full output: trailing bytes after the image, which libpng ignores with a warning)
    if (!(zr == Z_STREAM_END || (zr == Z_BUF_ERROR && full)) || !full) return KVFE_ERR_INVALID_ARG;
  }
  for (const Pass& P : passes) {
    const size_t rb = rowbytes_of(P.w);
    st = unfilter(raw.data() + P.off, P.h, rb, bpp);
    if (st != KVFE_OK) return st;
    for (size_t y = 0; y < P.h; y++) {
      const uint8_t* r = raw.data() + P.off + y * (rb + 1) + 1;
      uint8_t* o = dst + (size_t)(P.y0 + y * P.dy) * dst_stride;
      if (P.dx == 1 && H.color == 0 && H.depth == 8) {
        std::memcpy(o, r, P.w);   // the EuRoC case
      } else {
        for (size_t x = 0; x < P.w; x++) o[P.x0 + x * P.dx] = to_gray(H, r, x, pal, npal);
      }
    }
  }
  return KVFE_OK;
}

// ------------------------------------------------------------------------------------------------
// IMU buffer
// ------------------------------------------------------------------------------------------------
struct AccGyr {
  double v[6];
};

}  // namespace

struct kvfe_imu_buffer {
  mutable std::mutex mu;
  std::map<int64_t, AccGyr> values;     // ThreadsafeTemporalBuffer::values_
  int64_t buffer_length_ns = -1;
  std::atomic<bool> shutdown{false};

  // ThreadsafeImuBuffer::addMeasurement (ThreadsafeImuBuffer-inl.h:52-70: "Enforce strict time-wise ordering", a
  // sample that is not newer than the newest one is ignored) + ThreadsafeTemporalBuffer::addValue /
  // removeOutdatedItems
  void add(int64_t t, const double a[6]) {
    std::lock_guard<std::mutex> lk(mu);
    if (!values.empty() && t <= values.rbegin()->first) return;
    AccGyr x;
    std::memcpy(x.v, a, sizeof(x.v));
    values.emplace(t, x);
    if (values.empty() || buffer_length_ns <= 0) return;
    const int64_t thr = values.rbegin()->first - buffer_length_ns;
    if (values.begin()->first < thr) values.erase(values.begin(), values.lower_bound(thr));
  }
  // isDataAvailableUpToImpl
  int available(int64_t from, int64_t to) const {
    if (shutdown) return KVFE_IMU_QUEUE_SHUTDOWN;
    if (values.empty()) return KVFE_IMU_DATA_NOT_YET_AVAILABLE;
    if (values.rbegin()->first < to) return KVFE_IMU_DATA_NOT_YET_AVAILABLE;
    if (from < values.begin()->first) return KVFE_IMU_DATA_NEVER_AVAILABLE;
    return KVFE_IMU_DATA_AVAILABLE;
  }
  // getImuDataBtwTimestamps (lock held by the caller)
  int between(int64_t from, int64_t to, bool lower, std::vector<int64_t>& ts, std::vector<AccGyr>& vs) const {
    ts.clear();
    vs.clear();
    const int q = available(from, to);
    if (q != KVFE_IMU_DATA_AVAILABLE) return q;
    // getValuesBetweenTimes (the 100 % overlap condition holds after `available`)
    for (auto it = values.lower_bound(from); it != values.end() && it->first < to; ++it) {
      if (it->first == from && !lower) continue;
      ts.push_back(it->first);
      vs.push_back(it->second);
    }
    if (ts.empty()) return KVFE_IMU_TOO_FEW_MEASUREMENTS;
    return q;
  }
  // interpolateValueAtTimestamp: getValueAtOrBeforeTime / getValueAtOrAfterTime + linearInterpolate
  bool interpolate(int64_t t, AccGyr* out) const {
    auto lb = values.lower_bound(t);
    std::map<int64_t, AccGyr>::const_iterator pre, post;
    if (lb != values.end() && lb->first == t) {
      pre = post = lb;
    } else {
      if (values.empty() || lb == values.begin() || lb == values.end()) return false;   // (CHECK upstream)
      post = lb;
      pre = std::prev(lb);
    }
    kvfe_imu_linear_interpolate(pre->first, pre->second.v, post->first, post->second.v, t, out->v);
    return true;
  }
};

namespace {

int imu_emit(const std::vector<int64_t>& ts, const std::vector<AccGyr>& vs, int64_t* stamps, double* acc_gyr,
             int32_t capacity, int32_t* n) {
  if ((int64_t)ts.size() > capacity) {
    if (n) *n = (int32_t)ts.size();
    return -1;
  }
  for (size_t i = 0; i < ts.size(); i++) {
    stamps[i] = ts[i];
    std::memcpy(acc_gyr + 6 * i, vs[i].v, sizeof(double) * 6);
  }
  if (n) *n = (int32_t)ts.size();
  return KVFE_IMU_DATA_AVAILABLE;
}

int imu_query(kvfe_imu_buffer* b, int mode, int64_t from, int64_t to, bool lower, std::vector<int64_t>& ts,
              std::vector<AccGyr>& vs) {
  std::lock_guard<std::mutex> lk(b->mu);
  int q = b->between(from, to, mode == 1 ? true : (mode == 2 ? false : lower), ts, vs);
  if (q != KVFE_IMU_DATA_AVAILABLE) {
    ts.clear();
    vs.clear();
    return q;
  }
  if (mode == 1) {          // getImuDataInterpolatedUpperBorder
    AccGyr up;
    if (!b->interpolate(to, &up)) return KVFE_IMU_DATA_NEVER_AVAILABLE;
    ts.push_back(to);
    vs.push_back(up);
  } else if (mode == 2) {   // getImuDataInterpolatedBorders
    AccGyr lo, up;
    if (!b->interpolate(from, &lo) || !b->interpolate(to, &up)) return KVFE_IMU_DATA_NEVER_AVAILABLE;
    ts.insert(ts.begin(), from);
    vs.insert(vs.begin(), lo);
    ts.push_back(to);
    vs.push_back(up);
  }
  return q;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// stereo / IMU synchronisation
// ------------------------------------------------------------------------------------------------
struct kvfe_stereo_sync {
  struct FrameRef {
    int64_t t, tag;
  };
  std::mutex mu;
  std::deque<FrameRef> left, right;     // left_frame_queue_, right_frame_queue_
  kvfe_imu_buffer imu;                  // imu_data_.imu_buffer_
  bool have_cached = false;             // cached_left_frame_
  FrameRef cached{0, 0};
  int64_t timestamp_last_frame = 0;     // InvalidTimestamp = 0 (DataProviderModule.h:172)
  bool do_coarse_sync = false;
  int64_t imu_timestamp_correction = 0;
  std::atomic<int64_t> imu_time_shift_ns{0};
  std::atomic<bool> shutdown{false};
  int mode = KVFE_SYNC_MODE_STEREO;
};

namespace {

// ------------------------------------------------------------------------------------------------
// Worker pool of kvfe_png_decode_gray_batch.  Starting a thread costs 30-50 us, a 752x480 frame decodes in ~1.2 ms:
// with one thread per file and call the caller spends longer starting 64 threads than any of them works.  The pool
// is created on first use, grows to the largest thread count asked for (at most the hardware concurrency), lives for
// the rest of the process (its threads are detached: nothing to join at exit) and serves one batch at a time -- a
// second caller that finds it busy decodes on threads of its own.
//
// Round 5: a batch is handed over through ONE atomic word, and a worker that has just finished a batch spins on it for
// 50 us before it goes to sleep.  The first pool woke its workers through one mutex and condition variable: 63 threads
// taking the same lock one after the other (and idle cores coming out of their sleep states) cost ~3.7 ms per call of 64
// files on the 256-thread host of the GPU box -- three times the decode itself (bench.py input_side: 13.0 k frames/s on
// all threads against 0.83 k on one).  A decoder that feeds a running front-end calls again within that window, so its
// workers are awake when the next step's files arrive.
//   state_ = generation << 32 | OPEN | number of registered helpers.  A helper registers by compare-and-swap while the
//   batch is open and fewer than want_ have; the caller closes the batch (clears OPEN) when the files are handed out and
//   waits for exactly the helpers that registered: a late worker can neither touch the caller's stack nor be waited for.
// ------------------------------------------------------------------------------------------------
class DecodePool {
 public:
  static DecodePool& get() {
    // never destroyed (the detached workers may outlive static destruction); at exit they are told to stop, so none of
    // them is still spinning or decoding while the process tears its statics down
    static DecodePool* p = [] {
      DecodePool* q = new DecodePool();
      std::atexit([] { DecodePool::get().shutdown(); });
      return q;
    }();
    return *p;
  }
  // runs fn(i) for i in [0, n) on `threads` threads in total (the caller is one of them); false = pool busy
  template <class F>
  bool run(int n, int threads, const F& fn) {
    std::unique_lock<std::mutex> owner(run_mu_, std::try_to_lock);
    if (!owner.owns_lock()) return false;
    struct Ctx { const F* fn; std::atomic<int> next{0}; int n; } ctx;
    ctx.fn = &fn;
    ctx.n = n;
    auto body = [](void* c) noexcept {
      Ctx* x = static_cast<Ctx*>(c);
      for (;;) {
        const int i = x->next.fetch_add(1);
        if (i >= x->n) return;
        (*x->fn)(i);
      }
    };
    int helpers = 0;
    {
      std::lock_guard<std::mutex> lk(mu_);
      grow(threads - 1);
      helpers = std::min<int>(threads - 1, (int)started_);
    }
    job_ = body;
    job_ctx_ = &ctx;
    want_.store(helpers, std::memory_order_relaxed);
    done_.store(0);
    const uint64_t gen = ((state_.load() >> 32) + 1) & 0xffffffffull;
    state_.store(gen << 32 | (helpers > 0 ? kOpen : 0));   // (seq_cst: ordered against the sleepers_ read below)
    if (helpers > 0 && sleepers_.load() > 0) {
      { std::lock_guard<std::mutex> lk(mu_); }   // a worker between its predicate test and its wait holds mu_
      cv_job_.notify_all();
    }
    body(&ctx);
    // every file is handed out: close the batch, then wait for the helpers that got in
    uint64_t s = state_.load();
    while (!state_.compare_exchange_weak(s, s & ~kOpen)) {
    }
    const int registered = (int)(s & kCount);
    // a bounded spin (a helper is at most one file behind), then a blocking wait: under a cgroup CPU quota a descheduled
    // helper must not have the caller burn the very quota it is waiting to be given
    for (int spins = 0; done_.load(std::memory_order_acquire) != registered && spins < 20000; spins++) cpu_relax();
    if (done_.load(std::memory_order_acquire) != registered) {
      std::unique_lock<std::mutex> lk(mu_);
      caller_waits_.store(true);
      while (done_.load(std::memory_order_acquire) != registered)
        cv_done_.wait_for(lk, std::chrono::milliseconds(2));   // (the timeout covers a notify that raced the flag)
      caller_waits_.store(false);
    }
    job_ = nullptr;
    return true;
  }
  void shutdown() {
    stop_.store(true);
    { std::lock_guard<std::mutex> lk(mu_); }
    cv_job_.notify_all();
  }

 private:
  static constexpr uint64_t kOpen = 1ull << 31, kCount = kOpen - 1;
  static void cpu_relax() noexcept {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  void grow(int helpers) {   // mu_ held
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    const size_t cap = hw > 1 ? hw - 1 : 0;
    while (started_ < (size_t)std::max(helpers, 0) && started_ < cap) {
      try {
        std::thread([this] { worker(); }).detach();
      } catch (const std::exception&) {
        break;
      }
      started_++;
    }
  }
  void worker() {
    uint64_t seen = state_.load() >> 32;   // (a worker started for batch g + 1 sees generation g here)
    for (;;) {
      // the next batch: spin for a while (a running decoder comes back within a step), then sleep
      uint64_t s;
      auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(kSpinUs);
      for (int spins = 0;;) {
        if (stop_.load(std::memory_order_relaxed)) return;
        s = state_.load();
        if ((s >> 32) != seen) break;
        if ((++spins & 63) != 0 || std::chrono::steady_clock::now() < spin_until) {   // (the clock every 64th look)
          cpu_relax();
          continue;
        }
        std::unique_lock<std::mutex> lk(mu_);
        sleepers_.fetch_add(1);
        cv_job_.wait(lk, [&] { return (state_.load() >> 32) != seen || stop_.load(); });
        sleepers_.fetch_sub(1);
        spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(kSpinUs);
      }
      seen = s >> 32;
      bool in = false;
      while ((s & kOpen) && (s >> 32) == seen && (int)(s & kCount) < want_.load(std::memory_order_relaxed)) {
        if (state_.compare_exchange_weak(s, s + 1)) {
          in = true;
          break;
        }
      }
      if (!in) continue;   // (closed, full, or already the next batch: the loop above looks at it again)
      job_(job_ctx_);
      done_.fetch_add(1, std::memory_order_release);
      if (caller_waits_.load()) {
        { std::lock_guard<std::mutex> lk(mu_); }
        cv_done_.notify_one();
      }
    }
  }
  static constexpr int kSpinUs = 50;   // a worker spins this long for the next batch (a decoder that calls again at once finds it awake;
                                         // longer spins only burn the CPU quota of a container: 0.5 ms cost 12 % at 64 threads, tools/r5/gpu_ad.sh)
  std::mutex run_mu_, mu_;
  std::condition_variable cv_job_, cv_done_;
  std::atomic<bool> stop_{false}, caller_waits_{false};
  std::atomic<uint64_t> state_{0};
  std::atomic<int> done_{0}, sleepers_{0};
  // written by run() before it publishes the batch in state_, read by a worker after it has registered
  void (*job_)(void*) noexcept = nullptr;
  void* job_ctx_ = nullptr;
  std::atomic<int> want_{0};   // (atomic: a late worker may look at it while the next batch is being set up; its registration then fails)
  size_t started_ = 0;
};

// The processors this process can actually use: the hardware threads, cut down to its affinity mask and to its cgroup's
// CPU quota (a container on a 256-thread host is typically allowed a dozen: bench.py's input_side leg read the same
// 12 - 13 k frames/s from 16 threads as from 256 -- and 64 spinning or waking threads only eat that quota).  The default
// thread count of kvfe_png_decode_gray_batch (threads <= 0).
unsigned effective_cpus() {
  static const unsigned n = [] {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
#if defined(__linux__)
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
      const int c = CPU_COUNT(&set);
      if (c > 0) hw = std::min<unsigned>(hw, (unsigned)c);
    }
    auto read_ll = [](const char* path, long long* a, long long* b) -> int {   // "a b" or "max b" or "a"
      FILE* f = std::fopen(path, "r");
      if (!f) return 0;
      char buf[64] = {0};
      const size_t got = std::fread(buf, 1, sizeof(buf) - 1, f);
      std::fclose(f);
      if (got == 0) return 0;
      if (!std::strncmp(buf, "max", 3)) return -1;
      char* end = nullptr;
      *a = std::strtoll(buf, &end, 10);
      if (end == buf) return 0;
      if (b) *b = std::strtoll(end, nullptr, 10);
      return 1;
    };
    long long quota = 0, period = 0;
    if (read_ll("/sys/fs/cgroup/cpu.max", &quota, &period) == 1 && quota > 0 && period > 0) {   // cgroup v2
      hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    } else if (read_ll("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &quota, nullptr) == 1 && quota > 0 &&
               read_ll("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &period, nullptr) == 1 && period > 0) {   // cgroup v1
      hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    }
#endif
    return hw;
  }();
  return n;
}

}  // namespace

extern "C" {

kvfe_status kvfe_png_info(const uint8_t* data, size_t size, int32_t* width, int32_t* height, int32_t* channels) {
  PngHeader H;
  const kvfe_status st = png_header(data, size, &H);
  if (st != KVFE_OK) return st;
  if (width) *width = (int32_t)H.w;
  if (height) *height = (int32_t)H.h;
  if (channels) *channels = H.cv_channels();
  return KVFE_OK;
}

// the C boundary never throws: an allocation failure inside the decoder is reported, not propagated
static kvfe_status png_decode_noexcept(const uint8_t* data, size_t size, uint8_t* dst, size_t dst_stride,
                                       int32_t width, int32_t height) noexcept {
  try {
    return png_decode(data, size, dst, dst_stride, width, height);
  } catch (const std::bad_alloc&) {
    return KVFE_ERR_CAPACITY;
  } catch (...) {
    return KVFE_ERR_INVALID_ARG;
  }
}

kvfe_status kvfe_png_decode_gray(const uint8_t* data, size_t size, uint8_t* dst, size_t dst_stride, int32_t width,
                                 int32_t height) {
  return png_decode_noexcept(data, size, dst, dst_stride, width, height);
}

kvfe_status kvfe_png_decode_gray_batch(const uint8_t* const* data, const size_t* sizes, uint8_t* const* dst,
                                       size_t dst_stride, int32_t width, int32_t height, int32_t n, int32_t threads,
                                       kvfe_status* status) {
  if (n < 0 || (n > 0 && (!data || !sizes || !dst))) return KVFE_ERR_INVALID_ARG;
  if (n == 0) return KVFE_OK;
  const unsigned hw = effective_cpus();
  int nt = threads > 0 ? threads : (int)std::min<unsigned>(hw, (unsigned)n);
  nt = std::max(1, std::min(nt, n));
  try {
    std::vector<kvfe_status> local((size_t)n, KVFE_OK);
    auto one = [&](int i) noexcept {
      local[i] = png_decode_noexcept(data[i], sizes[i], dst[i], dst_stride, width, height);
    };
    if (nt == 1) {
      for (int i = 0; i < n; i++) one(i);
    } else if (!DecodePool::get().run(n, nt, one)) {
      // the pool serves another caller: threads of this call's own (the calling thread is worker 0; a thread that
      // cannot be started is done without)
      std::atomic<int> next{0};
      auto work = [&]() noexcept {
        for (;;) {
          const int i = next.fetch_add(1);
          if (i >= n) return;
          one(i);
        }
      };
      std::vector<std::thread> own;
      for (int t = 1; t < nt; t++) {
        try {
          own.emplace_back(work);
        } catch (const std::exception&) {
          break;
        }
      }
      work();
      for (auto& th : own) th.join();
    }
    kvfe_status first = KVFE_OK;
    for (int i = 0; i < n; i++) {
      if (status) status[i] = local[i];
      if (first == KVFE_OK && local[i] != KVFE_OK) first = local[i];
    }
    return first;
  } catch (const std::bad_alloc&) {
    return KVFE_ERR_CAPACITY;
  }
}

kvfe_imu_buffer* kvfe_imu_buffer_create(int64_t buffer_length_ns) {
  kvfe_imu_buffer* b = new (std::nothrow) kvfe_imu_buffer();
  if (b) b->buffer_length_ns = buffer_length_ns;
  return b;
}
void kvfe_imu_buffer_destroy(kvfe_imu_buffer* b) { delete b; }
void kvfe_imu_buffer_add(kvfe_imu_buffer* b, int64_t timestamp_ns, const double acc_gyr[6]) {
  if (!b || !acc_gyr) return;
  try {
    b->add(timestamp_ns, acc_gyr);
  } catch (const std::bad_alloc&) {   // out of memory: the sample is lost, like one that arrived out of order
  }
}
int64_t kvfe_imu_buffer_size(const kvfe_imu_buffer* b) {
  if (!b) return 0;
  std::lock_guard<std::mutex> lk(b->mu);
  return (int64_t)b->values.size();
}
void kvfe_imu_buffer_shutdown(kvfe_imu_buffer* b) {
  if (b) b->shutdown = true;
}

void kvfe_imu_linear_interpolate(int64_t t0, const double y0[6], int64_t t1, const double y1[6], int64_t t,
                                 double y[6]) {
  // *y = t0 == t1 ? y0 : y0 + (y1 - y0) * double(t - t0) / double(t1 - t0)   (Eigen: element-wise, left to right)
  for (int i = 0; i < 6; i++)
    y[i] = t0 == t1 ? y0[i] : y0[i] + (y1[i] - y0[i]) * static_cast<double>(t - t0) / static_cast<double>(t1 - t0);
}

static int32_t imu_query_c(kvfe_imu_buffer* b, int mode, int64_t t_from, int64_t t_to, int32_t lower,
                           int64_t* stamps, double* acc_gyr, int32_t capacity, int32_t* n) {
  if (n) *n = 0;
  if (!b || capacity < 0 || (capacity > 0 && (!stamps || !acc_gyr))) return KVFE_IMU_DATA_NEVER_AVAILABLE;
  if (t_from >= t_to) return KVFE_IMU_DATA_NEVER_AVAILABLE;   // (upstream: CHECK_LT(timestamp_ns_from, timestamp_ns_to) aborts)
  try {
    std::vector<int64_t> ts;
    std::vector<AccGyr> vs;
    const int q = imu_query(b, mode, t_from, t_to, lower != 0, ts, vs);
    if (q != KVFE_IMU_DATA_AVAILABLE) return q;
    return imu_emit(ts, vs, stamps, acc_gyr, capacity, n);
  } catch (const std::bad_alloc&) {
    return KVFE_IMU_TOO_FEW_MEASUREMENTS;
  }
}
int32_t kvfe_imu_buffer_between(kvfe_imu_buffer* b, int64_t t_from, int64_t t_to, int32_t get_lower_bound,
                                int64_t* stamps, double* acc_gyr, int32_t capacity, int32_t* n) {
  return imu_query_c(b, 0, t_from, t_to, get_lower_bound, stamps, acc_gyr, capacity, n);
}
int32_t kvfe_imu_buffer_interpolated_upper_border(kvfe_imu_buffer* b, int64_t t_from, int64_t t_to, int64_t* stamps,
                                                  double* acc_gyr, int32_t capacity, int32_t* n) {
  return imu_query_c(b, 1, t_from, t_to, 1, stamps, acc_gyr, capacity, n);
}
int32_t kvfe_imu_buffer_interpolated_borders(kvfe_imu_buffer* b, int64_t t_from, int64_t t_to, int64_t* stamps,
                                             double* acc_gyr, int32_t capacity, int32_t* n) {
  return imu_query_c(b, 2, t_from, t_to, 0, stamps, acc_gyr, capacity, n);
}

static inline void mat3_mul(const double* a, const double* b, double* c) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

kvfe_status kvfe_imu_preintegrate_rotation(const int64_t* stamps, const double* acc_gyr, int32_t n,
                                           const double gyro_bias[3], double deltaRij[9]) {
  if (!stamps || !acc_gyr || !deltaRij || n < 2) return KVFE_ERR_INVALID_ARG;
  for (int32_t i = 0; i + 1 < n; i++)
    if (stamps[i + 1] <= stamps[i]) return KVFE_ERR_INVALID_ARG;   // CHECK_GT(delta_t, 0.0) << "Imu delta is 0!"
  const double bz[3] = {0, 0, 0};
  const double* bg = gyro_bias ? gyro_bias : bz;
  for (int32_t i = 0; i + 1 < n; i++) {
    const double dt = static_cast<double>(stamps[i + 1] - stamps[i]) / 1e9;       // UtilsNumerical::NsecToSec
    const double* g = acc_gyr + 6 * (size_t)i + 3;
    const double w[3] = {(g[0] - bg[0]) * dt, (g[1] - bg[1]) * dt, (g[2] - bg[2]) * dt};
    // so3::ExpmapFunctor(omega).expmap()
    const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double W[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
    double E[9];
    if (theta2 <= 2.220446049250313e-16) {
      for (int k = 0; k < 9; k++) E[k] = W[k];
      E[0] += 1.0;
      E[4] += 1.0;
      E[8] += 1.0;
    } else {
      const double theta = std::sqrt(theta2), sin_theta = std::sin(theta), s2 = std::sin(theta / 2.0);
      const double one_minus_cos = 2.0 * s2 * s2;
      double K[9], KK[9];
      for (int k = 0; k < 9; k++) K[k] = W[k] / theta;
      mat3_mul(K, K, KK);
      for (int k = 0; k < 9; k++) E[k] = sin_theta * K[k] + one_minus_cos * KK[k];
      E[0] += 1.0;
      E[4] += 1.0;
      E[8] += 1.0;
    }
    double R[9];
    mat3_mul(deltaRij, E, R);   // deltaXij_.attitude().compose(dR)
    std::memcpy(deltaRij, R, sizeof(R));
  }
  return KVFE_OK;
}

void kvfe_keyframe_R_cur_frame(const double body_R_camLrect[9], const double deltaRij[9], double out[9]) {
  if (!body_R_camLrect || !deltaRij || !out) return;
  double cam_R_body[9], t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) cam_R_body[3 * i + j] = body_R_camLrect[3 * j + i];
  mat3_mul(cam_R_body, deltaRij, t);        // (cam_Rot_body * pim->deltaRij()) * body_Rot_cam
  mat3_mul(t, body_R_camLrect, out);
}

kvfe_stereo_sync* kvfe_stereo_sync_create(int64_t imu_buffer_length_ns) {
  kvfe_stereo_sync* s = new (std::nothrow) kvfe_stereo_sync();
  if (s) s->imu.buffer_length_ns = imu_buffer_length_ns;
  return s;
}
void kvfe_stereo_sync_destroy(kvfe_stereo_sync* s) { delete s; }
kvfe_status kvfe_stereo_sync_set_mode(kvfe_stereo_sync* s, int32_t mode) {
  if (!s || mode < KVFE_SYNC_MODE_STEREO || mode > KVFE_SYNC_MODE_RGBD) return KVFE_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lk(s->mu);
  s->mode = mode;
  return KVFE_OK;
}
void kvfe_stereo_sync_fill_left(kvfe_stereo_sync* s, int64_t timestamp_ns, int64_t tag) {
  if (!s) return;
  std::lock_guard<std::mutex> lk(s->mu);
  try {
    s->left.push_back({timestamp_ns, tag});
  } catch (const std::bad_alloc&) {   // out of memory: the frame is dropped
  }
}
void kvfe_stereo_sync_fill_right(kvfe_stereo_sync* s, int64_t timestamp_ns, int64_t tag) {
  if (!s) return;
  std::lock_guard<std::mutex> lk(s->mu);
  try {
    s->right.push_back({timestamp_ns, tag});
  } catch (const std::bad_alloc&) {   // out of memory: the frame is dropped
  }
}
void kvfe_stereo_sync_fill_imu(kvfe_stereo_sync* s, int64_t timestamp_ns, const double acc_gyr[6]) {
  if (s) kvfe_imu_buffer_add(&s->imu, timestamp_ns, acc_gyr);
}
void kvfe_stereo_sync_do_coarse_imu_camera_temporal_sync(kvfe_stereo_sync* s) {
  if (!s) return;
  std::lock_guard<std::mutex> lk(s->mu);
  s->do_coarse_sync = true;
}
void kvfe_stereo_sync_set_imu_time_shift(kvfe_stereo_sync* s, double imu_time_shift_s) {
  // UtilsNumerical::SecToNsec: nanoseconds of a std::chrono::duration<double>, truncated
  if (s) s->imu_time_shift_ns = (int64_t)(imu_time_shift_s * 1e9);
}
void kvfe_stereo_sync_shutdown(kvfe_stereo_sync* s) {
  if (!s) return;
  s->shutdown = true;
  s->imu.shutdown = true;
}

int32_t kvfe_stereo_sync_next(kvfe_stereo_sync* s, kvfe_sync_packet* packet, int64_t* imu_stamps,
                              double* imu_acc_gyr, int32_t capacity) {
  if (!s || !packet || capacity < 0 || (capacity > 0 && (!imu_stamps || !imu_acc_gyr))) return KVFE_SYNC_SHUTDOWN;
  std::lock_guard<std::mutex> lk(s->mu);
  std::memset(packet, 0, sizeof(*packet));
  if (s->shutdown) return KVFE_SYNC_SHUTDOWN;
  // ---- getMonoImuSyncPacket(cache_timestamp = false) ----------------------------------------------
  kvfe_stereo_sync::FrameRef lf;
  const bool from_cache = s->have_cached;
  if (from_cache) {
    lf = s->cached;
  } else {
    if (s->left.empty()) return KVFE_SYNC_EMPTY;
    lf = s->left.front();
  }
  auto consume_left = [&]() {
    if (from_cache)
      s->have_cached = false;
    else
      s->left.pop_front();
  };
  if (s->timestamp_last_frame >= lf.t) {
    consume_left();
    return KVFE_SYNC_DROP_OUT_OF_ORDER;
  }
  // ---- getTimeSyncedImuMeasurements -------------------------------------------------------------
  if (kvfe_imu_buffer_size(&s->imu) == 0) {
    consume_left();
    return KVFE_SYNC_DROP_NO_IMU;
  }
  if (s->timestamp_last_frame == 0) {
    s->timestamp_last_frame = lf.t;
    consume_left();
    return KVFE_SYNC_DROP_FIRST_FRAME;
  }
  bool coarse = s->do_coarse_sync;
  int64_t correction = s->imu_timestamp_correction;
  if (coarse) {
    std::lock_guard<std::mutex> lki(s->imu.mu);
    correction = s->imu.values.rbegin()->first - lf.t;   // newest_imu.timestamp_ - timestamp
  }
  const int64_t shift = s->imu_time_shift_ns;
  const int64_t t_last = s->timestamp_last_frame + correction + shift, t_cur = lf.t + correction + shift;
  std::vector<int64_t> ts;
  std::vector<AccGyr> vs;
  int q;
  try {
    q = imu_query(&s->imu, 2, t_last, t_cur, false, ts, vs);
  } catch (const std::bad_alloc&) {
    return KVFE_SYNC_WAIT_IMU;   // out of memory: nothing consumed, no state changed -- the caller may try again
  }
  if (q == KVFE_IMU_DATA_AVAILABLE && (int64_t)ts.size() > capacity) {
    packet->n_imu = (int32_t)ts.size();
    return -1;   // nothing consumed, no state changed
  }
  if (coarse) {   // (the correction is computed once, on the first frame that reaches this point)
    s->imu_timestamp_correction = correction;
    s->do_coarse_sync = false;
  }
  switch (q) {
    case KVFE_IMU_DATA_AVAILABLE: break;
    case KVFE_IMU_DATA_NOT_YET_AVAILABLE:   // FrameAction::Wait: cached_left_frame_ = the frame
      if (!from_cache) {
        s->left.pop_front();
        s->cached = lf;
        s->have_cached = true;
      }
      return KVFE_SYNC_WAIT_IMU;
    case KVFE_IMU_QUEUE_SHUTDOWN:
      s->shutdown = true;
      consume_left();
      return KVFE_SYNC_SHUTDOWN;
    case KVFE_IMU_DATA_NEVER_AVAILABLE:
      s->timestamp_last_frame = lf.t;
      consume_left();
      return KVFE_SYNC_DROP_IMU_NEVER;
    default:
      consume_left();
      return KVFE_SYNC_DROP_IMU_TOO_FEW;
  }
  for (int64_t& t : ts) t -= correction + shift;   // "adjust the timestamps for the frontend"
  consume_left();
  // getMonoImuSyncPacket(cache_timestamp): true for the mono and RGBD providers, false for the stereo one
  if (s->mode != KVFE_SYNC_MODE_STEREO) s->timestamp_last_frame = lf.t;
  // ---- syncQueue(timestamp, &right_frame_queue_ / &depth_frame_queue_) --------------------------
  bool found = s->mode == KVFE_SYNC_MODE_MONO;
  kvfe_stereo_sync::FrameRef rf{0, -1};
  while (!found && !s->right.empty()) {
    const kvfe_stereo_sync::FrameRef cur = s->right.front();
    if (cur.t > lf.t) break;          // "Could not retrieve exact timestamp requested": left in the queue
    s->right.pop_front();
    if (cur.t == lf.t) {
      rf = cur;
      found = true;
      break;
    }
  }
  if (!found) return KVFE_SYNC_DROP_NO_RIGHT;   // (timestamp_last_frame_ keeps its value)
  s->timestamp_last_frame = lf.t;
  packet->timestamp_ns = lf.t;
  packet->left_tag = lf.tag;
  packet->right_tag = rf.tag;
  int32_t n = 0;
  imu_emit(ts, vs, imu_stamps, imu_acc_gyr, capacity, &n);
  packet->n_imu = n;
  return KVFE_SYNC_PACKET;
}

// ------------------------------------------------------------------------------------------------
// EuRoC index files
// ------------------------------------------------------------------------------------------------
static bool next_line(const char* text, size_t size, size_t* pos, const char** line, size_t* len) {
  if (*pos >= size) return false;
  const char* b = text + *pos;
  const char* e = (const char*)std::memchr(b, '\n', size - *pos);
  const size_t l = e ? (size_t)(e - b) : size - *pos;
  *line = b;
  *len = l;
  *pos += l + (e ? 1 : 0);
  return true;
}

// std::stoll / std::stod of a field: leading whitespace skipped, trailing characters ignored; false = no conversion
static bool field_ll(const char* b, size_t n, int64_t* v) {
  char tmp[64];
  const size_t m = std::min(n, sizeof(tmp) - 1);
  std::memcpy(tmp, b, m);
  tmp[m] = 0;
  char* end = nullptr;
  errno = 0;
  const long long x = std::strtoll(tmp, &end, 10);
  if (end == tmp || errno == ERANGE) return false;
  *v = (int64_t)x;
  return true;
}
static bool field_d(const char* b, size_t n, double* v) {
  char tmp[128];
  const size_t m = std::min(n, sizeof(tmp) - 1);
  std::memcpy(tmp, b, m);
  tmp[m] = 0;
  char* end = nullptr;
  errno = 0;
  const double x = std::strtod(tmp, &end);
  if (end == tmp) return false;
  *v = x;
  return true;
}

kvfe_status kvfe_euroc_parse_camera_csv(const char* text, size_t size, int64_t* timestamps, int32_t capacity,
                                        int32_t* n) {
  if (!text || !n || capacity < 0 || (capacity > 0 && !timestamps)) return KVFE_ERR_INVALID_ARG;
  size_t pos = 0, len = 0;
  const char* line = nullptr;
  next_line(text, size, &pos, &line, &len);   // header
  int64_t count = 0;
  while (next_line(text, size, &pos, &line, &len)) {
    const char* c = (const char*)std::memchr(line, ',', len);
    int64_t t;
    if (!field_ll(line, c ? (size_t)(c - line) : len, &t)) {
      if (len == 0 || (len == 1 && line[0] == '\r')) continue;   // (a trailing blank line would throw upstream)
      return KVFE_ERR_INVALID_ARG;
    }
    if (count < capacity) timestamps[count] = t;
    count++;
  }
  *n = (int32_t)count;
  return count > capacity ? KVFE_ERR_CAPACITY : KVFE_OK;
}

kvfe_status kvfe_euroc_parse_imu_csv(const char* text, size_t size, int64_t* timestamps, double* acc_gyr,
                                     int32_t capacity, int32_t* n) {
  if (!text || !n || capacity < 0 || (capacity > 0 && (!timestamps || !acc_gyr))) return KVFE_ERR_INVALID_ARG;
  size_t pos = 0, len = 0;
  const char* line = nullptr;
  next_line(text, size, &pos, &line, &len);   // header
  int64_t count = 0, previous = -1;
  while (next_line(text, size, &pos, &line, &len)) {
    if (len == 0 || (len == 1 && line[0] == '\r')) continue;
    int64_t t = 0;
    double g[6];   // w_x w_y w_z a_x a_y a_z
    const char* b = line;
    size_t left = len;
    for (int i = 0; i < 7; i++) {
      const char* c = (const char*)std::memchr(b, ',', left);
      const size_t fl = c ? (size_t)(c - b) : left;
      if (i == 0 ? !field_ll(b, fl, &t) : !field_d(b, fl, &g[i - 1])) return KVFE_ERR_INVALID_ARG;
      if (c) {
        left -= fl + 1;
        b = c + 1;
      } else {   // line.substr(npos + 1) == the whole remainder again upstream; a short row is malformed here
        if (i < 6) return KVFE_ERR_INVALID_ARG;
        left = 0;
      }
    }
    if (t <= previous) return KVFE_ERR_INVALID_ARG;   // "Euroc IMU data is not in chronological order!"
    previous = t;
    if (count < capacity) {
      timestamps[count] = t;
      double* o = acc_gyr + 6 * count;   // imu_accgyr << gyr_acc_data.tail(3), gyr_acc_data.head(3)
      o[0] = g[3]; o[1] = g[4]; o[2] = g[5];
      o[3] = g[0]; o[4] = g[1]; o[5] = g[2];
    }
    count++;
  }
  *n = (int32_t)count;
  return count > capacity ? KVFE_ERR_CAPACITY : KVFE_OK;
}

}  // extern "C"
