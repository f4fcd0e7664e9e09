// Robustness of the input side at the C boundary ("never abort, never read or write out of bounds"): this program is
// built TOGETHER with kimera_vio_amd/csrc/host_input.cpp under -fsanitize=address,undefined and feeds kvfe_png_* with
// mutated PNG files (bit flips, truncations, header edits -- chunk CRCs re-computed so that the damage reaches the
// inflate / unfilter / sample-expansion code), the CSV parsers with mutated text and the synchroniser with random
// traffic.  Any sanitizer report or crash fails the test (tests/test_host_logic.py).
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "kvfe.h"

static void put32(std::vector<uint8_t>& v, size_t at, uint32_t x) {
  v[at] = x >> 24; v[at + 1] = x >> 16; v[at + 2] = x >> 8; v[at + 3] = x;
}
static uint32_t get32(const std::vector<uint8_t>& v, size_t at) {
  return ((uint32_t)v[at] << 24) | ((uint32_t)v[at + 1] << 16) | ((uint32_t)v[at + 2] << 8) | v[at + 3];
}
// re-compute every chunk CRC that can still be located
static void fix_crcs(std::vector<uint8_t>& f) {
  size_t off = 8;
  while (off + 12 <= f.size()) {
    const uint32_t len = get32(f, off);
    if ((size_t)len > f.size() - off - 12) break;
    put32(f, off + 8 + len, (uint32_t)crc32(crc32(0L, Z_NULL, 0), f.data() + off + 4, 4 + len));
    off += 12 + (size_t)len;
  }
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::mt19937 rng(12345);
  long decoded = 0, refused = 0;
  for (int a = 1; a < argc; a++) {
    std::ifstream in(argv[a], std::ios::binary);
    const std::vector<uint8_t> orig((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    int32_t w = 0, h = 0, c = 0;
    if (orig.size() > 2 && orig[0] == 0xFF && orig[1] == 0xD8) {
      // a JPEG file: byte damage anywhere (no checksums in the format), truncations, header edits
      if (kvfe_jpeg_info(orig.data(), orig.size(), &w, &h, &c) != KVFE_OK) return 11;
      std::vector<uint8_t> out((size_t)w * h + 64, 0xAB);
      if (kvfe_jpeg_decode_gray(orig.data(), orig.size(), out.data(), (size_t)w, w, h) != KVFE_OK) return 12;
      for (int it = 0; it < 1500; it++) {
        std::vector<uint8_t> f = orig;
        const int kind = it % 4;
        if (kind == 0) for (int k = 0; k < 1 + (int)(rng() % 6); k++) f[rng() % f.size()] = (uint8_t)rng();
        else if (kind == 1) f.resize(rng() % f.size());
        else if (kind == 2) for (int k = 0; k < 1 + (int)(rng() % 3); k++) f[2 + rng() % (f.size() < 700 ? f.size() - 2 : 700)] ^= (uint8_t)(1u << (rng() % 8));
        else { const size_t at = rng() % (f.size() - 1); f[at] = 0xFF; f[at + 1] = (uint8_t)(0xC0 + rng() % 0x40); }
        const kvfe_status sd = kvfe_jpeg_decode_gray(f.data(), f.size(), out.data(), (size_t)w, w, h);
        if (sd == KVFE_OK) decoded++; else refused++;
        for (size_t k = (size_t)w * h; k < out.size(); k++)
          if (out[k] != 0xAB) return 13;
      }
      continue;
    }
    if (kvfe_png_info(orig.data(), orig.size(), &w, &h, &c) != KVFE_OK) return 3;
    std::vector<uint8_t> out((size_t)w * h + 64, 0xAB);
    for (int it = 0; it < 300; it++) {
      std::vector<uint8_t> f = orig;
      const int kind = it % 6;
      if (kind == 0) {                                 // random byte damage, CRCs left broken
        for (int k = 0; k < 1 + (int)(rng() % 8); k++) f[rng() % f.size()] ^= (uint8_t)(1u << (rng() % 8));
      } else if (kind == 1) {                          // truncation
        f.resize(rng() % f.size());
      } else if (kind == 2) {                          // damage inside the compressed stream, CRCs repaired
        for (int k = 0; k < 1 + (int)(rng() % 16); k++) f[41 + rng() % (f.size() - 60)] ^= (uint8_t)(rng() | 1);
        fix_crcs(f);
      } else if (kind == 3) {                          // IHDR edits (depth, colour type, interlace, size), CRC repaired
        const int what = rng() % 5;
        if (what == 0) f[24] = (uint8_t)(rng() % 20);
        if (what == 1) f[25] = (uint8_t)(rng() % 8);
        if (what == 2) f[28] = (uint8_t)(rng() % 3);
        if (what == 3) put32(f, 16, (uint32_t)(rng() % 4000));
        if (what == 4) put32(f, 20, (uint32_t)(rng() % 4000));
        fix_crcs(f);
      } else if (kind == 4) {                          // chunk length edits
        put32(f, 33, (uint32_t)rng());
      } else {                                         // a foreign chunk type / palette in front of the data
        f[37] ^= 0x20;
        fix_crcs(f);
      }
      int32_t w2 = 0, h2 = 0, c2 = 0;
      const kvfe_status si = kvfe_png_info(f.data(), f.size(), &w2, &h2, &c2);
      // decode with the ORIGINAL geometry (the caller's buffer): a header that disagrees must be refused
      const kvfe_status sd = kvfe_png_decode_gray(f.data(), f.size(), out.data(), (size_t)w, w, h);
      if (sd == KVFE_OK) decoded++; else refused++;
      if (si != KVFE_OK && sd == KVFE_OK) return 4;    // decode accepted what info refused
      for (size_t k = (size_t)w * h; k < out.size(); k++)
        if (out[k] != 0xAB) return 5;                   // wrote past the image
    }
  }
  // well-formed containers around random scan-line bytes: every colour type / depth / interlacing, random filter
  // bytes (5 = invalid), random palette sizes -- the damage-free path into unfilter / sample expansion / Adam7
  {
    static const int combos[][2] = {{0, 1}, {0, 2}, {0, 4}, {0, 8}, {0, 16}, {2, 8}, {2, 16}, {3, 1}, {3, 2},
                                    {3, 4}, {3, 8}, {4, 8}, {4, 16}, {6, 8}, {6, 16}};
    auto chunk = [](std::vector<uint8_t>& f, const char* tag, const std::vector<uint8_t>& body) {
      const size_t at = f.size();
      f.resize(at + 12 + body.size());
      put32(f, at, (uint32_t)body.size());
      std::memcpy(&f[at + 4], tag, 4);
      if (!body.empty()) std::memcpy(&f[at + 8], body.data(), body.size());
      put32(f, at + 8 + body.size(), (uint32_t)crc32(crc32(0L, Z_NULL, 0), &f[at + 4], 4 + (uInt)body.size()));
    };
    long ok = 0, bad = 0;
    for (int it = 0; it < 3000; it++) {
      const int* cd = combos[rng() % 15];
      const uint32_t w = 1 + rng() % 40, h = 1 + rng() % 40;
      const int interlace = rng() % 2, channels = cd[0] == 0 ? 1 : cd[0] == 2 ? 3 : cd[0] == 3 ? 1 : cd[0] == 4 ? 2 : 4;
      const int bits = cd[1] * channels;
      size_t total = 0;
      static const int x0[7] = {0, 4, 0, 2, 0, 1, 0}, y0[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1},
                       dy[7] = {8, 8, 8, 4, 4, 2, 2};
      std::vector<size_t> row_at;   // offsets of the filter bytes
      if (!interlace) {
        for (size_t y = 0; y < h; y++) row_at.push_back(y * (((size_t)w * bits + 7) / 8 + 1));
        total = (size_t)h * (((size_t)w * bits + 7) / 8 + 1);
      } else {
        for (int p = 0; p < 7; p++) {
          const size_t pw = (w + dx[p] - 1 - x0[p]) / dx[p], ph = (h + dy[p] - 1 - y0[p]) / dy[p];
          if (!pw || !ph) continue;
          for (size_t y = 0; y < ph; y++) row_at.push_back(total + y * ((pw * bits + 7) / 8 + 1));
          total += ph * ((pw * bits + 7) / 8 + 1);
        }
      }
      if (rng() % 10 == 0) total += (rng() % 7) - 3;   // sometimes the wrong amount of data
      std::vector<uint8_t> raw(total);
      for (auto& b : raw) b = (uint8_t)rng();
      const bool bad_filter = rng() % 20 == 0;          // filter bytes: valid unless this file is a bad-filter case
      for (size_t at : row_at)
        if (at < raw.size()) raw[at] = (uint8_t)(rng() % (bad_filter ? 7 : 5));
      uLongf zl = compressBound((uLong)raw.size());
      std::vector<uint8_t> z(zl);
      compress2(z.data(), &zl, raw.data(), (uLong)raw.size(), 1);
      z.resize(zl);
      std::vector<uint8_t> f = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
      std::vector<uint8_t> ihdr(13);
      put32(ihdr, 0, w);
      put32(ihdr, 4, h);
      ihdr[8] = (uint8_t)cd[1];
      ihdr[9] = (uint8_t)cd[0];
      ihdr[12] = (uint8_t)interlace;
      chunk(f, "IHDR", ihdr);
      if (cd[0] == 3 || rng() % 8 == 0) {
        std::vector<uint8_t> pal(3 * (rng() % 257));
        for (auto& b : pal) b = (uint8_t)rng();
        chunk(f, "PLTE", pal);
      }
      chunk(f, "IDAT", z);
      chunk(f, "IEND", {});
      std::vector<uint8_t> out((size_t)w * h + 32, 0xCD);
      const kvfe_status sd = kvfe_png_decode_gray(f.data(), f.size(), out.data(), w, (int32_t)w, (int32_t)h);
      (sd == KVFE_OK ? ok : bad)++;
      for (size_t k = (size_t)w * h; k < out.size(); k++)
        if (out[k] != 0xCD) return 7;
    }
    if (ok < 2000) return 8;   // most of these are valid files
    std::printf("random containers: %ld decoded, %ld refused\n", ok, bad);
  }
  // the batch decoder from three caller threads at once (one gets the worker pool, the others find it busy and bring
  // threads of their own), every result checked
  {
    std::ifstream in(argv[1], std::ios::binary);   // (argv[1] is a PNG file)
    const std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    int32_t w = 0, h = 0, c = 0;
    kvfe_png_info(file.data(), file.size(), &w, &h, &c);
    std::vector<uint8_t> ref((size_t)w * h);
    if (kvfe_png_decode_gray(file.data(), file.size(), ref.data(), (size_t)w, w, h) != KVFE_OK) return 9;
    std::atomic<int> failures{0};
    auto caller = [&](int id) {
      const int n = 6 + id;
      std::vector<std::vector<uint8_t>> outs(n, std::vector<uint8_t>((size_t)w * h));
      std::vector<const uint8_t*> data(n, file.data());
      std::vector<size_t> sizes(n, file.size());
      std::vector<uint8_t*> dst(n);
      for (int rep = 0; rep < 6; rep++) {
        for (int i = 0; i < n; i++) {
          std::fill(outs[i].begin(), outs[i].end(), 0);
          dst[i] = outs[i].data();
        }
        if (kvfe_png_decode_gray_batch(data.data(), sizes.data(), dst.data(), (size_t)w, w, h, n, 2 + id, nullptr) != KVFE_OK)
          failures++;
        for (int i = 0; i < n; i++)
          if (outs[i] != ref) failures++;
      }
    };
    std::thread t0(caller, 0), t1(caller, 1), t2(caller, 2);
    t0.join();
    t1.join();
    t2.join();
    if (failures) return 10;
  }
  // CSV parsers on mutated text
  const std::string imu = "#h\n1,0.1,0.2,0.3,1,2,3\n2,0.1,0.2,0.3,1,2,3\n3,1e-3,-2E2,.5,9.81,0,-0\n";
  const std::string cam = "#h\n100,100.png\n200,200.png\r\n300,300.png\n";
  for (int it = 0; it < 2000; it++) {
    std::string t = it % 2 ? imu : cam;
    for (int k = 0; k < 1 + (int)(rng() % 4); k++) t[rng() % t.size()] = (char)(rng() % 256);
    if (rng() % 4 == 0) t.resize(rng() % t.size());
    int64_t ts[8];
    double ag[48];
    int32_t n = 0;
    kvfe_euroc_parse_camera_csv(t.data(), t.size(), ts, 8, &n);
    kvfe_euroc_parse_imu_csv(t.data(), t.size(), ts, ag, 8, &n);
    kvfe_euroc_parse_imu_csv(t.data(), t.size(), ts, ag, 1, &n);
  }
  // synchroniser under random traffic with small output capacities
  for (int mode = 0; mode < 3; mode++) {
    kvfe_stereo_sync* s = kvfe_stereo_sync_create(mode == 1 ? 40 : -1);
    kvfe_stereo_sync_set_mode(s, mode);
    int64_t ti = 1, tc = 1;
    for (int it = 0; it < 5000; it++) {
      const int ev = rng() % 10;
      const double v[6] = {1, 2, 3, 4, 5, 6};
      if (ev < 5) kvfe_stereo_sync_fill_imu(s, ti += (int)(rng() % 5) - 1, v);
      else if (ev < 7) { kvfe_stereo_sync_fill_left(s, tc += (int)(rng() % 8) - 2, it); if (rng() % 5) kvfe_stereo_sync_fill_right(s, tc, it); }
      else {
        int64_t st[4];
        double ag[24];
        kvfe_sync_packet p;
        int32_t r = kvfe_stereo_sync_next(s, &p, st, ag, (int32_t)(rng() % 4) + 1);
        if (r == -1) {                                  // capacity too small: the count is reported, retry with room
          std::vector<int64_t> st2((size_t)p.n_imu);
          std::vector<double> ag2((size_t)6 * p.n_imu);
          r = kvfe_stereo_sync_next(s, &p, st2.data(), ag2.data(), p.n_imu);
          if (r == -1) return 6;
        }
      }
    }
    kvfe_stereo_sync_destroy(s);
  }
  std::printf("OK decoded=%ld refused=%ld\n", decoded, refused);
  return 0;
}
