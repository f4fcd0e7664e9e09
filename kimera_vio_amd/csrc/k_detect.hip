// K2a-c  Shi-Tomasi min-eigenvalue + 3x3 local maxima + masked maximum   (cv::goodFeaturesToTrack)
// K2d    quality threshold, greedy min-distance filter, sort, maxCorners (cv::goodFeaturesToTrack)
//        + ANMS (AdaptiveNonMaximumSuppression::suppressNonMax: none / TopN / Binning)
// K3     cv::cornerSubPix + append to the frame (FeatureDetector::featureDetection)
// reference: src/frontend/feature-detector/FeatureDetector.cpp:94-299,
//            src/frontend/feature-detector/NonMaximumSuppression.cpp:33-169
//
// The dense kernel never materialises the eigenvalue map: whether a pixel is a 3x3 local maximum
// does not depend on the quality threshold (a neighbour larger than a pixel above the threshold
// is itself above it), so one pass emits (lambda, index) for every non-zero local maximum under
// the mask plus the masked global maximum; the per-stream select kernel applies
// lambda > maxVal*quality afterwards on the compacted list.  HBM traffic per image: one read of
// the image (+ the optional user mask); the detection mask "255 minus filled discs around the
// tracked keypoints" is evaluated analytically from the keypoint list (exact cv::circle spans).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include "kvfe_dev.hpp"

#include <type_traits>
#include <utility>

namespace kvfe {

// order-preserving float -> uint key (handles negative values), 0 is reserved for "none"
__device__ __forceinline__ unsigned fkey(float f) {
  unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
  if (k == 0) return 0.0f;
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ---------------------------------------------------------------------------------------------
// One wavefront streams down a strip of 58 output columns x ME_ROWS rows (64 lanes = 58 columns
// + 3 halo columns each side).  Nothing but the detection mask and the candidate buffer touches
// LDS: horizontal neighbours come from DPP wave shifts, vertical neighbours from three rotating
// register rows.  Per source row r the pipeline emits
//   pixel row r  ->  Sobel partials dh(r), sm(r)           (horizontal difference / smoothing)
//   cov row  r-1 ->  dx*dx, dx*dy, dy*dy and their horizontal 3-sums h(r-1) in float64
//   box row  r-2 ->  lambda(r-2) and its horizontal 3-max hm(r-2)
//   row      r-3 ->  3x3 local-maximum test, candidates appended to an LDS list
// with BORDER_REFLECT_101 applied exactly where cv::Sobel / cv::boxFilter apply it.
// The float operations and their order are those of cv::cornerMinEigenVal (see oracle).
// ---------------------------------------------------------------------------------------------
constexpr int ME_HALO = 3;
constexpr int ME_COLS = 64 - 2 * ME_HALO;  // 58 output columns per wave
constexpr int ME_ROWS = 64;                // most output rows per wave (rows per strip is a launch parameter)
constexpr int ME_LCAP = 256;               // LDS candidate buffer (flushed when nearly full); small = 8 waves per SIMD
constexpr int ME_NSLOT = 6;                // source rows in flight per lane (accumulation registers a0 .. a5)
constexpr int ME_WAVES_PER_SIMD = 5;       // resident waves of the min-eigenvalue launch (its register count admits 5)

__device__ __forceinline__ float dpp_from_left(float v) {   // lane i <- lane i-1
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float v) {  // lane i <- lane i+1
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// Correctly rounded sqrtf for 0 <= x < 2^96 (every lambda discriminant: sums of squares of scaled
// 8-bit gradients).  v_sqrt_f32 is good to 1 ulp; the two FMA residuals pick the neighbour exactly as
// the compiler's own IEEE expansion does.  The input is always scaled by 2^32 (exact; the expansion
// scales only below 2^-96 to keep the residuals normal) and zero needs no special case: its
// "neighbour below" is a NaN pattern whose residual compares false, its neighbour above leaves +0.
__device__ __forceinline__ float sqrt_rn_small(float x) {
  const float xs = x * 4294967296.0f;
  float s = __builtin_amdgcn_sqrtf(xs);
  const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
  const float ed = __builtin_fmaf(-sd, s, xs), eu = __builtin_fmaf(-su, s, xs);
  s = ed <= 0.f ? sd : s;
  s = eu > 0.f ? su : s;
  return s * 1.52587890625e-05f;
}
__device__ __forceinline__ float vmaxf(float a, float b) {  // v_max_f32 without canonicalising moves
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

struct MeRow {          // per-lane values of one image row at the different pipeline stages
  float dh, sm;         // Sobel partials of the pixel row
  double h0, h1, h2;    // horizontal 3-sums of dx*dx, dx*dy, dy*dy
  float hm, lam;        // horizontal 3-max of lambda, lambda
};

// ---------------------------------------------------------------------------------------------
// Rows nobody needs (shared by the kernel below and by its CPU statement in the comments): lambda matters only at
// pixels that pass the detection mask (masked maximum, candidates) and at their 8 neighbours (the 3x3 maximum):
// U(y) = row y of the strip has an unmasked output pixel; box row b is needed iff U(b-1) | U(b) | U(b+1), cov row c iff
// one of the box rows c-1 .. c+1 is.  With a few hundred tracked keypoints and discs of radius min_distance most of a
// frame is masked, and whole 58-pixel row segments drop out (cv::goodFeaturesToTrack computes them and throws them
// away).  Bit i + 2 of the 128-bit masks = strip row i; u_lo / u_hi = U of strip rows 0..63 / 64..127 (strip_rows <= 120,
// so nothing falls off the top).
// ---------------------------------------------------------------------------------------------
struct MeNeed {
  unsigned long long c0, c1, b0, b1, p0, p1;
};
__device__ __forceinline__ MeNeed me_need_masks(unsigned long long u_lo, unsigned long long u_hi) {
  const unsigned long long U0 = u_lo << 2, U1 = (u_hi << 2) | (u_lo >> 62);
  auto or3 = [](unsigned long long x0, unsigned long long x1, unsigned long long& y0, unsigned long long& y1) {
    y0 = x0 | (x0 << 1) | (x0 >> 1) | (x1 << 63);
    y1 = x1 | (x1 << 1) | (x0 >> 63) | (x1 >> 1);
  };
  MeNeed n;
  or3(U0, U1, n.c0, n.c1);
  or3(n.c0, n.c1, n.b0, n.b1);
  // cov row c (bit c - ys + 2 of needb) reads the pixel rows c-1 .. c+1: bits q-2, q-1, q of needb -> bit q of needp
  // (bit q = pixel row ys - 3 + q; the highest cov bit is 123, nothing falls off the top)
  n.p0 = n.b0 | (n.b0 << 1) | (n.b0 << 2);
  n.p1 = n.b1 | (n.b1 << 1) | (n.b1 << 2) | (n.b0 >> 63) | (n.b0 >> 62);
  return n;
}

// ---------------------------------------------------------------------------------------------
// Detection mask as a bitmap (round 5): one block per stream rasterises the cv::circle discs of the stream's tracked
// keypoints (FeatureDetector.cpp:185-203) into bits "pixel masked OUT" in HBM -- bit x + 64 of row y, rows of MW 64-bit
// words.  Reader: fast_kernel (one bit test per FAST corner).
// The min-eigenvalue launch does NOT use it.  Two forms of this round did (tools/r5/gpu_a.sh .. gpu_c.sh: the bitmap plus
// cost-ordered work items pulled by resident waves, first off one atomic counter -- 10 k returning device-scope atomics
// on one word take 0.11 ms, 88 per microsecond --, then off one counter per stream a memory channel apart): both were
// bit-exact and both lost to round 4's launch, 0.097 - 0.237 ms against 0.079 -- this extra launch and a wave's
// dependent round trips (stream costs -> item -> mask words -> rows) cost more than the mask phase they replace, and the
// launch was bound by the memory latency of its longest strip either way, which the six-deep row requests below address.
// ---------------------------------------------------------------------------------------------
constexpr int MEP_T = 1024;
__global__ __launch_bounds__(MEP_T) void detect_mask_kernel(
    int W, int H, int kcap, int radius, const int* __restrict__ circle_hw, const float2* __restrict__ kp_all,
    const long long* __restrict__ lmk_all, const int* __restrict__ kp_count, int use_discs,
    const int* __restrict__ flags, unsigned long long* __restrict__ maskbits, int MW, int band_rows) {
  const int s = blockIdx.x, tid = threadIdx.x;
  if (flags && !(flags[s] & FLAG_DETECT)) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char mep_lds[];
  __shared__ int hw_s[MAX_RADIUS + 1];
  unsigned long long* band = reinterpret_cast<unsigned long long*>(mep_lds);              // [band_rows][MW]
  int* cxy = reinterpret_cast<int*>(mep_lds + sizeof(unsigned long long) * (size_t)band_rows * MW);   // [kcap] x | y << 16
  unsigned long long* MB = maskbits + (size_t)s * H * MW;
  const int nk = use_discs ? min(kp_count[s], kcap) : 0;
  const int nr = 2 * radius + 1;
  for (int i = tid; i <= radius && i <= MAX_RADIUS; i += MEP_T) hw_s[i] = circle_hw[i];
  // only keypoints with a landmark mask (FeatureDetector.cpp:191); cv::Point(Point2f) rounds.  Centres far outside the
  // image cannot touch it and are dropped (the packed form holds +-16 K around the image)
  for (int i = tid; i < nk; i += MEP_T) {
    const float2 k = kp_all[(size_t)s * kcap + i];
    const int cx = __float2int_rn(k.x), cy = __float2int_rn(k.y);
    const bool on = lmk_all[(size_t)s * kcap + i] != -1 && cx + radius >= 0 && cx - radius < W && cy + radius >= 0 &&
                    cy - radius < H;
    cxy[i] = on ? ((cx + 16384) & 0xffff) | ((cy + 16384) << 16) : -1;
  }
  for (int y0 = 0; y0 < H; y0 += band_rows) {
    const int y1 = min(y0 + band_rows, H);
    __syncthreads();
    for (int i = tid; i < (y1 - y0) * MW; i += MEP_T) band[i] = 0ull;
    __syncthreads();
    for (int idx = tid; idx < nk * nr; idx += MEP_T) {
      const int d = idx / nr, dy = idx - d * nr - radius;
      const int c = cxy[d];
      if (c == -1) continue;
      const int cx = (c & 0xffff) - 16384, cy = (int)((unsigned)c >> 16) - 16384;
      const int y = cy + dy;
      if (y < y0 || y >= y1) continue;
      const int hw = hw_s[abs(dy)];
      const int pa = max(cx - hw, 0) + 64, pb = min(cx + hw, W - 1) + 64;
      if (pa > pb) continue;
      unsigned long long* row = band + (size_t)(y - y0) * MW;
      for (int w = pa >> 6; w <= (pb >> 6); w++) {
        const int lo = max(pa - (w << 6), 0), hi = min(pb - (w << 6), 63);
        const unsigned long long m = (hi - lo == 63) ? ~0ull : (((1ull << (hi - lo + 1)) - 1ull) << lo);
        atomicOr(&row[w], m);
      }
    }
    __syncthreads();
    for (int i = tid; i < (y1 - y0) * MW; i += MEP_T) MB[(size_t)y0 * MW + i] = band[i];
  }
}

// ---------------------------------------------------------------------------------------------
// mineig2_kernel: the same pipeline and the same arithmetic, restructured around what bounds it (instruction issue,
// every category: profiles/r2_v5_lk_analysis.md, tools/ubench/valu_rate.hip):
//   * the strip's rows are split into a prologue, a STEADY part and an epilogue.  Every stage is live and no row
//     reflects in the steady part, so its steps carry no range test, no border copy and no row-index arithmetic (the
//     old step spent ~45 scalar instructions and ~10 branches per row on them); the checked step runs only on the few
//     rows at either end of a strip and on strips at the top / bottom of the image;
//   * the source byte of a row comes from a raw buffer load: per-lane column offset in a VGPR, row offset in an SGPR that
//     advances by one scalar add;
//   * the row's detection-mask word is read from LDS one step ahead and stays in VGPRs until it is used;
//   * lane predicates that are wave-wide facts (detection mask, image columns 1..W-2) live as 64-bit scalar masks and
//     are combined with scalar ANDs; only the two float compares of the local-maximum test are vector instructions.
// ---------------------------------------------------------------------------------------------
typedef int me_v4i __attribute__((ext_vector_type(4)));
constexpr int ME2_ROWS = 120;             // most output rows per wave of mineig2_kernel (row-need masks: 128 bits incl. margins)

// -DKVFE_ME_PROF (debugging aid, tools/r4/gpu_ac.sh): cycle stamps per wave of the last launch.  Round 4, 64 x 752x480 with
// ~590 keypoints per stream: mean wave 73 k cycles (mask phase 26 k, row loop 43 k with 25 of 85 pixel rows needed,
// epilogue 4 k), SLOWEST wave 175 k -- a strip with no keypoint near it needs every row, and the launch (one round of
// waves) lasts as long as that wave.  Strips of 60 / 40 rows: slowest 154 / 155 k, launch 0.083 / 0.092 ms against 0.079 (the
// mask phase does not shrink with the strip: 23 k).  Two waves per block sharing one mask phase, each walking half the
// strip (tools/r4/gpu_ad.sh, 109 parity tests green): 0.087 against 0.080 ms, real frames 0.110 against 0.089,
// 1280x720 0.154 against 0.113 -- not kept.
#ifdef KVFE_ME_PROF
constexpr size_t KVFE_ME_PROF_WAVES = 8192;
__device__ unsigned long long kvfe_me_prof[KVFE_ME_PROF_WAVES * 8];   // per wave: cycles of the mask phase, the row loop, the epilogue; items; needed pixel rows; rows walked; start; end
#endif
// Launch shape: one single-wave block per (column strip, row strip, stream) item, every wave of the launch resident at
// once (rounds 3-5).  Round 5 measured three other shapes, all bit-exact, all slower (tools/r5/gpu_a.sh .. gpu_e.sh,
// profiles/r5_analysis.md): resident waves pulling cost-ordered items off one global counter (0.20 - 0.24 ms: 10 k
// returning device-scope atomics on one word take 0.11 ms), off one counter per stream with a per-stream disc bitmap from
// a prep launch (0.097 - 0.114 ms), and two 8-wave blocks per compute unit pulling items off an LDS counter (0.085 - 0.104
// ms) -- against 0.079 ms.  The launch ends with its longest item either way (a strip that needs all of its rows: 85
// rows at ~1.75 k cycles each, the issue rate of a SIMD shared by five waves -- not memory latency: six rows in flight
// instead of three changed nothing), and shorter strips pay the mask phase (22 k cycles, ~600 keypoint tests) once more
// per item.
#define KVFE_ME_WAVE_SYNC() __syncthreads()   /* (a block is one wave) */
template <bool HAS_MASK, bool HARRIS>
__global__ __launch_bounds__(64) void mineig2_kernel(
    const unsigned char* __restrict__ img, size_t row_stride, size_t img_stride,
    const unsigned char* __restrict__ user_mask, int W, int H, int kcap, int ccap, int radius,
    const int* __restrict__ circle_hw, const float2* __restrict__ kp_all,
    const long long* __restrict__ lmk_all, const int* __restrict__ kp_count, int use_discs,
    const int* __restrict__ flags,
    unsigned long long* __restrict__ cand_all, int* __restrict__ cand_count,
    unsigned int* __restrict__ maxkey, int strip_rows, int B, int nx, int ny, float harris_kf, double harris_kd) {
  // block -> (column strip, row strip, stream).  (An XCD-banded 1-D order was measured in round 3: 0.084 against 0.082 ms.)
  const int s = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
  if (flags && !(flags[s] & FLAG_DETECT)) return;
#ifdef KVFE_ME_PROF
  const unsigned long long me_t0 = __builtin_readcyclecounter();
#endif
  __shared__ unsigned long long rowmask[132];  // [row of the strip] bit l set: lane l's column is masked OUT; 128: read-ahead slot
  __shared__ unsigned long long lcand[ME_LCAP];
  __shared__ int hw_s[MAX_RADIUS + 1];
  const int lane = threadIdx.x;
  for (int i = lane; i <= radius && i <= MAX_RADIUS; i += 64) hw_s[i] = circle_hw[i];
  __syncthreads();

  const unsigned char* I = img + (size_t)s * img_stride;
  const unsigned char* M = HAS_MASK ? user_mask + (size_t)s * W * H : nullptr;
  const int xs = bx * ME_COLS, ys = by * strip_rows;
  const int ye = min(ys + strip_rows, H);
  const int x0 = xs - ME_HALO;          // column of lane 0
  const int gx = x0 + lane;
  const int cxr = reflect101(gx, W);    // source column (BORDER_REFLECT_101)
  const bool out_col = lane >= ME_HALO && lane < 64 - ME_HALO && gx < W;
  const bool at_left = gx == 0, at_right = gx == W - 1;

  unsigned long long needc0 = ~0ull, needc1 = ~0ull, needb0 = ~0ull, needb1 = ~0ull;   // wave-uniform
  unsigned long long needp0 = ~0ull, needp1 = ~0ull;   // pixel rows somebody needs: bit q = row ys - 3 + q
  // detection mask: the cv::circle discs that touch this strip, rasterised into row bit-masks.  Lane = row of the strip
  // (two rows per lane for strips above 64 rows): the keypoints are tested 64 at a time, then every hit is replayed for all
  // rows at once from scalar registers -- no atomics, no per-row loop (the per-keypoint row loop of the first kernel
  // was a quarter of its instructions).
  {
    unsigned long long mrow0 = 0ull, mrow1 = 0ull;
    if (use_discs) {
      const float2* kp = kp_all + (size_t)s * kcap;
      const int nk = kp_count[s];
      const long long* lmk = lmk_all + (size_t)s * kcap;
      const int gy0 = ys + lane, gy1 = ys + 64 + lane;
      // the keypoint list is read in chunks of 8 x 64 entries, all requests of a chunk in flight at once: one memory
      // round trip per chunk instead of one per 64 keypoints (ten dependent round trips were a third of a wave's life)
      constexpr int CH = 8;
      for (int base = 0; base < nk; base += 64 * CH) {
        float2 pk[CH];
        long long lk[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
          if (base + 64 * j < nk) {   // (wave-uniform: the groups past the end of the list are neither read nor tested)
            const int i = min(base + 64 * j + lane, nk - 1);
            lk[j] = lmk[i];
            pk[j] = kp[i];
          }
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
          if (base + 64 * j >= nk) break;
          const int i = base + 64 * j + lane;
          // only keypoints with a landmark mask (FeatureDetector.cpp:191); cv::Point(Point2f) rounds
          const int cx = __float2int_rn(pk[j].x), cy = __float2int_rn(pk[j].y);
          const bool hit = i < nk && lk[j] != -1 &&
                           !(cx + radius < x0 || cx - radius >= x0 + 64 || cy + radius < ys || cy - radius >= ye);
          unsigned long long bal = __ballot(hit);
          while (bal) {
            const int l = __builtin_ctzll(bal);
            bal &= bal - 1;
            const int ccx = __builtin_amdgcn_readlane(cx, l), ccy = __builtin_amdgcn_readlane(cy, l);
            auto span = [&](int gy) -> unsigned long long {
              const int dy = abs(gy - ccy);
              if (dy > radius) return 0ull;
              const int hw = hw_s[dy];
              const int xa = max(ccx - hw, x0) - x0, xb = min(ccx + hw, x0 + 63) - x0;
              if (xa > xb) return 0ull;
              return (~0ull >> (63 - (xb - xa))) << xa;   // bits xa .. xb (0 <= xa <= xb <= 63)
            };
            // (a disc of a strip above 64 rows touches its rows 0..63, its rows 64.., or both: wave-uniform tests)
            if (ccy - radius <= ys + 63) mrow0 |= span(gy0);
            if (strip_rows > 64 && ccy + radius >= ys + 64) mrow1 |= span(gy1);
          }
        }
      }
    }
    rowmask[lane] = mrow0;
    rowmask[64 + lane] = mrow1;   // (rows past the strip: all zero; slot 128 is the read-ahead slot)
    if (lane == 0) rowmask[128] = 0ull;
    // rows nobody needs: me_need_masks above
    const unsigned long long om = __ballot(out_col);
    const unsigned long long u_lo = __ballot(lane < strip_rows && (om & ~mrow0) != 0ull);
    const unsigned long long u_hi = __ballot(64 + lane < strip_rows && (om & ~mrow1) != 0ull);
    const MeNeed nd = me_need_masks(u_lo, u_hi);
    needc0 = nd.c0, needc1 = nd.c1, needb0 = nd.b0, needb1 = nd.b1, needp0 = nd.p0, needp1 = nd.p1;
  }
  KVFE_ME_WAVE_SYNC();
#ifdef KVFE_ME_PROF
  const unsigned long long me_t1 = __builtin_readcyclecounter();
#endif

  const float f1 = (float)(1.0 / (4.0 * 3.0 * 255.0));  // (float)scale, blockSize 3, ksize 3
  const float f0 = 2.0f * f1;
  // rows: hm rows b0..b1, cov/h rows c0..c1, pixel rows c0-1..c1+1 (reflected outside the image)
  const int b0 = max(ys - 1, 0), b1 = min(ye, H - 1);
  const int c0 = max(b0 - 1, 0), c1 = min(b1 + 1, H - 1);
  const int r_first = c0 - 1, r_last = min(ye + 2, H + 1);
  const int lm0 = max(ys, 1), lm1 = min(ye - 1, H - 2);
  // steady steps r: cov row r-1, box row r-2 (inside the strip, not an image border row), local-max row r-3 are all
  // live and the row requested by the step, r + ME_NSLOT, is a plain image row
  const int rs = max(max(c0 + 1, max(max(b0, ys), 1) + 2), lm0 + 3);
  const int re = min(min(c1 + 1, min(min(b1, ye - 1), H - 2) + 2), min(lm1 + 3, H - 1 - ME_NSLOT));

  float bestv = -__builtin_inff();  // masked maximum of lambda (no pixel has lambda = -inf)
  int n_loc = 0;                    // wave-uniform
  auto flush = [&]() {
    int base = 0;
    if (lane == 0 && n_loc > 0) base = atomicAdd(&cand_count[s], n_loc);
    base = __shfl(base, 0);
    for (int i = lane; i < n_loc; i += 64) {
      const int pos = base + i;
      if (pos < ccap) cand_all[(size_t)s * ccap + pos] = lcand[i];
    }
    n_loc = 0;
  };

  // source rows: raw buffer, per-lane column in a VGPR, row offset in an SGPR (an image is < 2 GiB).  SIX rows in flight
  // per lane (one per rotating slot), issued and awaited by hand: vmcnt counts in issue order, so "at most five
  // outstanding" means this slot's load has landed.  (Rounds 3-4 kept three in flight; six measured the same launch
  // time, 0.078 against 0.079 ms, tools/r5/gpu_d.sh: a row costs ~1.75 k cycles because five waves share the SIMD's
  // issue slots, not because its byte is late.)
  const unsigned stride_u = (unsigned)row_stride;
  const unsigned long long ibase = (unsigned long long)(size_t)I;
  const me_v4i rsrc = {(int)(unsigned)ibase, (int)(unsigned)((ibase >> 32) & 0xffffu),
                       (int)(stride_u * (unsigned)(H - 1) + (unsigned)W), 0x00027000};
  const unsigned voff = (unsigned)cxr;
  const unsigned char* mcol = HAS_MASK ? M + min(max(gx, 0), W - 1) : nullptr;
  auto row_of = [&](int r) {  // BORDER_REFLECT_101 for r in [-1, H], clamped for the unused rows beyond
    int rr = r < 0 ? -r : r;
    rr = rr >= H ? 2 * H - 2 - rr : rr;
    return (unsigned)min(max(rr, 0), H - 1);
  };
  // The destination of a request is an ACCUMULATION register (a0 / a1 / a2, one per rotating slot; gfx950 loads can
  // target them): hipcc never allocates AGPRs in this kernel, so nothing it generates can touch a register with a request
  // in flight.  (With a VGPR destination bound to a C++ variable hipcc is free to copy that variable -- a phi move at a
  // loop header, a tied asm operand -- before the byte has landed: it cannot know that the asm statement's output is
  // written late.  tools/check_inflight_regs.py checks the ISA.)
#define KVFE_ME_ISSUE(N_) asm volatile("buffer_load_ubyte a" #N_ ", %0, %1, %2 offen" : : "v"(voff), "s"(rsrc), "s"(soff) : "a" #N_, "memory")
#define KVFE_ME_TAKE(N_) asm volatile("s_waitcnt vmcnt(5)\n\tv_accvgpr_read_b32 %0, a" #N_ "\n\tv_cvt_f32_ubyte0 %0, %0" : "=v"(p) : : "a" #N_, "memory")
  auto issue_off = [&](auto slot_tag, unsigned soff) {
    constexpr int SLOT = decltype(slot_tag)::value;
    static_assert(SLOT >= 0 && SLOT < ME_NSLOT, "slot");
    if constexpr (SLOT == 0) KVFE_ME_ISSUE(0);
    else if constexpr (SLOT == 1) KVFE_ME_ISSUE(1);
    else if constexpr (SLOT == 2) KVFE_ME_ISSUE(2);
    else if constexpr (SLOT == 3) KVFE_ME_ISSUE(3);
    else if constexpr (SLOT == 4) KVFE_ME_ISSUE(4);
    else KVFE_ME_ISSUE(5);
  };
  // wait for the slot's request (at most ME_NSLOT - 1 younger ones outstanding: vmcnt counts in issue order) and convert
  // the byte
  auto take = [&](auto slot_tag) -> float {
    constexpr int SLOT = decltype(slot_tag)::value;
    float p;
    if constexpr (SLOT == 0) KVFE_ME_TAKE(0);
    else if constexpr (SLOT == 1) KVFE_ME_TAKE(1);
    else if constexpr (SLOT == 2) KVFE_ME_TAKE(2);
    else if constexpr (SLOT == 3) KVFE_ME_TAKE(3);
    else if constexpr (SLOT == 4) KVFE_ME_TAKE(4);
    else KVFE_ME_TAKE(5);
    return p;
  };
#undef KVFE_ME_ISSUE
#undef KVFE_ME_TAKE
  using SL0 = std::integral_constant<int, 0>;
  using SL1 = std::integral_constant<int, 1>;
  using SL2 = std::integral_constant<int, 2>;
  using SL3 = std::integral_constant<int, 3>;
  using SL4 = std::integral_constant<int, 4>;
  using SL5 = std::integral_constant<int, 5>;
  // the six requests of a (re)start at pixel row r, in row order; hi: row r belongs to slot 3 (the slots rotate with the
  // row index, slot(r) = (r - r_first) % 6, and the walk advances in threes, so a start is on slot 0 or on slot 3)
  auto issue_six = [&](int r, bool hi) {
    if (!hi) {
      issue_off(SL0{}, row_of(r) * stride_u);
      issue_off(SL1{}, row_of(r + 1) * stride_u);
      issue_off(SL2{}, row_of(r + 2) * stride_u);
      issue_off(SL3{}, row_of(r + 3) * stride_u);
      issue_off(SL4{}, row_of(r + 4) * stride_u);
      issue_off(SL5{}, row_of(r + 5) * stride_u);
    } else {
      issue_off(SL3{}, row_of(r) * stride_u);
      issue_off(SL4{}, row_of(r + 1) * stride_u);
      issue_off(SL5{}, row_of(r + 2) * stride_u);
      issue_off(SL0{}, row_of(r + 3) * stride_u);
      issue_off(SL1{}, row_of(r + 4) * stride_u);
      issue_off(SL2{}, row_of(r + 5) * stride_u);
    }
  };
  issue_six(r_first, false);
  const bool edge_strip = x0 <= 0 || x0 + 64 >= W;  // strip contains column 0 or W-1 (wave-uniform)
  // wave-wide lane facts as scalar masks
  const unsigned long long out_mask = __ballot(out_col);
  const unsigned long long colok_mask = __ballot(gx >= 1 && gx < W - 1);
  unsigned long long in_b_mask = 0ull;   // lanes whose pixel of the last box row passes the detection mask
  // the detection-mask word of the next box row, read one step ahead; the per-lane zero keeps the value in VGPRs (a
  // uniform LDS read would be moved to SGPRs -- and waited for -- right where it is issued)
  unsigned lane_zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
  unsigned long long mk_v = 0ull;
  auto fetch_mask = [&](int b) {   // row b of the strip's mask (clamped: rows outside the strip are never used)
    const int idx = min(max(b - ys, 0), 128);
    mk_v = rowmask[idx + lane_zero];
  };
  unsigned soff_next = 0;   // steady part: row offset of the next request

  auto row_needed = [&](unsigned long long m0, unsigned long long m1, int y) {   // bit (y - ys + 2) of a 128-bit row mask
    const int pos = y - ys + 2;
    if (pos < 0 || pos > 127) return false;
    return (((pos < 64 ? m0 : m1) >> (pos & 63)) & 1ull) != 0ull;
  };
  // one pipeline step; X = slot of pixel row r, Y = row r-1, Z = row r-2.  CHECK: the stages test their row ranges and
  // copy border rows (prologue, epilogue, image-border strips); otherwise every stage is live.  Only wave-uniform
  // branches: per-lane conditions are predicated, so every DPP sees all 64 lanes.
  auto step = [&](auto check_tag, auto slot_tag, int r, MeRow& X, MeRow& Y, MeRow& Z) {
    constexpr bool CHECK = decltype(check_tag)::value;
    const unsigned long long in_m_mask = in_b_mask;  // row m = r-3 was the box row of the previous step
    // ---- pixel row r -> Sobel partials (row r+1 is already being fetched) ------------------------
    {
      const float p = take(slot_tag);
      if (CHECK) {
        issue_off(slot_tag, row_of(r + ME_NSLOT) * stride_u);
      } else {
        issue_off(slot_tag, soff_next);
        soff_next += stride_u;
      }
      const float pl = dpp_from_left(p), pr = dpp_from_right(p);
      X.dh = pr - pl;
      float t = f1 * pl;
      t += f0 * p;
      t += f1 * pr;
      X.sm = t;
    }
    // the mask word of box row r-2 was requested by the previous step; request the next one now
    const unsigned long long mk_cur_v = mk_v;
    fetch_mask(r - 1);
    // ---- cov row c = r-1 -> horizontal float64 sums --------------------------------------------
    const int c = r - 1;
    if ((!CHECK || (c >= c0 && c <= c1)) && row_needed(needb0, needb1, c)) {
      const float dx = (Z.dh + X.dh) * f1 + Y.dh * f0;
      const float dy = X.sm - Z.sm;
      const float cxx = dx * dx, cxy = dx * dy, cyy = dy * dy;
      const float lxx = dpp_from_left(cxx), rxx = dpp_from_right(cxx);
      const float lxy = dpp_from_left(cxy), rxy = dpp_from_right(cxy);
      const float lyy = dpp_from_left(cyy), ryy = dpp_from_right(cyy);
      // BORDER_REFLECT_101 of cv::boxFilter: column -1 is column 1, column W is column W-2
      float axx = lxx, bxx = rxx, axy = lxy, bxy = rxy, ayy = lyy, byy = ryy;
      if (edge_strip) {
        axx = at_left ? rxx : lxx, bxx = at_right ? lxx : rxx;
        axy = at_left ? rxy : lxy, bxy = at_right ? lxy : rxy;
        ayy = at_left ? ryy : lyy, byy = at_right ? lyy : ryy;
      }
      // cv::boxFilter accumulates 0 + a + b + c in float64; "0 +" is dropped: it is exact for the
      // non-negative dx*dx / dy*dy sums and can only change the sign of a zero dx*dy sum, which
      // enters lambda squared.
      Y.h0 = ((double)axx + (double)cxx) + (double)bxx;
      Y.h1 = ((double)axy + (double)cxy) + (double)bxy;
      Y.h2 = ((double)ayy + (double)cyy) + (double)byy;
    }
    // ---- box row b = r-2 -> lambda, horizontal max; masked maximum ------------------------------
    const int b = r - 2;
    if (!row_needed(needc0, needc1, b)) {
      in_b_mask = 0ull;   // (no unmasked pixel in rows b-1 .. b+1: lambda of this row is never read)
    } else if (!CHECK || (b >= b0 && b <= b1)) {
      // rows b-1, b, b+1 = slots X, Z, Y; BORDER_REFLECT_101 at the image border: row -1 is row 1
      // (slot Y), row H is row H-2 (slot X) -- wave-uniform branches taken once per strip
      if (CHECK && b == 0) {
        asm volatile("");  // keep the once-per-strip border copies out of the row loop's selects
        X.h0 = Y.h0;
        X.h1 = Y.h1;
        X.h2 = Y.h2;
      }
      if (CHECK && b == H - 1) {
        asm volatile("");
        Y.h0 = X.h0;
        Y.h1 = X.h1;
        Y.h2 = X.h2;
      }
      const double s0 = (X.h0 + Z.h0) + Y.h0, s1 = (X.h1 + Z.h1) + Y.h1, s2 = (X.h2 + Z.h2) + Y.h2;
      float lam;
      if constexpr (HARRIS) {
        // cv::cornerHarris (use_harris_corner_detector_, FeatureDetector.cpp:78-79): calcHarris runs 4-wide float lanes
        // over the image as ONE row of W * H pixels -- a*c - b*b - (float)k*(a+c)*(a+c), every operation in float -- and
        // the (W * H) % 4 pixels at the very end go through its scalar tail, whose k is a double
        const float ha = (float)s0, hb = (float)s1, hc = (float)s2;
        const float ac = ha + hc;
        lam = (ha * hc - hb * hb) - (harris_kf * ac) * ac;
        if (CHECK && b == H - 1 && ((W * H) & 3) != 0) {
          const float ld = (float)((double)(ha * hc - hb * hb) - harris_kd * (double)ac * (double)ac);
          lam = gx >= W - ((W * H) & 3) ? ld : lam;
        }
      } else {
        const float fa = (float)s0 * 0.5f, fb = (float)s1, fc = (float)s2 * 0.5f;
        lam = (fa + fc) - sqrt_rn_small((fa - fc) * (fa - fc) + fb * fb);
      }
      Z.lam = lam;
      Z.hm = fmaxf(fmaxf(dpp_from_left(lam), lam), dpp_from_right(lam));
      if (!CHECK || (b >= ys && b < ye)) {
        // the row's bit mask is wave-uniform: it IS the lane predicate
        const unsigned long long mk = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(mk_cur_v >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mk_cur_v);
        unsigned long long in_mask = out_mask & ~mk;
        if (HAS_MASK) in_mask &= __ballot(mcol[(unsigned)(b * W)] != 0);
        in_b_mask = in_mask;
        // masked maximum under the mask as EXEC (all 64 lanes are live here): one VALU instruction instead of select + max
        asm volatile("s_mov_b64 exec, %2\n\tv_max_f32 %0, %0, %1\n\ts_mov_b64 exec, -1" : "+v"(bestv) : "v"(lam), "s"(in_mask));
      } else if (CHECK) {
        in_b_mask = 0ull;   // (box rows outside the strip feed only the 3x3 maximum)
      }
    }
    // ---- row m = r-3: 3x3 local maximum (rows m-1, m, m+1 = slots Y, X, Z) ----------------------
    const int m = r - 3;
    if (in_m_mask != 0ull && (!CHECK || (m >= lm0 && m <= lm1))) {
      const float v = X.lam;
      // the two float compares write scalar masks directly (the ballot idiom costs a select and a third compare)
      const float mx3 = fmaxf(fmaxf(Y.hm, X.hm), Z.hm);
      unsigned long long mne, meq;
      asm("v_cmp_neq_f32_e64 %0, 0, %1" : "=s"(mne) : "v"(v));
      asm("v_cmp_eq_f32_e64 %0, %1, %2" : "=s"(meq) : "v"(v), "v"(mx3));
      const unsigned long long bal = mne & meq & in_m_mask & colok_mask;
      if (bal) {
        if (__builtin_amdgcn_inverse_ballot_w64(bal)) {
          const int pos = n_loc + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
          lcand[pos] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(m * W + gx);
        }
        n_loc += __popcll(bal);
        if (n_loc > ME_LCAP - 64) {
          KVFE_ME_WAVE_SYNC();
          flush();
          KVFE_ME_WAVE_SYNC();
        }
      }
    }
  };
  using CK = std::integral_constant<bool, true>;
  using NC = std::integral_constant<bool, false>;

  MeRow S0, S1, S2;
  S0 = S1 = S2 = MeRow{0.f, 0.f, 0., 0., 0., 0.f, 0.f};
  // the row registers rotate with the row index in threes (S0, S1, S2), the request slots in sixes: slot(r) = (r - r_first) % 6
  int r = r_first;
  fetch_mask(r - 2);   // (the first step's "previous" request)
  {
    // RUNS OF NEEDED ROWS (round 4).  With a few hundred tracked keypoints and discs of radius min_distance a frame is
    // almost entirely masked (the 600-feature benchmark streams: 3 % of the pixels pass the mask, a quarter of a strip's
    // rows is needed by anybody), so the wave walks only the runs of pixel rows that some unmasked pixel needs -- it
    // does not even fetch the others -- instead of stepping through every row and gating the stages.  A run restarts the
    // pipeline: six row requests, the mask word one step ahead; the registers keep their phase (a run starts on a multiple
    // of three steps from r_first, at most two rows early), and what they hold from before the gap is never read: a
    // needed cov / box / local-maximum row has all of its source rows inside the same run (needp is needb widened by one
    // row, needb is needc widened by one, needc the unmasked rows widened by one).
    auto next_needed = [&](int from) -> int {   // first needed pixel row >= from (wave-uniform); INT_MAX: none
      int q = from - (ys - 3);
      q = q < 0 ? 0 : q;
      if (q < 64) {
        const unsigned long long m = (needp0 >> q) << q;
        if (m) return ys - 3 + __builtin_ctzll(m);
        if (needp1) return ys - 3 + 64 + __builtin_ctzll(needp1);
        return 0x7fffffff;
      }
      if (q >= 128) return 0x7fffffff;
      const unsigned long long m = (needp1 >> (q - 64)) << (q - 64);
      return m ? ys - 3 + 64 + __builtin_ctzll(m) : 0x7fffffff;
    };
    // three rows of the walk; hi = they sit in request slots 3 .. 5 (wave-uniform, alternates every three rows)
    auto three = [&](auto ck, int r0, bool hi) {
      constexpr bool CHECK = decltype(ck)::value;
      if (!hi) {
        step(ck, SL0{}, r0, S0, S2, S1);
        if (!CHECK || r0 + 1 <= r_last) step(ck, SL1{}, r0 + 1, S1, S0, S2);
        if (!CHECK || r0 + 2 <= r_last) step(ck, SL2{}, r0 + 2, S2, S1, S0);
      } else {
        step(ck, SL3{}, r0, S0, S2, S1);
        if (!CHECK || r0 + 1 <= r_last) step(ck, SL4{}, r0 + 1, S1, S0, S2);
        if (!CHECK || r0 + 2 <= r_last) step(ck, SL5{}, r0 + 2, S2, S1, S0);
      }
    };
    bool hi = false;
    // Structure as without the walk (checked prologue / unchecked steady part / checked epilogue: with both step variants
    // in ONE loop hipcc needs 96 instead of 62 VGPRs); the gaps are skipped inside the steady part, where almost all rows
    // of a strip lie.  A jump re-issues the six row requests (into the same accumulation registers: whatever is still in
    // flight for them lands first, requests return in order) and the mask word one step ahead.
    const int n_pro = rs > re ? 0x3fffffff : ((rs - r_first + 2) / 3) * 3;
    for (; r <= r_last && r - r_first < n_pro; r += 3, hi = !hi) three(CK{}, r, hi);
    if (r + 2 <= re) {
      soff_next = (unsigned)(r + ME_NSLOT) * stride_u;
      while (r + 2 <= re) {
        const int rn = next_needed(r);
        if (rn >= r + 3) {
          if (rn > r_last) {   // nothing needed any more in this strip
            r = r_last + 1;
            break;
          }
          const int jump = (rn - r) / 3;
          r += jump * 3;
          hi = hi != ((jump & 1) != 0);
          issue_six(r, hi);
          soff_next = (unsigned)(r + ME_NSLOT) * stride_u;
          fetch_mask(r - 2);
          in_b_mask = 0ull;
          continue;
        }
        three(NC{}, r, hi);
        r += 3;
        hi = !hi;
      }
    }
    for (; r <= r_last; r += 3, hi = !hi) three(CK{}, r, hi);
  }
  // the row requests issued beyond the last step are still in flight: they land before their registers are re-used
  asm volatile("s_waitcnt vmcnt(0)" : : : "a0", "a1", "a2", "a3", "a4", "a5", "memory");
#ifdef KVFE_ME_PROF
  const unsigned long long me_t2 = __builtin_readcyclecounter();
#endif
  KVFE_ME_WAVE_SYNC();
  flush();
  for (int off = 32; off > 0; off >>= 1) bestv = fmaxf(bestv, __shfl_xor(bestv, off));
  const unsigned bestkey = bestv == -__builtin_inff() ? 0u : fkey(bestv);
  // masked maximum: one global atomic per item, skipped when it cannot raise the maximum
  if (lane == 0 && bestkey &&
      bestkey > __hip_atomic_load(&maxkey[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(&maxkey[s], bestkey);
#ifdef KVFE_ME_PROF
  if (lane == 0) {   // one record per wave of the LAST launch (no atomics: they would be what is measured)
    const unsigned long long me_t3 = __builtin_readcyclecounter();
    const size_t w = ((size_t)s * gridDim.y + by) * gridDim.x + bx;
    if (w < KVFE_ME_PROF_WAVES) {
      unsigned long long* o = kvfe_me_prof + w * 8;
      o[0] = me_t1 - me_t0;
      o[1] = me_t2 - me_t1;
      o[2] = me_t3 - me_t2;
      o[3] = 1ull;
      o[4] = (unsigned long long)(__popcll(needp0) + __popcll(needp1));
      o[5] = (unsigned long long)(r_last - r_first + 1);
      o[6] = me_t0;
      o[7] = me_t3;
    }
  }
#endif
}
#undef KVFE_ME_WAVE_SYNC

void launch_mineig(const KParams& P, const Tables& T, const unsigned char* img, size_t row_stride,
                   size_t img_stride, const unsigned char* user_mask, const FrameTab& k,
                   const StreamState& S, const DetectScratch& D, int use_discs, hipStream_t st) {
  // D.cand_count / D.maxkey are zero here: allocated zeroed, re-zeroed by every select_kernel
  if (P.detector == 0) return launch_fast(P, T, img, row_stride, img_stride, user_mask, k, S, D, use_discs, st);   // FeatureDetectorType::FAST
#ifdef KVFE_ME_PROF
  {
    static bool reg = false;
    if (!reg) {
      reg = true;
      std::atexit([] {
        static unsigned long long h[KVFE_ME_PROF_WAVES * 8];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(kvfe_me_prof), sizeof(h)) != hipSuccess) return;
        double a[6] = {0, 0, 0, 0, 0, 0}, longest = 0, life = 0, waves = 0, idle = 0, first = 1e300, last = 0;
        for (size_t w = 0; w < KVFE_ME_PROF_WAVES; w++) {
          const unsigned long long* o = h + w * 8;
          if (!o[7]) continue;
          waves += 1;
          idle += o[3] ? 0 : 1;
          for (int i = 0; i < 6; i++) a[i] += (double)o[i];
          life += (double)(o[7] - o[6]);
          longest = std::max(longest, (double)(o[7] - o[6]));
          first = std::min(first, (double)o[6]);
          last = std::max(last, (double)o[7]);
        }
        if (a[3] > 0)
          std::fprintf(stderr, "KVFE_ME_PROF last launch: waves %.0f (%.0f without an item), items %.0f, cycles per item: mask phase %.0f | "
                       "row loop %.0f | epilogue %.0f; wave lifetime mean %.0f longest %.0f, first start to last end %.0f (100 MHz ticks x 24); pixel rows needed %.1f of %.1f per item\n",
                       waves, idle, a[3], a[0] / a[3], a[1] / a[3], a[2] / a[3], life / waves, longest, last - first, a[4] / a[3], a[5] / a[3]);
      });
    }
  }
#endif
  const int simds = 4 * (P.n_cu > 0 ? P.n_cu : 256);   // (of the context's device, KParams)
  // Strip height.  Every wave of the launch is resident at once (5 waves per SIMD) and a SIMD works through its waves'
  // rows, so the launch lasts (waves per SIMD, rounded UP) x (rows per wave); every strip pays 5 rows of overlap.  The
  // product is smallest for 6 strips of 80 rows at 64 x 752x480 (4992 waves, 5 per SIMD x 85 rows = 425 row-times;
  // 4 strips: 4 x 125 = 500, 8 strips: 7 x 65 = 455); a few streams get short strips (more waves than SIMDs).
  // Measured in round 3, again with the run walk in round 4 (40 / 60 / 80 / 120 rows -> 0.090 / 0.088 / 0.082 / 0.105 ms)
  // and in round 5 with six rows in flight (48 / 60 / 80 / 96 / 120 -> 0.092 / 0.101 / 0.078 / 0.079 / 0.087 ms).
  const int nx = (P.W + ME_COLS - 1) / ME_COLS;
  int strip_rows = min(P.H, ME2_ROWS);
  double best = 1e30;
  for (int ns = 1; ns <= (P.H + 15) / 16; ns++) {
    const int rws = (P.H + ns - 1) / ns;
    if (rws > ME2_ROWS) continue;
    const long long waves = (long long)nx * ns * P.B;
    const double cost = (double)((waves + simds - 1) / simds) * (rws + 5);
    if (cost < best) best = cost, strip_rows = rws;
  }
#ifdef KVFE_ME_ROWS_OVERRIDE   // (with KVFE_ME_PROF: strip height of the A/B builds)
  strip_rows = KVFE_ME_ROWS_OVERRIDE;
#endif
  const int ny = (P.H + strip_rows - 1) / strip_rows;
  const dim3 grid((unsigned)nx, (unsigned)ny, (unsigned)P.B);
  auto go = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, grid, dim3(64), 0, st, img, row_stride, img_stride, user_mask, P.W, P.H,
                       P.kcap, P.ccap, P.min_distance, T.circle_hw, k.kp, k.lmk, k.count, use_discs, S.flags, D.cand, D.cand_count,
                       D.maxkey, strip_rows, P.B, nx, ny, (float)P.harris_k, P.harris_k);
  };
  if (P.use_harris) {
    if (user_mask) go(mineig2_kernel<true, true>);
    else go(mineig2_kernel<false, true>);
  } else {
    if (user_mask) go(mineig2_kernel<true, false>);
    else go(mineig2_kernel<false, false>);
  }
}

// =============================================================================================
// FeatureDetectorType::FAST (FeatureDetector.cpp:35-40): cv::FastFeatureDetector::create(fast_thresh_, true) ->
// cv::FAST(TYPE_9_16) with non-maximum suppression + KeyPointsFilter::runByPixelsMask.
// Integer work, one image read: a block stages a (64 + 8) x (16 + 8) pixel tile in LDS, scores the (64 + 2) x (16 + 2)
// pixels whose 3 x 3 neighbourhood the tile's output pixels look at (corner test: 9 contiguous pixels of the 16-pixel
// circle all darker than v - t or all brighter than v + t, as bit tricks on the two 16-bit class masks; score = the
// largest threshold that keeps the pixel a corner, minus 1 = max over the 16 arcs of the arc's smallest one-sided
// difference, cornerScore<16>), keeps the strict 3 x 3 maxima under the mask (the stream's disc bitmap of
// detect_mask_kernel and the optional user mask) and appends (pixel index, score) to the stream's candidate list.
// The keypoint ORDER of cv::FAST (raster) is restored by the select kernel's sort.
// =============================================================================================
constexpr int FAST_TW = 64, FAST_TH = 16, FAST_T = 256;
__device__ __forceinline__ int fast_score16(const int (&d)[16]) {
  int best = -256;
#pragma unroll
  for (int a = 0; a < 16; a++) {
    int mn = 255, mx = -255;
#pragma unroll
    for (int j = 0; j < 9; j++) {
      mn = min(mn, d[(a + j) & 15]);
      mx = max(mx, d[(a + j) & 15]);
    }
    best = max(best, max(mn, -mx));
  }
  return best - 1;
}
template <bool HAS_MASK>
__global__ __launch_bounds__(FAST_T) void fast_kernel(const unsigned char* __restrict__ img, size_t row_stride,
                                                     size_t img_stride, const unsigned char* __restrict__ user_mask,
                                                     int W, int H, int ccap, int threshold,
                                                     const unsigned long long* __restrict__ maskbits, int MW,
                                                     const int* __restrict__ flags,
                                                     unsigned long long* __restrict__ cand_all,
                                                     int* __restrict__ cand_count) {
  const int s = blockIdx.z;
  if (flags && !(flags[s] & FLAG_DETECT)) return;
  constexpr int PW = FAST_TW + 8, PH = FAST_TH + 8, SW = FAST_TW + 2, SH = FAST_TH + 2;
  __shared__ unsigned char px[PH][PW + 4];
  __shared__ unsigned char sc[SH][SW + 2];
  __shared__ unsigned long long lc[FAST_TW * FAST_TH];   // (a tile cannot hold more maxima than a quarter of its pixels)
  __shared__ int n_loc, g_base;
  const unsigned char* I = img + (size_t)s * img_stride;
  const int tid = threadIdx.x;
  const int tx0 = blockIdx.x * FAST_TW, ty0 = blockIdx.y * FAST_TH;
  if (tid == 0) n_loc = 0;
  for (int e = tid; e < PH * PW; e += FAST_T) {
    const int yy = e / PW, xx = e - yy * PW;
    const int gy = min(max(ty0 - 4 + yy, 0), H - 1), gx = min(max(tx0 - 4 + xx, 0), W - 1);   // (clamped: never scored)
    px[yy][xx] = I[(size_t)gy * row_stride + gx];
  }
  __syncthreads();
  // circle offsets of cv::FAST's 16-pixel pattern (dx, dy), fast.cpp makeOffsets
  constexpr int OX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  constexpr int OY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
  const int t = min(max(threshold, 0), 255);
  for (int e = tid; e < SH * SW; e += FAST_T) {
    const int yy = e / SW, xx = e - yy * SW;          // score cell (yy, xx) = pixel (ty0 - 1 + yy, tx0 - 1 + xx)
    const int gy = ty0 - 1 + yy, gx = tx0 - 1 + xx;
    int score = 0;
    if (gy >= 3 && gy < H - 3 && gx >= 3 && gx < W - 3) {
      const int cy = yy + 3, cx = xx + 3;            // in the pixel tile
      const int v = px[cy][cx];
      int d[16];
      unsigned dark = 0, bright = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        d[k] = v - (int)px[cy + OY[k]][cx + OX[k]];
        dark |= (d[k] > t ? 1u : 0u) << k;
        bright |= (d[k] < -t ? 1u : 0u) << k;
      }
      auto nine = [](unsigned m) {   // 9 contiguous set bits in the cyclic 16-bit mask
        unsigned x = m | (m << 16);
        x &= x >> 1;
        x &= x >> 2;
        x &= x >> 4;
        x &= (m | (m << 16)) >> 8;
        return (x & 0xffffu) != 0;
      };
      if (nine(dark) || nine(bright)) score = fast_score16(d);
    }
    sc[yy][xx] = (unsigned char)score;
  }
  __syncthreads();
  const unsigned long long* MB = maskbits + (size_t)s * H * MW;
  for (int e = tid; e < FAST_TH * FAST_TW; e += FAST_T) {
    const int yy = e / FAST_TW, xx = e - yy * FAST_TW;
    const int gy = ty0 + yy, gx = tx0 + xx;
    if (gy >= H || gx >= W) continue;
    const int v = sc[yy + 1][xx + 1];
    if (v == 0) continue;   // (a corner's score is at least the threshold: > 0 unless t = 0 and the arc's margin is 1 ... see below)
    const bool is_max = v > sc[yy][xx] && v > sc[yy][xx + 1] && v > sc[yy][xx + 2] && v > sc[yy + 1][xx] &&
                        v > sc[yy + 1][xx + 2] && v > sc[yy + 2][xx] && v > sc[yy + 2][xx + 1] && v > sc[yy + 2][xx + 2];
    if (!is_max) continue;
    const int p = gx + 64;
    if ((MB[(size_t)gy * MW + (p >> 6)] >> (p & 63)) & 1ull) continue;                       // inside a tracked keypoint's disc
    if (HAS_MASK && user_mask[(size_t)s * W * H + (size_t)gy * W + gx] == 0) continue;        // runByPixelsMask
    const int pos = atomicAdd(&n_loc, 1);
    lc[pos] = ((unsigned long long)(~(unsigned)(gy * W + gx)) << 32) | (unsigned)v;   // descending key order = raster order
  }
  __syncthreads();
  if (tid == 0 && n_loc > 0) g_base = atomicAdd(&cand_count[s], n_loc);
  __syncthreads();
  for (int i = tid; i < n_loc; i += FAST_T) {
    const int pos = g_base + i;
    if (pos < ccap) cand_all[(size_t)s * ccap + pos] = lc[i];
  }
}

void launch_fast(const KParams& P, const Tables& T, const unsigned char* img, size_t row_stride, size_t img_stride,
                 const unsigned char* user_mask, const FrameTab& k, const StreamState& S, const DetectScratch& D,
                 int use_discs, hipStream_t st) {
  // the disc bitmap of the tracked keypoints
  const int MW = me_mask_words(P.W);
  static const int prep_budget = lds_dynamic_budget(reinterpret_cast<const void*>(detect_mask_kernel));
  const size_t fixed = sizeof(int) * (size_t)P.kcap + 16;
  const int band_rows = (int)std::min<long long>(P.H, ((long long)prep_budget - (long long)fixed) / ((long long)MW * 8));
  if (band_rows < 1) {
    std::fprintf(stderr, "kvfe: detect_mask_kernel does not fit in LDS (kcap %d, %d x %d)\n", P.kcap, P.W, P.H);
    return;
  }
  hipLaunchKernelGGL(detect_mask_kernel, dim3(P.B), dim3(MEP_T), (size_t)band_rows * MW * 8 + fixed, st, P.W, P.H, P.kcap,
                     P.min_distance, T.circle_hw, k.kp, k.lmk, k.count, use_discs, S.flags, D.me_maskbits, MW, band_rows);
  const dim3 grid((P.W + FAST_TW - 1) / FAST_TW, (P.H + FAST_TH - 1) / FAST_TH, P.B);
  if (user_mask)
    hipLaunchKernelGGL(fast_kernel<true>, grid, dim3(FAST_T), 0, st, img, row_stride, img_stride, user_mask, P.W, P.H, P.ccap,
                       P.fast_thresh, D.me_maskbits, MW, S.flags, D.cand, D.cand_count);
  else
    hipLaunchKernelGGL(fast_kernel<false>, grid, dim3(FAST_T), 0, st, img, row_stride, img_stride, user_mask, P.W, P.H, P.ccap,
                       P.fast_thresh, D.me_maskbits, MW, S.flags, D.cand, D.cand_count);
}

// =============================================================================================
// select: one 1024-thread workgroup per stream
// =============================================================================================
constexpr int SEL_T = 1024;
constexpr int MAX_CELLS = 12288;
constexpr int LDS_SORT_CAP = 8192;
constexpr int GREEDY_CELLS = 3072;  // 16-byte accepted-corner slots of the in-order greedy filter (LDS)
constexpr int SEL_LDS_BASE = (int)(sizeof(unsigned long long) * LDS_SORT_CAP + sizeof(int) * (MAX_CELLS + 1));
constexpr int SEL_LDS_MAX = 160 * 1024 - 2048;  // dynamic LDS ceiling (statics of the kernel stay below 2 KB)
// "blocked" bitmap of the in-order minimum-distance filter: one bit per pixel, rows of `rw` 64-bit words (odd, so
// that the rows of a disc land in different banks, and with at least one spare word at the end of a row so that a
// span may always be written as two words); it sits behind the packed (x | y << 16) candidate list
__host__ __device__ inline int sel_bitmap_row_words(int W) { return (((W + 63) >> 6) + 1) | 1; }  // >= 1 pad word
__host__ __device__ inline long long sel_bitmap_lds_bytes(int W, int H) {
  return (long long)sizeof(unsigned) * LDS_SORT_CAP + (long long)H * sel_bitmap_row_words(W) * 8;
}
// LDS layout when the bitmap fits (every image up to about 1280x720):
//   [0, 32 K)        xy      u32 [8192]  candidates in rank order, packed x | y << 16 (accepted ones in place)
//   [32 K, 96 K)     skeys   u64 [8192]  candidate keys as compacted (unordered)          } the bitmap overlays these
//   [96 K, 112 K)    hist    int [4096]  radix-rank sort: keys per bin / cursors           } once the ranks are known
//   [112 K, 128 K)   start   int [4096]  first rank of each bin                            }
//   [128 K, 144 K)   tmpidx  u16 [8192]  key positions grouped by bin                      }
// the other paths of the kernel use skeys and, behind it, cell_start [MAX_CELLS + 1] as before (ends at 147 460).
constexpr int SEL_XY_BYTES = (int)sizeof(unsigned) * LDS_SORT_CAP;
constexpr int SEL_RADIX_BINS = 4096;
constexpr int SEL_LDS_BITMAP_MIN = SEL_XY_BYTES + SEL_LDS_BASE + 12;  // 147 472
__host__ __device__ inline bool sel_use_bitmap(int W, int H, int md) {
  return md >= 1 && md <= 127 && W < 65536 && H < 65536 && sel_bitmap_lds_bytes(W, H) <= SEL_LDS_MAX;
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* wave_tot /*[16]*/, int* total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wave_tot[wv] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < SEL_T / 64; i++) {
    const int t = wave_tot[i];
    if (i < wv) base += t;
    tot += t;
  }
  __syncthreads();
  if (total) *total = tot;
  return base + inc - v;
}

// Bitonic sort, descending, n = EPT * 1024 keys with EPT in registers per thread (element i lives in
// thread i % 1024, slot i / 1024).  Of the log2(n)(log2(n)+1)/2 compare-exchange steps only those with
// partner distance 64 <= j < 1024 cross wavefronts (through `a` and a barrier: 22 of 91 for n = 8192);
// j >= 1024 stays inside the thread, j < 64 inside the wavefront (shuffles).  Equal keys only occur
// as zero padding, so max/min exchanges are exact.
template <int EPT>
__device__ void block_bitonic_desc_reg(unsigned long long* a) {
  const int tid = threadIdx.x;
  unsigned long long v[EPT];
#pragma unroll
  for (int m = 0; m < EPT; m++) v[m] = a[tid + m * SEL_T];
  constexpr int n = EPT * SEL_T;
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= SEL_T) {
        const int jm = j / SEL_T;
#pragma unroll
        for (int JM = 1; JM < EPT; JM <<= 1) {
          if (jm != JM) continue;
#pragma unroll
          for (int m = 0; m < EPT; m++) {
            if ((m & JM) == 0) {
              const bool desc = (((m * SEL_T) & k) == 0);
              const unsigned long long x = v[m], y = v[m | JM];
              const unsigned long long hi = x > y ? x : y, lo = x > y ? y : x;
              v[m] = desc ? hi : lo;
              v[m | JM] = desc ? lo : hi;
            }
          }
        }
      } else if (j >= 64) {
        __syncthreads();  // the previous exchange's readers are done
#pragma unroll
        for (int m = 0; m < EPT; m++) a[tid + m * SEL_T] = v[m];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < EPT; m++) {
          const int i = tid + m * SEL_T;
          const unsigned long long o = a[i ^ j];
          const bool keepmax = ((i & j) == 0) == ((i & k) == 0);
          v[m] = ((o > v[m]) == keepmax) ? o : v[m];  // equal keys (zero padding): either copy
        }
      } else {
#pragma unroll
        for (int m = 0; m < EPT; m++) {
          const int i = tid + m * SEL_T;
          const unsigned long long o = __shfl_xor(v[m], j);
          const bool keepmax = ((i & j) == 0) == ((i & k) == 0);
          v[m] = ((o > v[m]) == keepmax) ? o : v[m];  // equal keys (zero padding): either copy
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < EPT; m++) a[tid + m * SEL_T] = v[m];
  __syncthreads();
}

// bitonic sort, descending, n = power of two, keys in LDS or global memory
__device__ void block_bitonic_desc(unsigned long long* a, int n) {
  if (n == SEL_T) return block_bitonic_desc_reg<1>(a);
  if (n == 2 * SEL_T) return block_bitonic_desc_reg<2>(a);
  if (n == 4 * SEL_T) return block_bitonic_desc_reg<4>(a);
  if (n == 8 * SEL_T) return block_bitonic_desc_reg<8>(a);
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += SEL_T) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = a[i], y = a[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) {
            a[i] = y;
            a[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// anms::BrownANMS (anms/anms.cpp:51-81) orders its (radius, index) pairs with
//     std::sort(results.begin(), results.end(), sort_pred())        sort_pred: left.first > right.first
// Radii between integer pixel positions tie all the time and std::sort is not stable, so WHICH of the tied
// keypoints make the cut is decided by libstdc++'s introsort (bits/stl_algo.h: __introsort_loop with the
// median-of-three moved to the front, __unguarded_partition, depth limit 2 lg n with a heap-sort fallback, then
// __final_insertion_sort; threshold 16).  That algorithm is restated here step for step on the device; the oracle
// calls the real std::sort of the host's libstdc++, and the parity tests compare the two.
// ---------------------------------------------------------------------------------------------
#define KVFE_HD __device__
#include "kvfe_stdsort.inl"
#undef KVFE_HD
#include "kvfe_blocksort.inl"

// KVFE_SELECT_PROF=1 (debugging aid): phase time stamps of stream 0's block, summed on the host and printed at exit
__device__ unsigned long long kvfe_select_stamps[16];
#define SEL_STAMP(i)                                                                        \
  do {                                                                                      \
    if (prof && s == 0 && threadIdx.x == 0) kvfe_select_stamps[i] = __builtin_readcyclecounter(); \
  } while (0)

__global__ __launch_bounds__(SEL_T) void select_kernel(KParams P, Tables T, FrameTab K,
                                                       StreamState S, DetectScratch D,
                                                       int fixed_need, int prof) {
  const int s = blockIdx.x;
  if (!(S.flags[s] & FLAG_DETECT)) return;
  SEL_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // LDS carve-up: [xy u32 [LDS_SORT_CAP] when the blocked-pixel bitmap fits |] sortkeys [LDS_SORT_CAP] u64 |
  // cell_start [MAX_CELLS+1] int | small
  const bool bitmap_cfg = sel_use_bitmap(P.W, P.H, P.min_distance);
  unsigned char* lds_keys = lds_raw + (bitmap_cfg ? SEL_XY_BYTES : 0);
  unsigned long long* skeys = reinterpret_cast<unsigned long long*>(lds_keys);
  int* cell_start = reinterpret_cast<int*>(lds_keys + sizeof(unsigned long long) * LDS_SORT_CAP);
  __shared__ int wave_tot[SEL_T / 64];
  __shared__ int sh_cnt, sh_flag, sh_n;
  __shared__ int bin_cnt[MAX_BINS];

  const int tid = threadIdx.x;
  const int W = P.W, H = P.H;
  unsigned long long* cand = D.cand + (size_t)s * P.ccap;
  unsigned long long* work = D.sortbuf + (size_t)s * D.sort_cap;
  unsigned int* items = D.cell_items + (size_t)s * P.ccap;
  unsigned char* state = D.state + (size_t)s * P.ccap;
  int overflow = 0;
  int C = D.cand_count[s];
  if (C > P.ccap) {
    C = P.ccap;
    overflow = 1;
  }
  // FeatureDetectorType::FAST: the candidates ARE the keypoints (cv::FAST has no quality threshold, no minimum distance
  // and no maxCorners); they only have to be put back into cv::FAST's raster order, and the (int)response sort of
  // AdaptiveNonMaximumSuppression::suppressNonMax works on real scores instead of all-equal keys
  const bool fast_det = P.detector == 0;
  // ---- quality threshold (cv::threshold THRESH_TOZERO with maxVal*qualityLevel) -------------
  const float maxVal = fkey_inv(D.maxkey[s]);
  const float thr = (float)((double)maxVal * P.quality);
  if (tid == 0) sh_cnt = 0;
  __syncthreads();
  // every thread has read the counters: leave them zeroed for the next mineig launch of this
  // stream (invariant: cand_count / maxkey are zero between a select and the next mineig)
  if (tid == 0) {
    D.cand_count[s] = 0;
    D.maxkey[s] = 0u;
  }
  // (the order of the survivors is irrelevant: keys are distinct and everything downstream ranks by key, so the
  // list is appended wave by wave with one LDS atomic per wave instead of block scans and barriers)
  for (int base = 0; base < C; base += SEL_T) {
    const int i = base + tid;
    bool keep = false;
    unsigned long long key = 0;
    if (i < C) {
      key = cand[i];
      const float v = __uint_as_float((unsigned)(key >> 32));
      keep = fast_det || ((v > thr) && (v != 0.0f));   // (FAST: every candidate is a keypoint, key = ~index << 32 | score)
    }
    const unsigned long long km = __ballot(keep);
    if (km) {
      const int lane = tid & 63;
      int wbase = 0;
      if (lane == 0) wbase = atomicAdd(&sh_cnt, __popcll(km));
      wbase = __builtin_amdgcn_readfirstlane(wbase);
      if (keep) {
        const int pos = wbase + __popcll(km & ((1ull << lane) - 1ull));
        work[pos] = key;
        if (pos < LDS_SORT_CAP) skeys[pos] = key;
      }
    }
  }
  __syncthreads();
  const int C2 = sh_cnt;
  __syncthreads();
  SEL_STAMP(1);
  if (prof && s == 0 && tid == 0) {
    kvfe_select_stamps[10] = C;
    kvfe_select_stamps[11] = C2;
  }

  int A = 0;  // accepted corners, keys in `akeys`
  unsigned long long* akeys = skeys;
  const int md = P.min_distance;
  bool sorted_accepted = false, corners_written = false;
  // bin of the radix-rank sort = monotone non-increasing function of the key: the value's float bits relative to the
  // threshold's, scaled so that [threshold, maximum] spans the bins; bin 0 holds the largest values
  const unsigned lo_bits = __float_as_uint(fmaxf(thr, 0.0f));
  const unsigned hi_bits = max(__float_as_uint(maxVal), lo_bits + 1u);
  int shift = 0;
  while (((hi_bits - lo_bits) >> shift) >= (unsigned)SEL_RADIX_BINS) shift++;
  auto bin_of = [&](unsigned long long key) -> int {
    const unsigned vb = (unsigned)(key >> 32);
    const unsigned d = vb > lo_bits ? vb - lo_bits : 0u;
    return max(SEL_RADIX_BINS - 1 - (int)min(d >> shift, (unsigned)(SEL_RADIX_BINS - 1)), 0);
  };
  // More candidates than the LDS list holds (1280x720 keyframes reach 15 k): TWO passes.  The histogram of all keys
  // gives the largest prefix of bins with at most LDS_SORT_CAP keys; those are ranked and filtered first -- every key
  // of a later bin ranks below all of them --, then the remaining keys are tested against the bitmap in parallel and
  // only the survivors (a few hundred: the image is covered by then) are ranked and walked, continuing the same
  // accepted list.  NL = keys of the first pass, cut_bin = first bin of the second.
  int NL = C2, cut_bin = SEL_RADIX_BINS;
  bool two_pass = false;
  if (!fast_det && bitmap_cfg && C2 > LDS_SORT_CAP) {
    int* hist = cell_start;
    for (int i = tid; i < SEL_RADIX_BINS; i += SEL_T) hist[i] = 0;
    if (tid == 0) {
      sh_n = 0;
      sh_flag = SEL_RADIX_BINS;
    }
    __syncthreads();
    for (int i = tid; i < C2; i += SEL_T) atomicAdd(&hist[bin_of(work[i])], 1);
    __syncthreads();
    {
      constexpr int PER = SEL_RADIX_BINS / SEL_T;
      int c[PER], sum = 0;
#pragma unroll
      for (int q = 0; q < PER; q++) {
        c[q] = hist[tid * PER + q];
        sum += c[q];
      }
      int run = block_exclusive_scan(sum, wave_tot, nullptr);
#pragma unroll
      for (int q = 0; q < PER; q++) {   // first bin whose end passes the capacity
        if (run <= LDS_SORT_CAP && run + c[q] > LDS_SORT_CAP) {
          sh_flag = tid * PER + q;
          sh_n = run;
        }
        run += c[q];
      }
    }
    __syncthreads();
    cut_bin = sh_flag;
    NL = sh_n;
    __syncthreads();
    if (NL > 0) {   // (NL == 0: more than LDS_SORT_CAP keys in the very first bin -> the generic path below)
      two_pass = true;
      if (tid == 0) sh_cnt = 0;
      __syncthreads();
      for (int base = 0; base < C2; base += SEL_T) {   // the first pass's keys, unordered, into the LDS list
        const int i = base + tid;
        unsigned long long key = 0;
        const bool keep = i < C2 && bin_of(key = work[i]) < cut_bin;
        const unsigned long long km = __ballot(keep);
        if (km) {
          const int lane = tid & 63;
          int wbase = 0;
          if (lane == 0) wbase = atomicAdd(&sh_cnt, __popcll(km));
          wbase = __builtin_amdgcn_readfirstlane(wbase);
          if (keep) skeys[wbase + __popcll(km & ((1ull << lane) - 1ull))] = key;
        }
      }
      __syncthreads();
    }
  }
  if (!fast_det && bitmap_cfg && C2 > 0 && (C2 <= LDS_SORT_CAP || two_pass)) {
    // ---- cv::goodFeaturesToTrack's greedy minimum-distance filter, in its own (sequential) order ----
    // 1. the candidates are RANKED by (value, index) descending: radix-rank sort -- histogram over the top bits of the
    //    value, prefix sum, grouping by bin, then every key counts the larger keys of its own bin (two or three on real
    //    images); a bin with more than 64 keys (flat synthetic patterns) falls back to the block-wide bitonic sort;
    // 2. blocked-pixel bitmap: accepting a corner marks the open disc of radius minDistance around it, a candidate is
    //    tested with ONE bit lookup (the reference's 3x3-cell search evaluates exactly this predicate: dx^2 + dy^2 <
    //    minDistance^2 against every accepted corner).  One wavefront walks the ranked list 64 at a time; survivors of a
    //    batch are resolved in rank order with ballots.  The sequential wave spends its time on the ~n_corners
    //    acceptances, not on the C2 candidates, and stops at maxCorners exactly where the reference loop breaks.
    unsigned* xy = reinterpret_cast<unsigned*>(lds_raw);  // candidates in rank order; accepted ones in place
    unsigned long long* bm = reinterpret_cast<unsigned long long*>(lds_raw + SEL_XY_BYTES);
    const int rw = sel_bitmap_row_words(W);
    auto pack_xy = [&](unsigned long long key) -> unsigned {
      const unsigned idx = (unsigned)key;
      const unsigned y = idx / (unsigned)W;
      return (idx - y * (unsigned)W) | (y << 16);
    };
    if (NL <= 256) {
      // short list: rank = number of larger keys
      for (int i = tid; i < NL; i += SEL_T) {
        const unsigned long long k = skeys[i];
        int rank = 0;
        for (int q = 0; q < NL; q++) rank += skeys[q] > k ? 1 : 0;
        xy[rank] = pack_xy(k);
      }
    } else {
      int* hist = cell_start;                        // [SEL_RADIX_BINS]
      int* start = hist + SEL_RADIX_BINS;            // [SEL_RADIX_BINS]
      unsigned short* tmpidx = reinterpret_cast<unsigned short*>(start + SEL_RADIX_BINS);  // [LDS_SORT_CAP]
      for (int i = tid; i < SEL_RADIX_BINS; i += SEL_T) hist[i] = 0;
      if (tid == 0) sh_flag = 0;
      __syncthreads();
      for (int i = tid; i < NL; i += SEL_T) atomicAdd(&hist[bin_of(skeys[i])], 1);
      __syncthreads();
      {
        constexpr int PER = SEL_RADIX_BINS / SEL_T;
        int c[PER], sum = 0;
        bool big = false;
#pragma unroll
        for (int q = 0; q < PER; q++) {
          c[q] = hist[tid * PER + q];
          sum += c[q];
          big |= c[q] > 64;
        }
        if (big) sh_flag = 1;
        int run = block_exclusive_scan(sum, wave_tot, nullptr);
#pragma unroll
        for (int q = 0; q < PER; q++) {
          start[tid * PER + q] = run;
          hist[tid * PER + q] = run;  // becomes the bin's cursor
          run += c[q];
        }
      }
      __syncthreads();
      if (sh_flag == 0) {
        for (int i = tid; i < NL; i += SEL_T) {
          const int pos = atomicAdd(&hist[bin_of(skeys[i])], 1);
          tmpidx[pos] = (unsigned short)i;
        }
        __syncthreads();
        for (int p0 = tid; p0 < NL; p0 += SEL_T) {
          const unsigned long long k = skeys[tmpidx[p0]];
          const int b = bin_of(k);
          const int b0 = start[b], b1 = hist[b];  // the cursor ended at the bin's end
          int rank = b0;
          for (int q = b0; q < b1; q++) rank += skeys[tmpidx[q]] > k ? 1 : 0;
          xy[rank] = pack_xy(k);
        }
      } else {
        // many equal or nearly equal values: block-wide bitonic sort of the zero-padded list
        constexpr int EPT = LDS_SORT_CAP / SEL_T;
        unsigned long long v[EPT];
#pragma unroll
        for (int m = 0; m < EPT; m++) {
          const int i = tid + m * SEL_T;
          v[m] = i < NL ? skeys[i] : 0ull;
        }
        __syncthreads();
        blocksort::sort_desc_blocked<EPT>(v, skeys);
#pragma unroll
        for (int m = 0; m < EPT; m++) {
          const int r = EPT * tid + m;
          if (r < NL) xy[r] = pack_xy(v[m]);
        }
      }
    }
    __syncthreads();
    SEL_STAMP(6);
    for (int i = tid; i < H * rw; i += SEL_T) bm[i] = 0ull;
    __syncthreads();
    SEL_STAMP(2);
    // the in-order filter over the ranked list positions [lb, le), continuing an accepted list of acc0 entries
    auto greedy = [&](const int lb, const int le, const int acc0) {
    if (tid < 64) {
      const int lane = tid;
      const int md2i = md * md;
      // half width of the disc on the row lane + 64 p of its 2 md - 1 rows: largest h with h^2 + dy^2 < md^2
      int hw[4];
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const int dy = lane + 64 * p - (md - 1);
        int h = -1;
        if (dy < md) {
          const int t = md2i - dy * dy;  // > 0
          h = (int)sqrtf((float)t);
          while (h * h >= t) h--;
          while ((h + 1) * (h + 1) < t) h++;
        }
        hw[p] = h;
      }
      int acc = acc0;
      bool done = false;
      if (md <= 32) {
        // one pass of rows (2 md - 1 <= 63) and a span of at most 63 bits = two words, all straight-line: the wave
        // runs alone, every instruction and above all every taken branch of the per-acceptance path costs latency
        const int h = hw[0];
        const int row0 = lane - (md - 1);
        const int rowoff = row0 * rw;  // word offset of this lane's row of the disc relative to the centre row
        unsigned long long t_test = 0, t_res = 0, t_prev = prof ? __builtin_readcyclecounter() : 0;
        // software pipeline: the candidates of batch b + 2 and the bitmap words of batch b + 1 are fetched while
        // batch b is resolved; the speculative bitmap word stays valid as long as batch b accepts nothing (the usual
        // case once the image is covered), otherwise it is read again behind the marks
        auto word_of = [&](unsigned c) -> const unsigned long long* {
          return &bm[(int)(c >> 16) * rw + (int)((c & 0xffffu) >> 6)];
        };
        unsigned v = lb + lane < le ? xy[lb + lane] : 0u;
        unsigned vn = lb + 64 + lane < le ? xy[lb + 64 + lane] : 0u;
        unsigned long long wbits = *word_of(v);
        for (int base = lb; base < le && !done; base += 64) {
          unsigned long long wn = *word_of(vn);
          const int nnb = base + 128 + lane;
          const unsigned vnn = nnb < le ? xy[nnb] : 0u;
          const bool valid = base + lane < le;
          const int x = (int)(v & 0xffffu), y = (int)(v >> 16);
          const bool ok = valid && !((wbits >> (x & 63)) & 1ull);
          unsigned long long mask = __ballot(ok);
          if (prof) {
            const unsigned long long t = __builtin_readcyclecounter();
            t_test += t - t_prev;
            t_prev = t;
          }
          if (mask) {
            do {
              const int l = __ffsll((long long)mask) - 1;
              const int lx = __builtin_amdgcn_readlane(x, l), ly = __builtin_amdgcn_readlane(y, l);
              if (lane == l) xy[acc] = v;  // in place: acc <= base + l, and batches b + 1, b + 2 are in registers
              acc++;
              if (P.max_corners > 0 && acc == P.max_corners) {
                done = true;
                break;
              }
              const int yy = ly + row0;
              if (h >= 0 && (unsigned)yy < (unsigned)H) {
                const int x0 = max(lx - h, 0), x1 = min(lx + h, W - 1);
                const int sh = x0 & 63;
                const unsigned long long span = (2ull << (x1 - x0)) - 1ull;  // x1 - x0 + 1 <= 63 ones
                unsigned long long* pw = &bm[ly * rw + rowoff + (x0 >> 6)];
                __hip_atomic_fetch_or(pw, span << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(pw + 1, (span >> 1) >> (63 - sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
              const int dx = x - lx, dy = y - ly;
              const bool near = ok && (__mul24(dx, dx) + __mul24(dy, dy) < md2i);
              mask &= ~__ballot(near);
            } while (mask);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);  // the lookups below follow the marks (LDS is in order)
            wn = *static_cast<const volatile unsigned long long*>(word_of(vn));
          }
          v = vn;
          vn = vnn;
          wbits = wn;
          if (prof) {
            const unsigned long long t = __builtin_readcyclecounter();
            t_res += t - t_prev;
            t_prev = t;
          }
        }
        if (prof && s == 0 && lane == 0) {
          kvfe_select_stamps[8] = t_test;
          kvfe_select_stamps[9] = t_res;
        }
      } else {
      for (int base = lb; base < le && !done; base += 64) {
        const int i = base + lane;
        const bool valid = i < le;
        const unsigned v = valid ? xy[i] : 0u;
        const int x = (int)(v & 0xffffu), y = (int)(v >> 16);
        const unsigned long long wbits = bm[y * rw + (x >> 6)];
        const bool ok = valid && !((wbits >> (x & 63)) & 1ull);
        unsigned long long mask = __ballot(ok);
        while (mask) {
          const int l = __ffsll((long long)mask) - 1;  // highest-ranked survivor of the batch
          const int lx = __builtin_amdgcn_readlane(x, l), ly = __builtin_amdgcn_readlane(y, l);
          if (lane == l) xy[acc] = v;  // in place: acc <= base + l, the batch itself is in registers
          acc++;
          if (P.max_corners > 0 && acc == P.max_corners) {
            done = true;
            break;
          }
#pragma unroll
          for (int p = 0; p < 4; p++) {
            if (64 * p < 2 * md - 1) {  // wave-uniform
              const int yy = ly - (md - 1) + lane + 64 * p;
              const int h = hw[p];
              if (h >= 0 && yy >= 0 && yy < H) {
                const int x0 = max(lx - h, 0), x1 = min(lx + h, W - 1);
                const int w0 = x0 >> 6, w1 = x1 >> 6;
                for (int w = w0; w <= w1; w++) {
                  const int lo = w == w0 ? (x0 & 63) : 0, hi = w == w1 ? (x1 & 63) : 63;
                  const unsigned long long bits = (~0ull >> (63 - hi)) & (~0ull << lo);
                  __hip_atomic_fetch_or(&bm[yy * rw + w], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
              }
            }
          }
          const int dx = x - lx, dy = y - ly;
          const bool near = ok && (dx * dx + dy * dy < md2i);
          mask &= ~__ballot(near);
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);  // the next batch's lookups follow the marks (LDS is in order)
      }
      }
      if (lane == 0) {
        sh_cnt = acc;
        sh_flag = done ? 1 : 0;
      }
    }
    __syncthreads();
    };
    // (Round 3 also wrote the filter by ROUNDS over all threads of the block -- the lexicographically-first maximal
    // independent set of the "closer than minDistance" graph over a cell grid.  Round 4 ran it on the hardware: bit-exact
    // on every detection / sequence test and 2.2 x SLOWER at 64 streams (select 0.079 against 0.036 ms), 3.4 x on the
    // single EuRoC stream (1.25 against 0.37 ms per pair: the rounds are long dependency chains in clustered corners, each
    // a block barrier).  Deleted; tests/test_select_rounds_algebra.py keeps the statement of the algorithm.)
    greedy(0, NL, 0);
    A = sh_cnt;
    if (two_pass && !sh_flag) {
      // second pass: the keys of the bins from cut_bin on that the bitmap has not blocked yet
      unsigned long long* surv = cand;   // (the raw candidate list is dead since the compaction)
      const int acc1 = A;
      __syncthreads();
      if (tid == 0) sh_n = 0;
      __syncthreads();
      for (int base = 0; base < C2; base += SEL_T) {
        const int i = base + tid;
        unsigned long long key = 0;
        bool keep = false;
        if (i < C2) {
          key = work[i];
          if (bin_of(key) >= cut_bin) {
            const unsigned c = pack_xy(key);
            const unsigned long long wb = bm[(int)(c >> 16) * rw + (int)((c & 0xffffu) >> 6)];
            keep = !((wb >> (c & 63u)) & 1ull);
          }
        }
        const unsigned long long km = __ballot(keep);
        if (km) {
          const int lane = tid & 63;
          int wbase = 0;
          if (lane == 0) wbase = atomicAdd(&sh_n, __popcll(km));
          wbase = __builtin_amdgcn_readfirstlane(wbase);
          if (keep) surv[wbase + __popcll(km & ((1ull << lane) - 1ull))] = key;
        }
      }
      __syncthreads();
      int n2 = sh_n;
      if (acc1 + n2 > LDS_SORT_CAP) {   // (cannot rank more than the list holds: flagged, not silently dropped)
        n2 = LDS_SORT_CAP - acc1;
        overflow = 1;
      }
      // rank = number of larger survivors (a few hundred keys, read through L2)
      for (int i = tid; i < n2; i += SEL_T) {
        const unsigned long long k = surv[i];
        int rank = 0;
        for (int q = 0; q < n2; q++) rank += surv[q] > k ? 1 : 0;
        xy[acc1 + rank] = pack_xy(k);
      }
      __syncthreads();
      greedy(acc1, acc1 + n2, acc1);
      A = sh_cnt;
    }
    // corners are written from the packed list (the u64 key area is gone)
    {
      int nc = A;
      if (P.max_corners > 0) nc = min(nc, P.max_corners);
      nc = min(nc, P.acap);
      float2* cw = D.corners + (size_t)s * P.acap;
      for (int i = tid; i < nc; i += SEL_T) {
        const unsigned v = xy[i];
        cw[i] = make_float2((float)(v & 0xffffu), (float)(v >> 16));
      }
    }
    corners_written = true;
    sorted_accepted = true;
  } else if (!fast_det && md >= 1 && C2 > 0 && C2 <= LDS_SORT_CAP &&
      ((W + md - 1) / md) * ((H + md - 1) / md) <= GREEDY_CELLS && W < 65536 && H < 65536) {
    // ---- cv::goodFeaturesToTrack's greedy minimum-distance filter, in its own (sequential) order ----
    // 1. all candidates sorted by (value, index) descending in LDS;
    // 2. one wavefront walks them 64 at a time: every lane tests its candidate against the grid of
    //    already accepted corners (3x3 cells of side minDistance, at most four accepted corners per
    //    cell), then the survivors of the batch are resolved in rank order with ballots -- the
    //    lowest surviving lane is accepted and knocks out the later lanes closer than minDistance.
    // The accepted list comes out in quality order (= the function's output order) and stops at
    // maxCorners exactly where the reference loop breaks.
    const int cell = md;
    const int gw = (W + cell - 1) / cell, gh = (H + cell - 1) / cell;
    const int ncell = gw * gh;
    unsigned* grid = reinterpret_cast<unsigned*>(cell_start);  // [ncell][4] packed (x | y << 16)
    if (C2 <= 512) {
      // rank sort: keys are distinct, rank = number of larger keys (O(n^2): only for short lists --
      // 2048 keys cost 80 us this way, 10 us with the register-resident bitonic network below)
      unsigned long long* tmp = reinterpret_cast<unsigned long long*>(cell_start);
      for (int i = tid; i < C2; i += SEL_T) tmp[i] = skeys[i];
      __syncthreads();
      for (int i = tid; i < C2; i += SEL_T) {
        const unsigned long long k = tmp[i];
        int rank = 0;
        for (int q = 0; q < C2; q++) rank += tmp[q] > k ? 1 : 0;
        skeys[rank] = k;
      }
      __syncthreads();
    } else {
      int n2 = 1;
      while (n2 < C2) n2 <<= 1;
      for (int i = C2 + tid; i < n2; i += SEL_T) skeys[i] = 0ull;  // (the keys are in skeys since the compaction)
      __syncthreads();
      block_bitonic_desc(skeys, n2);
    }
    SEL_STAMP(6);
    {
    for (int i = tid; i < ncell * 4; i += SEL_T) grid[i] = 0xffffffffu;
    if (tid == 0) sh_flag = 0;
    __syncthreads();
    SEL_STAMP(2);
    if (tid < 64) {
      const int lane = tid;
      const float md2 = (float)((double)md * (double)md);
      int acc = 0;
      bool done = false;
      for (int base = 0; base < C2 && !done; base += 64) {
        const int i = base + lane;
        const bool valid = i < C2;
        const unsigned long long key = valid ? skeys[i] : 0ull;
        const unsigned idx = (unsigned)key;
        const int y = idx / W, x = idx - y * W;
        bool ok = valid;
        const int xc = x / cell, yc = y / cell;
        const int mycell = yc * gw + xc;
        int myslot = 0;  // accepted corners already in this candidate's own cell
        {
          // 3x3 cells, one 16-byte LDS read each and no early exit: nine independent reads in flight
          // instead of up to 36 dependent ones (border cells are clamped = tested twice, harmless)
          const uint4* g4 = reinterpret_cast<const uint4*>(grid);
          bool clear = true;
#pragma unroll
          for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
              const int cy = min(max(yc + dy, 0), gh - 1), cx = min(max(xc + dx, 0), gw - 1);
              const uint4 g = g4[cy * gw + cx];
              const unsigned e[4] = {g.x, g.y, g.z, g.w};
              int cnt = 0;
#pragma unroll
              for (int q = 0; q < 4; q++) {
                const bool used = e[q] != 0xffffffffu;
                const float fx = (float)(x - (int)(e[q] & 0xffffu)), fy = (float)(y - (int)(e[q] >> 16));
                clear &= !(used && (fx * fx + fy * fy < md2));
                cnt += used ? 1 : 0;
              }
              if (dy == 0 && dx == 0) myslot = cnt;  // slots fill in order
            }
          ok = valid && clear;
        }
        // The batch is resolved in rank order without a round trip through LDS: the accepted lane
        // writes its own key and grid slot (fire and forget), everybody else only needs its position
        // (readlane) -- later lanes of the same cell count it into their slot index.
        unsigned long long mask = __ballot(ok);
        while (mask) {
          const int l = __ffsll((long long)mask) - 1;  // highest-ranked survivor of the batch
          const int lx = __builtin_amdgcn_readlane(x, l), ly = __builtin_amdgcn_readlane(y, l);
          const int lcell = __builtin_amdgcn_readlane(mycell, l);
          if (lane == l) {
            skeys[acc] = key;  // in place: acc <= base, the batch itself is in registers
            if (myslot < 4)
              grid[mycell * 4 + myslot] = (unsigned)x | ((unsigned)y << 16);
            else
              sh_flag = 2;  // cannot happen: five corners >= minDistance apart in one cell
          }
          acc++;
          if (P.max_corners > 0 && acc == P.max_corners) {
            done = true;
            break;
          }
          const float dx = (float)(x - lx), dy = (float)(y - ly);
          const bool near = ok && (dx * dx + dy * dy < md2);
          mask &= ~__ballot(near);
          myslot += mycell == lcell ? 1 : 0;
        }
      }
      if (lane == 0) sh_cnt = acc;
    }
    __syncthreads();
    A = sh_cnt;
    if (sh_flag == 2) overflow = 1;
    sorted_accepted = true;
    }
  } else if (!fast_det && md >= 1 && C2 > 0) {
    // grid of cells of side >= minDistance (cvRound(minDistance) for an integer distance): the 3x3 block
    // around a candidate holds every corner closer than minDistance.  For a small distance on a large
    // image the side grows until the grid fits the LDS work area (the grid only accelerates the search).
    int cell = md;
    while ((long long)((W + cell - 1) / cell) * ((H + cell - 1) / cell) > MAX_CELLS) cell++;
    const int gw = (W + cell - 1) / cell, gh = (H + cell - 1) / cell;
    const int ncell = gw * gh;
    if (ncell > MAX_CELLS) {
      overflow = 1;
    } else {
      for (int i = tid; i <= ncell; i += SEL_T) cell_start[i] = 0;
      __syncthreads();
      for (int i = tid; i < C2; i += SEL_T) {
        const unsigned idx = (unsigned)work[i];
        const int y = idx / W, x = idx - y * W;
        atomicAdd(&cell_start[(y / cell) * gw + (x / cell)], 1);
        state[i] = 0;
      }
      __syncthreads();
      // exclusive scan over cells (serial chunks per thread + block scan)
      {
        const int per = (ncell + SEL_T - 1) / SEL_T;
        const int b = tid * per, e = min(ncell, b + per);
        int sum = 0;
        for (int i = b; i < e; i++) sum += cell_start[i];
        int tot;
        int run = block_exclusive_scan(sum, wave_tot, &tot);
        for (int i = b; i < e; i++) {
          const int c = cell_start[i];
          cell_start[i] = run;
          run += c;
        }
        if (tid == 0) cell_start[ncell] = tot;
        __syncthreads();
      }
      // fill: second LDS-free pass using a global cursor array would need atomics on cell_start;
      // instead use `items` with per-cell cursors kept in the upper half of state (int view).
      int* cursor = reinterpret_cast<int*>(items + C2 + 1);  // scratch behind the item list
      // (cell_items has ccap entries; ncell cursors fit when C2 + 1 + ncell <= ccap, else fall back)
      const bool cursor_ok = (long long)C2 + 1 + ncell <= (long long)P.ccap;
      if (!cursor_ok) {
        overflow = 1;
      } else {
        for (int i = tid; i < ncell; i += SEL_T) cursor[i] = 0;
        __syncthreads();
        for (int i = tid; i < C2; i += SEL_T) {
          const unsigned idx = (unsigned)work[i];
          const int y = idx / W, x = idx - y * W;
          const int c = (y / cell) * gw + (x / cell);
          const int p = atomicAdd(&cursor[c], 1);
          items[cell_start[c] + p] = (unsigned)i;
        }
        __syncthreads();
        // parallel evaluation of the sequential greedy filter: a candidate is accepted iff every
        // higher-ranked candidate closer than minDistance (searched in the 3x3 cell block, as
        // OpenCV does) is rejected; rejected iff one of them is accepted.
        SEL_STAMP(2);
        const float md2 = (float)((double)md * (double)md);
        volatile unsigned char* vstate = state;
        for (int round = 0; round < 8192; round++) {
          if (tid == 0) sh_flag = 0;
          __syncthreads();
          for (int i = tid; i < C2; i += SEL_T) {
            if (vstate[i] != 0) continue;
            const unsigned long long ki = work[i];
            const unsigned idx = (unsigned)ki;
            const int y = idx / W, x = idx - y * W;
            const int xc = x / cell, yc = y / cell;
            const int x1 = max(0, xc - 1), y1 = max(0, yc - 1);
            const int x2 = min(gw - 1, xc + 1), y2 = min(gh - 1, yc + 1);
            bool rejected = false, pending = false;
            for (int yy = y1; yy <= y2 && !rejected; yy++)
              for (int xx = x1; xx <= x2 && !rejected; xx++) {
                const int c = yy * gw + xx;
                for (int q = cell_start[c]; q < cell_start[c + 1]; q++) {
                  const unsigned j = items[q];
                  const unsigned long long kj = work[j];
                  if (kj <= ki) continue;  // only higher-ranked (value desc, index desc)
                  const unsigned jdx = (unsigned)kj;
                  const int jy = jdx / W, jx = jdx - jy * W;
                  const float dx = (float)(x - jx), dy = (float)(y - jy);
                  if (dx * dx + dy * dy < md2) {
                    const unsigned char sj = vstate[j];
                    if (sj == 1) {
                      rejected = true;
                      break;
                    }
                    if (sj == 0) pending = true;
                  }
                }
              }
            if (rejected)
              vstate[i] = 2;
            else if (!pending)
              vstate[i] = 1;
            else
              sh_flag = 1;
          }
          __syncthreads();
          const int again = sh_flag;
          __syncthreads();
          if (!again) break;
        }
        // compact accepted keys
        if (tid == 0) sh_cnt = 0;
        __syncthreads();
        for (int base = 0; base < C2; base += SEL_T) {
          const int i = base + tid;
          const bool acc = (i < C2) && state[i] == 1;
          int tot;
          const int pos = block_exclusive_scan(acc ? 1 : 0, wave_tot, &tot);
          const int off = sh_cnt;
          if (acc) {
            if (off + pos < LDS_SORT_CAP) skeys[off + pos] = work[i];
          }
          __syncthreads();
          if (tid == 0) sh_cnt = off + tot;
          __syncthreads();
        }
        A = sh_cnt;
        if (A > LDS_SORT_CAP) {
          A = LDS_SORT_CAP;
          overflow = 1;
        }
      }
    }
    // sort accepted (LDS)
    int n2 = 1;
    while (n2 < A) n2 <<= 1;
    for (int i = A + tid; i < n2; i += SEL_T) skeys[i] = 0;
    __syncthreads();
    block_bitonic_desc(skeys, n2);
  } else if (C2 > 0 && !sorted_accepted) {
    // no minimum distance: all candidates, sorted (global memory when they do not fit in LDS)
    int n2 = 1;
    while (n2 < C2) n2 <<= 1;
    if (n2 <= LDS_SORT_CAP) {
      for (int i = tid; i < n2; i += SEL_T) skeys[i] = i < C2 ? work[i] : 0ull;
      __syncthreads();
      block_bitonic_desc(skeys, n2);
    } else {
      for (int i = C2 + tid; i < n2; i += SEL_T) work[i] = 0ull;
      __syncthreads();
      block_bitonic_desc(work, n2);
      akeys = work;
    }
    A = C2;
  }
  __syncthreads();
  SEL_STAMP(3);
  int n_corners = A;
  if (P.max_corners > 0 && !fast_det) n_corners = min(n_corners, P.max_corners);
  n_corners = min(n_corners, P.acap);
  if (A > P.acap && (fast_det || P.max_corners <= 0 || P.max_corners > P.acap)) overflow = 1;
  float2* corners = D.corners + (size_t)s * P.acap;
  unsigned short* perm_dyn = reinterpret_cast<unsigned short*>(items);   // FAST: cv::sortIdx((int)response, DESCENDING)
  if (fast_det) {
    // keypoints in raster order + the permutation cv::sortIdx gives their integer responses: libstdc++'s std::sort of
    // the indices by `response[a] < response[b]` (restated on one lane, kvfe_stdsort.inl: the comparator `a.r > b.r` on
    // r = -response orders exactly like it), reversed.  FAST scores tie all the time, so the unstable sort's tie
    // order is part of the result (binning and the Bailo ANMS variants walk the list in order).
    constexpr int PER = LDS_SORT_CAP / SEL_T;
    int sc_reg[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const int i = tid + q * SEL_T;
      sc_reg[q] = 0;
      if (i < n_corners) {
        const unsigned long long key = akeys[i];
        const unsigned idx = ~(unsigned)(key >> 32);
        const int y = idx / W, x = idx - y * W;
        corners[i] = make_float2((float)x, (float)y);
        sc_reg[q] = (int)(key & 0xffu);
      }
    }
    __syncthreads();
    BrownRI* res = reinterpret_cast<BrownRI*>(skeys);   // [n_corners <= LDS_SORT_CAP]
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const int i = tid + q * SEL_T;
      if (i < n_corners) {
        res[i].r = -(float)sc_reg[q];
        res[i].i = i;
      }
    }
    __syncthreads();
    if (P.sortidx_policy == 0) {
      if (tid == 0) brown_std_sort(res, n_corners, cell_start);
      __syncthreads();
      for (int i = tid; i < n_corners; i += SEL_T) perm_dyn[i] = (unsigned short)res[n_corners - 1 - i].i;
    } else {   // KVFE_SORTIDX_STABLE: descending response, equal responses in keypoint order
      for (int i = tid; i < n_corners; i += SEL_T) {
        const float ri = res[i].r;
        int rank = 0;
        for (int j = 0; j < n_corners; j++) rank += (res[j].r < ri || (res[j].r == ri && j < i)) ? 1 : 0;
        perm_dyn[rank] = (unsigned short)i;
      }
    }
    __syncthreads();
  } else {
  for (int i = tid; i < n_corners && !corners_written; i += SEL_T) {
    const unsigned idx = (unsigned)akeys[i];
    const int y = idx / W, x = idx - y * W;
    corners[i] = make_float2((float)x, (float)y);
  }
  }
  __syncthreads();

  // ---- FeatureDetector::featureDetection(Frame*) bookkeeping (FeatureDetector.cpp:101-115) ----
  int need = fixed_need;
  int n_existing_total = 0;
  if (fixed_need < 0) {
    const int cnt = K.count[s];
    n_existing_total = cnt;
    int local = 0;
    for (int i = tid; i < cnt; i += SEL_T) {
      if (K.lmk[(size_t)s * P.kcap + i] != -1) local++;
      K.age[(size_t)s * P.kcap + i] += 1;
    }
    int tot;
    block_exclusive_scan(local, wave_tot, &tot);
    need = max(P.max_features - tot, 0);
  }

  SEL_STAMP(4);
  // ---- ANMS (NonMaximumSuppression.cpp:33-169) -----------------------------------------------
  float2* newc = D.newc + (size_t)s * P.acap;
  int n_new = 0;
  if (n_corners == 0) {
    n_new = 0;
  } else if (!P.enable_anms) {
    for (int i = tid; i < n_corners; i += SEL_T) newc[i] = corners[i];
    n_new = n_corners;
  } else if (P.anms_type == 0 /* TopN: receives the UNSORTED keypoints */) {
    n_new = need > n_corners ? n_corners : need;
    for (int i = tid; i < n_new; i += SEL_T) newc[i] = corners[i];
  } else if (P.anms_type == 1 /* BrownANMS: receives the UNSORTED keypoints (NonMaximumSuppression.cpp:74) */) {
    if (need > n_corners) {
      for (int i = tid; i < n_corners; i += SEL_T) newc[i] = corners[i];
      n_new = n_corners;
    } else {
      // radius of keypoint i = distance to the nearest keypoint that precedes it (float, as upstream)
      BrownRI* res = reinterpret_cast<BrownRI*>(skeys);          // [n_corners <= LDS_SORT_CAP]
      int* stack = cell_start;                                    // introsort ranges
      __syncthreads();
      for (int i = tid; i < n_corners; i += SEL_T) {
        float minDist = 3.402823466e+38f;  // FLT_MAX
        const float2 ci = corners[i];
        for (int j = 0; j < i; j++) {
          const float2 cj = corners[j];
          const float exp1 = cj.x - ci.x, exp2 = cj.y - ci.y;
          const float curDist = sqrtf(exp1 * exp1 + exp2 * exp2);
          minDist = fminf(curDist, minDist);
        }
        res[i].r = minDist;
        res[i].i = i;
      }
      __syncthreads();
      if (tid == 0) brown_std_sort(res, n_corners, stack);
      __syncthreads();
      n_new = need;
      for (int i = tid; i < n_new; i += SEL_T) newc[i] = corners[res[i].i];
    }
  } else if (P.anms_type >= 2 && P.anms_type <= 5) {
    // ---- anms::Sdc / KdTree / RangeTree / Ssc (anms/anms.cpp:83-436) on the permuted keypoints -----
    // Binary search on the suppression radius; every probe is the greedy sweep "keep a keypoint
    // unless an already kept one covers it", run by one wave 64 candidates at a time (test against
    // the kept list, then resolve the batch in order with ballots).  The spatial indices of the
    // reference only accelerate the covering relation, which is evaluated directly:
    //   SDC (2)       cells of side c = 0.25 r / sqrt 2 : sqrt(dr^2 + dc^2) <= r / c
    //   KdTree (3)    dx^2 + dy^2 < r^2 on integer pixels
    //   RangeTree (4) |dx| <= w and |dy| <= w
    //   SSC (5)       cells of side c = (double)(w / 2) : |dr|, |dc| <= floor(w / c)
    // The sweep that defines the result is replayed once at the end to write the corners out.
    const unsigned short* perm = fast_det ? perm_dyn : T.sortidx + T.sortidx_off[n_corners];
    int* pxy = reinterpret_cast<int*>(cell_start);          // [n] x | y << 16 of the permuted keypoints
    int* sel = pxy + n_corners;                              // [n] kept keypoints (pixel or cell coords)
    const int n = n_corners;
    const int type = P.anms_type;
    for (int i = tid; i < n; i += SEL_T) {
      const float2 c = corners[perm[i]];
      pxy[i] = (int)c.x | ((int)c.y << 16);
    }
    if (tid == 0) sh_flag = 0;
    __syncthreads();
    // numRetPoints < 2: KdTree / RangeTree / Ssc divide by (numRetPoints - 1) resp. by numRetPoints and convert
    // the infinite result to int (undefined upstream): no new corners, like the oracle.  Sdc has no such term
    // and runs for 0 and 1 as well.
    const bool search = need >= 2 || type == 2;
    if (tid < 64 && search && 3 * n <= MAX_CELLS + 1) {
      const int lane = tid;
      int low, high;
      if (type == 2) {
        low = 1;
        high = W;
      } else {
        const int exp1 = H + W + 2 * need;
        const long long exp2 = ((long long)4 * W + (long long)4 * need + (long long)4 * H * need +
                                (long long)H * H + (long long)W * W - (long long)2 * H * W +
                                (long long)4 * H * W * need);
        const double exp3 = sqrt((double)exp2);
        const double exp4 = need - 1;
        const double sol1 = -round((exp1 + exp3) / exp4);
        const double sol2 = -round((exp1 - exp3) / exp4);
        high = (int)((sol1 > sol2) ? sol1 : sol2);
        low = (int)floor(sqrt((double)n / need));
      }
      const float Kf = (float)(unsigned)need;
      const unsigned Kmin = (unsigned)roundf(Kf - (Kf * 0.1f));
      const unsigned Kmax = (unsigned)roundf(Kf + (Kf * 0.1f));
      // one greedy sweep for radius r; write != nullptr: the kept keypoints go to newc
      auto sweep = [&](int r, bool write) -> int {
        double c = 0.0, reach = 0.0;
        if (type == 2) {
          c = 0.25 * r / sqrt(2.0);
          reach = ((double)r) / c;
        } else if (type == 5) {
          c = (double)(r / 2);
          reach = floor(r / c);
        }
        const int ireach = (int)reach;
        const int r2 = r * r;
        int nsel = 0;
        for (int base = 0; base < n; base += 64) {
          const int i = base + lane;
          const bool valid = i < n;
          const int v = valid ? pxy[i] : 0;
          int a = v & 0xffff, b = v >> 16;  // pixel (x, y) ...
          if (type == 2 || type == 5) {     // ... or cell (col, row)
            a = (int)floor((float)(v & 0xffff) / c);
            b = (int)floor((float)(v >> 16) / c);
          }
          auto covers = [&](int sa, int sb) -> bool {
            const int da = a - sa, db = b - sb;
            if (type == 2) return sqrt((double)(db * db + da * da)) <= reach;
            if (type == 3) return da * da + db * db < r2;
            if (type == 4) return abs(da) <= r && abs(db) <= r;
            return abs(db) <= ireach && abs(da) <= ireach;
          };
          bool ok = valid;
          for (int q = 0; q < nsel && __ballot(ok); q++) {
            const int sv = sel[2 * q], sw = sel[2 * q + 1];
            if (ok && covers(sv, sw)) ok = false;
          }
          unsigned long long mask = __ballot(ok);
          while (mask) {
            const int l = __ffsll((long long)mask) - 1;
            const int la = __shfl(a, l), lb = __shfl(b, l);
            if (lane == 0) {
              sel[2 * nsel] = la;
              sel[2 * nsel + 1] = lb;
              if (write) newc[nsel] = corners[perm[base + l]];
            }
            nsel++;
            const bool near = ok && covers(la, lb);
            mask &= ~__ballot(near);
            mask &= ~(1ull << l);
          }
        }
        return nsel;
      };
      bool complete = false;
      int r = 0, prev = -1, r_result = -1, cnt_result = 0;
      int last_r = -1, last_cnt = 0;  // the previous sweep (what `result` holds in the reference)
      while (!complete) {
        r = low + (high - low) / 2;
        if (r == prev || low > high || (type == 5 && r / 2 == 0)) {  // (SSC: c = 0, see the oracle)
          r_result = last_r;
          cnt_result = last_cnt;
          break;
        }
        const int cnt = sweep(r, false);
        last_r = r;
        last_cnt = cnt;
        if ((unsigned)cnt >= Kmin && (unsigned)cnt <= Kmax) {
          r_result = r;
          cnt_result = cnt;
          complete = true;
        } else if ((unsigned)cnt < Kmin) {
          high = r - 1;
        } else {
          low = r + 1;
        }
        if (type != 2) prev = r;  // Sdc never updates prevradius
      }
      int nn = 0;
      if (r_result >= 0 && cnt_result > 0) nn = sweep(r_result, true);
      if (lane == 0) sh_cnt = nn;
    } else if (tid == 0) {
      sh_cnt = 0;
      if (search) sh_flag = 3;  // keypoint list does not fit the LDS work area
    }
    __syncthreads();
    n_new = sh_cnt;
    if (sh_flag == 3) overflow = 1;
  } else {  // Binning on the cv::sortIdx-permuted keypoints
    const unsigned short* perm = fast_det ? perm_dyn : T.sortidx + T.sortidx_off[n_corners];
    if (need > n_corners) {
      for (int i = tid; i < n_corners; i += SEL_T) newc[i] = corners[perm[i]];
      n_new = n_corners;
    } else {
      const int hb = P.hbins, vb = P.vbins;
      const float binRowSize = (float)H / (float)vb;
      const float binColSize = (float)W / (float)hb;
      int active = 0;
      for (int i = 0; i < hb * vb; i++) active += T.binning_mask[i];
      const float nrActiveBins = (float)active;
      const int quota = (int)roundf((float)need / nrActiveBins);
      // bin of every permuted keypoint, kept in LDS behind the sort keys (u16)
      unsigned short* bins = reinterpret_cast<unsigned short*>(cell_start);
      for (int i = tid; i < n_corners; i += SEL_T) {
        const float2 c = corners[perm[i]];
        const long long br = (long long)(c.y / binRowSize), bc = (long long)(c.x / binColSize);
        unsigned short b = 0xffff;
        if (br >= 0 && br < vb && bc >= 0 && bc < hb && T.binning_mask[br * hb + bc] == 1)
          b = (unsigned short)(br * hb + bc);
        bins[i] = b;
      }
      __syncthreads();
      if (tid == 0) sh_cnt = 0;
      __syncthreads();
      for (int base = 0; base < n_corners; base += SEL_T) {
        const int i = base + tid;
        bool keep = false;
        if (i < n_corners) {
          const unsigned short b = bins[i];
          if (b != 0xffff) {
            int rank = 0;
            for (int q = 0; q < i; q++) rank += (bins[q] == b);
            keep = rank < quota;
          }
        }
        int tot;
        const int pos = block_exclusive_scan(keep ? 1 : 0, wave_tot, &tot);
        const int off = sh_cnt;
        if (keep) newc[off + pos] = corners[perm[i]];
        __syncthreads();
        if (tid == 0) sh_cnt = off + tot;
        __syncthreads();
      }
      n_new = sh_cnt;
    }
  }
  // capacity of the frame table
  if (fixed_need < 0 && n_existing_total + n_new > P.kcap) {
    n_new = P.kcap - n_existing_total;
    overflow = 1;
  }
  __syncthreads();
  SEL_STAMP(5);
  if (prof && s == 0 && tid == 0) {
    kvfe_select_stamps[12] = n_corners;
    kvfe_select_stamps[13] = n_new;
  }
  if (tid == 0) {
    D.n_corners[s] = n_corners;
    D.n_new[s] = n_new;
    D.sp_next[s] = 0;   // (the refinement's work counter of this stream)
    D.need[s] = need;
    S.n_detected[s] = n_new;
    if (overflow) S.flags[s] |= FLAG_OVERFLOW;
  }
}

void launch_select(const KParams& P, const Tables& T, const FrameTab& k, const StreamState& S,
                   const DetectScratch& D, int fixed_need, hipStream_t st) {
  size_t lds = SEL_LDS_BASE;
  if (sel_use_bitmap(P.W, P.H, P.min_distance))
    lds = std::max((size_t)SEL_LDS_BITMAP_MIN, (size_t)sel_bitmap_lds_bytes(P.W, P.H));
  // the > 64 KB dynamic-LDS opt-in is a per-device function attribute: apply it once on every device a
  // context of this process launches on (contexts may live on different GPUs / threads, kvfe.h)
  static std::mutex mu;
  static unsigned long long done_mask[4] = {0, 0, 0, 0};   // 256 devices
  int dev = 0;
  hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < 256 && !((done_mask[dev >> 6] >> (dev & 63)) & 1ull)) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(select_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, SEL_LDS_MAX);
      done_mask[dev >> 6] |= 1ull << (dev & 63);
    }
  }
  static const bool prof = std::getenv("KVFE_SELECT_PROF") != nullptr;
  hipLaunchKernelGGL(select_kernel, dim3(P.B), dim3(SEL_T), lds, st, P, T, k, S, D, fixed_need, prof ? 1 : 0);
  if (prof) {
    static double acc[16];
    static long n = 0;
    static bool reg = false;
    unsigned long long h[16];
    hipStreamSynchronize(st);
    hipMemcpyFromSymbol(h, HIP_SYMBOL(kvfe_select_stamps), sizeof(h));
    if (h[6] < h[1] || h[2] < h[6] || h[3] < h[2]) return;  // another path ran: the stamps of this one are stale
    acc[1] += (double)(h[1] - h[0]);   // threshold + compaction
    acc[2] += (double)(h[6] - h[1]);   // sort
    acc[6] += (double)(h[2] - h[6]);   // pack + clear
    acc[3] += (double)(h[3] - h[2]);   // greedy filter
    acc[4] += (double)(h[4] - h[3]);
    acc[5] += (double)(h[5] - h[4]);
    acc[7] += (double)(h[5] - h[0]);
    acc[8] += (double)h[8];
    acc[9] += (double)h[9];
    for (int i = 10; i < 14; i++) acc[i] += (double)h[i];
    n++;
    if (!reg) {
      reg = true;
      std::atexit([] {
        std::fprintf(stderr, "KVFE_SELECT_PROF n=%ld cycles: total %.0f (greedy: test %.0f resolve %.0f) threshold+compact %.0f sort %.0f pack+clear %.0f greedy %.0f bookkeeping %.0f anms %.0f | C %.0f C2 %.0f n_corners %.0f n_new %.0f\n",
                     n, acc[7] / n, acc[8] / n, acc[9] / n, acc[1] / n, acc[2] / n, acc[6] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[10] / n, acc[11] / n, acc[12] / n, acc[13] / n);
      });
    }
  }
}

// =============================================================================================
// cv::cornerSubPix: one wavefront per corner.
// The five normal-equation sums are accumulated in float64 in OpenCV's row-major order (lanes 0-4
// each own one accumulator) so results are bit-identical to the CPU path.
// =============================================================================================
#include "kvfe_subpix.inl"

// cv::undistortPoints of one pixel (double), shared with k_stereo.hip via kvfe_undistort.inl
#include "kvfe_undistort.inl"

// KVFE_SUBPIX_STATS=1 (debugging aid): per-corner cycle counts of the refinement launches, printed at exit
// New corners of ALL detecting streams of the launch (every wave computes it for itself: one coalesced read per 64
// streams and a wave reduction).  The two cornerSubPix kernels share a launch slot: few corners in total -> the chip is
// not full, an iteration's latency decides -> one corner per block, four float64 chains per wave; many corners -> the
// number of instructions decides -> SPG_G corners per block, one chain per lane.  Both are bit-identical, so the choice
// is invisible in the results; it is made on the device because the count is only known there.
constexpr int SPG_MIN_TOTAL = 3000;
__device__ __forceinline__ int subpix_total_new(const StreamState& S, const DetectScratch& D, int B) {
  int v = 0;
  for (int t = threadIdx.x & 63; t < B; t += 64) v += (S.flags[t] & FLAG_DETECT) ? D.n_new[t] : 0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

__device__ unsigned long long kvfe_subpix_stats[8];   // corners, sum cycles, max cycles, launches' first/last stamp

// Launch bound of the two-wave variant (round 4): hipcc's own allocation is 172 VGPRs -- 2 waves per SIMD, four corners
// per CU, although LDS (21.5 KB per corner) admits seven.  Bounded to 4 waves per SIMD it is 128 VGPRs + 30 spilled
// to scratch, seven corners per CU: cornerSubPix 0.431 -> 0.377 ms in the 64-stream step, 1.878 -> 1.604 ms on real
// frames (A/B of two builds on one box, tools/r4/gpu_j.sh); (128, 3) = 168 VGPRs gave 0.402 / 1.626 ms.
template <int WIN, int NW>
__global__ __launch_bounds__(64 * NW, NW == 2 ? 4 : 2) void subpix_append_kernel(KParams P, Tables T,
                                                           const unsigned char* __restrict__ img,
                                                           size_t row_stride, size_t img_stride,
                                                           FrameTab K, StreamState S,
                                                           DetectScratch D, int append) {
  // grid = (streams, corner slots): the stream is the FAST index, so the launch starts with every stream's first corners
  // (with the corner index fast, the last stream's corners sat behind ~50 k blocks that have nothing to do), and a block
  // walks its stream's corners with the stride of the grid (the launcher sizes the grid to what the device holds at once)
  const int s = blockIdx.x;
  if (!(S.flags[s] & FLAG_DETECT)) return;
  const int n_new = D.n_new[s];
  if ((int)blockIdx.y >= n_new) return;
  if ((append & 128) && subpix_total_new(S, D, P.B) >= SPG_MIN_TOTAL) return;   // (bit 7: the group kernel's launch follows)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x;
  const bool stats = (append & 16) != 0;
  // (raised wave priority -- s_setprio 3 -- was measured in round 3: +0.6 % on the step, and the rectification beside it
  // 0.098 -> 0.108 ms: the dense kernel pays for the latency-bound one; removed.  Round 6, with the matching kernels beside it
  // instead of the rectification: 82.3 / 81.8 k with against 82.4 / 82.8 k without, tools/r6/gpu_ab_env.sh)
  append &= 15;
  for (int ci = blockIdx.y; ci < n_new; ci += gridDim.y) {
  float2 c = D.newc[(size_t)s * P.acap + ci];
  const unsigned long long t_begin = stats ? __builtin_readcyclecounter() : 0ull;
  if (P.subpix_enable) {
    __syncthreads();   // (the previous corner's readers of the LDS areas are done)
    c = corner_subpix_wave<WIN, NW>(img + (size_t)s * img_stride, row_stride, P.W, P.H, c, P.subpix_win,
                               P.subpix_iters, P.subpix_eps2, T.subpix_mask, lds_raw, lane);
  }
  if (stats && lane == 0) {
    const unsigned long long dt = __builtin_readcyclecounter() - t_begin;
    atomicAdd(&kvfe_subpix_stats[0], 1ull);
    atomicAdd(&kvfe_subpix_stats[1], dt);
    atomicMax(&kvfe_subpix_stats[2], dt);
    atomicAdd(&kvfe_subpix_stats[3 + (dt < 100000 ? 0 : (dt < 200000 ? 1 : (dt < 400000 ? 2 : 3)))], 1ull);
  }
  if (lane == 0) {
    if (append) {
      const int base = S.n_tracked[s];
      const size_t o = (size_t)s * P.kcap + base + ci;
      K.kp[o] = c;
      K.lmk[o] = S.lmk_counter[s] + ci;
      K.age[o] = 1;   // (the bearing vector: detect_commit_kernel, all corners of the stream side by side)
    } else {
      D.newc[(size_t)s * P.acap + ci] = c;
    }
  }
  }
}

// ---------------------------------------------------------------------------------------------
// cv::cornerSubPix for a GROUP of corners per block (round 4; FeatureDetector.cpp:283-296, kvfe_subpix.inl).
//
// Why: a step is bound by the number of instructions its kernels issue (profiles/r4_analysis.md: every kernel of the
// step runs at ~1 instruction per SIMD and 4 cycles, whatever its category), and the one-corner-per-block kernel above
// spends 900 of its 1 900 vector instructions per corner and iteration on the five sequential float64 chains: 448
// v_fmac_f64_dpp per wave with FOUR useful lanes in one wave and ONE in the other.  Here a lane IS a chain:
//   * block = SPG_G corner SLOTS of one stream, 512 threads; a slot whose corner is done takes the stream's next one;
//   * patch phase: one wave per slot computes cv::getRectSubPix's 23 x 23 float patch from the corner's u8 stage in LDS
//     (rect_subpix_from_stage_w64 / rect_subpix_border_from_stage, 9 entries per lane);
//   * term phase: waves 1-7 turn patches into the five float64 terms of every window pixel, SPG_KC window pixels of all
//     corners per chunk (one term set per thread), into one of two LDS chunk buffers laid out [pixel][chain][corner];
//   * chain phase: wave 0, lane = 8 chain + corner (40 lanes), adds its chain's terms in window order -- one
//     ds_read_b64 + one v_add_f64 per term for ALL corners of the block -- while waves 1-7 produce the next chunk;
//   * solve: lanes 0..7 of wave 0, one corner each, the 2 x 2 system, the convergence test and the refill.
// Same operations in the same order per corner as corner_subpix_wave_t => bit-identical
// (tests/test_gpu_bench_configs.py::test_grouped_corner_subpix_kernel_forced runs the small configurations through this
// kernel, ::test_c3e_real_frames_64_streams takes it by itself).  ~600 instead of ~1 900 instructions per corner and
// iteration.  What an iteration costs and what bounds it: profiles/r5_analysis.md sections 10 and 12.
// Used for many streams; a few streams (latency bound, few corners) keep the four-waves-per-corner kernel, and so do
// windows other than 10 and images smaller than the 36 x 36 stage.
// ---------------------------------------------------------------------------------------------
__device__ unsigned long long kvfe_spg_stats[8];   // KVFE_SUBPIX_STATS: iterations, cycles of the phases (wave 0 of every block)
constexpr int SPG_G = 8;     // corners per block
constexpr int SPG_T = 512;   // threads per block: 64 per corner in the patch phase; wave 0 walks the chains, waves 1-7 make the terms
constexpr int SPG_KC = 56;   // window pixels per chunk: 56 x 8 corners = 448 term sets = ONE per producing thread and chunk,
                             // and 8 chunks are exactly the 448 zero-padded terms of a chain
struct SpgGeom {
  int pw, nt, rs;
  size_t mask_off, state_off, stage_off, patch_off, terms_off, bytes;
};
__host__ __device__ inline SpgGeom spg_geom(int win) {
  SpgGeom g;
  const SubpixGeom s = subpix_geom(win);
  g.pw = s.pw;
  g.nt = s.nt;
  g.rs = s.rs;
  size_t o = 0;
  g.mask_off = o;   o += sizeof(uint2) * (size_t)SPG_KC * ((g.nt + SPG_KC - 1) / SPG_KC);   // [window pixel, zero padded] {mask weight, packed patch offset and window coordinates}
  g.state_off = o;  o += 16 * sizeof(int) * SPG_G;                                         // [corner][16] ints / floats
  g.stage_off = o;  o += (size_t)SPG_G * (((size_t)g.rs * g.rs + 15) & ~(size_t)15);
  g.patch_off = o;  o += (size_t)SPG_G * sizeof(float) * g.pw * g.pw;
  o = (o + 15) & ~(size_t)15;
  g.terms_off = o;  o += 2 * sizeof(double) * SPG_KC * 5 * SPG_G;
  g.bytes = o;
  return g;
}
enum { SPG_CIX = 0, SPG_CIY, SPG_CTX, SPG_CTY, SPG_ACTIVE, SPG_STAGED, SPG_SX0, SPG_SY0, SPG_ITER, SPG_CI };

template <int WIN>
__global__ __launch_bounds__(SPG_T, 4) void subpix_group_kernel(KParams P, Tables T, const unsigned char* __restrict__ img,
                                                             size_t row_stride, size_t img_stride, FrameTab K,
                                                             StreamState S, DetectScratch D, int append) {
  const int s = blockIdx.x;
  if (!(S.flags[s] & FLAG_DETECT)) return;
  const int n_new = D.n_new[s];
  const int c_first = blockIdx.y * SPG_G;
  if (c_first >= n_new) return;
  if (!(append & 64) && subpix_total_new(S, D, P.B) < SPG_MIN_TOTAL) return;   // (bit 6: forced, A/B and tests)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const SpgGeom G = spg_geom(WIN);
  constexpr int pw = 2 * WIN + 3, ww = 2 * WIN + 1, nt = ww * ww, rs = pw + 1 + 12;
  constexpr int TPC = SPG_T / SPG_G;   // threads per corner in the patch phase
  constexpr int MAXP = (pw * pw + TPC - 1) / TPC;
  constexpr int NCH = (nt + SPG_KC - 1) / SPG_KC;
  constexpr int NPROD = SPG_T - 64;    // term-producing threads
  // per window pixel k (zero padded to whole chunks): .x = the mask weight's bits (0 in the padding: such a term set is
  // +-0, and adding +-0 leaves a float64 sum that started at +0 as it is), .y = byte offset of the pixel in a corner's
  // patch | (j - WIN) << 16 | (i - WIN) << 24 -- everything a producing thread needs about its pixel in ONE ds_read_b64
  // (it used to divide k by the window width in every chunk)
  uint2* ktab = reinterpret_cast<uint2*>(lds_raw + G.mask_off);
  int* state = reinterpret_cast<int*>(lds_raw + G.state_off);
  float* statef = reinterpret_cast<float*>(lds_raw + G.state_off);
  const size_t stage_stride = ((size_t)rs * rs + 15) & ~(size_t)15;
  double* terms = reinterpret_cast<double*>(lds_raw + G.terms_off);
  __shared__ int sh_active;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int cs = tid / TPC, sub = tid % TPC;   // corner slot and thread of the slot (patch phase)
  const unsigned char* I = img + (size_t)s * img_stride;
  const int W = P.W, H = P.H;
  const int max_iters = P.subpix_iters;
  const double eps2 = P.subpix_eps2;

  for (int k = tid; k < NCH * SPG_KC; k += SPG_T) {
    const int kq = min(k, nt - 1), i = kq / ww, jj = kq - i * ww;   // (padding: the last pixel's addresses, weight 0)
    const unsigned off = (unsigned)(((i + 1) * pw + (jj + 1)) * (int)sizeof(float));
    ktab[k] = make_uint2(k < nt ? __float_as_uint(T.subpix_mask[k]) : 0u,
                         off | ((unsigned)((jj - WIN) & 0xff) << 16) | ((unsigned)((i - WIN) & 0xff) << 24));
  }
  // A finished corner is written out at once and its slot takes the stream's next corner off a counter (round 5).  Until
  // round 4 a block kept its eight corners to the end, i.e. for as many iterations as its slowest one: corners need 19
  // iterations on average and 40 % of them more than 20 (KVFE_SUBPIX_STATS), so most of a block's lanes idled most of
  // the time.  Same arithmetic per corner; which block refines a corner is invisible in the result.
  //
  // A refill is on the critical path of the WHOLE block (the other seven corners wait at the next barrier), and one
  // happens every ~2.4 iterations.  Its first form cost ~10 dependent memory round trips and ~400 dependent float64
  // instructions each time: the counter, the corner's coordinates, the per-stream append offsets, the bearing vector of
  // the finished corner, six round trips of the byte-wise stage load.  Now:
  //   * the block keeps one corner ON DECK -- index and coordinates requested one / two iterations before a slot needs
  //     them (wave 0; the barriers of the loop wait for LDS only, so the requests stay in flight across them);
  //   * the append offsets are read once, the bearing vectors are left to detect_commit_kernel (all of a stream's corners
  //     side by side instead of one lane at a time in here);
  //   * the stage comes in one round trip (subpix_load_stage).
  const bool do_append = (append & 15) != 0;
  const int app_base = S.n_tracked[s];
  const long long app_lmk0 = S.lmk_counter[s];
  auto finish = [&](int ci, float2 c, float2 cT) {   // FeatureDetector.cpp:141-160
    if (P.subpix_enable && (fabsf(c.x - cT.x) > WIN || fabsf(c.y - cT.y) > WIN)) c = cT;
    if (do_append) {
      const size_t o = (size_t)s * P.kcap + app_base + ci;
      K.kp[o] = c;
      K.lmk[o] = app_lmk0 + ci;
      K.age[o] = 1;
    } else {
      D.newc[(size_t)s * P.acap + ci] = c;
    }
  };
  auto take = [&](int slot, int ci, float2 c0) {   // corner ci moves into the slot
    statef[slot * 16 + SPG_CIX] = c0.x;
    statef[slot * 16 + SPG_CIY] = c0.y;
    statef[slot * 16 + SPG_CTX] = c0.x;
    statef[slot * 16 + SPG_CTY] = c0.y;
    state[slot * 16 + SPG_ACTIVE] = 1;
    state[slot * 16 + SPG_STAGED] = 0;
    state[slot * 16 + SPG_SX0] = 0;
    state[slot * 16 + SPG_SY0] = 0;
    state[slot * 16 + SPG_ITER] = 0;
    state[slot * 16 + SPG_CI] = ci;
  };
  auto pull = [&](int slot) -> bool {   // the slot's next corner, waited for; false: the stream has none left
    for (;;) {
      const int ci = atomicAdd(&D.sp_next[s], 1);
      if (ci >= n_new) {
        state[slot * 16 + SPG_ACTIVE] = 0;
        return false;
      }
      const float2 c0 = D.newc[(size_t)s * P.acap + ci];
      if (!P.subpix_enable) {   // nothing to refine: the corner goes out as it is
        finish(ci, c0, c0);
        continue;
      }
      take(slot, ci, c0);
      return true;
    }
  };
  // the corner on deck (wave 0, the same values in every lane): 0 none, 1 index requested (deck_raw of lane 0; at the end
  // of phase C), 2 index known and coordinates requested (in the middle of the next chunk loop), 3 the stream has no
  // corner left
  int deck_state = 0, deck_raw = 0, deck_ci = 0;
  float2 deck_c = make_float2(0.f, 0.f);
  // (the counter's address through an opaque VGPR: with an address it can prove uniform hipcc rewrites the atomic as
  // "first active lane adds for all, v_readfirstlane, per-lane prefix" -- and the v_readfirstlane waits for the answer on
  // the spot, which is exactly what the deck is there to avoid)
  typedef __attribute__((address_space(1))) int glb_int_t;   // (global, not generic: a flat atomic would count as LDS traffic too)
  glb_int_t* deck_ctr = (glb_int_t*)&D.sp_next[s];
  asm volatile("" : "+v"(deck_ctr));
  if (wave == 0) {
    const bool on = lane < SPG_G && pull(lane);
    const unsigned long long bal = __ballot(on);
    if (lane == 0) sh_active = __popcll(bal);
    if (__popcll(bal) == SPG_G) {
      if (lane == 0) deck_raw = __hip_atomic_fetch_add(deck_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      deck_state = 1;
    } else {
      deck_state = 3;   // (a slot found the counter past the end)
    }
  }
  int eij[MAXP];   // patch entries e = sub + 64 t of the (2w+3)^2 window
#pragma unroll
  for (int t = 0; t < MAXP; t++) {
    const int e = sub + TPC * t, i = e / pw;
    eij[t] = (i << 8) | (e - i * pw);
  }
#if defined(KVFE_SPG_DIAG) && KVFE_SPG_DIAG == 2
  for (int k = tid; k < 2 * SPG_KC * 5 * SPG_G; k += SPG_T) terms[k] = 0.0;   // (nobody produces: every sum is 0, every corner stays where it is)
#endif
  static_assert(SPG_KC * SPG_G == NPROD, "one term set per producing thread and chunk");
  static_assert(TPC == 64, "a slot's patch threads are one wave (its stage is written and read without a barrier)");
  const int prod_c = (tid - 64) & (SPG_G - 1), prod_kk = (tid - 64) >> 3;   // producing thread: corner slot, window pixel of the chunk
  __syncthreads();

  const bool stats = (append & 16) != 0;
  append &= 15;
  unsigned long long st_acc[5] = {0, 0, 0, 0, 0}, st_t = stats ? __builtin_readcyclecounter() : 0ull;
#define SPG_STAMP(i)                                           \
  do {                                                         \
    if (stats) {                                               \
      const unsigned long long t_ = __builtin_readcyclecounter(); \
      st_acc[i] += t_ - st_t;                                  \
      st_t = t_;                                               \
    }                                                          \
  } while (0)
  unsigned long long st_iters = 0;
  while (sh_active > 0) {
    st_iters++;
    SPG_STAMP(4);
    // ---- A: the corner's u8 stage and cv::getRectSubPix patch, 64 threads (one wave) per corner ---------------------
    if (state[cs * 16 + SPG_ACTIVE]) {
      const float cIx = statef[cs * 16 + SPG_CIX], cIy = statef[cs * 16 + SPG_CIY];
      unsigned char* stage = lds_raw + G.stage_off + (size_t)cs * stage_stride;
      float* patch = reinterpret_cast<float*>(lds_raw + G.patch_off) + (size_t)cs * pw * pw;
      const float ccx = cIx - (pw - 1) * 0.5f, ccy = cIy - (pw - 1) * 0.5f;
      const int ipx = cv_floorf(ccx), ipy = cv_floorf(ccy);
      const bool interior = 0 <= ipx && ipx + pw < W && 0 <= ipy && ipy + pw < H;
      const int fx0 = min(max(ipx, 0), W - 1), fx1 = min(max(ipx + pw, 0), W - 1);
      const int fy0 = min(max(ipy, 0), H - 1), fy1 = min(max(ipy + pw, 0), H - 1);
      int sx0 = state[cs * 16 + SPG_SX0], sy0 = state[cs * 16 + SPG_SY0];
      bool staged = state[cs * 16 + SPG_STAGED] != 0;
      if (!staged || fx0 < sx0 || fy0 < sy0 || fx1 >= sx0 + rs || fy1 >= sy0 + rs) {
        const int nx0 = min(max(ipx - 6, 0), W - rs), ny0 = min(max(ipy - 6, 0), H - rs);
        subpix_load_stage<rs, TPC>(I + (size_t)ny0 * row_stride + nx0, row_stride, stage, sub);
        sx0 = nx0;
        sy0 = ny0;
        staged = true;
        if (sub == 0) {
          state[cs * 16 + SPG_SX0] = nx0;
          state[cs * 16 + SPG_SY0] = ny0;
          state[cs * 16 + SPG_STAGED] = 1;
        }
      }
      const bool use_stage = staged && fx0 >= sx0 && fy0 >= sy0 && fx1 < sx0 + rs && fy1 < sy0 + rs;
      if (use_stage && interior)
        rect_subpix_from_stage_w64<MAXP>(stage, rs, ipx - sx0, ipy - sy0, ccx, ccy, ipx, ipy, pw, eij, patch, sub);
      else if (use_stage)
        rect_subpix_border_from_stage<MAXP, TPC>(stage, rs, sx0, sy0, W, H, ccx, ccy, ipx, ipy, pw, eij, patch, sub);
      else
        rect_subpix_8u32f(I, row_stride, W, H, cIx, cIy, pw, patch, sub, TPC);
    }
    SPG_STAMP(0);
    lds_block_sync();
    SPG_STAMP(1);
    // ---- B: terms (waves 1-7) and chains (wave 0), software pipelined over chunks of SPG_KC window pixels ------------
    // (whether the thread's corner slot is in use is read once per iteration: the slots change hands in phase C only)
    const bool prod_on = wave > 0 && state[prod_c * 16 + SPG_ACTIVE] != 0;
    // -DKVFE_SPG_DIAG=1 / 2 (profiling aid, never in the product build, WRONG results): the chain wave adds nothing / the
    // producers produce nothing -- what the chunk loop costs when only the other side works (tools/r5/gpu_ab.sh)
    auto produce = [&](int kc, int buf) {
#if defined(KVFE_SPG_DIAG) && KVFE_SPG_DIAG == 2
      return;
#endif
      if (!prod_on) return;
      const uint2 e = ktab[kc * SPG_KC + prod_kk];
      double* o = terms + (size_t)buf * SPG_KC * 5 * SPG_G + (size_t)prod_kk * 5 * SPG_G + prod_c;
      const float* sp = reinterpret_cast<const float*>(lds_raw + G.patch_off + (size_t)prod_c * pw * pw * sizeof(float) + (e.y & 0xffffu));
      const double m = (double)__uint_as_float(e.x);
      const double tgx = (double)(sp[1] - sp[-1]);
      const double tgy = (double)(sp[pw] - sp[-pw]);
      const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
      const double px = (double)((int)(e.y << 8) >> 24), py = (double)((int)e.y >> 24);
      o[0] = gxx;
      o[SPG_G] = gxy;
      o[2 * SPG_G] = gyy;
      o[3 * SPG_G] = gxx * px + gxy * py;
      o[4 * SPG_G] = gxy * px + gyy * py;
    };
    double acc = 0.0;   // wave 0, lane = 8 chain + corner: the chain's running float64 sum, in window order
    produce(0, 0);
    lds_block_sync();
    SPG_STAMP(2);
    // (raised wave priority -- s_setprio 3 -- for the chain wave, the block's critical path, was measured in round 5:
    // cornerSubPix on real frames 1.560 ms with and without it, tools/r5/gpu_e.sh)
    for (int kc = 0; kc < NCH; kc++) {
      if (wave == 0) {
        if (kc == NCH / 2 && deck_state == 1) {
          // the index asked for at the end of the last phase C has long arrived: ask for the coordinates now, half an
          // iteration before phase C can want them
          deck_ci = __builtin_amdgcn_readfirstlane(deck_raw);
          if (deck_ci >= n_new) {
            deck_state = 3;
          } else {
            deck_c = D.newc[(size_t)s * P.acap + deck_ci];
            deck_state = 2;
          }
        }
#if defined(KVFE_SPG_DIAG) && KVFE_SPG_DIAG == 1
        if (false) {
#else
        if (lane < 5 * SPG_G) {
#endif
          // the chain is a string of dependent additions (~8 cycles each for a lone wave): the reads run SPG_PF terms
          // ahead of them so that no addition waits for LDS
          const double* tb = terms + (size_t)(kc & 1) * SPG_KC * 5 * SPG_G + lane;
          constexpr int SPG_PF = 16;
          double r[SPG_KC];
#pragma unroll
          for (int kk = 0; kk < SPG_PF; kk++) r[kk] = tb[(size_t)kk * 5 * SPG_G];
          __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise sinks the reads to ~6 terms ahead of their additions)
#pragma unroll
          for (int kk = 0; kk < SPG_KC; kk += 2) {
            if (kk + SPG_PF < SPG_KC) {
              r[kk + SPG_PF] = tb[(size_t)(kk + SPG_PF) * 5 * SPG_G];
              r[kk + SPG_PF + 1] = tb[(size_t)(kk + SPG_PF + 1) * 5 * SPG_G];
            }
            acc += r[kk];
            acc += r[kk + 1];
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else if (kc + 1 < NCH) {
        produce(kc + 1, (kc + 1) & 1);
      }
      if (kc + 1 < NCH) lds_block_sync();
    }
    SPG_STAMP(3);
    // ---- C: the 2 x 2 system, one lane per corner (wave 0); finished corners out, their slots refilled -------------
    if (wave == 0) {
      const int c = lane & (SPG_G - 1);
      const double a = __shfl(acc, c), b = __shfl(acc, SPG_G + c), cc = __shfl(acc, 2 * SPG_G + c);
      const double bb1 = __shfl(acc, 3 * SPG_G + c), bb2 = __shfl(acc, 4 * SPG_G + c);
      bool still = false, need = false;
      if (lane < SPG_G && state[lane * 16 + SPG_ACTIVE]) {
        float2 cI = make_float2(statef[lane * 16 + SPG_CIX], statef[lane * 16 + SPG_CIY]);
        int iter = state[lane * 16 + SPG_ITER];
        still = true;
        const double det = a * cc - b * b;
        if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) {
          still = false;
        } else {
          const double scale = 1.0 / det;
          float2 cI2;
          cI2.x = (float)(cI.x + cc * scale * bb1 - b * scale * bb2);
          cI2.y = (float)(cI.y - b * scale * bb1 + a * scale * bb2);
          const double err = (double)((cI2.x - cI.x) * (cI2.x - cI.x) + (cI2.y - cI.y) * (cI2.y - cI.y));
          cI = cI2;
          if (cI.x < 0 || cI.x >= W || cI.y < 0 || cI.y >= H) still = false;
          else still = (++iter < max_iters && err > eps2);
        }
        statef[lane * 16 + SPG_CIX] = cI.x;
        statef[lane * 16 + SPG_CIY] = cI.y;
        state[lane * 16 + SPG_ITER] = iter;
        if (!still) {   // done: out it goes, and the slot wants the stream's next corner
          finish(state[lane * 16 + SPG_CI], cI, make_float2(statef[lane * 16 + SPG_CTX], statef[lane * 16 + SPG_CTY]));
          need = true;
        }
      }
      const unsigned long long need_m = __ballot(need);
      if (need_m != 0ull) {
        // the first slot that wants a corner takes the one on deck; a second one in the same iteration (rare) asks the
        // counter itself and waits.  A corner on deck is never left behind: the loop ends when the last slot finds the
        // stream empty, and a slot that finds a corner on deck stays active.
        const int first = __builtin_ctzll(need_m);
        const bool exhausted = deck_state == 3;
        if (lane == first) {
          if (deck_state == 2) {
            take(lane, deck_ci, deck_c);
            still = true;
          } else if (exhausted) {
            state[lane * 16 + SPG_ACTIVE] = 0;
          } else {
            still = pull(lane);
          }
        } else if (need) {
          if (exhausted) state[lane * 16 + SPG_ACTIVE] = 0;
          else still = pull(lane);
        }
        if (deck_state == 2) deck_state = 0;
      }
      if (deck_state == 0) {   // the next corner on deck: ask for its index now, for its coordinates next time round
        if (lane == 0) deck_raw = __hip_atomic_fetch_add(deck_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        deck_state = 1;
      }
      const unsigned long long bal = __ballot(still);
      if (lane == 0) sh_active = __popcll(bal);
    }
    lds_block_sync();
  }
  if (stats && tid == 0) {
    for (int i = 0; i < 5; i++) atomicAdd(&kvfe_spg_stats[i], st_acc[i]);
    atomicAdd(&kvfe_spg_stats[5], 1ull);
    atomicAdd(&kvfe_spg_stats[6], st_iters);
  }
#undef SPG_STAMP
}

// after the append: counts and the per-stream landmark-id counter -- and the two pieces of per-stream state the NEXT
// step's tracking reads (they depend on this step's flags only): keyframe_R_ref_frame_ and the "initialised" flag.
// They used to be written by step_finalize, at the very end of the step; written here, the next step's predictor and
// tracking launch depend on the corner refinement only, and this step's tail (stereo matching of the new corners,
// measurements, lkf <- k) runs next to them instead of in front of them.
// what: bit 0 = the state the next step's tracking reads (keyframe_R_ref_frame_, "initialised", the keypoint count: all
// known once the corners are selected), bit 1 = the landmark-id counter (read by the append of THIS frame's corners, so
// it moves after them).  Every caller runs both behind the corner refinement.
// (Round 5 measured a step whose tracking is split -- the points frame k-1 already had tracked while its new corners are
// still refined on a third stream, the new corners in a second small launch behind the refinement: bit exact, and 8 %
// SLOWER, tools/r5/gpu_e.sh, gpu_f.sh.  The window the refinement's latency leaves is not idle: the rectify / match /
// reject chain fills it, and with the tracking launch on top all three only share the chip.)
// bit 2 = the bearing vectors of the appended corners (UndistorterRectifier::GetBearingVector, FeatureDetector.cpp:149-150):
// a block per stream, a thread per corner.  Until round 5 the refinement kernels computed them, one lane at a time at the
// end of a corner -- ~400 dependent float64 instructions that, in the grouped kernel, the whole block waited for.
constexpr int DC_T = 128;
__global__ __launch_bounds__(DC_T) void detect_commit_kernel(KParams P, Tables T, FrameTab K, StreamState S, DetectScratch D,
                                                           int what) {
  const int s = blockIdx.x;
  if (P.quiet_gate && !kvfe_all_quiet(S.flags, P.B)) return;
  const int flags = S.flags[s];
  if (threadIdx.x == 0 && (what & 1)) {
    if (flags & FLAG_KEYFRAME) {   // keyframe_R_ref_frame_ = identity (StereoVisionImuFrontend.cpp:203,225)
      for (int i = 0; i < 9; i++) S.kf_R_ref[(size_t)s * 9 + i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else if ((flags & FLAG_INIT) && !(flags & FLAG_DETECT)) {
      // non-keyframe of the normal path: keyframe_R_ref_frame_ = keyframe_R_cur_frame; the "all tracks lost" early
      // return (StereoVisionImuFrontend.cpp:313-323) leaves it untouched
      for (int i = 0; i < 9; i++) S.kf_R_ref[(size_t)s * 9 + i] = S.kf_R_cur[(size_t)s * 9 + i];
    }
    S.flags[s] = flags | FLAG_INIT;
  }
  if (!(flags & FLAG_DETECT)) return;
  const int n_new = D.n_new[s];
  const int base = S.n_tracked[s];
  if (what & 4) {
    for (int ci = threadIdx.x; ci < n_new; ci += DC_T) {
      const size_t o = (size_t)s * P.kcap + base + ci;
      const float2 c = K.kp[o];
      double v[3];
      bearing_vector(T.und_left_R, c.x, c.y, v);
      K.versor[o * 3] = v[0];
      K.versor[o * 3 + 1] = v[1];
      K.versor[o * 3 + 2] = v[2];
    }
  }
  if (threadIdx.x == 0) {
    if (what & 1) K.count[s] = base + n_new;
    if (what & 2) S.lmk_counter[s] += n_new;
  }
}

int detect_new_bound(const KParams& P) {
  int bound = P.max_corners > 0 ? P.max_corners : P.acap;
  if (P.enable_anms && (P.anms_type == 0 || P.anms_type == 6))
    bound = min(bound, P.max_features + P.hbins * P.vbins + P.max_features / 4 + 8);
  return min(bound, P.acap);
}

void launch_subpix_append(const KParams& P, const Tables& T, const unsigned char* img,
                          size_t row_stride, size_t img_stride, const FrameTab& k,
                          const StreamState& S, const DetectScratch& D, int append,
                          hipStream_t st) {
  // (Fewer corners per compute unit -- extra LDS per block as a probe, tools/r5/gpu_z.sh -- make a corner faster, mean 183 k ->
  // 154 k cycles and slowest 586 k -> 401 k at two per unit, and the launch slower, 0.27 -> 0.42 ms per step on average: the
  // steps behind a feature-age burst bring thousands of corners and need the slots.)
  // append bit 9 (do_step, a few streams, synchronous call): the host has read this step's flags and no stream of the batch
  // detects -- the refinement launches would find nothing to do; detect_commit_kernel still has its per-step state to write
  if (append & 512) {
    if (append & 15) hipLaunchKernelGGL(detect_commit_kernel, dim3(P.B), dim3(DC_T), 0, st, P, T, k, S, D, 7);
    return;
  }
  const size_t lds = subpix_geom(P.subpix_win).bytes;
  const int bound = detect_new_bound(P);
  // waves per corner of the one-corner-per-block kernel: 2 (DPP broadcast chains, kvfe_subpix.inl), and 4 for a few
  // streams -- the patch and term work of an iteration spread over four SIMDs (5.4 k instead of 6.3 k cycles per
  // iteration as long as the corners are few; with ~1000 corners in flight two waves are faster).  One block per
  // (stream, corner slot) up to the bound of the new corners, the stream is the fast grid index (see the kernel); a
  // grid sized to what the device holds at once, every block walking ~10 corners, lost 24 % on real frames (round 3:
  // the corners' iteration counts differ too much for a static assignment).
  // (Round 6, four waves per corner at 64 streams: 0.45 against 0.40 ms over a feature-age period, and the quiet steps alone --
  // ~640 corners in the launch -- 225 - 270 against 180 - 190 us: tools/r6/gpu_subpix_nw.sh, profiles/r6_analysis.md section 8.)
  const int nw = P.B <= 4 ? 4 : 2;
  // (corner slots per stream: a block walks its stream's corners with the stride of the grid, so the grid only has to
  // hold the corners this kernel is FOR -- fewer than SPG_MIN_TOTAL over the whole launch, otherwise the grouped kernel
  // works and these blocks return -- instead of one block per possible corner, 50 k mostly empty blocks at 64 streams)
  static const int group_env = std::getenv("KVFE_SUBPIX_GROUP") ? std::atoi(std::getenv("KVFE_SUBPIX_GROUP")) : -1;
  const bool group_ok = P.subpix_win == 10 && P.W >= subpix_geom(10).rs && P.H >= subpix_geom(10).rs;
  const int group_mode = !group_ok ? 0 : (group_env == 0 ? 0 : (group_env == 1 ? 2 : (P.B > 4 ? 1 : 0)));   // 0 never, 1 by count, 2 always
  const dim3 grid(P.B, group_mode == 1 ? std::min(bound, std::max(64, (SPG_MIN_TOTAL + P.B - 1) / P.B)) : bound);
  static const bool stats_on = std::getenv("KVFE_SUBPIX_STATS") != nullptr;
  const int kappend = append | (stats_on ? 16 : 0);   // (bit 4: per-corner cycle statistics)
  if (stats_on) {
    static bool reg = false;
    if (!reg) {
      reg = true;
      std::atexit([] {
        unsigned long long h[8];
        unsigned long long g[8];
        if (hipMemcpyFromSymbol(g, HIP_SYMBOL(kvfe_spg_stats), sizeof(g)) == hipSuccess && g[5])
          std::fprintf(stderr, "KVFE_SUBPIX_STATS (group kernel) blocks %llu, cycles per block: patch %.0f | barrier %.0f | first chunk %.0f | chunk loop %.0f | solve + loop %.0f | iterations per block %.1f, chunk loop per iteration %.0f\n",
                       g[5], (double)g[0] / g[5], (double)g[1] / g[5], (double)g[2] / g[5], (double)g[3] / g[5], (double)g[4] / g[5],
                       (double)g[6] / g[5], g[6] ? (double)g[3] / g[6] : 0.0);
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(kvfe_subpix_stats), sizeof(h)) == hipSuccess && h[0])
          std::fprintf(stderr, "KVFE_SUBPIX_STATS corners %llu, cycles per corner: mean %.0f max %llu; < 100 k: %llu, < 200 k: %llu, < 400 k: %llu, more: %llu\n",
                       h[0], (double)h[1] / (double)h[0], h[2], h[3], h[4], h[5], h[6]);
      });
    }
  }
  // Two kernels share the launch slot when there are many streams (see subpix_total_new): the one-corner-per-block
  // kernel below works when the launch holds fewer than SPG_MIN_TOTAL new corners, the grouped kernel after it when it
  // holds more; the other one's blocks return at once.  KVFE_SUBPIX_GROUP = 0 / 1 forces one of them (A/B, tests).
  int kapp = kappend | (group_mode == 1 ? 128 : 0);
  if (group_mode != 2) {
  if (P.subpix_win == 10 && nw == 4)
    hipLaunchKernelGGL((subpix_append_kernel<10, 4>), grid, dim3(256), lds, st, P, T, img,
                       row_stride, img_stride, k, S, D, kapp);
  else if (P.subpix_win == 10 && nw == 2)
    hipLaunchKernelGGL((subpix_append_kernel<10, 2>), grid, dim3(128), lds, st, P, T, img,
                       row_stride, img_stride, k, S, D, kapp);
  else if (P.subpix_win == 10)
    hipLaunchKernelGGL((subpix_append_kernel<10, 1>), grid, dim3(64), lds, st, P, T, img,
                       row_stride, img_stride, k, S, D, kapp);
  else
    hipLaunchKernelGGL((subpix_append_kernel<0, 1>), grid, dim3(64), lds, st, P, T, img,
                       row_stride, img_stride, k, S, D, kapp);
  }
  if (group_mode != 0) {
    static const int budget = lds_dynamic_budget(reinterpret_cast<const void*>(subpix_group_kernel<10>));
    const size_t glds = spg_geom(10).bytes;
    if ((long long)glds > budget) std::fprintf(stderr, "kvfe: subpix_group_kernel needs %zu B of LDS, %d available\n", glds, budget);
    // blocks per stream: what the device holds at once (two 65 KB blocks per compute unit) shared out over the streams --
    // every block resident from the start, its eight slots refilled from the stream's counter as corners finish
    const int cus = P.n_cu > 0 ? P.n_cu : 256;   // (of the context's device, KParams)
    const int gy = std::max(1, std::min((bound + SPG_G - 1) / SPG_G, (2 * cus + P.B - 1) / P.B));
    hipLaunchKernelGGL((subpix_group_kernel<10>), dim3(P.B, gy), dim3(SPG_T), glds, st, P, T, img,
                       row_stride, img_stride, k, S, D, append | (stats_on ? 16 : 0) | (group_mode == 2 ? 64 : 0));
  }
  if (append) hipLaunchKernelGGL(detect_commit_kernel, dim3(P.B), dim3(DC_T), 0, st, P, T, k, S, D, 7);
}

template <int WIN, int NW>
__global__ __launch_bounds__(64 * NW) void subpix_points_kernel(const float* __restrict__ mask,
                                                           const unsigned char* __restrict__ img,
                                                           size_t row_stride, int W, int H,
                                                           float2* pts, int n, int win,
                                                           int max_iters, double eps2) {
  const int ci = blockIdx.x;
  if (ci >= n) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const float2 c = corner_subpix_wave<WIN, NW>(img, row_stride, W, H, pts[ci], win, max_iters, eps2, mask,
                                          lds_raw, threadIdx.x);
  if (threadIdx.x == 0) pts[ci] = c;
}

void launch_subpix_points(const KParams& P, const float* mask_tab, const unsigned char* img,
                          size_t row_stride, int W, int H, float2* pts, int n, int win,
                          int max_iters, double eps2, hipStream_t st) {
  if (n <= 0) return;
  const size_t lds = subpix_geom(win).bytes;
  const int nw = n <= 256 ? 4 : 2;
  if (win == 10 && nw == 4)
    hipLaunchKernelGGL((subpix_points_kernel<10, 4>), dim3(n), dim3(256), lds, st, mask_tab, img, row_stride,
                       W, H, pts, n, win, max_iters, eps2);
  else if (win == 10 && nw == 2)
    hipLaunchKernelGGL((subpix_points_kernel<10, 2>), dim3(n), dim3(128), lds, st, mask_tab, img, row_stride,
                       W, H, pts, n, win, max_iters, eps2);
  else if (win == 10)
    hipLaunchKernelGGL((subpix_points_kernel<10, 1>), dim3(n), dim3(64), lds, st, mask_tab, img, row_stride,
                       W, H, pts, n, win, max_iters, eps2);
  else
    hipLaunchKernelGGL((subpix_points_kernel<0, 1>), dim3(n), dim3(64), lds, st, mask_tab, img, row_stride,
                       W, H, pts, n, win, max_iters, eps2);
}

__global__ void undistort_points_kernel(UndistortDev U, const float2* __restrict__ in, int n,
                                        float2* out, double* versors) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float2 p = in[i];
  if (out) {
    float ox, oy;
    undistort_point_dev(U, p.x, p.y, &ox, &oy);
    out[i] = make_float2(ox, oy);
  }
  if (versors) bearing_vector(U, p.x, p.y, versors + (size_t)i * 3);
}

void launch_undistort_points(const UndistortDev& U, const float2* in, int n, float2* out,
                             double* versors, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(undistort_points_kernel, dim3((n + 63) / 64), dim3(64), 0, st, U, in, n, out,
                     versors);
}

}  // namespace kvfe
