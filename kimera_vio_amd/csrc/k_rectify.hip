// K1  undistort-rectify  == cv::remap(INTER_LINEAR, BORDER_REPLICATE) on 8-bit images
//     reference: UndistorterRectifier::undistortRectifyImage, src/frontend/UndistorterRectifier.cpp:115-128
// K4a pyramid level     == cv::pyrDown inside cv::buildOpticalFlowPyramid
//     reference: cv::calcOpticalFlowPyrLK call at src/frontend/Tracker.cpp:137-146
//
// Both are HBM-bound gather/stencil kernels: 16 B/lane coalesced map reads, 4 output pixels per
// lane (one 4-byte store), source gathers served by L1/L2 (a rectification map is locally smooth,
// so a wave touches ~2-3 source rows).  Integer arithmetic only => bit-exact vs the CPU path.
#include "kvfe_dev.hpp"

#include <cstdint>
#include <cstdlib>

namespace kvfe {

__device__ __forceinline__ int clipi(int x, int b) { return x >= 0 ? (x < b ? x : b - 1) : 0; }

// cv::remap taps of one output pixel: 1/32-px fixed-point coordinates, 15-bit weights, the four
// source offsets already clamped (BORDER_REPLICATE).  Depends only on the map, not on the image:
// a block computes the taps once and applies them to SPB streams.
struct RemapTap {
  int o00, o01, o10, o11;   // byte offsets into the source image
  int w00, w01, w10, w11;
};

__device__ __forceinline__ RemapTap remap_tap(int W, int H, int stride, float mx, float my) {
  const int sxf = __float2int_rn(mx * 32.0f);
  const int syf = __float2int_rn(my * 32.0f);
  const int ax = sxf & 31, ay = syf & 31;
  int sx = sxf >> 5, sy = syf >> 5;
  sx = max(-32768, min(32767, sx));  // saturate_cast<short>
  sy = max(-32768, min(32767, sy));
  RemapTap t;
  t.w00 = (32 - ax) * (32 - ay) * 32;
  t.w01 = ax * (32 - ay) * 32;
  t.w10 = (32 - ax) * ay * 32;
  t.w11 = ax * ay * 32;
  if ((ax | ay) == 0) {  // table entry (0,0) after saturation and fix-up: {32767,0,0,1}
    t.w00 = 32767;
    t.w11 = 1;
  }
  const int sx0 = clipi(sx, W), sx1 = clipi(sx + 1, W);
  const int sy0 = clipi(sy, H) * stride, sy1 = clipi(sy + 1, H) * stride;
  t.o00 = sy0 + sx0;
  t.o01 = sy0 + sx1;
  t.o10 = sy1 + sx0;
  t.o11 = sy1 + sx1;
  return t;
}

__device__ __forceinline__ unsigned remap_apply(const unsigned char* __restrict__ src, const RemapTap& t) {
  const int r = ((int)src[t.o00] * t.w00 + (int)src[t.o01] * t.w01 + (int)src[t.o10] * t.w10 +
                 (int)src[t.o11] * t.w11 + (1 << 14)) >> 15;
  // the four weights sum to 2^15, so 0 <= r <= 255 without the saturate_cast of the reference.
  // (Deliberately no min/max here: hipcc 7.2 fuses "ashr, clamp, pack" into v_ashr_pk_u8_i32,
  // which leaves the upper half of its destination unwritten -> corrupted third pixel.)
  return (unsigned)r & 0xffu;
}

constexpr int RECT_SPB = 8;  // streams per block: the map is read once per RECT_SPB images

template <bool VEC4>
__global__ __launch_bounds__(256) void rectify_kernel(
    const unsigned char* __restrict__ src0, const unsigned char* __restrict__ src1,
    size_t src_row_stride, size_t src_img_stride, unsigned char* __restrict__ dst0,
    unsigned char* __restrict__ dst1, const float2* __restrict__ map0,
    const float2* __restrict__ map1, int W, int H, int B, const int* __restrict__ flags,
    int act_flag, int n_tiles, int gz, int mode, const int* __restrict__ skip) {
  // stream s is worked on iff (flags == null or flags[s] & act_flag) and not (skip != null and skip[s])
  auto inactive = [&](int s) { return (flags && !(flags[s] & act_flag)) || (skip && skip[s]); };
  // XCD-aware block -> tile map.  Workgroups are dealt round-robin to the 8 XCDs (each with its
  // own L2), so linear block id L runs on XCD L & 7.  XCD k owns the k-th horizontal band of the
  // image for ALL streams and both cameras, and walks it tile-major / stream-group-minor: a map
  // tile comes from HBM once (the other stream groups hit it in this XCD's L2) and the source rows
  // shared by vertically adjacent tiles are fetched by one L2 only.  (Placement is a speed
  // assumption, never a correctness one.)
  const int L = blockIdx.x;
  int tile, cam, s_begin;
  if (mode == 3) {  // 3-D grid (tile, camera, stream group)
    tile = blockIdx.x;
    cam = blockIdx.y;
    s_begin = blockIdx.z * RECT_SPB;
  } else if (mode == 0) {  // plain: tile fastest, then camera, then stream group (XCD-oblivious)
    const int bt8 = 8 * ((n_tiles + 7) >> 3);
    tile = L % bt8;
    const int rem = L / bt8;
    cam = rem & 1;
    s_begin = (rem >> 1) * RECT_SPB;
    if (tile >= n_tiles) return;
  } else {
    const int xcd = L & 7, j = L >> 3;
    const int band0 = (xcd * n_tiles) >> 3, band1 = ((xcd + 1) * n_tiles) >> 3;
    const int band_tiles = (n_tiles + 7) >> 3;
    int rem;
    if (mode == 1) {  // tile-major, stream-group-minor
      const int per_tile = 2 * gz;
      tile = band0 + j / per_tile;
      rem = j % per_tile;
    } else {          // stream-group-major, tile-minor
      tile = band0 + j % band_tiles;
      rem = j / band_tiles;
    }
    if (tile >= band1) return;
    cam = rem / gz;
    s_begin = (rem - cam * gz) * RECT_SPB;
  }
  const int s_end = min(B, s_begin + RECT_SPB);
  const unsigned char* src = cam == 0 ? src0 : src1;
  unsigned char* dst = cam == 0 ? dst0 : dst1;
  const float2* map = cam == 0 ? map0 : map1;
  const int N = W * H;
  const int stride = (int)src_row_stride;
  if (VEC4) {
    const int i = (tile * 256 + threadIdx.x) * 4;
    if (i >= N) return;
    const float4 m01 = *reinterpret_cast<const float4*>(map + i);
    const float4 m23 = *reinterpret_cast<const float4*>(map + i + 2);
    const RemapTap t0 = remap_tap(W, H, stride, m01.x, m01.y);
    const RemapTap t1 = remap_tap(W, H, stride, m01.z, m01.w);
    const RemapTap t2 = remap_tap(W, H, stride, m23.x, m23.y);
    const RemapTap t3 = remap_tap(W, H, stride, m23.z, m23.w);
    // A rectification map is locally smooth: the 16 taps of four neighbouring output pixels
    // normally lie in two source rows within 8 bytes.  Then two unaligned 8-byte loads per image
    // replace 16 byte gathers; bytes are picked with v_perm_b32 and blended with v_dot2_i32_i16.
    const int r0 = t0.o00 - (t0.o00 % stride), r1 = t0.o10 - (t0.o10 % stride);
    const int xmin = min(min(t0.o00, t1.o00), min(t2.o00, t3.o00)) - r0;
    auto in_rows = [&](const RemapTap& t) {
      const int x0 = t.o00 - r0, x1 = t.o01 - r0;
      return t.o10 - r1 == x0 && t.o11 - r1 == x1 && x0 >= xmin && x0 < xmin + 8 && x1 >= xmin &&
             x1 < xmin + 8 && x0 < stride && x1 < stride;
    };
    const bool fast = in_rows(t0) && in_rows(t1) && in_rows(t2) && in_rows(t3) &&
                      (long long)r1 + xmin + 8 <= (long long)stride * H &&
                      (long long)r0 + xmin + 8 <= (long long)stride * H;
    if (fast) {
      auto sel_of = [&](const RemapTap& t) {  // (byte x0) | 0 << 8 | (byte x1) << 16 | 0 << 24
        const unsigned b0 = (unsigned)(t.o00 - r0 - xmin), b1 = (unsigned)(t.o01 - r0 - xmin);
        return b0 | (0x0cu << 8) | (b1 << 16) | (0x0cu << 24);
      };
      typedef short v2s_ __attribute__((ext_vector_type(2)));
      auto pk = [](int lo, int hi) { return (lo & 0xffff) | (hi << 16); };
      const unsigned e0 = sel_of(t0), e1 = sel_of(t1), e2 = sel_of(t2), e3 = sel_of(t3);
      const int a0 = pk(t0.w00, t0.w01), b0 = pk(t0.w10, t0.w11), a1 = pk(t1.w00, t1.w01),
                b1 = pk(t1.w10, t1.w11), a2 = pk(t2.w00, t2.w01), b2 = pk(t2.w10, t2.w11),
                a3 = pk(t3.w00, t3.w01), b3 = pk(t3.w10, t3.w11);
      const int off0 = r0 + xmin, off1 = r1 + xmin;
      auto blend = [&](uint2 u, uint2 v, unsigned sel, int wa, int wb) {
        const unsigned p = __builtin_amdgcn_perm(u.y, u.x, sel), q = __builtin_amdgcn_perm(v.y, v.x, sel);
        const int r = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s_, p), __builtin_bit_cast(v2s_, wa),
                                             __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s_, q),
                                                                    __builtin_bit_cast(v2s_, wb),
                                                                    1 << 14, false),
                                             false) >> 15;
        return (unsigned)r & 0xffu;  // weights sum to 2^15: already within 0..255 (see remap_apply)
      };
      // all RECT_SPB streams' source words are requested before the first one is used (the loop
      // is otherwise one exposed memory latency per stream); out-of-range streams re-read the
      // last one and are not stored
      uint2 u[RECT_SPB], v[RECT_SPB];
#pragma unroll
      for (int k = 0; k < RECT_SPB; k++) {
        const unsigned char* S = src + (size_t)min(s_begin + k, B - 1) * src_img_stride;
        __builtin_memcpy(&u[k], S + off0, 8);
        __builtin_memcpy(&v[k], S + off1, 8);
      }
#pragma unroll
      for (int k = 0; k < RECT_SPB; k++) {
        const int s = s_begin + k;
        if (s >= s_end || inactive(s)) continue;
        const unsigned p0 = blend(u[k], v[k], e0, a0, b0), p1 = blend(u[k], v[k], e1, a1, b1),
                       p2 = blend(u[k], v[k], e2, a2, b2), p3 = blend(u[k], v[k], e3, a3, b3);
        *reinterpret_cast<unsigned*>(dst + (size_t)s * N + i) = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
      }
    } else {
      for (int s = s_begin; s < s_end; s++) {
        if (inactive(s)) continue;
        const unsigned char* S = src + (size_t)s * src_img_stride;
        const unsigned p0 = remap_apply(S, t0), p1 = remap_apply(S, t1), p2 = remap_apply(S, t2),
                       p3 = remap_apply(S, t3);
        *reinterpret_cast<unsigned*>(dst + (size_t)s * N + i) = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
      }
    }
  } else {
    const int i = tile * 256 + threadIdx.x;
    if (i >= N) return;
    const float2 m = map[i];
    const RemapTap t = remap_tap(W, H, stride, m.x, m.y);
    for (int s = s_begin; s < s_end; s++) {
      if (inactive(s)) continue;
      dst[(size_t)s * N + i] = (unsigned char)remap_apply(src + (size_t)s * src_img_stride, t);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Tiled cv::remap: one block = one 128 x 16 output tile of one camera for SPB streams.
//   phase A  the block reads its 16 KB map tile once (32 B per lane and row, coalesced), turns every
//            pixel into (clamped source position, four 15-bit weights) -- BORDER_REPLICATE is folded
//            into the weights: a tap that clamps onto its neighbour hands its weight over, so every
//            pixel reads the 2x2 block at its clamped position -- and reduces the bounding box of
//            the source positions over the block;
//   phase B  the bounding box (16-byte aligned in x; <= 160 B x 32 rows for the EuRoC / D455 maps,
//            whose rows drift ~7 px per 128 columns) of each of the SPB streams is copied into LDS
//            with global_load_lds_dwordx4 (LDS-DMA: 1 KiB per wave instruction, no VGPR round trip,
//            every byte of a source row segment requested exactly once per block);
//   phase C  16 ds_read_u8 + 8 v_dot2_u32_u16 per lane and stream blend 8 pixels; two coalesced dword
//            stores per lane (a wave writes 2 x 128 B row segments per store instruction).
// The taps are computed once per block and reused for all SPB streams (registers: 3 per pixel).
// Blocks whose box does not fit (exotic maps) take the gather path of the same kernel, so any map
// is handled; images whose rows are not 16-byte aligned use rectify_kernel above.
// ---------------------------------------------------------------------------------------------
constexpr int RT_W = 128;                        // tile width; the tile height TH (16 | 32) is a template parameter
constexpr int RT_CPR = 16;                       // 16-byte chunk slots per staged source row: the row pitch of 256 B
constexpr int RT_PITCH = RT_CPR * 16;            //   = 64 banks keeps the 32 lanes of a row group on 32 different banks
                                                 //   however the map drifts across source rows (a 160-B pitch cost
                                                 //   2.2 bank-conflict cycles per LDS cycle)
constexpr int rt_rows(int th) { return th + 16; }          // staged source rows per stream
constexpr int rt_patch(int th) { return RT_PITCH * (rt_rows(th) + 1) + 16; }  // folded taps read one row / byte further

typedef __attribute__((address_space(3))) unsigned char lds_u8_t;
typedef __attribute__((address_space(1))) const unsigned char glb_cu8_t;
typedef unsigned short v2u16_ __attribute__((ext_vector_type(2)));

struct RTap {
  int cx, cy;          // clamped source position of the 2x2 block
  unsigned w0, w1;     // w00 | w01 << 16, w10 | w11 << 16 (after folding; each <= 32768)
};

// the two halves of a tap: where the 2 x 2 block sits and which of its columns / rows fold onto each other (rtap_parts),
// and the four 15-bit weights that follow from the 5-bit fractions and the folds (rtap_weights).  The packed tap table of
// the tile kernel stores the first half per pixel and recomputes the second -- through these same functions.
__device__ __forceinline__ void rtap_parts(int W, int H, float mx, float my, int& cx, int& cy, int& ax, int& ay,
                                           bool& foldx, bool& foldy) {
  const int sxf = __float2int_rn(mx * 32.0f);
  const int syf = __float2int_rn(my * 32.0f);
  ax = sxf & 31;
  ay = syf & 31;
  int sx = sxf >> 5, sy = syf >> 5;
  sx = max(-32768, min(32767, sx));  // saturate_cast<short>
  sy = max(-32768, min(32767, sy));
  const int sx0 = clipi(sx, W), sx1 = clipi(sx + 1, W), sy0 = clipi(sy, H), sy1 = clipi(sy + 1, H);
  cx = sx0;
  cy = sy0;
  foldx = sx1 == sx0;  // both columns clamp onto the same source column
  foldy = sy1 == sy0;
}
__device__ __forceinline__ void rtap_weights(int ax, int ay, bool foldx, bool foldy, unsigned& w0, unsigned& w1) {
  int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
  if ((ax | ay) == 0) {  // table entry (0,0) after saturation and fix-up: {32767,0,0,1}
    w00 = 32767;
    w11 = 1;
  }
  if (foldx) {
    w00 += w01;
    w10 += w11;
    w01 = w11 = 0;
  }
  if (foldy) {
    w00 += w10;
    w01 += w11;
    w10 = w11 = 0;
  }
  w0 = (unsigned)w00 | ((unsigned)w01 << 16);
  w1 = (unsigned)w10 | ((unsigned)w11 << 16);
}
__device__ __forceinline__ RTap rtap(int W, int H, float mx, float my) {
  int ax, ay;
  bool foldx, foldy;
  RTap t;
  rtap_parts(W, H, mx, my, t.cx, t.cy, ax, ay, foldx, foldy);
  rtap_weights(ax, ay, foldx, foldy, t.w0, t.w1);
  return t;
}

__device__ __forceinline__ unsigned rblend(unsigned p00, unsigned p01, unsigned p10, unsigned p11, unsigned w0,
                                           unsigned w1) {
  const unsigned p = p00 | (p01 << 16), q = p10 | (p11 << 16);
  const unsigned r = __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16_, p), __builtin_bit_cast(v2u16_, w0),
                                            __builtin_amdgcn_udot2(__builtin_bit_cast(v2u16_, q),
                                                                   __builtin_bit_cast(v2u16_, w1), 1u << 14, false),
                                            false);
  return r >> 15;   // the weights sum to 2^15: 0 <= r <= 255
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
  return v;
}

// the block's source box from the lanes' tap extents: x_lo (16-byte aligned), y_lo, 16-byte chunks per row and rows that
// hold needed bytes; ncx = 0 when the tile has no pixel or its box does not fit the LDS stage
template <int TH>
__device__ __forceinline__ void rt_block_box(int mnx, int mxx, int mny, int mxy, int* bounds, int lane, int wave, int W,
                                             int H, int& x_lo, int& y_lo, int& ncx, int& ph) {
  mnx = wave_min(mnx);
  mxx = wave_max(mxx);
  mny = wave_min(mny);
  mxy = wave_max(mxy);
  if (lane == 0) {
    bounds[wave * 4 + 0] = mnx;
    bounds[wave * 4 + 1] = mxx;
    bounds[wave * 4 + 2] = mny;
    bounds[wave * 4 + 3] = mxy;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 4; w++) {
    mnx = min(mnx, bounds[w * 4 + 0]);
    mxx = max(mxx, bounds[w * 4 + 1]);
    mny = min(mny, bounds[w * 4 + 2]);
    mxy = max(mxy, bounds[w * 4 + 3]);
  }
  x_lo = mnx & ~15;
  y_lo = mny;
  ncx = ((min(mxx + 1, W - 1) - x_lo) >> 4) + 1;
  ph = min(mxy + 1, H - 1) - y_lo + 1;
  if (!(mxx >= 0 && ncx <= RT_CPR && ph <= rt_rows(TH))) ncx = 0;
}

// the boxes of all 128 x 16 tiles of one camera (context creation): one block per tile
__global__ __launch_bounds__(256) void rectify_box_kernel(const float2* __restrict__ map, int W, int H, int tiles_x,
                                                          int4* __restrict__ box) {
  __shared__ int bounds[16];
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x = tx * RT_W + (tid & 31) * 4, y0 = ty * 16 + (tid >> 5);
  int mnx = 1 << 30, mxx = -1, mny = 1 << 30, mxy = -1;
  for (int r = 0; r < 2; r++) {
    const int y = y0 + 8 * r;
    if (x < W && y < H)
      for (int q = 0; q < 4 && x + q < W; q++) {   // (a width that is no multiple of 4: the last lane's columns end at W --
        const float2 m = map[y * W + x + q];        // round 6, KVFE_GUARD_ALLOC=1: the last row read 16 bytes past the map)
        const RTap t = rtap(W, H, m.x, m.y);
        mnx = min(mnx, t.cx);
        mxx = max(mxx, t.cx);
        mny = min(mny, t.cy);
        mxy = max(mxy, t.cy);
      }
  }
  int x_lo, y_lo, ncx, ph;
  rt_block_box<16>(mnx, mxx, mny, mxy, bounds, lane, wave, W, H, x_lo, y_lo, ncx, ph);
  if (tid == 0) box[blockIdx.x] = make_int4(x_lo, y_lo, ncx, ph);
}

void launch_rectify_boxes(const float2* map, int W, int H, int4* box, hipStream_t st) {
  const int tiles_x = (W + RT_W - 1) / RT_W, tiles_y = (H + 15) / 16;
  hipLaunchKernelGGL(rectify_box_kernel, dim3(tiles_x * tiles_y), dim3(256), 0, st, map, W, H, tiles_x, box);
}

size_t rectify_box_count(int W, int H) { return (size_t)((W + RT_W - 1) / RT_W) * ((H + 15) / 16); }

// Packed taps of the staged tiles (context creation, behind the boxes): per output pixel ONE dword instead of the 8-byte
// map entry --
//   bits 0..13  byte offset of the 2 x 2 block inside the tile's staged source box ((cy - y_lo) * RT_PITCH + cx - x_lo),
//   bits 14..18 / 19..23  the 5-bit fractions ax / ay,   bit 24 / 25  columns / rows fold (BORDER_REPLICATE)
// -- everything rtap() derives from the map entry that does not depend on the frame.  The tile kernel reads 8 KB instead
// of 16 KB of map per tile and stream group (the map was a quarter of the launch's fetched bytes) and skips the float
// conversions and clamps of phase A.  Tiles whose box does not fit the LDS stage keep the float map (gather path).
constexpr unsigned RT_TAP_FOLDX = 1u << 24, RT_TAP_FOLDY = 1u << 25;
static_assert(rt_patch(16) <= (1 << 14), "box offsets fit 14 bits");
__global__ __launch_bounds__(256) void rectify_pack_kernel(const float2* __restrict__ map, int W, int H, int tiles_x,
                                                           const int4* __restrict__ box, unsigned* __restrict__ tap) {
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int tid = threadIdx.x;
  const int x = tx * RT_W + (tid & 31) * 4, y0 = ty * 16 + (tid >> 5);
  const int4 bx = box[blockIdx.x];
  for (int r = 0; r < 2; r++) {
    const int y = y0 + 8 * r;
    if (x < W && y < H)
      for (int q = 0; q < 4 && x + q < W; q++) {
        const float2 m = map[y * W + x + q];
        int cx, cy, ax, ay;
        bool fx, fy;
        rtap_parts(W, H, m.x, m.y, cx, cy, ax, ay, fx, fy);
        unsigned t = 0;
        if (bx.z > 0)
          t = (unsigned)((cy - bx.y) * RT_PITCH + (cx - bx.x)) | ((unsigned)ax << 14) | ((unsigned)ay << 19) |
              (fx ? RT_TAP_FOLDX : 0u) | (fy ? RT_TAP_FOLDY : 0u);
        tap[y * W + x + q] = t;
      }
  }
}

void launch_rectify_taps(const float2* map, int W, int H, const int4* box, unsigned* tap, hipStream_t st) {
  const int tiles_x = (W + RT_W - 1) / RT_W, tiles_y = (H + 15) / 16;
  hipLaunchKernelGGL(rectify_pack_kernel, dim3(tiles_x * tiles_y), dim3(256), 0, st, map, W, H, tiles_x, box, tap);
}

template <int TH, int SPB, int NSUB, int MINW>
__global__ __launch_bounds__(256, MINW) void rectify_tile_kernel(
    const unsigned char* __restrict__ src0, const unsigned char* __restrict__ src1, size_t src_row_stride,
    size_t src_img_stride, unsigned char* __restrict__ dst0, unsigned char* __restrict__ dst1,
    const float2* __restrict__ map0, const float2* __restrict__ map1, int W, int H, int B,
    const int* __restrict__ flags, int act_flag, int tiles_x, int tiles_y, int gz, int mode,
    const int* __restrict__ skip, const int4* __restrict__ box0, const int4* __restrict__ box1,
    const unsigned* __restrict__ tap0, const unsigned* __restrict__ tap1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  constexpr int NBUF = NSUB > 1 ? 2 : 1;   // LDS boxes: SPB streams x NBUF buffers
  constexpr int NR = TH / 8;               // tile rows per lane
  constexpr int RT_ROWS = rt_rows(TH), RT_PATCH = rt_patch(TH);
  constexpr int NIT = (RT_ROWS * RT_CPR + 255) / 256;   // DMA wave-instructions per lane and box
  int* bounds = reinterpret_cast<int*>(sm + (size_t)SPB * NBUF * RT_PATCH);   // [4 waves][4]
  int tx, ty, cam, s_begin;
  if ((mode & 1) == 0) {  // 3-D grid (tile, camera, stream group)
    tx = blockIdx.x % tiles_x;
    ty = blockIdx.x / tiles_x;
    cam = blockIdx.y;
    s_begin = blockIdx.z * (SPB * NSUB);
  } else if (mode & 8) {
    // 1-D grid, an XCD per STREAM GROUP (round 6).  Workgroups are dealt round-robin to the 8 XCDs (block L runs on XCD
    // L & 7, a speed assumption only): XCD k takes the stream groups k, k + 8, .. and walks a group's tiles in raster order,
    // camera by camera, so the source rows two neighbouring tiles share (a 128 x 16 tile stages ~24 rows of ~160 bytes:
    // 1.9 x its own pixels) are fetched by ONE L2 and are still there when the neighbour asks.  Every XCD does the same
    // amount of work whatever the image height (the banded order below gives 3 or 4 of 30 tile rows to an XCD).
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int per = tiles_x * tiles_y * 2;
    const int g = xcd + 8 * (j / per);
    if (g >= gz) return;
    const int t = j % per;
    cam = t / (tiles_x * tiles_y);
    const int tt = t - cam * (tiles_x * tiles_y);
    ty = tt / tiles_x;
    tx = tt - ty * tiles_x;
    s_begin = g * (SPB * NSUB);
  } else {
    // XCD-banded 1-D grid.  Workgroups are dealt round-robin to the 8 XCDs (block L runs on XCD L & 7, a speed
    // assumption only): XCD k owns the k-th band of tile rows of both cameras for ALL streams and walks it
    // tile-major / stream-group-minor, so a map tile comes from HBM once (the other stream groups hit it in this
    // XCD's L2) and the source rows shared by neighbouring tiles are fetched by one L2 only.
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int r0 = (xcd * tiles_y) >> 3, r1 = ((xcd + 1) * tiles_y) >> 3;
    const int nb = (r1 - r0) * tiles_x;
    const int g = j % gz, t = j / gz;
    if (t >= 2 * nb) return;
    cam = t / nb;
    const int tt = t - cam * nb;
    ty = r0 + tt / tiles_x;
    tx = tt % tiles_x;
    s_begin = g * (SPB * NSUB);
  }
  const unsigned char* src = cam == 0 ? src0 : src1;
  unsigned char* dst = cam == 0 ? dst0 : dst1;
  const float2* map = cam == 0 ? map0 : map1;
  const int N = W * H;
  const int stride = (int)src_row_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lx = tid & 31, ly = tid >> 5;
  const int x = tx * RT_W + lx * 4;
  const int y0 = ty * TH + ly;
  // stream activity (block-uniform), one bit per stream of the block
  unsigned act = 0;
#pragma unroll
  for (int k = 0; k < SPB * NSUB; k++) {
    const int s = s_begin + k;
    if (s < B && (!flags || (flags[s] & act_flag)) && !(skip && skip[s])) act |= 1u << k;
  }
  if (!act) return;
  // The source box of a tile depends on the maps only: for TH = 16 it is tabulated when the context is created
  // (rectify_box_kernel, the same taps and the same reduction), so the first boxes are requested BEFORE the map tile is
  // read and the taps are computed -- one memory round trip of the five a block used to make in sequence.
  const int4* box = cam == 0 ? box0 : box1;
  const bool tabulated = TH == 16 && box != nullptr;
  int x_lo = 0, y_lo = 0, ncx = 0, ph = 0;
  bool staged = false;
  if (tabulated) {
    const int4 bx = box[ty * tiles_x + tx];
    x_lo = bx.x;
    y_lo = bx.y;
    ncx = bx.z;
    ph = bx.w;
    staged = ncx > 0 && !(mode & 2);   // mode bit 1: force the gather path (tests)
  }
  // this lane's (at most NIT) 16-byte chunks of a box: offsets are the same for every stream
  int goff[NIT];
  bool gok[NIT];
  int n = 0;
  const unsigned char* Sbase = src;
  auto plan = [&]() {
    n = ph * RT_CPR;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int c = wave * 64 + it * 256 + lane;
      const int row = c / RT_CPR, ch = c - row * RT_CPR;
      gok[it] = c < n && ch < ncx;
      goff[it] = row * stride + ch * 16;
    }
    Sbase = src + (size_t)y_lo * stride + x_lo;
  };
  auto issue = [&](int j) {
    unsigned char* pb = sm + (j & (NBUF - 1)) * (SPB * RT_PATCH);
#pragma unroll
    for (int k = 0; k < SPB; k++) {
      if (!((act >> (j * SPB + k)) & 1u)) continue;
      const unsigned char* S = Sbase + (size_t)(s_begin + j * SPB + k) * src_img_stride;
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        if (wave * 64 + it * 256 < n && gok[it])
          __builtin_amdgcn_global_load_lds((glb_cu8_t*)(S + goff[it]),
                                           (lds_u8_t*)(pb + k * RT_PATCH + (wave * 64 + it * 256) * 16), 16, 0, 0);
      }
    }
  };
  const bool pairs = NSUB > 1 && (mode & 4);
  if (tabulated && staged) {
    plan();
    issue(0);
    if (pairs) issue(1);
  }
  // ---- phase A: taps of this lane's NR x 4 pixels ----------------------------------------------------------
  RTap tp[4 * NR];
  int mnx = 1 << 30, mxx = -1, mny = 1 << 30, mxy = -1;
  bool okr[NR];
  // packed taps (tabulated boxes only: see rectify_pack_kernel): cx holds the byte offset inside the staged box
  const unsigned* tap = cam == 0 ? tap0 : tap1;
  const bool packed = tabulated && staged && tap != nullptr;
  if (packed) {
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int y = y0 + 8 * r;
      okr[r] = x < W && y < H;
      uint4 t4 = make_uint4(0u, 0u, 0u, 0u);
      if (okr[r]) t4 = *reinterpret_cast<const uint4*>(tap + (size_t)y * W + x);
      const unsigned tq[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        RTap t;
        t.cx = (int)(tq[q] & 0x3fffu);
        t.cy = 0;
        rtap_weights((int)((tq[q] >> 14) & 31u), (int)((tq[q] >> 19) & 31u), (tq[q] & RT_TAP_FOLDX) != 0u,
                     (tq[q] & RT_TAP_FOLDY) != 0u, t.w0, t.w1);
        if (!okr[r]) t = RTap{0, 0, 0u, 0u};
        tp[4 * r + q] = t;
      }
    }
  } else {
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int y = y0 + 8 * r;
    okr[r] = x < W && y < H;
    if (okr[r]) {
      const int i = y * W + x;
      const float4 m01 = *reinterpret_cast<const float4*>(map + i);
      const float4 m23 = *reinterpret_cast<const float4*>(map + i + 2);
      tp[4 * r + 0] = rtap(W, H, m01.x, m01.y);
      tp[4 * r + 1] = rtap(W, H, m01.z, m01.w);
      tp[4 * r + 2] = rtap(W, H, m23.x, m23.y);
      tp[4 * r + 3] = rtap(W, H, m23.z, m23.w);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        mnx = min(mnx, tp[4 * r + q].cx);
        mxx = max(mxx, tp[4 * r + q].cx);
        mny = min(mny, tp[4 * r + q].cy);
        mxy = max(mxy, tp[4 * r + q].cy);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) tp[4 * r + q] = RTap{0, 0, 0u, 0u};
    }
  }
  }
  if (!tabulated) {
    rt_block_box<TH>(mnx, mxx, mny, mxy, bounds, lane, wave, W, H, x_lo, y_lo, ncx, ph);
    staged = ncx > 0 && !(mode & 2);
    if (staged) {
      plan();
      issue(0);
      if (pairs) issue(1);
    }
  }
  if (staged) {
    // ---- phase B: LDS-DMA of the source boxes, phase C: blend; software pipeline over sub-chunks of SPB streams:
    //      the boxes of sub-chunk j+1 are in flight while sub-chunk j is blended (two LDS buffers) ----------------
    // LDS byte address of every pixel's 2 x 2 block in buffer 0, stream 0 -- ABSOLUTE (the base of the dynamic LDS area
    // added once here: added per read it cost one v_add_u32 per pixel and stream, hipcc does not fold the relocated
    // base into the address); the buffer / stream offsets of a read are compile-time constants and ride in the
    // instruction's 16-bit offset field
    typedef __attribute__((address_space(3))) const unsigned char lds_cu8_t;
    const unsigned sm_base = (unsigned)(size_t)(lds_u8_t*)sm;
    unsigned ad[4 * NR];
#pragma unroll
    for (int q = 0; q < 4 * NR; q++)
      ad[q] = sm_base + (packed ? (unsigned)tp[q].cx : (unsigned)((tp[q].cy - y_lo) * RT_PITCH + (tp[q].cx - x_lo)));
    auto blend = [&](int j) {
      const int pb_off = (j & (NBUF - 1)) * (SPB * RT_PATCH);
#pragma unroll
      for (int k = 0; k < SPB; k++) {
        if (!((act >> (j * SPB + k)) & 1u)) continue;
        const int pk_off = pb_off + k * RT_PATCH;
        unsigned char* D = dst + (size_t)(s_begin + j * SPB + k) * N;
#pragma unroll
        for (int r = 0; r < NR; r++) {
          if (!okr[r]) continue;
          unsigned px[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            // (round 6: the two pixels of a row in ONE 16-bit read at the byte address -- hipcc emits ds_read_u16 for an
            // align-1 load on gfx950 -- and a byte permute into the dot product's lanes: bit-exact, and the launch takes
            // 0.174 ms instead of 0.042: a 16-bit LDS read at an odd address is replayed; tools/r6/gpu_rect_xcd.sh)
            lds_cu8_t* a = (lds_cu8_t*)(size_t)ad[4 * r + q] + pk_off;
            px[q] = rblend(a[0], a[1], a[RT_PITCH], a[RT_PITCH + 1], tp[4 * r + q].w0, tp[4 * r + q].w1);
          }
          *reinterpret_cast<unsigned*>(D + (size_t)(y0 + 8 * r) * W + x) = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
        }
      }
    };
    if (pairs) {
      // both buffers are requested together and blended together: two memory round trips of 2 SPB boxes per block
      // instead of four of SPB boxes (the barrier drains ALL outstanding requests, so a pipeline of depth two cannot
      // keep more than one sub-chunk in flight while it blends)
#pragma unroll
      for (int j = 0; j < NSUB; j += 2) {
        if (!(act >> (j * SPB))) break;
        if (j > 0) {
          __syncthreads();   // everybody is done with both buffers
          issue(j);
          if (j + 1 < NSUB) issue(j + 1);
        }
        __syncthreads();     // (drains vmcnt: both sub-chunks have landed)
        blend(j);
        if (j + 1 < NSUB) blend(j + 1);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NSUB; j++) {
        if (!(act >> (j * SPB))) break;   // no active stream in this or any later sub-chunk (block-uniform)
        __syncthreads();   // hipcc drains vmcnt before the barrier: sub-chunk j has landed; everybody is done with j-1
        if (j + 1 < NSUB) issue(j + 1);
        blend(j);
      }
    }
  } else {
    // gather path: the box of this tile does not fit the LDS stage (exotic maps only, so it is written for
    // few registers, not for speed: the taps are recomputed pixel by pixel from the map)
#pragma unroll 1
    for (int k = 0; k < SPB * NSUB; k++) {
      if (!((act >> k) & 1u)) continue;
      const unsigned char* S = src + (size_t)(s_begin + k) * src_img_stride;
      unsigned char* D = dst + (size_t)(s_begin + k) * N;
#pragma unroll 1
      for (int e = 0; e < 4 * NR; e++) {
        const int y = y0 + 8 * (e >> 2);
        if (x >= W || y >= H) continue;
        const int i = y * W + x + (e & 3);
        const float2 m = map[i];
        const RTap t = rtap(W, H, m.x, m.y);
        const int xa = t.cx, xb = min(t.cx + 1, W - 1);
        const size_t ra = (size_t)t.cy * stride, rb = (size_t)min(t.cy + 1, H - 1) * stride;
        D[i] = (unsigned char)rblend(S[ra + xa], S[ra + xb], S[rb + xa], S[rb + xb], t.w0, t.w1);
      }
    }
  }
}

void launch_rectify(const KParams& P, const Tables& T, const unsigned char* const src[2],
                    size_t src_row_stride, size_t src_img_stride, unsigned char* const dst[2],
                    const int* flags, int act_flag, hipStream_t st, const int* skip) {
  const int N = P.W * P.H;
  // LDS-staged tiles where the source rows are 16-byte aligned, per-lane gathers (rectify_kernel) otherwise.
  // KVFE_RECT_FORCE_GATHER (debugging aid, tests/test_gpu_parity.py): every tile takes the gather fallback of the tile
  // kernel.  Measured and removed in round 4 (profiles/r3_analysis.md section 6): tile height 32, 1 / 4 streams per LDS
  // buffer, both buffers requested together (-2 %), the XCD-banded block order (-8 % alone, +6 % inside the step: 3 or 4 of
  // the 30 tile rows per XCD).
  static const int fmode = std::getenv("KVFE_RECT_FORCE_GATHER") ? 2 : 0;
  // KVFE_RECT_XCD (A/B switch): 0 = 3-D grid (tile, camera, stream group), 1 = XCD-banded, 2 = an XCD per stream group
  static const int xcd_env = [] { const char* e = std::getenv("KVFE_RECT_XCD"); return e ? std::atoi(e) : -1; }();
  int tmode = fmode;
  const int impl = 1, spb = 2, nsub = 4, th = 16;
  const bool aligned = P.W % 4 == 0 && src_row_stride % 16 == 0 && src_img_stride % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(src[0]) % 16 == 0 && reinterpret_cast<uintptr_t>(src[1]) % 16 == 0;
  if (impl == 1 && aligned) {
    const int TH = th == 16 ? 16 : 32;
    const int tiles_x = (P.W + RT_W - 1) / RT_W, tiles_y = (P.H + TH - 1) / TH;
    // streams per block = SPB (streams per LDS buffer) x NSUB (pipelined sub-chunks; taps computed once for all)
    // (16 streams per block -- taps and map tile shared by twice as many streams -- was measured in round 4: the launch
    // inside the step 0.060 against 0.068 ms, alone 0.048 against 0.044 ms, the step 1.2035 against 1.195 ms: not kept)
    int S = spb == 1 ? 1 : 2, NS = nsub == 1 ? 1 : (nsub == 2 ? 2 : 4);
    if (P.B <= S) NS = 1;
    else if (P.B <= 2 * S && NS > 2) NS = 2;
    const int per_block = S * NS;
    const int gz = (P.B + per_block - 1) / per_block;
    dim3 grid(tiles_x * tiles_y, 2, gz);
    // Round 6 (tools/r6/gpu_rect_xcd.sh, gpu_rect_pmc.sh; 64 x 752x480, same call): 3-D grid 0.0421 / 0.0425 ms alone,
    // FETCH_SIZE 62.4 MB (x 2: 124.8) + WRITE_SIZE 50.9 MB = 1.90 x the algorithmic bytes; XCD-banded 0.0398 / 0.0394; an
    // XCD per stream group 0.0393 / 0.0392 ms, 32.8 MB (x 2: 65.6) + 45.3 MB = 1.20 x -- a third less traffic buys 7 %:
    // the launch is bound by its LDS gather and instruction issue, not by HBM.
    const int xcd_mode = xcd_env >= 0 ? xcd_env : 2;
    if (xcd_mode == 1) tmode |= 1;
    if (xcd_mode == 2 && gz % 8 == 0) tmode |= 1 | 8;   // (fewer groups than XCDs: the 3-D grid)
    if (tmode & 8) grid = dim3(8 * tiles_x * tiles_y * 2 * (gz / 8));
    else if (tmode & 1) grid = dim3(8 * ((tiles_y + 7) / 8) * tiles_x * 2 * gz);
    const size_t lds = (size_t)S * (NS > 1 ? 2 : 1) * rt_patch(TH) + 64;
    const bool use_box = TH == 16 && T.rect_box[0] && T.rect_box[1];
    // KVFE_RECT_FLOAT_MAP (debugging aid, A/B): the tiles read the float map although the packed taps exist
    static const bool float_map = std::getenv("KVFE_RECT_FLOAT_MAP") != nullptr;
    const bool use_tap = use_box && T.rect_tap[0] && T.rect_tap[1] && P.W % 4 == 0 && !float_map;
#define KVFE_RT_LAUNCH(TH_, SPB_, NSUB_, MINW_)                                                                     \
  hipLaunchKernelGGL((rectify_tile_kernel<TH_, SPB_, NSUB_, MINW_>), grid, dim3(256), lds, st, src[0], src[1],        \
                     src_row_stride, src_img_stride, dst[0], dst[1], T.map[0], T.map[1], P.W, P.H, P.B, flags, act_flag, \
                     tiles_x, tiles_y, gz, tmode, skip, use_box ? T.rect_box[0] : nullptr,                               \
                     use_box ? T.rect_box[1] : nullptr, use_tap ? T.rect_tap[0] : nullptr,                               \
                     use_tap ? T.rect_tap[1] : nullptr)
#define KVFE_RT_DISPATCH(TH_, W1_, W2_)                 \
  do {                                                  \
    if (NS == 1) KVFE_RT_LAUNCH(TH_, 2, 1, W1_);        \
    else if (NS == 2) KVFE_RT_LAUNCH(TH_, 2, 2, W2_);   \
    else KVFE_RT_LAUNCH(TH_, 2, 4, W2_);                \
  } while (0)
    KVFE_RT_DISPATCH(16, 7, 4);
#undef KVFE_RT_DISPATCH
#undef KVFE_RT_LAUNCH
    return;
  }
  const int gz = (P.B + RECT_SPB - 1) / RECT_SPB;
  const bool vec4 = N % 4 == 0;
  const int n_tiles = vec4 ? (N / 4 + 255) / 256 : (N + 255) / 256;
  // 1-D grid: 8 XCDs x (tiles of the largest band) x 2 cameras x stream groups (see the kernel)
  const int band_tiles = (n_tiles + 7) / 8;
  dim3 grid(8 * band_tiles * 2 * gz);
  const int mode = 3;   // 3-D grid (tile, camera, stream group); the XCD-banded orders of round 1 are kept in the kernel
  grid = dim3(n_tiles, 2, gz);
  if (vec4)
    hipLaunchKernelGGL(rectify_kernel<true>, grid, dim3(256), 0, st, src[0], src[1],
                       src_row_stride, src_img_stride, dst[0], dst[1], T.map[0], T.map[1], P.W,
                       P.H, P.B, flags, act_flag, n_tiles, gz, mode, skip);
  else
    hipLaunchKernelGGL(rectify_kernel<false>, grid, dim3(256), 0, st, src[0], src[1],
                       src_row_stride, src_img_stride, dst[0], dst[1], T.map[0], T.map[1], P.W,
                       P.H, P.B, flags, act_flag, n_tiles, gz, mode, skip);
}

// ---------------------------------------------------------------------------------------------
// cv::pyrDown: separable [1 4 6 4 1]/16, BORDER_REFLECT_101, (sum + 128) >> 8, dst = (src+1)/2.
// One block = 64x8 output tile; the (2*64+3)x(2*8+3) source patch is staged in LDS, the
// horizontal pass writes 19 x 64 ints to LDS, the vertical pass produces the outputs.
// ---------------------------------------------------------------------------------------------
constexpr int PT_W = 64, PT_H = 8;
constexpr int PS_W = 2 * PT_W + 3, PS_H = 2 * PT_H + 3;

// COPY: the block also writes the 128 x 16 source pixels its tile covers into a dense copy of the source level (the
// context's own level 0 of this frame: next step's LK reads it, so the caller's image pointer is not kept past the step)
template <bool COPY>
__global__ __launch_bounds__(256) void pyrdown_kernel(const unsigned char* __restrict__ src,
                                                      size_t src_row_stride,
                                                      size_t src_img_stride, int sw, int sh,
                                                      unsigned char* __restrict__ dst,
                                                      size_t dst_img_stride, int dw, int dh,
                                                      unsigned char* __restrict__ copy_dst,
                                                      size_t copy_img_stride) {
  __shared__ unsigned char tile[PS_H][PS_W + 1];
  __shared__ int hrow[PS_H][PT_W];
  const int s = blockIdx.z;
  const unsigned char* S = src + (size_t)s * src_img_stride;
  unsigned char* D = dst + (size_t)s * dst_img_stride;
  const int ox = blockIdx.x * PT_W, oy = blockIdx.y * PT_H;
  const int sx0 = 2 * ox - 2, sy0 = 2 * oy - 2;
  for (int i = threadIdx.x; i < PS_W * PS_H; i += 256) {
    const int ty = i / PS_W, tx = i - ty * PS_W;
    const int gx = reflect101(sx0 + tx, sw), gy = reflect101(sy0 + ty, sh);
    tile[ty][tx] = S[(size_t)gy * src_row_stride + gx];
  }
  __syncthreads();
  if (COPY) {
    unsigned char* C = copy_dst + (size_t)s * copy_img_stride;
    for (int i = threadIdx.x; i < 2 * PT_H * (2 * PT_W / 4); i += 256) {   // 16 rows x 32 dwords
      const int ry = i / (2 * PT_W / 4), q = i - ry * (2 * PT_W / 4);
      const int gy = 2 * oy + ry, gx = 2 * ox + 4 * q;
      if (gy < sh && gx < sw) {
        const unsigned char* r = &tile[ry + 2][4 * q + 2];
        unsigned char* o = C + (size_t)gy * sw + gx;
        if (gx + 3 < sw && ((sw & 3) == 0)) {
          *reinterpret_cast<unsigned*>(o) = (unsigned)r[0] | ((unsigned)r[1] << 8) | ((unsigned)r[2] << 16) | ((unsigned)r[3] << 24);
        } else {
          for (int k = 0; k < 4 && gx + k < sw; k++) o[k] = r[k];
        }
      }
    }
  }
  for (int i = threadIdx.x; i < PS_H * PT_W; i += 256) {
    const int ty = i / PT_W, x = i - ty * PT_W;
    const unsigned char* r = &tile[ty][2 * x];
    hrow[ty][x] = r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PT_H * PT_W; i += 256) {
    const int y = i / PT_W, x = i - y * PT_W;
    const int gx = ox + x, gy = oy + y;
    if (gx < dw && gy < dh) {
      const int v = hrow[2 * y + 2][x] * 6 + (hrow[2 * y + 1][x] + hrow[2 * y + 3][x]) * 4 +
                    hrow[2 * y][x] + hrow[2 * y + 4][x];
      D[(size_t)gy * dw + gx] = (unsigned char)((v + 128) >> 8);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// pyr2_kernel: TWO cv::pyrDown levels in one streaming pass, no barrier, no LDS staging of pixels.
//   wave  = one horizontal strip (T2 rows of level l+2 = 2 T2 rows of level l+1 = 4 T2 source rows + halo) of one stream
//   lane  = 16 source columns = 8 level-(l+1) columns = 4 level-(l+2) columns: ONE aligned 16-byte load per lane and
//           source row (a wave reads 1 KB row segments), one 16-byte store of the level-0 copy, one 8-byte store per
//           level-(l+1) row, one 4-byte store per level-(l+2) row -- every byte is requested once per strip;
//   rows  = the wave walks down its strip with a five-row register window of horizontal sums; the columns a lane
//           needs from its neighbours (2 on the left, 1 on the right) come by DPP wave shifts, BORDER_REFLECT_101 at
//           the image border is a byte permutation of the lane's own dwords (columns) and index arithmetic (rows);
//   sums  = horizontal 1-4-6-4-1 by two v_dot4_u32_u8 per output, two outputs per register; the vertical pass in packed
//           16-bit arithmetic (v_pk_add_u16 / v_pk_mad_u16: the full 5x5 sum + 128 is at most 65 408 < 2^16; the
//           rounding constant rides in as +16 on the odd rows, whose weight is always 4); the second level's horizontal
//           sums are taken from the registers that hold the first level's row and parked in a per-lane LDS ring (rows of
//           level l+1 reflect at the image border, so they are addressed by row index), its vertical pass reads five
//           ring rows.  Integer arithmetic, any order of the 25 taps gives the same sum => bit-exact.
// Requirements (else the tile kernel above is used): source width % 16 == 0, 16-byte aligned rows, height >= 16.
// Waves along x only for more than 62 lanes (992 columns): a wave then owns lanes 2..61, the two lanes on either
// side recompute their neighbours' values.
// ---------------------------------------------------------------------------------------------
typedef unsigned short pk16_t __attribute__((ext_vector_type(2)));
constexpr unsigned PW_A = 0x04060401u;   // taps (1,4,6,4) on bytes 0..3
constexpr unsigned PW_B = 0x00000001u;   // tap 1 on byte 0
constexpr unsigned PW_C = 0x04010000u;   // taps (1,4) on bytes 2,3
constexpr unsigned PW_D = 0x00010406u;   // taps (6,4,1) on bytes 0..2
constexpr unsigned PSEL_LEFT = 0x01020c0cu;    // columns -2, -1 of BORDER_REFLECT_101 = columns 2, 1
constexpr unsigned PSEL_RIGHT = 0x0c0c0c02u;   // column w = column w - 2
constexpr unsigned PSEL_ROUND = 0x07050301u;   // (sum + 128) >> 8 of four 16-bit sums = their high bytes

struct PyrH {   // horizontal sums of the 8 outputs of one lane and source row, two per register
  pk16_t a, b, c, d;
};

__device__ __forceinline__ unsigned pyr_dot4(unsigned x, unsigned w, unsigned acc) {
  return __builtin_amdgcn_udot4(x, w, acc, false);
}
__device__ __forceinline__ unsigned pyr_shr1(unsigned v, unsigned fill) {   // lane i <- lane i-1, lane 0 <- fill
  return (unsigned)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned pyr_shl1(unsigned v, unsigned fill) {   // lane i <- lane i+1, lane 63 <- fill
  return (unsigned)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ pk16_t pyr_pk(unsigned lo, unsigned hi) { return __builtin_bit_cast(pk16_t, lo | (hi << 16)); }

// eight outputs from the 16 bytes of a lane (+ 2 bytes of the left, 1 byte of the right neighbour)
__device__ __forceinline__ PyrH pyr_hpass16(const uint4& d, unsigned sel_right, unsigned bias) {
  const unsigned pd = pyr_shr1(d.w, __builtin_amdgcn_perm(d.x, d.x, PSEL_LEFT));
  const unsigned nd = __builtin_amdgcn_perm(pyr_shl1(d.x, 0u), d.w, sel_right);
  PyrH h;
  h.a = pyr_pk(pyr_dot4(pd, PW_C, pyr_dot4(d.x, PW_D, bias)), pyr_dot4(d.x, PW_A, pyr_dot4(d.y, PW_B, bias)));
  h.b = pyr_pk(pyr_dot4(d.x, PW_C, pyr_dot4(d.y, PW_D, bias)), pyr_dot4(d.y, PW_A, pyr_dot4(d.z, PW_B, bias)));
  h.c = pyr_pk(pyr_dot4(d.y, PW_C, pyr_dot4(d.z, PW_D, bias)), pyr_dot4(d.z, PW_A, pyr_dot4(d.w, PW_B, bias)));
  h.d = pyr_pk(pyr_dot4(d.z, PW_C, pyr_dot4(d.w, PW_D, bias)), pyr_dot4(d.w, PW_A, pyr_dot4(nd, PW_B, bias)));
  return h;
}
// four outputs of the second level from the 8 bytes (lo, hi) of a lane's first-level row
__device__ __forceinline__ uint2 pyr_hpass8(unsigned lo, unsigned hi, unsigned sel_right, unsigned bias) {
  const unsigned ph = pyr_shr1(hi, __builtin_amdgcn_perm(lo, lo, PSEL_LEFT));
  const unsigned nl = __builtin_amdgcn_perm(pyr_shl1(lo, 0u), hi, sel_right);
  const unsigned g0 = pyr_dot4(ph, PW_C, pyr_dot4(lo, PW_D, bias)), g1 = pyr_dot4(lo, PW_A, pyr_dot4(hi, PW_B, bias));
  const unsigned g2 = pyr_dot4(lo, PW_C, pyr_dot4(hi, PW_D, bias)), g3 = pyr_dot4(hi, PW_A, pyr_dot4(nl, PW_B, bias));
  return make_uint2(g0 | (g1 << 16), g2 | (g3 << 16));
}
__device__ __forceinline__ pk16_t pyr_vert(pk16_t r0, pk16_t r1, pk16_t r2, pk16_t r3, pk16_t r4) {
  const pk16_t four = {4, 4}, six = {6, 6};
  return r2 * six + ((r1 + r3) * four + (r0 + r4));
}
// the 8 first-level pixels of a lane from five rows of horizontal sums
__device__ __forceinline__ void pyr_vert8(const PyrH& r0, const PyrH& r1, const PyrH& r2, const PyrH& r3, const PyrH& r4,
                                          unsigned& lo, unsigned& hi) {
  const unsigned ua = __builtin_bit_cast(unsigned, pyr_vert(r0.a, r1.a, r2.a, r3.a, r4.a));
  const unsigned ub = __builtin_bit_cast(unsigned, pyr_vert(r0.b, r1.b, r2.b, r3.b, r4.b));
  const unsigned uc = __builtin_bit_cast(unsigned, pyr_vert(r0.c, r1.c, r2.c, r3.c, r4.c));
  const unsigned ud = __builtin_bit_cast(unsigned, pyr_vert(r0.d, r1.d, r2.d, r3.d, r4.d));
  lo = __builtin_amdgcn_perm(ub, ua, PSEL_ROUND);
  hi = __builtin_amdgcn_perm(ud, uc, PSEL_ROUND);
}
__device__ __forceinline__ int pyr_reflect(int k, int n) {   // BORDER_REFLECT_101 for -n < k < 2n - 1
  return k < 0 ? -k : (k >= n ? 2 * n - 2 - k : k);
}

typedef unsigned pyr_v4u __attribute__((ext_vector_type(4)));
typedef unsigned pyr_v2u __attribute__((ext_vector_type(2)));

// T2: second-level rows per strip (compile time: the strip is fully unrolled -- ALL its source rows are requested
// before the first one is used, so a wave waits for memory once; the five-row window rotates by register renaming).
struct Pyr2Args {
  const unsigned char* src;
  unsigned srow;
  size_t simg;
  int w0, h0;
  unsigned char *dst1, *dst2;
  size_t dimg;
  unsigned char* copy_dst;
  size_t cimg;
};

// One strip.  INTERIOR: no source / first-level row of the strip reflects at the image border and every row it owns
// exists -- row offsets are then plain multiples and every bounds test folds away; the border strips take the general
// instantiation (a wave-uniform choice).
template <int T2, bool COPY, bool TWO, bool INTERIOR>
__device__ __forceinline__ void pyr2_strip(const Pyr2Args& A, uint2* ring, int s, int strip, int gl, int nl, bool owner) {
  constexpr int NJ = TWO ? 2 * T2 + 3 : 2 * T2;   // first-level rows a strip computes (TWO: 2 + 1 rows of halo)
  constexpr int NR = 2 * NJ + 3;                  // source rows it reads
  constexpr int J0 = TWO ? 2 : 0;                 // first first-level row it owns (stores)
  constexpr int I0 = TWO ? 6 : 2;                 // first source row whose level-0 copy it stores
  const int lane = threadIdx.x;
  const int w0 = A.w0, h0 = A.h0;
  const int w1 = w0 >> 1, h1 = (h0 + 1) >> 1, w2 = w0 >> 2, h2 = (h1 + 1) >> 1;
  const int Y1 = strip * 2 * T2;                  // first owned first-level row (even)
  const int y1b = TWO ? Y1 - 2 : Y1;              // first-level row of j = 0 (may be virtual: -2)
  const int kb = 2 * y1b - 2;                     // source row of i = 0 (may be virtual)
  // right image border: column w is column w - 2; every other lane takes its neighbour's first dword (one v_perm_b32
  // with a per-lane selector instead of a select)
  const unsigned sel_right = gl == nl - 1 ? PSEL_RIGHT : 0x07060504u;
  // raw buffers: per-lane byte offset in a VGPR, the row offset in an SGPR (one scalar op per row, no 64-bit math);
  // lanes that own nothing store at offset 2^32 - 1, which the buffer bounds check drops
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(A.src) + (size_t)s * A.simg, 0,
                                                                     (int)(A.srow * (unsigned)(h0 - 1) + (unsigned)w0), 0x00027000);
  const unsigned voff = (unsigned)min(gl, nl - 1) * 16u;
  pyr_v4u R[NR];
#pragma unroll
  for (int i = 0; i < NR; i++) {
    const unsigned r = INTERIOR ? (unsigned)(kb + i) : (unsigned)pyr_reflect(kb + i, h0);
    R[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, r * A.srow, 0);
  }
  __builtin_amdgcn_sched_barrier(0);   // (keep the requests ahead of the arithmetic)
  // rows are consumed in request order (vmcnt counts in order): row i is worked on while rows i+1.. are still in flight
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(COPY ? A.copy_dst + (size_t)s * A.cimg : nullptr, 0,
                                                                     COPY ? h0 * w0 : 0, 0x00027000);
  // The 16-byte store takes its row offset in the VECTOR offset, the scalar offset is the constant 0 -- NOT the row offset
  // in an SGPR as the loads and the narrower stores do.  gfx950 reads the data registers of a store of more than 64 bits
  // after the instruction has issued: a VALU write of one of them in the very next issue slot changes what lanes 12 - 15
  // of every row of 16 store (measured, round 6: tools/fuzz_batched.py 120 2 79 -- the first dword of the copy's 16 bytes
  // replaced by the next row's v_perm_b32 result when a wave runs alone on its SIMD, one run in three; tools/r6/
  // gpu_pyr_probe.sh reads the copy back).  The compiler pads that hazard with an s_nop only for a store WITHOUT an SGPR
  // offset (its rule: "a buffer store that uses an SGPR offset needs no wait state") -- so this store has none.
  // tests/test_isa_hazards.py scans the compiled code of every kernel for the pattern.  (Lanes that own nothing: offset
  // 2^31 + row offset, which the bounds check drops -- an image is smaller than 2^31 bytes.)
  const unsigned coff = owner ? (unsigned)gl * 16u : 0x80000000u;
  auto keep = [&](int i) {   // level-0 copy of source row kb + i
    if (COPY && i >= I0 && i < I0 + 4 * T2 && (INTERIOR || kb + i < h0))
      __builtin_amdgcn_raw_buffer_store_b128(R[i], rc, coff + (unsigned)(kb + i) * (unsigned)w0, 0, 0);
  };
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(A.dst1 + (size_t)s * A.dimg, 0, h1 * w1, 0x00027000);
  const unsigned o1 = owner ? (unsigned)gl * 8u : 0xffffffffu;
  PyrH H[5];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const uint4 d = make_uint4(R[i].x, R[i].y, R[i].z, R[i].w);
    H[i] = pyr_hpass16(d, sel_right, (i & 1) ? 16u : 0u);
    keep(i);
  }
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    // first-level row y1b + j from source rows 2 j .. 2 j + 4 (window slots rotate: slot of source row i is i % 5)
#pragma unroll
    for (int i = 2 * j + 3; i <= 2 * j + 4; i++) {
      const uint4 d = make_uint4(R[i].x, R[i].y, R[i].z, R[i].w);
      H[i % 5] = pyr_hpass16(d, sel_right, (i & 1) ? 16u : 0u);
      keep(i);
    }
    unsigned lo, hi;
    pyr_vert8(H[(2 * j) % 5], H[(2 * j + 1) % 5], H[(2 * j + 2) % 5], H[(2 * j + 3) % 5], H[(2 * j + 4) % 5], lo, hi);
    if (j >= J0 && j < J0 + 2 * T2 && (INTERIOR || y1b + j < h1)) {
      const pyr_v2u v = {lo, hi};
      __builtin_amdgcn_raw_buffer_store_b64(v, r1, o1, (unsigned)(y1b + j) * (unsigned)w1, 0);
    }
    if (TWO) ring[j * 64 + lane] = pyr_hpass8(lo, hi, sel_right, (j & 1) ? 16u : 0u);
  }
  if (TWO) {
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(A.dst2 + (size_t)s * A.dimg, 0, h2 * w2, 0x00027000);
    const unsigned o2 = owner ? (unsigned)gl * 4u : 0xffffffffu;
#pragma unroll
    for (int t = 0; t < T2; t++) {
      const int y2 = strip * T2 + t;
      if (INTERIOR || y2 < h2) {
        uint2 g[5];
#pragma unroll
        for (int d = 0; d < 5; d++) {
          const int slot = INTERIOR ? 2 * t + d : pyr_reflect(2 * y2 - 2 + d, h1) - y1b;
          g[d] = ring[slot * 64 + lane];
        }
        const unsigned ua = __builtin_bit_cast(unsigned, pyr_vert(__builtin_bit_cast(pk16_t, g[0].x), __builtin_bit_cast(pk16_t, g[1].x),
                                                                  __builtin_bit_cast(pk16_t, g[2].x), __builtin_bit_cast(pk16_t, g[3].x),
                                                                  __builtin_bit_cast(pk16_t, g[4].x)));
        const unsigned ub = __builtin_bit_cast(unsigned, pyr_vert(__builtin_bit_cast(pk16_t, g[0].y), __builtin_bit_cast(pk16_t, g[1].y),
                                                                  __builtin_bit_cast(pk16_t, g[2].y), __builtin_bit_cast(pk16_t, g[3].y),
                                                                  __builtin_bit_cast(pk16_t, g[4].y)));
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_amdgcn_perm(ub, ua, PSEL_ROUND), r2, o2, (unsigned)y2 * (unsigned)w2, 0);
      }
    }
  }
}

// T2: second-level rows per strip (compile time: the strip is fully unrolled -- ALL its source rows are requested
// before the first one is used, so a wave waits for memory once; the five-row window rotates by register renaming).
template <int T2, bool COPY, bool TWO>
__global__ __launch_bounds__(64) void pyr2_kernel(Pyr2Args A, int B, int NS, int nwx, int xcd_mode) {
  constexpr int NJ = TWO ? 2 * T2 + 3 : 2 * T2;
  __shared__ uint2 ring[TWO ? NJ * 64 : 1];       // second-level horizontal sums of the strip's first-level rows
  int s, strip, wx;
  {
    const int per_stream = NS * nwx;
    int b = blockIdx.x, rem;
    if (xcd_mode) {   // workgroup b runs on XCD b & 7 (a speed assumption only): a stream's strips share one L2
      const int j = b >> 3;
      s = (b & 7) + 8 * (j / per_stream);
      rem = j % per_stream;
      if (s >= B) return;
    } else {
      s = b / per_stream;
      rem = b - s * per_stream;
    }
    strip = rem / nwx;
    wx = rem - strip * nwx;
  }
  const int lane = threadIdx.x;
  const int nl = A.w0 >> 4;
  const int gl = 60 * wx + lane;
  const bool owner = gl < nl && lane < 62 && (wx == 0 || lane >= 2);
  const int h1 = (A.h0 + 1) >> 1, h2 = (h1 + 1) >> 1;
  const int Y1 = strip * 2 * T2, kb = 2 * (TWO ? Y1 - 2 : Y1) - 2;
  const bool interior = kb >= 0 && kb + 2 * NJ + 3 <= A.h0 && Y1 + 2 * T2 + (TWO ? 1 : 0) <= h1 &&
                        (!TWO || (strip + 1) * T2 <= h2);
  if (interior) pyr2_strip<T2, COPY, TWO, true>(A, ring, s, strip, gl, nl, owner);
  else pyr2_strip<T2, COPY, TWO, false>(A, ring, s, strip, gl, nl, owner);
}

// can level `l` (and l+1) be produced by pyr2_kernel from level l-1?
static bool pyr2_ok(const unsigned char* src, size_t srow, size_t simg, int w0, int h0, const unsigned char* d1,
                    const unsigned char* d2, size_t dimg) {
  return w0 % 16 == 0 && h0 >= 16 && srow % 16 == 0 && simg % 16 == 0 && reinterpret_cast<uintptr_t>(src) % 16 == 0 &&
         reinterpret_cast<uintptr_t>(d1) % 8 == 0 && reinterpret_cast<uintptr_t>(d2) % 4 == 0 && dimg % 8 == 0;
}

void launch_pyramid(const KParams& P, const unsigned char* img, size_t row_stride,
                    size_t img_stride, unsigned char* pyr, hipStream_t st, unsigned char* level0_copy) {
  if (level0_copy && P.nlevels < 2)   // (klt_max_level 0: no level-1 launch to piggyback on)
    for (int s = 0; s < P.B; s++)
      (void)hipMemcpy2DAsync(level0_copy + (size_t)s * P.W * P.H, (size_t)P.W, img + (size_t)s * img_stride, row_stride,
                             (size_t)P.W, (size_t)P.H, hipMemcpyDeviceToDevice, st);
  // the streaming two-level kernel where the geometry allows, the tile kernel per level otherwise
  const int impl = 1;
  for (int l = 1; l < P.nlevels; l++) {
    const unsigned char* src = l == 1 ? img : pyr + P.loff[l - 1];
    const size_t srow = l == 1 ? row_stride : (size_t)P.lw[l - 1];
    const size_t simg = l == 1 ? img_stride : (size_t)P.pyr_stride;
    const bool two = l + 1 < P.nlevels;
    const bool copy = l == 1 && level0_copy;
    if (impl == 1 && pyr2_ok(src, srow, simg, P.lw[l - 1], P.lh[l - 1], pyr + P.loff[l], two ? pyr + P.loff[l + 1] : pyr,
                             (size_t)P.pyr_stride) &&
        (!copy || (reinterpret_cast<uintptr_t>(level0_copy) % 16 == 0 && ((size_t)P.W * P.H) % 16 == 0))) {
      const int w0 = P.lw[l - 1], h0 = P.lh[l - 1], h1 = (h0 + 1) / 2, h2 = (h1 + 1) / 2;
      const int nl = w0 / 16, nwx = nl <= 62 ? 1 : 1 + (nl - 62 + 59) / 60;
      // strip height: the halo costs (4 T2 + 9) / (4 T2) source rows (re-read from the XCD's L2, not from HBM: a stream's
      // strips run on one XCD), against waves in flight: a strip is one wave and its source rows live in registers.
      // Round 6 (tools/r6/gpu_pyr2_t2.sh, 64 x 752x480, same call, copy probe 4.4 - 4.7 TB/s): T2 = 4 (1 920 waves) 16.9 /
      // 16.6 us alone, T2 = 2 (3 840 waves, ~110 VGPRs) 16.0 / 15.8, T2 = 1 (7 680 waves) 17.6 / 17.4 -- 53 MB of real traffic
      // at 3.3 TB/s = 0.72 of what the plain copy kernel reaches on that box.  T2 = 2 everywhere; KVFE_PYR2_T2 is the A/B
      // switch.  (T2 = 8 was removed in round 4: its first test at more than 100 streams failed the level-0 copy.)
      int T2 = 2;
      static const int t2_env = [] { const char* e = std::getenv("KVFE_PYR2_T2"); return e ? std::atoi(e) : 0; }();
      if (t2_env == 1 || t2_env == 2 || t2_env == 4) T2 = t2_env;
      const int NS = two ? (h2 + T2 - 1) / T2 : (h1 + 2 * T2 - 1) / (2 * T2);
      const int xcd = P.B >= 8 ? 1 : 0;
      const int nblk = xcd ? 8 * ((P.B + 7) / 8) * NS * nwx : P.B * NS * nwx;
      const Pyr2Args pa{src, (unsigned)srow, simg, w0, h0, pyr + P.loff[l], two ? pyr + P.loff[l + 1] : (unsigned char*)nullptr,
                        (size_t)P.pyr_stride, level0_copy, (size_t)P.W * P.H};
#define KVFE_PYR2(T2_, COPY_, TWO_)                                                                                    \
  hipLaunchKernelGGL((pyr2_kernel<T2_, COPY_, TWO_>), dim3(nblk), dim3(64), 0, st, pa, P.B, NS, nwx, xcd)
#define KVFE_PYR2_T(T2_)                             \
  do {                                               \
    if (copy && two) KVFE_PYR2(T2_, true, true);     \
    else if (copy) KVFE_PYR2(T2_, true, false);      \
    else if (two) KVFE_PYR2(T2_, false, true);       \
    else KVFE_PYR2(T2_, false, false);               \
  } while (0)
      if (T2 == 4) KVFE_PYR2_T(4);
      else if (T2 == 2) KVFE_PYR2_T(2);
      else KVFE_PYR2_T(1);
#undef KVFE_PYR2_T
#undef KVFE_PYR2
      if (two) l++;
      continue;
    }
    dim3 grid((P.lw[l] + PT_W - 1) / PT_W, (P.lh[l] + PT_H - 1) / PT_H, P.B);
    if (l == 1 && level0_copy)
      hipLaunchKernelGGL(pyrdown_kernel<true>, grid, dim3(256), 0, st, src, srow, simg, P.lw[l - 1], P.lh[l - 1],
                         pyr + P.loff[l], (size_t)P.pyr_stride, P.lw[l], P.lh[l], level0_copy,
                         (size_t)P.W * P.H);
    else
      hipLaunchKernelGGL(pyrdown_kernel<false>, grid, dim3(256), 0, st, src, srow, simg, P.lw[l - 1], P.lh[l - 1],
                         pyr + P.loff[l], (size_t)P.pyr_stride, P.lw[l], P.lh[l], (unsigned char*)nullptr, (size_t)0);
  }
}

// ---------------------------------------------------------------------------------------------
// cv::equalizeHist (reference: UtilsOpenCV::ReadAndConvertToGrayScale, src/utils/UtilsOpenCV.cpp:398-401)
// pass 1: 256-bin histogram per image (LDS histogram per block, one global atomic per occupied bin)
// pass 2: LUT = saturate_cast<uchar>(cumulative count above the first occupied bin * scale) rebuilt
//         by every block from the 1 KB histogram, applied to 16 pixels per lane
// ---------------------------------------------------------------------------------------------
constexpr int EQ_T = 256;
constexpr int EQ_PX = EQ_T * 16 * 4;  // pixels per block

__global__ __launch_bounds__(EQ_T) void equalize_count_kernel(const unsigned char* __restrict__ src,
                                                              size_t row_stride, size_t img_stride,
                                                              int W, int H, int* __restrict__ hist) {
  __shared__ int lh[256];
  const int s = blockIdx.y, tid = threadIdx.x;
  lh[tid] = 0;
  __syncthreads();
  const unsigned char* S = src + (size_t)s * img_stride;
  const int N = W * H;
  const int p0 = blockIdx.x * EQ_PX;
  for (int p = p0 + tid * 4; p < min(N, p0 + EQ_PX); p += EQ_T * 4) {
    // 4 consecutive pixels (a row may have any stride; W % 4 == 0 is not assumed)
    for (int k = 0; k < 4 && p + k < N; k++) {
      const int y = (p + k) / W, x = (p + k) - y * W;
      atomicAdd(&lh[S[(size_t)y * row_stride + x]], 1);
    }
  }
  __syncthreads();
  const int c = lh[tid];
  if (c) atomicAdd(&hist[s * 256 + tid], c);
}

__global__ __launch_bounds__(EQ_T) void equalize_apply_kernel(const unsigned char* __restrict__ src,
                                                              size_t row_stride, size_t img_stride,
                                                              int W, int H,
                                                              const int* __restrict__ hist,
                                                              unsigned char* __restrict__ dst) {
  __shared__ int lh[256];
  __shared__ unsigned char lut[256];
  __shared__ int sh_first;
  const int s = blockIdx.y, tid = threadIdx.x;
  lh[tid] = hist[s * 256 + tid];
  if (tid == 0) sh_first = 256;
  __syncthreads();
  if (lh[tid]) atomicMin(&sh_first, tid);
  __syncthreads();
  const int first = sh_first, total = W * H;
  if (lh[first] == total) {
    lut[tid] = (unsigned char)first;  // dst.setTo(i)
  } else {
    const float scale = (256 - 1.f) / (float)(total - lh[first]);
    int sum = 0;
    for (int i = first + 1; i <= tid; i++) sum += lh[i];
    const int v = __float2int_rn((float)sum * scale);
    lut[tid] = tid <= first ? 0 : (unsigned char)min(max(v, 0), 255);
  }
  __syncthreads();
  const unsigned char* S = src + (size_t)s * img_stride;
  unsigned char* D = dst + (size_t)s * total;
  const int p0 = blockIdx.x * EQ_PX;
  for (int p = p0 + tid * 4; p < min(total, p0 + EQ_PX); p += EQ_T * 4)
    for (int k = 0; k < 4 && p + k < total; k++) {
      const int y = (p + k) / W, x = (p + k) - y * W;
      D[p + k] = lut[S[(size_t)y * row_stride + x]];
    }
}

void launch_equalize_hist(int W, int H, int B, const unsigned char* src, size_t src_row_stride,
                          size_t src_img_stride, unsigned char* dst, int* hist, hipStream_t st) {
  hipMemsetAsync(hist, 0, sizeof(int) * 256 * (size_t)B, st);
  const dim3 grid((W * H + EQ_PX - 1) / EQ_PX, B);
  hipLaunchKernelGGL(equalize_count_kernel, grid, dim3(EQ_T), 0, st, src, src_row_stride, src_img_stride,
                     W, H, hist);
  hipLaunchKernelGGL(equalize_apply_kernel, grid, dim3(EQ_T), 0, st, src, src_row_stride, src_img_stride,
                     W, H, hist, dst);
}

}  // namespace kvfe
