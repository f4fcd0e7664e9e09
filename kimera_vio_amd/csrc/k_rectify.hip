// K1  undistort-rectify  == cv::remap(INTER_LINEAR, BORDER_REPLICATE) on 8-bit images
//     reference: UndistorterRectifier::undistortRectifyImage, src/frontend/UndistorterRectifier.cpp:115-128
// K4a pyramid level     == cv::pyrDown inside cv::buildOpticalFlowPyramid
//     reference: cv::calcOpticalFlowPyrLK call at src/frontend/Tracker.cpp:137-146
//
// Both are HBM-bound gather/stencil kernels: 16 B/lane coalesced map reads, 4 output pixels per
// lane (one 4-byte store), source gathers served by L1/L2 (a rectification map is locally smooth,
// so a wave touches ~2-3 source rows).  Integer arithmetic only => bit-exact vs the CPU path.
#include "kvfe_dev.hpp"

namespace kvfe {

__device__ __forceinline__ int clipi(int x, int b) { return x >= 0 ? (x < b ? x : b - 1) : 0; }

// one output pixel of cv::remap: 1/32-px fixed-point coordinates, 15-bit weights
__device__ __forceinline__ unsigned remap_px(const unsigned char* __restrict__ src, int W, int H,
                                             size_t stride, float mx, float my) {
  int sxf = __float2int_rn(mx * 32.0f);
  int syf = __float2int_rn(my * 32.0f);
  const int ax = sxf & 31, ay = syf & 31;
  int sx = sxf >> 5, sy = syf >> 5;
  sx = max(-32768, min(32767, sx));  // saturate_cast<short>
  sy = max(-32768, min(32767, sy));
  int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32,
      w11 = ax * ay * 32;
  if ((ax | ay) == 0) {  // table entry (0,0) after saturation and fix-up: {32767,0,0,1}
    w00 = 32767;
    w11 = 1;
  }
  int v0, v1, v2, v3;
  if ((unsigned)sx < (unsigned)max(W - 1, 0) && (unsigned)sy < (unsigned)max(H - 1, 0)) {
    const unsigned char* S = src + (size_t)sy * stride + sx;
    v0 = S[0];
    v1 = S[1];
    v2 = S[stride];
    v3 = S[stride + 1];
  } else {
    const int sx0 = clipi(sx, W), sx1 = clipi(sx + 1, W);
    const int sy0 = clipi(sy, H), sy1 = clipi(sy + 1, H);
    v0 = src[(size_t)sy0 * stride + sx0];
    v1 = src[(size_t)sy0 * stride + sx1];
    v2 = src[(size_t)sy1 * stride + sx0];
    v3 = src[(size_t)sy1 * stride + sx1];
  }
  int r = (v0 * w00 + v1 * w01 + v2 * w10 + v3 * w11 + (1 << 14)) >> 15;
  return (unsigned)max(0, min(255, r));
}

template <bool VEC4>
__global__ __launch_bounds__(256) void rectify_kernel(
    const unsigned char* __restrict__ src0, const unsigned char* __restrict__ src1,
    size_t src_row_stride, size_t src_img_stride, unsigned char* __restrict__ dst0,
    unsigned char* __restrict__ dst1, const float2* __restrict__ map0,
    const float2* __restrict__ map1, int W, int H, const int* __restrict__ flags, int act_flag) {
  const int s = blockIdx.z, cam = blockIdx.y;
  if (flags && !(flags[s] & act_flag)) return;
  const unsigned char* src = (cam == 0 ? src0 : src1) + (size_t)s * src_img_stride;
  unsigned char* dst = (cam == 0 ? dst0 : dst1) + (size_t)s * W * H;
  const float2* map = cam == 0 ? map0 : map1;
  const int N = W * H;
  if (VEC4) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= N) return;
    const float4 m01 = *reinterpret_cast<const float4*>(map + i);
    const float4 m23 = *reinterpret_cast<const float4*>(map + i + 2);
    unsigned p0 = remap_px(src, W, H, src_row_stride, m01.x, m01.y);
    unsigned p1 = remap_px(src, W, H, src_row_stride, m01.z, m01.w);
    unsigned p2 = remap_px(src, W, H, src_row_stride, m23.x, m23.y);
    unsigned p3 = remap_px(src, W, H, src_row_stride, m23.z, m23.w);
    *reinterpret_cast<unsigned*>(dst + i) = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
  } else {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float2 m = map[i];
    dst[i] = (unsigned char)remap_px(src, W, H, src_row_stride, m.x, m.y);
  }
}

void launch_rectify(const KParams& P, const Tables& T, const unsigned char* const src[2],
                    size_t src_row_stride, size_t src_img_stride, unsigned char* const dst[2],
                    const int* flags, int act_flag, hipStream_t st) {
  const int N = P.W * P.H;
  if (N % 4 == 0) {
    dim3 grid((N / 4 + 255) / 256, 2, P.B);
    hipLaunchKernelGGL(rectify_kernel<true>, grid, dim3(256), 0, st, src[0], src[1],
                       src_row_stride, src_img_stride, dst[0], dst[1], T.map[0], T.map[1], P.W,
                       P.H, flags, act_flag);
  } else {
    dim3 grid((N + 255) / 256, 2, P.B);
    hipLaunchKernelGGL(rectify_kernel<false>, grid, dim3(256), 0, st, src[0], src[1],
                       src_row_stride, src_img_stride, dst[0], dst[1], T.map[0], T.map[1], P.W,
                       P.H, flags, act_flag);
  }
}

// ---------------------------------------------------------------------------------------------
// cv::pyrDown: separable [1 4 6 4 1]/16, BORDER_REFLECT_101, (sum + 128) >> 8, dst = (src+1)/2.
// One block = 64x8 output tile; the (2*64+3)x(2*8+3) source patch is staged in LDS, the
// horizontal pass writes 19 x 64 ints to LDS, the vertical pass produces the outputs.
// ---------------------------------------------------------------------------------------------
constexpr int PT_W = 64, PT_H = 8;
constexpr int PS_W = 2 * PT_W + 3, PS_H = 2 * PT_H + 3;

__global__ __launch_bounds__(256) void pyrdown_kernel(const unsigned char* __restrict__ src,
                                                      size_t src_row_stride,
                                                      size_t src_img_stride, int sw, int sh,
                                                      unsigned char* __restrict__ dst,
                                                      size_t dst_img_stride, int dw, int dh) {
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

void launch_pyramid(const KParams& P, const unsigned char* img, size_t row_stride,
                    size_t img_stride, unsigned char* pyr, hipStream_t st) {
  for (int l = 1; l < P.nlevels; l++) {
    const unsigned char* src = l == 1 ? img : pyr + P.loff[l - 1];
    const size_t srow = l == 1 ? row_stride : (size_t)P.lw[l - 1];
    const size_t simg = l == 1 ? img_stride : (size_t)P.pyr_stride;
    dim3 grid((P.lw[l] + PT_W - 1) / PT_W, (P.lh[l] + PT_H - 1) / PT_H, P.B);
    hipLaunchKernelGGL(pyrdown_kernel, grid, dim3(256), 0, st, src, srow, simg, P.lw[l - 1],
                       P.lh[l - 1], pyr + P.loff[l], (size_t)P.pyr_stride, P.lw[l], P.lh[l]);
  }
}

}  // namespace kvfe
