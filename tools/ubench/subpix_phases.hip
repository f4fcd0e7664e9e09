// Micro-benchmark: where one cv::cornerSubPix iteration of the wave kernel spends its cycles
// (patch interpolation | term products | the five float64 chains | broadcast + barrier | solve | loop head).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I kimera_vio_amd/csrc tools/ubench/subpix_phases.hip -o tools/ubench/subpix_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <utility>
#include <vector>
#define KVFE_SUBPIX_PROF 1
__device__ unsigned long long* kvfe_sp_out = nullptr;
namespace kvfe {
#include "kvfe_subpix.inl"
template <int WIN, int NW>
__global__ __launch_bounds__(64 * NW) void k(const float* mask, const unsigned char* img, size_t step, int W, int H, float2* pts,
                                        int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  if (blockIdx.x == 0) kvfe_sp_out = out;   // (one writer; the ubench launches one block per measurement)
  __syncthreads();
  const float2 c = corner_subpix_wave<WIN, NW>(img, step, W, H, pts[blockIdx.x], 10, iters, 0.0, mask, lds_raw, threadIdx.x);
  if (threadIdx.x == 0) pts[blockIdx.x] = c;
}
}  // namespace kvfe
int main() {
  const int W = 752, H = 480, win = 10, ww = 21;
  std::vector<unsigned char> img(W * H);
  unsigned s = 12345;
  for (auto& v : img) { s = s * 1664525u + 1013904223u; v = (unsigned char)(s >> 24); }
  // a checker corner so that the iteration does something sensible
  for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) if (((x / 37) + (y / 29)) & 1) img[y * W + x] = (unsigned char)(200 + (img[y * W + x] & 15)); else img[y * W + x] &= 31;
  std::vector<float> mask(ww * ww);
  for (int i = 0; i < ww; i++) for (int j = 0; j < ww; j++) {
    const float y = (float)(i - win) / win, x = (float)(j - win) / win;
    mask[i * ww + j] = (float)std::exp(-y * y) * (float)std::exp(-x * x);
  }
  unsigned char* dimg; float* dmask; float2* dpts; unsigned long long* dout;
  hipMalloc(&dimg, W * H); hipMalloc(&dmask, mask.size() * 4); hipMalloc(&dpts, 1024 * 8); hipMalloc(&dout, 64);
  hipMemcpy(dimg, img.data(), W * H, hipMemcpyHostToDevice);
  hipMemcpy(dmask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice);
  const size_t lds = kvfe::subpix_geom(win).bytes;
  for (int nw : {1, 2, 4})
  for (int nblk : {1, 256, 1024}) {
    for (int rep = 0; rep < 2; rep++) {
      std::vector<float2> pts(1024);
      for (int i = 0; i < 1024; i++) pts[i] = make_float2(74.3f + 37 * (i % 16), 58.2f + 29 * ((i / 16) % 12));
      hipMemcpy(dpts, pts.data(), 1024 * 8, hipMemcpyHostToDevice);
      hipMemset(dout, 0, 64);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (nw == 1)
        hipLaunchKernelGGL((kvfe::k<10, 1>), dim3(nblk), dim3(64), lds, 0, dmask, dimg, (size_t)W, W, H, dpts, 40, dout);
      else if (nw == 2)
        hipLaunchKernelGGL((kvfe::k<10, 2>), dim3(nblk), dim3(128), lds, 0, dmask, dimg, (size_t)W, W, H, dpts, 40, dout);
      else
        hipLaunchKernelGGL((kvfe::k<10, 4>), dim3(nblk), dim3(256), lds, 0, dmask, dimg, (size_t)W, W, H, dpts, 40, dout);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[8]; hipMemcpy(h, dout, 64, hipMemcpyDeviceToHost);
      const double it = (double)h[6];
      if (rep == 1)
        printf("waves/corner %d, blocks %4d: kernel %.1f us, %g iterations; cycles per iteration: patch %.0f | terms %.0f | chains %.0f | bcast+barrier %.0f | solve %.0f | loop %.0f | total %.0f\n",
               nw, nblk, ms * 1e3, it, h[0] / it, h[1] / it, h[2] / it, h[3] / it, h[4] / it, h[5] / it,
               (h[0] + h[1] + h[2] + h[3] + h[4] + h[5]) / it);
    }
  }
  return 0;
}
