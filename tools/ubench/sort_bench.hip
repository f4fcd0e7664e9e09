// kimera_vio_amd/csrc/kvfe_blocksort.inl against std::sort, and its single-block latency.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/sort_bench.hip -o tools/ubench/sort_bench && tools/ubench/sort_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../kimera_vio_amd/csrc/kvfe_blocksort.inl"

template <int EPT>
__global__ __launch_bounds__(1024) void sort_kernel(const unsigned long long* in, unsigned long long* out, int reps,
                                                    unsigned long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned long long* a = reinterpret_cast<unsigned long long*>(lds);
  const int t = threadIdx.x;
  unsigned long long v[EPT];
  unsigned long long t0 = 0, acc = 0;
  for (int r = 0; r < reps; r++) {
    for (int m = 0; m < EPT; m++) a[t + m * 1024] = in[t + m * 1024];
    __syncthreads();
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int m = 0; m < EPT; m++) v[m] = a[t + m * 1024];
    blocksort::sort_desc_blocked<EPT>(v, a);
    acc += __builtin_readcyclecounter() - t0;
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < EPT; m++) out[EPT * t + m] = v[m];
  if (t == 0) *cycles = acc / reps;
}

template <int EPT>
static int run(int reps) {
  const int n = EPT * 1024;
  std::vector<unsigned long long> h(n), ref, got(n);
  unsigned long long s = 0x1234567ull + EPT;
  for (int i = 0; i < n; i++) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    h[i] = (i % 7 == 0 && i > n / 2) ? 0ull : ((s >> 8) | 1ull) + (unsigned long long)i;  // some zero padding
  }
  ref = h;
  std::sort(ref.begin(), ref.end(), [](unsigned long long x, unsigned long long y) { return x > y; });
  unsigned long long *din, *dout, *dc;
  hipMalloc(&din, n * 8);
  hipMalloc(&dout, n * 8);
  hipMalloc(&dc, 8);
  hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(sort_kernel<EPT>), hipFuncAttributeMaxDynamicSharedMemorySize, n * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(sort_kernel<EPT>, dim3(1), dim3(1024), n * 8, 0, din, dout, 1, dc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(sort_kernel<EPT>, dim3(1), dim3(1024), n * 8, 0, din, dout, reps, dc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long cyc = 0;
  hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
  hipMemcpy(got.data(), dout, n * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; i++) bad += got[i] != ref[i];
  printf("n=%5d  %s  %8.2f us per sort (events, incl. reload)  %llu cycles in the sort\n", n, bad ? "MISMATCH" : "ok", 1e3 * ms / reps, cyc);
  return bad;
}

int main() {
  int bad = 0;
  bad += run<1>(200);
  bad += run<2>(200);
  bad += run<4>(200);
  bad += run<8>(200);
  return bad ? 1 : 0;
}
