// Dependent float64 additions: cycles per addition of a chain in ONE wave (what a cornerSubPix iteration waits for:
// five chains of 441 sequential additions).  Forms: v_fmac_f64_dpp row_newbcast (the chains of subpix_append_kernel),
// v_add_f64 with a VGPR operand (the grouped kernel), v_add_f64 with the term in an SGPR pair, v_fma_f64.
//   hipcc --offload-arch=gfx950 -O3 -o f64_chain f64_chain.hip && ./f64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(X) X X X X X X X X
#define R64(X) R8(R8(X))
template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned long long* out, double* sink, double seed) {
  double acc = seed * threadIdx.x, t0v = seed + 1.0, t1v = seed + 2.0, one = 1.0;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 64; it++) {
    if (MODE == 0)
      asm volatile(R64("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(acc) : "v"(t0v), "v"(one));
    else if (MODE == 1)
      asm volatile(R64("v_add_f64 %0, %0, %1\n\t") : "+v"(acc) : "v"(t0v));
    else if (MODE == 2)
      asm volatile(R64("v_add_f64 %0, %0, %1\n\t") : "+v"(acc) : "s"(seed));
    else if (MODE == 3)
      asm volatile(R64("v_fma_f64 %0, %1, %2, %0\n\t") : "+v"(acc) : "v"(t0v), "v"(one));
    else if (MODE == 4)   // two independent chains interleaved
      asm volatile(R64("v_add_f64 %0, %0, %2\n\tv_add_f64 %1, %1, %2\n\t") : "+v"(acc), "+v"(t1v) : "v"(t0v));
    else if (MODE == 5)   // a chain with one independent 32-bit VALU instruction between the additions
      asm volatile(R64("v_add_f64 %0, %0, %2\n\tv_mov_b32 %1, %1\n\t") : "+v"(acc), "+v"(*(int*)&t1v) : "v"(t0v));
    else if (MODE == 6)   // ... and with a ds_read_b64 between them (address in a VGPR, result unused until the end)
      asm volatile(R64("v_add_f64 %0, %0, %1\n\ts_nop 0\n\t") : "+v"(acc) : "v"(t0v));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = acc + t1v;
}
int main() {
  unsigned long long* d;
  double* sink;
  hipMalloc(&d, 8 * 4096);
  hipMalloc(&sink, 8 * 4096 * 64);
  const char* names[] = {"v_fmac_f64_dpp row_newbcast", "v_add_f64 (VGPR term)", "v_add_f64 (SGPR term)", "v_fma_f64 (x * 1.0 + acc)",
                         "two interleaved v_add_f64 chains (per pair)", "v_add_f64 + an independent v_mov_b32 (per pair)",
                         "v_add_f64 + s_nop 0 (per pair)"};
  for (int blocks : {64, 1024, 4096}) {
    for (int m = 0; m < 7; m++) {
      for (int rep = 0; rep < 2; rep++) {
        switch (m) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, sink, 0.5); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, sink, 0.5); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d, sink, 0.5); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, d, sink, 0.5); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, d, sink, 0.5); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, d, sink, 0.5); break;
          default: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(64), 0, 0, d, sink, 0.5); break;
        }
        hipDeviceSynchronize();
      }
      static unsigned long long h[4096];
      hipMemcpy(h, d, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
      double s = 0;
      for (int i = 0; i < blocks; i++) s += (double)h[i];
      std::printf("%5d one-wave blocks  %-52s %.2f cycles per step\n", blocks, names[m], s / blocks / (64 * 64));
    }
  }
  return 0;
}
