// Micro-benchmark: dependent-chain latency of v_add_f64 / v_add_f32 / v_pk_add_f32 on one wave
// (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off chain_latency.hip -o chain_latency)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, double x) {
  double a = out[threadIdx.x], b = a + 1.0, c = a + 2.0, d = a + 3.0;
  float fa = (float)a, fb = fa + 1.f;
  v2f pa = {fa, fb}, px = {(float)x, (float)x};
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < 64; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (MODE == 0) a += x;
      if (MODE == 1) { a += x; b += x; }
      if (MODE == 2) { a += x; b += x; c += x; d += x; }
      if (MODE == 3) fa += (float)x;
      if (MODE == 4) pa = pa + px;
      if (MODE == 5) a = __builtin_fma(a, x, x);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a + b + c + d + fa + pa.x + pa.y;
  if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  double* o; unsigned long long* c;
  hipMalloc(&o, 64 * 8); hipMalloc(&c, 64); hipMemset(o, 0, 512); hipMemset(c, 0, 64);
  for (int rep = 0; rep < 2; rep++) {
    k<0><<<1, 64>>>(o, c, 1.0000001); k<1><<<1, 64>>>(o, c, 1.0000001); k<2><<<1, 64>>>(o, c, 1.0000001);
    k<3><<<1, 64>>>(o, c, 1.0000001); k<4><<<1, 64>>>(o, c, 1.0000001); k<5><<<1, 64>>>(o, c, 1.0000001);
  }
  unsigned long long h[8]; hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
  const char* n[6] = {"v_add_f64 x1 chain", "v_add_f64 x2 chains", "v_add_f64 x4 chains", "v_add_f32 x1 chain", "v_pk_add_f32 x1 chain", "v_fma_f64 x1 chain"};
  for (int i = 0; i < 6; i++) printf("%-24s %6.2f cycles per loop step (1024 steps)\n", n[i], h[i] / 1024.0);
  return 0;
}
