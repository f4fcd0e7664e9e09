// Micro-benchmark: issue rate of the instructions the dense kernels are made of, on gfx950, as SIMD cycles per
// wave-instruction at 1 / 2 / 4 / 8 resident waves per SIMD (every CU busy).  Eight independent accumulators per wave,
// so a wave alone is bound by issue, not by dependent latency.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
// Output: cycles per wave-instruction seen by ONE wave (s_memtime), and the SIMD-level figure (that / waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

constexpr int INNER = 8 * 8;   // instructions per loop trip
constexpr int TRIPS = 512;

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float seed) {
  float a[8];
  double d[8];
  unsigned u[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    a[i] = out[threadIdx.x] + i;
    d[i] = a[i];
    u[i] = __float_as_uint(a[i]) | 1u;
  }
  float x = seed;
  double xd = seed;
  unsigned xu = __float_as_uint(seed);
  unsigned sacc = 0;
  unsigned long long smask = __ballot(threadIdx.x & 1);
  __shared__ float4 lbuf[1024];
  typedef unsigned uq __attribute__((ext_vector_type(4)));
  uq q[8];
  lbuf[threadIdx.x] = make_float4(seed, seed, seed, seed);
  const unsigned ldsaddr = (unsigned)(size_t)lbuf + threadIdx.x * 4 % 1024, ldsaddr16 = (unsigned)(size_t)lbuf + (threadIdx.x & 63) * 16;
#pragma unroll
  for (int i = 0; i < 8; i++) q[i] = uq{1u, 2u, 3u, 4u};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int t = 0; t < TRIPS; t++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
#define A(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define B(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(xd));
#define C(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(x));
#define D(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(xd));
#define E(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(xd));
#define F(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(xd));
#define G(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define H(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
#define I(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define J(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x));
#define K(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(x));
#define L(i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(u[i]) : "v"(xu));
#define M(i) asm volatile("v_pk_mad_u16 %0, %0, %1, %0" : "+v"(u[i]) : "v"(xu));
#define N(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(xu));
#define O(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(xu));
#define P(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(x) : "vcc");
#define Q(i) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x));
#define R(i) asm volatile("v_add_f32 %0, %0, %2\n s_add_u32 %1, %1, 1" : "+v"(a[i]), "+s"(sacc) : "v"(x));
#define S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(xd));
#define T(i) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define U(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(xd));
#define V(i) asm volatile("v_add_f32 %0, %0, %2\n s_add_u32 %1, %1, 1\n s_add_u32 %1, %1, 1" : "+v"(a[i]), "+s"(sacc) : "v"(x));
#define W(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(xu));
#define Y(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(xu), "s"(smask));
#define Z(i) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(u[i]) : "v"(xu));
#define AA(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(xu));
#define AB(i) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(u[i]));
#define AC(i) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(u[i]) : "v"(xu));
#define AD(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(u[i]) : "v"(xu));
#define AE(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(u[i]) : "v"(xu));
#define AF(i) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(u[i]) : "v"(xu));
#define AG(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sacc) : "v"(u[i]));
#define AH(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
#define AI(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
#define AJ(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc)); asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
#define AK(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define AL(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x), "v"(a[(i + 1) & 7]) : "vcc");
#define AM(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
#define AN(i) asm volatile("ds_read_b32 %0, %1" : "=v"(u[i]) : "v"(ldsaddr) : "memory");
#define AO(i) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(ldsaddr) : "memory");
#define AP(i) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"(ldsaddr16) : "memory");
#define AQ(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u[i]));
#define AR(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      if (OP == 0) { REP8(A) }
      if (OP == 1) { REP8(B) }
      if (OP == 2) { REP8(C) }
      if (OP == 3) { REP8(D) }
      if (OP == 4) { REP8(E) }
      if (OP == 5) { REP8(F) }
      if (OP == 6) { REP8(G) }
      if (OP == 7) { REP8(H) }
      if (OP == 8) { REP8(I) }
      if (OP == 9) { REP8(J) }
      if (OP == 10) { REP8(K) }
      if (OP == 11) { REP8(L) }
      if (OP == 12) { REP8(M) }
      if (OP == 13) { REP8(N) }
      if (OP == 14) { REP8(O) }
      if (OP == 15) { REP8(P) }
      if (OP == 16) { REP8(Q) }
      if (OP == 17) { REP8(R) }
      if (OP == 18) { REP8(S) }
      if (OP == 19) { REP8(T) }
      if (OP == 20) { REP8(U) }
      if (OP == 21) { REP8(V) }
      if (OP == 22) { REP8(W) }
      if (OP == 23) { REP8(X) }
      if (OP == 24) { REP8(Y) }
      if (OP == 25) { REP8(Z) }
      if (OP == 26) { REP8(AA) }
      if (OP == 27) { REP8(AB) }
      if (OP == 28) { REP8(AC) }
      if (OP == 29) { REP8(AD) }
      if (OP == 30) { REP8(AE) }
      if (OP == 31) { REP8(AF) }
      if (OP == 32) { REP8(AG) }
      if (OP == 33) { REP8(AH) }
      if (OP == 34) { REP8(AI) }
      if (OP == 35) { REP8(AJ) }
      if (OP == 36) { REP8(AK) }
      if (OP == 37) { REP8(AL) }
      if (OP == 38) { REP8(AM) }
      if (OP == 39) { REP8(AN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      if (OP == 40) { REP8(AO) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      if (OP == 41) { REP8(AP) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      if (OP == 42) { REP8(AQ) }
      if (OP == 43) { REP8(AR) }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = (float)sacc;
#pragma unroll
  for (int i = 0; i < 8; i++) r += a[i] + (float)d[i] + __uint_as_float(u[i]) + __uint_as_float(q[i].x + q[i].w);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char* name, int instr_per_unit, float* o, unsigned long long* c) {
  printf("%-34s", name);
  for (int wps : {1, 2, 4, 8}) {
    // 256 CUs x 4 SIMDs x wps waves: blocks of 256 threads (one wave per SIMD), wps blocks per CU
    const int blocks = 256 * wps;
    k<OP><<<blocks, 256>>>(o, c, 1.0000001f);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(o, c, 1.0000001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h = 0;
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double n = (double)TRIPS * INNER * instr_per_unit;
    // s_memtime runs at a fixed 100 MHz on gfx9: report wall-clock based SIMD cycles at an assumed 2.4 GHz too
    printf("  w%d: %6.2f cyc/instr/SIMD (wall@2.4GHz), memtime/instr %.3f |", wps, ms * 1e-3 * 2.4e9 / (n * wps), h / n);
  }
  printf("\n");
}

int main() {
  float* o;
  unsigned long long* c;
  hipMalloc(&o, 256 * 8 * 256 * 4);
  hipMalloc(&c, 64);
  hipMemset(o, 0, 256 * 8 * 256 * 4);
  run<0>("v_add_f32", 1, o, c);
  run<22>("v_mul_f32", 1, o, c);
  run<2>("v_fma_f32", 1, o, c);
  run<1>("v_pk_add_f32", 1, o, c);
  run<20>("v_pk_mul_f32", 1, o, c);
  run<3>("v_pk_fma_f32", 1, o, c);
  run<4>("v_add_f64", 1, o, c);
  run<18>("v_mul_f64", 1, o, c);
  run<5>("v_fma_f64", 1, o, c);
  run<6>("v_cvt_f64_f32", 1, o, c);
  run<7>("v_cvt_f32_f64", 1, o, c);
  run<8>("v_sqrt_f32", 1, o, c);
  run<9>("v_mov_b32_dpp wave_shr:1", 1, o, c);
  run<16>("v_add_f32_dpp wave_shr:1", 1, o, c);
  run<10>("v_max3_f32", 1, o, c);
  run<11>("v_dot4_u32_u8", 1, o, c);
  run<12>("v_pk_mad_u16", 1, o, c);
  run<13>("v_perm_b32", 1, o, c);
  run<23>("v_add_u32", 1, o, c);
  run<14>("v_cndmask_b32", 1, o, c);
  run<15>("v_cmp_lt_f32", 1, o, c);
  run<19>("v_cvt_f32_ubyte0", 1, o, c);
  run<24>("v_cndmask_b32_e64 (sgpr mask)", 1, o, c);
  run<37>("v_cmp + v_cndmask (pair)", 1, o, c);
  run<36>("v_max_f32", 1, o, c);
  run<43>("v_sub_f32", 1, o, c);
  run<38>("v_mul_f32 ; v_add_f32 dependent (pair)", 1, o, c);
  run<25>("v_lshl_or_b32", 1, o, c);
  run<26>("v_and_b32", 1, o, c);
  run<27>("v_bfe_u32", 1, o, c);
  run<28>("v_alignbyte_b32", 1, o, c);
  run<29>("v_mad_u32_u24", 1, o, c);
  run<30>("v_pk_add_u16", 1, o, c);
  run<31>("v_add_u32_sdwa", 1, o, c);
  run<42>("v_cvt_f32_ubyte1", 1, o, c);
  run<32>("v_readlane_b32", 1, o, c);
  run<33>("s_add_u32", 1, o, c);
  run<34>("v_add_f32 ; s_add_u32 (pair)", 1, o, c);
  run<35>("v_add_f32 ; 2 s_add_u32 (triple)", 1, o, c);
  run<39>("ds_read_b32", 1, o, c);
  run<40>("ds_read_b64", 1, o, c);
  run<41>("ds_read_b128", 1, o, c);
  return 0;
}
