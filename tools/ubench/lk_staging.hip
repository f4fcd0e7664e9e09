// Micro-benchmark for the open question of profiles/r2_v5_lk_analysis.md: the batched LK launch runs at 75-80
// points/us whatever its instruction count or occupancy, and every phase that touches LDS or memory is 2-2.6x slower
// under load.  Which part of the window staging is it?  One wave per point (64 streams x 640 points, 752x480 images,
// three pyramid levels like lk_kernel_sys<24>), each level = stage a 27x27 byte window of image A + a 31x31 window of
// image B into LDS, then a dummy compute phase of `valu` dependent VALU steps that reads the staged data back from LDS
// (`lds_reads` ds_read_b32 per step block).  Variants of the staging:
//   0  byte loads, one byte per lane and iteration          (the kernel of profiles/r2_v4_*)
//   1  dword loads at byte addresses, 4 / 8 bytes per task   (the kernel of profiles/r2_v5_*)
//   2  aligned 16-byte loads, <= 3 per window row, raw rows to LDS with ds_write_b128
//   3  no staging at all (the compute phase alone: the floor)
// and of the launch: 4.8 / 7.7 KB of LDS per point, 64 / 128 threads per workgroup (two points per workgroup).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value lk_staging.hip -o lk_staging && ./lk_staging
// (variant 2 reads 16-byte aligned addresses of the packed pyramid, i.e. not exactly the window at the odd-width levels:
// this is a timing experiment, the values are never checked)
// Prints one line per (variant, valu, lds_reads): kernel time, points/us.  Round 3 runs this first.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int W = 752, H = 480, B = 64, P = 640, LEVELS = 3;
typedef int int_u __attribute__((aligned(1)));

struct Img {
  const unsigned char* p;
  int w, h;
};

__device__ __forceinline__ Img level(const unsigned char* base, int s, int l) {
  // levels of one stream packed one after the other: 752x480, 376x240, 188x120
  size_t off = (size_t)s * (W * H + W * H / 4 + W * H / 16);
  int w = W, h = H;
  for (int i = 0; i < l; i++) {
    off += (size_t)w * h;
    w >>= 1;
    h >>= 1;
  }
  return Img{base + off, w, h};
}

template <int VARIANT>
__device__ __forceinline__ void stage(const Img& I, int x0, int y0, int rows, int cols, unsigned char* lds, int lane) {
  // window [y0, y0 + rows) x [x0, x0 + cols) -> lds, row stride 32 bytes (cols <= 31)
  if (VARIANT == 0) {
    for (int e = lane; e < rows * cols; e += 64) {
      const int y = e / cols, x = e - y * cols;
      lds[y * 32 + x] = I.p[(size_t)(y0 + y) * I.w + x0 + x];
    }
  } else if (VARIANT == 1) {
    const int per = (cols + 3) / 4;
    for (int e = lane; e < rows * per; e += 64) {
      const int y = e / per, t = e - y * per;
      *reinterpret_cast<int*>(lds + y * 32 + 4 * t) =
          *reinterpret_cast<const int_u*>(I.p + (size_t)(y0 + y) * I.w + x0 + 4 * t);
    }
  } else if (VARIANT == 2) {
    // aligned 16-byte chunks covering [x0, x0 + cols): at most 3 per row; the row lands in LDS from its aligned start
    // (row stride 48 bytes), the consumer adds x0 & 15
    const int a0 = x0 & ~15;
    for (int e = lane; e < rows * 3; e += 64) {
      const int y = e / 3, c = e - y * 3;
      if (a0 + 16 * c < x0 + cols) {
        const int4 v = *reinterpret_cast<const int4*>(I.p + (((size_t)(y0 + y) * I.w + a0 + 16 * c) & ~(size_t)15));
        *reinterpret_cast<int4*>(lds + y * 48 + 16 * c) = v;
      }
    }
  }
}

template <int VARIANT, int LDS_BYTES, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lk_like(const unsigned char* A, const unsigned char* Bimg,
                                                     const short2* pts, float* out, int valu, int lds_reads) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[LDS_BYTES * WAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s = blockIdx.y, pt = blockIdx.x * WAVES + wave;
  if (pt >= P) return;
  unsigned char* lds = lds_all + wave * LDS_BYTES;
  const short2 p0 = pts[s * P + pt];
  float acc = (float)lane;
  for (int l = LEVELS - 1; l >= 0; l--) {
    const Img IA = level(A, s, l), IB = level(Bimg, s, l);
    int x0 = (p0.x >> l) - 13, y0 = (p0.y >> l) - 13;
    x0 = min(max(x0, 0), IA.w - 48);
    y0 = min(max(y0, 0), IA.h - 32);
    __builtin_amdgcn_wave_barrier();
    stage<VARIANT>(IA, x0, y0, 27, 27, lds, lane);
    stage<VARIANT>(IB, x0 + 1, y0 + 1, 31, 31, lds + 27 * 48, lane);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // compute phase: dependent VALU steps with an LDS read every valu / lds_reads steps
    const int* ldw = reinterpret_cast<const int*>(lds);
    const int every = lds_reads > 0 ? max(valu / lds_reads, 1) : (1 << 30);
    int idx = lane, left = 0;
    for (int i = 0; i < valu; i++) {
      acc = acc * 1.0000001f + 0.5f;
      if (--left < 0) {   // (uniform branch)
        left = every - 1;
        idx = (idx * 5 + 3) & 511;
        acc += (float)(ldw[idx] & 255);
      }
    }
  }
  if (lane == 0) out[s * P + pt] = acc;
}

template <int VARIANT, int LDS_BYTES, int WAVES>
static void run(const unsigned char* A, const unsigned char* Bi, const short2* pts, float* out, int valu, int reads,
                const char* name) {
  dim3 grid((P + WAVES - 1) / WAVES, B), block(64 * WAVES);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((lk_like<VARIANT, LDS_BYTES, WAVES>), grid, block, 0, 0, A, Bi, pts, out, valu, reads);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((lk_like<VARIANT, LDS_BYTES, WAVES>), grid, block, 0, 0, A, Bi, pts, out, valu, reads);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  std::printf("%-34s lds/pt %5d B  waves/wg %d  valu %5d  lds_reads %4d : %8.1f us  %6.1f points/us\n", name, LDS_BYTES,
              WAVES, valu, reads, ms * 1e3, (double)B * P / (ms * 1e3));
}

int main() {
  const size_t per_stream = (size_t)W * H + W * H / 4 + W * H / 16;
  const size_t bytes = per_stream * B + 4096;
  unsigned char *A, *Bi;
  short2* pts;
  float* out;
  hipMalloc(&A, bytes);
  hipMalloc(&Bi, bytes);
  hipMalloc(&pts, sizeof(short2) * B * P);
  hipMalloc(&out, sizeof(float) * B * P);
  std::vector<unsigned char> h(bytes);
  srand(1);
  for (auto& v : h) v = (unsigned char)rand();
  hipMemcpy(A, h.data(), bytes, hipMemcpyHostToDevice);
  hipMemcpy(Bi, h.data(), bytes, hipMemcpyHostToDevice);
  std::vector<short2> hp((size_t)B * P);
  for (auto& p : hp) p = short2{(short)(20 + rand() % (W - 40)), (short)(20 + rand() % (H - 40))};
  hipMemcpy(pts, hp.data(), sizeof(short2) * hp.size(), hipMemcpyHostToDevice);
  const int valus[3] = {0, 1500, 5000}, reads[2] = {0, 120};
  for (int v : valus)
    for (int r : reads) {
      if (v == 0 && r) continue;
      run<0, 4864, 1>(A, Bi, pts, out, v, r, "0 byte loads");
      run<1, 4864, 1>(A, Bi, pts, out, v, r, "1 dword loads at byte addresses");
      run<2, 4864, 1>(A, Bi, pts, out, v, r, "2 aligned 16-byte loads");
      run<3, 4864, 1>(A, Bi, pts, out, v, r, "3 no staging");
      run<1, 7744, 1>(A, Bi, pts, out, v, r, "1 dword loads, 7.7 KB LDS");
      run<1, 4864, 2>(A, Bi, pts, out, v, r, "1 dword loads, 2 points / workgroup");
      run<2, 4864, 2>(A, Bi, pts, out, v, r, "2 16-byte loads, 2 points / workgroup");
    }
  return 0;
}
