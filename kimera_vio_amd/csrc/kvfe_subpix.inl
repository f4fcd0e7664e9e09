// cv::cornerSubPix for one corner by one wavefront (included inside namespace kvfe by
// k_detect.hip and k_stereo.hip).  reference call sites: FeatureDetector.cpp:288-292,
// StereoMatcher.cpp:404-413.  The five normal-equation sums are accumulated in float64 in
// OpenCV's row-major order (lanes 0-4 each own one accumulator) => bit-identical to the CPU path.
__device__ __forceinline__ int cv_floorf(float v) { return (int)floorf(v); }

// phase stamps for tools/ubench/subpix_phases.hip (cycle counter deltas per phase of an iteration); no-ops in the
// library build
#ifdef KVFE_SUBPIX_PROF
#define KVFE_SP_T(i) do { const unsigned long long _t = __builtin_readcyclecounter(); kvfe_sp_acc[i] += _t - kvfe_sp_last; kvfe_sp_last = _t; } while (0)
#else
#define KVFE_SP_T(i) do { } while (0)
#endif

// cv::getRectSubPix(u8 -> f32) of a (n x n) window centred at c, into LDS `dst` (row stride n)
static __device__ void rect_subpix_8u32f(const unsigned char* __restrict__ img, size_t step, int W, int H,
                                  float cx, float cy, int n, float* dst, int lane, int nthr = 64) {
  const float ccx = cx - (n - 1) * 0.5f, ccy = cy - (n - 1) * 0.5f;
  const int ipx = cv_floorf(ccx), ipy = cv_floorf(ccy);
  if (0 <= ipx && ipx + n < W && 0 <= ipy && ipy + n < H) {
    float a = ccx - ipx;
    const float b = ccy - ipy;
    a = fmaxf(a, 0.0001f);
    const float a12 = a * (1.f - b), a22 = a * b, b1 = 1.f - b, b2 = b;
    const double sc = (1. - a) / a;
    const unsigned char* S = img + (size_t)ipy * step + ipx;
    for (int e = lane; e < n * n; e += nthr) {
      const int i = e / n, j = e - i * n;
      const unsigned char* R = S + (size_t)i * step;
      float prev;
      if (j == 0) {
        prev = (1 - a) * (b1 * R[0] + b2 * R[step]);
      } else {
        const float tp = a12 * R[j] + a22 * R[j + step];
        prev = (float)(tp * sc);
      }
      const float t = a12 * R[j + 1] + a22 * R[j + 1 + step];
      dst[e] = prev + t;
    }
  } else {
    const float a = ccx - ipx, b = ccy - ipy;
    const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
    const float b1 = 1.f - b, b2 = b;
    int rx, rw, ry, rh;
    if (ipx >= 0)
      rx = 0;
    else {
      rx = -ipx;
      if (rx > n) rx = n;
    }
    if (ipx < W - n)
      rw = n;
    else {
      rw = W - ipx - 1;
      if (rw < 0) rw = 0;
    }
    ry = ipy >= 0 ? 0 : -ipy;
    if (ipy < H - n)
      rh = n;
    else {
      rh = H - ipy - 1;
      if (rh < 0) rh = 0;
    }
    const int row0 = ipy >= 0 ? ipy : 0;
    for (int e = lane; e < n * n; e += nthr) {
      const int i = e / n, j = e - i * n;
      // running source row after i window rows: advances once per row with r.y <= row < r.height
      int adv = min(i, rh) - min(i, ry);
      if (adv < 0) adv = 0;
      int row = row0 + adv;
      int row2 = (i < ry || i >= rh) ? row : row + 1;
      row = min(max(row, 0), H - 1);
      row2 = min(max(row2, 0), H - 1);
      const unsigned char* S = img + (size_t)row * step;
      const unsigned char* S2 = img + (size_t)row2 * step;
      auto col = [&](int jj) {
        const int c = ipx + jj;
        return c < 0 ? 0 : (c > W - 1 ? W - 1 : c);
      };
      float v;
      if (j < rx) {
        v = S[col(rx)] * b1 + S2[col(rx)] * b2;
      } else if (j >= rw) {
        v = S[col(rw)] * b1 + S2[col(rw)] * b2;
      } else {
        v = S[col(j)] * a11 + S[col(j + 1)] * a12 + S2[col(j)] * a21 + S2[col(j + 1)] * a22;
      }
      dst[e] = v;
    }
  }
}

// LDS carve-up of corner_subpix_wave (bytes), shared by the kernels and their launchers
struct SubpixGeom {
  int ww, pw, nt, ntp, ts, rs;
  size_t terms_off, patch_off, stage_off, res_off, bytes;
};
__host__ __device__ inline SubpixGeom subpix_geom(int win) {
  SubpixGeom g;
  g.ww = 2 * win + 1;
  g.pw = g.ww + 2;
  g.nt = g.ww * g.ww;
  g.ntp = (g.nt + 31) & ~31;          // chains are walked in batches of 32, zero padded
  g.ts = g.ntp + 2;                   // chain stride (float64s): +16 B so that the five chain
                                      // lanes read five different groups of four LDS banks
  g.rs = g.pw + 1 + 2 * 6;            // staged source side: bilinear footprint + 6 px margin
  g.terms_off = 0;
  g.patch_off = sizeof(double) * 5 * (size_t)g.ts;
  g.stage_off = g.patch_off + sizeof(float) * (size_t)g.pw * g.pw;
  g.res_off = (g.stage_off + (size_t)g.rs * g.rs + 15) & ~(size_t)15;   // five float64 sums (two-wave variant)
  g.bytes = g.res_off + 64;
  return g;
}

// The rs x rs byte neighbourhood of a corner, image -> LDS stage (row stride rs), by NTHR threads (`t` = index of the
// thread among them).  rs % 4 == 0 (the reference's half size 10: rs = 36): rows are whole dwords, read as dwords at
// byte addresses with EVERY request of the thread in flight before the first LDS write -- ONE memory round trip.  (Until
// round 5 this was a byte loop; hipcc turned it into groups of four loads with a wait behind each: six dependent round
// trips per stage, and in the grouped kernel the seven other corners of the block wait at the barrier behind them.)
typedef int kvfe_int_u __attribute__((aligned(1)));   // dword at any byte address
template <int RS, int NTHR>
static __device__ __forceinline__ void subpix_load_stage(const unsigned char* __restrict__ src, size_t step,
                                                         unsigned char* stage, int t) {
  static_assert(RS % 4 == 0, "rows of whole dwords");
  constexpr int ND = RS / 4, NTASK = RS * ND, NPT = (NTASK + NTHR - 1) / NTHR;
  int v[NPT];
#pragma unroll
  for (int k = 0; k < NPT; k++) {
    const int d = t + NTHR * k;
    const int y = d / ND, xd = d - y * ND;
    v[k] = 0;
    if (d < NTASK) v[k] = *reinterpret_cast<const kvfe_int_u*>(src + (size_t)y * step + 4 * xd);
  }
#pragma unroll
  for (int k = 0; k < NPT; k++) {
    const int d = t + NTHR * k;
    if (d < NTASK) reinterpret_cast<int*>(stage)[d] = v[k];   // (y * RS + 4 * xd) / 4 == d
  }
}

// Block barrier of kernels whose waves talk through LDS only: waits for this wave's LDS traffic, not for its global
// memory requests (__syncthreads() drains vmcnt as well, which would end every prefetch at the next barrier).
static __device__ __forceinline__ void lds_block_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// interior branch of cv::getRectSubPix reading the source from the LDS stage (row stride rs);
// eij[t] = (i << 8) | j of this lane's patch entries e = lane + 64 t (fixed per corner)
template <int MAXP, int NTHR = 64>
static __device__ __forceinline__ void rect_subpix_from_stage(const unsigned char* stage, int rs,
                                                              int dx0, int dy0, float ccx, float ccy,
                                                              int ipx, int ipy, int n,
                                                              const int (&eij)[MAXP], float* dst,
                                                              int lane) {
  float a = ccx - ipx;
  const float b = ccy - ipy;
  a = fmaxf(a, 0.0001f);
  const float a12 = a * (1.f - b), a22 = a * b, b1 = 1.f - b, b2 = b;
  const double sc = (1. - a) / a;
  const unsigned char* S = stage + dy0 * rs + dx0;
#pragma unroll
  for (int t = 0; t < MAXP; t++) {
    const int e = lane + NTHR * t;
    if (e < n * n) {
      const int i = eij[t] >> 8, j = eij[t] & 255;
      const unsigned char* R = S + i * rs;
      // j == 0: prev = (1-a) * (b1*R[0] + b2*R[rs]); else prev = (float)((a12*R[j] + a22*R[j+rs]) * sc)
      const float r0 = (float)R[j], r1 = (float)R[j + rs];
      const float first = (1 - a) * (b1 * r0 + b2 * r1);
      const float tp = a12 * r0 + a22 * r1;
      const float prev = j == 0 ? first : (float)(tp * sc);
      const float tt = a12 * R[j + 1] + a22 * R[j + 1 + rs];
      dst[e] = prev + tt;
    }
  }
}

// The same branch for ONE wave per patch (lane = entry e = lane + 64 t, row-major): cv::getRectSubPix walks a row as
//   prev = (1-a) * (b1*R[0] + b2*R[rs]);  for j: t = a12*R[j+1] + a22*R[j+1+rs]; dst[j] = prev + t; prev = (float)(t * s)
// so entry (i, j)'s `prev` is entry (i, j-1)'s `t` scaled -- the value the lane to the left has just computed.  Every
// lane computes its own t (two bytes of the stage) and takes its neighbour's through a DPP wave shift (lane 0: lane 63
// of the previous batch); the 23 row heads (j = 0) are written by the first lanes on their own: two stage bytes and ~17
// instructions per entry instead of four bytes and ~30.  Same operations on the same operands in the same order as
// rect_subpix_from_stage => the same bits.
// Round 6: a patch shared by TWO waves (the one-corner kernel, NW = 2): wave w takes the entries e_base + lane + 64 t with
// e_base = 320 w (eij is built for that mapping), the second wave starts from the t of entry e_base - 1 -- which every lane
// computes for itself (two uniform byte reads) -- and `heads` says which wave writes the 23 row heads.
template <int MAXP>
static __device__ __forceinline__ void rect_subpix_from_stage_w64(const unsigned char* stage, int rs, int dx0, int dy0,
                                                                  float ccx, float ccy, int ipx, int ipy, int n,
                                                                  const int (&eij)[MAXP], float* dst, int lane,
                                                                  int e_base = 0, bool heads = true) {
  float a = ccx - ipx;
  const float b = ccy - ipy;
  a = fmaxf(a, 0.0001f);
  const float a12 = a * (1.f - b), a22 = a * b, b1 = 1.f - b, b2 = b;
  const double sc = (1. - a) / a;
  const unsigned char* S = stage + dy0 * rs + dx0;
  if (heads && lane < n) {   // entry (lane, 0)
    const unsigned char* R = S + lane * rs;
    const float r0 = (float)R[0], r1 = (float)R[rs];
    const float first = (1 - a) * (b1 * r0 + b2 * r1);
    const float t0 = a12 * R[1] + a22 * R[1 + rs];
    dst[lane * n] = first + t0;
  }
  float tt_last = 0.f;   // t of the previous batch (its lane 63 feeds lane 0)
  if (e_base > 0) {
    const int ip = (e_base - 1) / n, jp = (e_base - 1) - ip * n;
    const unsigned char* R = S + min(ip, n - 1) * rs;
    tt_last = a12 * R[jp + 1] + a22 * R[jp + 1 + rs];
  }
  // every byte of the thread first, then the arithmetic (round 6): `dst` and the stage are both LDS and may alias as far as
  // the compiler knows, so a batch's reads stayed behind the store of the batch before -- a read latency per batch
  unsigned char ra[MAXP], rb[MAXP];
#pragma unroll
  for (int t = 0; t < MAXP; t++) {
    // (entries past the patch -- the last batch's upper lanes -- compute on the last row's bytes and store nothing)
    const int i = min(eij[t] >> 8, n - 1), j = eij[t] & 255;
    const unsigned char* R = S + i * rs;
    ra[t] = R[j + 1];
    rb[t] = R[j + 1 + rs];
  }
#pragma unroll
  for (int t = 0; t < MAXP; t++) {
    const int e = e_base + lane + 64 * t;
    const int j = eij[t] & 255;
    const float tt = a12 * ra[t] + a22 * rb[t];
    float tp = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tt), 0x138, 0xf, 0xf, true));   // wave_shr:1
    const float t63 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tt_last), 63));
    tp = lane == 0 ? t63 : tp;
    const float prev = (float)(tp * sc);
    if (e < n * n && j != 0) dst[e] = prev + tt;
    tt_last = tt;
  }
}

// border branch of cv::getRectSubPix (the window leaves the image: rows and columns are replicated with OpenCV's
// r.x / r.width / r.y / r.height rules, rect_subpix_8u32f above) reading the source from the LDS stage, whose window
// [sx0, sx0 + rs) x [sy0, sy0 + rs) lies inside the image and holds every row and column the branch can address.
// New corners are found where new scene content enters the image, i.e. near its border: without this a border
// corner reads global memory in every one of its up to 40 dependent iterations and bounds the launch.
template <int MAXP, int NTHR = 64>
static __device__ __forceinline__ void rect_subpix_border_from_stage(const unsigned char* stage, int rs, int sx0,
                                                                     int sy0, int W, int H, float ccx, float ccy,
                                                                     int ipx, int ipy, int n, const int (&eij)[MAXP],
                                                                     float* dst, int lane) {
  const float a = ccx - ipx, b = ccy - ipy;
  const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
  const float b1 = 1.f - b, b2 = b;
  int rx, rw, ry, rh;
  if (ipx >= 0)
    rx = 0;
  else {
    rx = -ipx;
    if (rx > n) rx = n;
  }
  if (ipx < W - n)
    rw = n;
  else {
    rw = W - ipx - 1;
    if (rw < 0) rw = 0;
  }
  ry = ipy >= 0 ? 0 : -ipy;
  if (ipy < H - n)
    rh = n;
  else {
    rh = H - ipy - 1;
    if (rh < 0) rh = 0;
  }
  const int row0 = ipy >= 0 ? ipy : 0;
  auto col = [&](int jj) {
    const int c = ipx + jj;
    return (c < 0 ? 0 : (c > W - 1 ? W - 1 : c)) - sx0;
  };
#pragma unroll
  for (int t = 0; t < MAXP; t++) {
    const int e = lane + NTHR * t;
    if (e < n * n) {
      const int i = eij[t] >> 8, j = eij[t] & 255;
      int adv = min(i, rh) - min(i, ry);
      if (adv < 0) adv = 0;
      int row = row0 + adv;
      int row2 = (i < ry || i >= rh) ? row : row + 1;
      row = min(max(row, 0), H - 1);
      row2 = min(max(row2, 0), H - 1);
      const unsigned char* S = stage + (row - sy0) * rs;
      const unsigned char* S2 = stage + (row2 - sy0) * rs;
      float v;
      if (j < rx) {
        v = S[col(rx)] * b1 + S2[col(rx)] * b2;
      } else if (j >= rw) {
        v = S[col(rw)] * b1 + S2[col(rw)] * b2;
      } else {
        v = S[col(j)] * a11 + S[col(j + 1)] * a12 + S2[col(j)] * a21 + S2[col(j + 1)] * a22;
      }
      dst[e] = v;
    }
  }
}

// refines one corner; all 64 lanes of the wave call it with identical arguments.
// LDS (subpix_geom): terms 5 x ntp float64 | patch (2w+3)^2 float | stage rs^2 bytes.
// Latency matters here (40 strictly sequential iterations for a corner that does not converge):
// the u8 neighbourhood is staged in LDS once and re-used while the corner stays inside it, the
// Gaussian mask weights and window coordinates live in registers, and the five float64 chains
// are walked by lanes 0-4 with the LDS reads of the next batch in flight.
// Sequential float64 sum of NV2 double2 values in LDS, in index order (one chain per lane).
// A dependent v_add_f64 issues every ~6 cycles once its operand is in a register, but an LDS read
// takes ~100 cycles; hipcc keeps only three reads ahead of the adds.  Here 15 ds_read_b128 stay in
// flight (the lgkmcnt limit) in a 16-slot register ring; the waits are explicit, tied to the slot
// register so that the consuming add cannot be scheduled above its wait.
typedef double kvfe_d2 __attribute__((ext_vector_type(2)));
template <int IDX>
static __device__ __forceinline__ void lds_chain_read(kvfe_d2 (&ring)[16], unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[IDX % 16]) : "v"(addr), "n"(IDX * 16));
}
template <int NV2, int I>
static __device__ __forceinline__ void lds_chain_step(kvfe_d2 (&ring)[16], unsigned addr, double& acc) {
  constexpr int kAhead = NV2 - 1 - I < 14 ? NV2 - 1 - I : 14;  // reads issued after read I
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[I % 16]) : "n"(kAhead));
  acc += ring[I % 16].x;
  acc += ring[I % 16].y;
  if constexpr (I + 15 < NV2) lds_chain_read<I + 15>(ring, addr);
}
template <int NV2, int... Is>
static __device__ __forceinline__ void lds_chain_prologue(kvfe_d2 (&ring)[16], unsigned addr,
                                                          std::integer_sequence<int, Is...>) {
  (lds_chain_read<Is>(ring, addr), ...);
}
template <int NV2, int... Is>
static __device__ __forceinline__ void lds_chain_body(kvfe_d2 (&ring)[16], unsigned addr, double& acc,
                                                      std::integer_sequence<int, Is...>) {
  (lds_chain_step<NV2, Is>(ring, addr, acc), ...);
}
template <int NV2>
static __device__ __forceinline__ double lds_chain_sum(const double* base_lds) {
  static_assert(NV2 >= 16, "ring deeper than the chain");
  const unsigned addr = (unsigned)(size_t)base_lds;  // LDS byte address (low 32 bits of the pointer)
  kvfe_d2 ring[16];
  lds_chain_prologue<NV2>(ring, addr, std::make_integer_sequence<int, 15>{});
  double acc = 0;
  lds_chain_body<NV2>(ring, addr, acc, std::make_integer_sequence<int, NV2>{});
  return acc;
}

// Two-wave variant of the chains (WIN = 10: 448 zero-padded terms per chain).  A lone wave issues one instruction
// per ~5 cycles whatever its type (tools/ubench/subpix_phases.hip: the LDS-ring walk above costs 6.2-7.2 k cycles
// per iteration, 2 instructions per add), and a dependent v_add_f64 has 6.1 cycles of latency, so the floor is one
// instruction per add.  gfx950 has it: v_fmac_f64 is a VOP2 instruction and DP-ALU DPP supports row_newbcast, so
//     v_fmac_f64_dpp acc, T[m], 1.0 row_newbcast:p          acc = fma(T[m] of lane p of this row, 1.0, acc)
// adds, in every lane of a DPP row, the term held by lane p: fma(t, 1, acc) is the correctly rounded acc + t, i.e.
// bit-identical to v_add_f64.  Lane p of row r holds terms 28 p .. 28 p + 27 of chain r in registers (14
// ds_read_b128), the 448 adds of a chain are 448 instructions with no LDS traffic and no waits, four chains ride in
// the four DPP rows of wave 0 and the fifth in wave 1 on another SIMD.
// One asm statement per source lane p: 28 dependent v_fmac_f64_dpp (hipcc pads every asm statement with an s_nop,
// which is an issue slot of the lone wave: one statement per add would double the chain).  Operands: %0 acc, %1 1.0,
// %2..%29 the 28 terms -- 30 operands, the most an asm statement takes.
#define KVFE_FM1(P_, I_) "v_fmac_f64_dpp %0, %" #I_ ", %1 row_newbcast:" #P_ " row_mask:0xf bank_mask:0xf\n\t"
#define KVFE_FM28(P_)                                                                                                 \
  KVFE_FM1(P_, 2) KVFE_FM1(P_, 3) KVFE_FM1(P_, 4) KVFE_FM1(P_, 5) KVFE_FM1(P_, 6) KVFE_FM1(P_, 7) KVFE_FM1(P_, 8)         \
  KVFE_FM1(P_, 9) KVFE_FM1(P_, 10) KVFE_FM1(P_, 11) KVFE_FM1(P_, 12) KVFE_FM1(P_, 13) KVFE_FM1(P_, 14) KVFE_FM1(P_, 15)   \
  KVFE_FM1(P_, 16) KVFE_FM1(P_, 17) KVFE_FM1(P_, 18) KVFE_FM1(P_, 19) KVFE_FM1(P_, 20) KVFE_FM1(P_, 21) KVFE_FM1(P_, 22) \
  KVFE_FM1(P_, 23) KVFE_FM1(P_, 24) KVFE_FM1(P_, 25) KVFE_FM1(P_, 26) KVFE_FM1(P_, 27) KVFE_FM1(P_, 28) KVFE_FM1(P_, 29)
#define KVFE_CHAIN_LANE(P_)                                                                                           \
  asm volatile(KVFE_FM28(P_)                                                                                          \
               : "+v"(acc)                                                                                            \
               : "v"(one), "v"(T[0]), "v"(T[1]), "v"(T[2]), "v"(T[3]), "v"(T[4]), "v"(T[5]), "v"(T[6]), "v"(T[7]),     \
                 "v"(T[8]), "v"(T[9]), "v"(T[10]), "v"(T[11]), "v"(T[12]), "v"(T[13]), "v"(T[14]), "v"(T[15]),        \
                 "v"(T[16]), "v"(T[17]), "v"(T[18]), "v"(T[19]), "v"(T[20]), "v"(T[21]), "v"(T[22]), "v"(T[23]),      \
                 "v"(T[24]), "v"(T[25]), "v"(T[26]), "v"(T[27]))
// chain_base: this lane's 28 consecutive terms (16-byte aligned LDS address)
static __device__ __forceinline__ double dpp_chain_sum448(const double* chain_base) {
  const kvfe_d2* src = reinterpret_cast<const kvfe_d2*>(chain_base);
  double T[28];
#pragma unroll
  for (int m = 0; m < 14; m++) {
    const kvfe_d2 v = src[m];
    T[2 * m] = v.x;
    T[2 * m + 1] = v.y;
  }
  double one = 1.0, acc = 0.0;
  // the terms come from LDS loads (no VALU write feeds a DPP read); `one` and `acc` are not DPP operands
  asm volatile("s_nop 1" : "+v"(one), "+v"(acc));
  KVFE_CHAIN_LANE(0);
  KVFE_CHAIN_LANE(1);
  KVFE_CHAIN_LANE(2);
  KVFE_CHAIN_LANE(3);
  KVFE_CHAIN_LANE(4);
  KVFE_CHAIN_LANE(5);
  KVFE_CHAIN_LANE(6);
  KVFE_CHAIN_LANE(7);
  KVFE_CHAIN_LANE(8);
  KVFE_CHAIN_LANE(9);
  KVFE_CHAIN_LANE(10);
  KVFE_CHAIN_LANE(11);
  KVFE_CHAIN_LANE(12);
  KVFE_CHAIN_LANE(13);
  KVFE_CHAIN_LANE(14);
  KVFE_CHAIN_LANE(15);
  return acc;
}

// WIN > 0: window half size fixed at compile time (all loops unroll, divisions fold); WIN == 0:
// any half size up to 15 at run time.
// NW = number of wavefronts working on the corner (block of 64 NW threads, `lane` = thread index in the block):
// 1, or 2 for WIN = 10 (patch and term work split between the waves, chains by dpp_chain_sum448).
template <int WIN, int NW = 1>
static __device__ float2 corner_subpix_wave_t(const unsigned char* __restrict__ img, size_t step, int W,
                                       int H, float2 cT, int win_rt, int max_iters, double eps2,
                                       const float* __restrict__ mask, unsigned char* lds, int lane) {
  static_assert(NW == 1 || ((NW == 2 || NW == 4) && WIN == 10), "the multi-wave variants are written for half size 10");
  constexpr int NTHR = 64 * NW;
  const int win = WIN > 0 ? WIN : win_rt;
  const SubpixGeom G = subpix_geom(win);
  const int ww = G.ww, pw = G.pw, nt = G.nt, ntp = G.ntp, ts = G.ts, rs = G.rs;
  constexpr int MAXT = WIN > 0 ? ((2 * WIN + 1) * (2 * WIN + 1) + NTHR - 1) / NTHR : 16;
  constexpr int MAXP = WIN > 0 ? ((2 * WIN + 3) * (2 * WIN + 3) + NTHR - 1) / NTHR : 18;
  double* terms = reinterpret_cast<double*>(lds + G.terms_off);
  float* patch = reinterpret_cast<float*>(lds + G.patch_off);
  unsigned char* stage = lds + G.stage_off;
  // per-lane term slots k = lane + 64 t: mask weight and window coordinates (fixed per corner)
  float mk[MAXT];
  int pij[MAXT];            // (i << 8) | j
#pragma unroll
  for (int t = 0; t < MAXT; t++) {
    const int k = lane + NTHR * t;
    mk[t] = 0.f;
    pij[t] = 0;
    if (k < nt) {
      const int i = k / ww, j = k - i * ww;
      mk[t] = mask[k];
      pij[t] = (i << 8) | j;
    }
  }
  // patch entries of this thread in the (2w+3)^2 window: e = pe0 + PES t.  One or four waves: thread index + NTHR t.  Two
  // waves (round 6): wave w owns the CONTIGUOUS entries 320 w + lane + 64 t, so that an entry's left neighbour is the lane
  // before it and the interior branch can take the neighbour's product through a DPP shift (rect_subpix_from_stage_w64:
  // two stage bytes and ~17 instructions per entry instead of four and ~30)
  constexpr bool SPLIT2 = NW == 2;
  constexpr int PES = SPLIT2 ? 64 : NTHR;
  const int pe0 = SPLIT2 ? (lane & 63) + 320 * (lane >> 6) : lane;
  int eij[MAXP];
#pragma unroll
  for (int t = 0; t < MAXP; t++) {
    const int e = pe0 + PES * t;
    const int i = e / pw;
    eij[t] = (i << 8) | (e - i * pw);
  }
  for (int k = nt + lane; k < ntp; k += NTHR)
    for (int q = 0; q < 5; q++) terms[q * ts + k] = 0.0;
  int sx0 = 0, sy0 = 0;
  bool staged = false;
  float2 cI = cT;
  int iter = 0;
  double err = 0;
#ifdef KVFE_SUBPIX_PROF
  unsigned long long kvfe_sp_acc[6] = {0, 0, 0, 0, 0, 0}, kvfe_sp_last = __builtin_readcyclecounter();
#endif
  do {
    KVFE_SP_T(5);
    {
      const float ccx = cI.x - (pw - 1) * 0.5f, ccy = cI.y - (pw - 1) * 0.5f;
      const int ipx = cv_floorf(ccx), ipy = cv_floorf(ccy);
      const bool interior = 0 <= ipx && ipx + pw < W && 0 <= ipy && ipy + pw < H;
      bool use_stage = false;
      // footprint inside the image: columns fx0 .. fx1, rows fy0 .. fy1 (interior: ipx .. ipx+pw, ipy .. ipy+pw; a
      // window that leaves the image addresses the clamped range, see rect_subpix_border_from_stage)
      const int fx0 = min(max(ipx, 0), W - 1), fx1 = min(max(ipx + pw, 0), W - 1);
      const int fy0 = min(max(ipy, 0), H - 1), fy1 = min(max(ipy + pw, 0), H - 1);
      if (W >= rs && H >= rs) {
        if (!staged || fx0 < sx0 || fy0 < sy0 || fx1 >= sx0 + rs || fy1 >= sy0 + rs) {
          const int nx0 = min(max(ipx - 6, 0), W - rs), ny0 = min(max(ipy - 6, 0), H - rs);
          __syncthreads();
          if constexpr (WIN > 0 && (2 * WIN + 16) % 4 == 0) {
            subpix_load_stage<2 * WIN + 16, NTHR>(img + (size_t)ny0 * step + nx0, step, stage, lane);
          } else {
            for (int e = lane; e < rs * rs; e += NTHR) {
              const int y = e / rs, x = e - y * rs;
              stage[e] = img[(size_t)(ny0 + y) * step + nx0 + x];
            }
          }
          __syncthreads();
          sx0 = nx0;
          sy0 = ny0;
          staged = true;
        }
        use_stage = staged && fx0 >= sx0 && fy0 >= sy0 && fx1 < sx0 + rs && fy1 < sy0 + rs;
      }
      if (use_stage && interior) {
        if constexpr (SPLIT2)
          rect_subpix_from_stage_w64<MAXP>(stage, rs, ipx - sx0, ipy - sy0, ccx, ccy, ipx, ipy, pw, eij, patch, lane & 63,
                                           320 * (lane >> 6), (lane >> 6) == 1);
        else
          rect_subpix_from_stage<MAXP, NTHR>(stage, rs, ipx - sx0, ipy - sy0, ccx, ccy, ipx, ipy, pw, eij, patch, lane);
      } else if (use_stage) {   // (the functions index entries as `lane + NTHR t`: here pe0 + PES t)
        rect_subpix_border_from_stage<MAXP, PES>(stage, rs, sx0, sy0, W, H, ccx, ccy, ipx, ipy, pw, eij, patch, pe0);
      }
      else
        rect_subpix_8u32f(img, step, W, H, cI.x, cI.y, pw, patch, lane, NTHR);
    }
    __syncthreads();
    KVFE_SP_T(0);
    // (the patch values of all slots first: `terms` and `patch` are both LDS, and as far as the compiler knows a slot's reads
    // must stay behind the stores of the slot before -- round 6)
    float pxr[MAXT], pxl[MAXT], pyd[MAXT], pyu[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
      const int i = pij[t] >> 8, j = pij[t] & 255;   // (slots past the window hold (0, 0): a valid address, unused)
      const float* sp = patch + (i + 1) * pw + (j + 1);
      pxr[t] = sp[1];
      pxl[t] = sp[-1];
      pyd[t] = sp[pw];
      pyu[t] = sp[-pw];
    }
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
      const int k = lane + NTHR * t;
      if (k < nt) {
        const int i = pij[t] >> 8, j = pij[t] & 255;
        const double m = (double)mk[t];
        const double tgx = (double)(pxr[t] - pxl[t]);
        const double tgy = (double)(pyd[t] - pyu[t]);
        const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
        const double px = (double)(j - win), py = (double)(i - win);
        terms[k] = gxx;
        terms[ts + k] = gxy;
        terms[2 * ts + k] = gyy;
        terms[3 * ts + k] = gxx * px + gxy * py;
        terms[4 * ts + k] = gxy * px + gyy * py;
      }
    }
    __syncthreads();
    KVFE_SP_T(1);
    // the five sequential float64 chains (lanes 0-4): a dependent v_add_f64 issues every ~6 cycles
    // as long as its operand has landed, so the loop only has to keep LDS reads ahead of the adds
    double a, b, c, bb1, bb2;
    if constexpr (NW >= 2) {
      // wave 0: DPP row r walks chain r (a, b, c, bb1); wave 1: chain 4 (bb2) -- see dpp_chain_sum448; waves 2, 3 of
      // the four-wave variant only share the patch and term work
      double* res = reinterpret_cast<double*>(lds + G.res_off);
      const int wv = lane >> 6, l = lane & 63, q = wv == 0 ? (l >> 4) : 4;
      if (wv < 2) {
        const double acc = dpp_chain_sum448(terms + q * ts + 28 * (l & 15));
        if ((l & 15) == 0 && (wv == 0 || l == 0)) res[q] = acc;
      }
      KVFE_SP_T(2);
      __syncthreads();
      a = res[0];
      b = res[1];
      c = res[2];
      bb1 = res[3];
      bb2 = res[4];
    } else {
      double acc = 0;
      if (lane < 5) {
        const double2* t2 = reinterpret_cast<const double2*>(terms + lane * ts);
        if (WIN > 0) {
          constexpr int NTP = (((2 * WIN + 1) * (2 * WIN + 1) + 31) & ~31);
          acc = lds_chain_sum<(WIN > 0 ? NTP / 2 : 16)>(terms + lane * ts);
        } else {
#pragma unroll 16
          for (int k2 = 0; k2 < ntp / 2; k2++) {
            const double2 v = t2[k2];
            acc += v.x;
            acc += v.y;
          }
        }
      }
      KVFE_SP_T(2);
      a = __shfl(acc, 0);
      b = __shfl(acc, 1);
      c = __shfl(acc, 2);
      bb1 = __shfl(acc, 3);
      bb2 = __shfl(acc, 4);
      __syncthreads();
    }
    KVFE_SP_T(3);
    const double det = a * c - b * b;
    if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) break;
    const double scale = 1.0 / det;
    float2 cI2;
    cI2.x = (float)(cI.x + c * scale * bb1 - b * scale * bb2);
    cI2.y = (float)(cI.y - b * scale * bb1 + a * scale * bb2);
    err = (double)((cI2.x - cI.x) * (cI2.x - cI.x) + (cI2.y - cI.y) * (cI2.y - cI.y));
    cI = cI2;
    if (cI.x < 0 || cI.x >= W || cI.y < 0 || cI.y >= H) break;
    KVFE_SP_T(4);
  } while (++iter < max_iters && err > eps2);
#ifdef KVFE_SUBPIX_PROF
  if (lane == 0 && kvfe_sp_out)
    for (int i = 0; i < 6; i++) kvfe_sp_out[i] = kvfe_sp_acc[i];
  if (lane == 0 && kvfe_sp_out) kvfe_sp_out[6] = (unsigned long long)iter;
#endif
  if (fabsf(cI.x - cT.x) > win || fabsf(cI.y - cT.y) > win) cI = cT;
  return cI;
}

// The reference uses half size 10 everywhere (FeatureDetector.cpp:288-292,
// StereoMatcher.cpp:406-411): kernels are instantiated for WIN = 10 (everything unrolled, window
// coordinates and mask weights register resident) and WIN = 0 (any half size at run time), and the
// launcher picks one, so the hot instantiation does not pay the registers of the generic one.
template <int WIN, int NW = 1>
static __device__ __forceinline__ float2 corner_subpix_wave(const unsigned char* __restrict__ img,
                                                            size_t step, int W, int H, float2 cT,
                                                            int win, int max_iters, double eps2,
                                                            const float* __restrict__ mask,
                                                            unsigned char* lds, int lane) {
  return corner_subpix_wave_t<WIN, NW>(img, step, W, H, cT, win, max_iters, eps2, mask, lds, lane);
}
