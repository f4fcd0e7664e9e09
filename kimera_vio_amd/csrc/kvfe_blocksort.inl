// Block-wide bitonic sort (descending) of EPT * 1024 distinct 64-bit keys held in registers, for the 1024-thread
// select kernel.  Included by k_detect.hip and by tools/ubench/sort_bench.hip (checks it against std::sort and
// times one block).
//
// Layout: BLOCKED -- thread t owns the elements EPT t .. EPT t + EPT - 1.  Of the compare-exchange distances j of
// the network, j < EPT stay inside the thread (36 of the 91 steps for 8192 keys), EPT <= j < 64 EPT stay inside the
// wavefront and go through DPP / ds_swizzle / v_permlane{16,32}_swap (45 steps, no LDS bank traffic, no address
// arithmetic), and only j >= 64 EPT cross wavefronts through LDS (10 steps, slot-major so every access is
// conflict-free).  The striped layout this replaces (element i in thread i % 1024) had 6 / 63 / 22 steps of the three
// kinds, all 63 through ds_bpermute, and took 215 k cycles for 8192 keys.
// Equal keys only occur as zero padding, so max/min exchanges are exact.
#pragma once

namespace blocksort {

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long x) {
  const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)x, CTRL, 0xF, 0xF, false);
  const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(x >> 32), CTRL, 0xF, 0xF, false);
  return ((unsigned long long)hi << 32) | lo;
}

// value of lane (lane ^ LX)
template <int LX>
__device__ __forceinline__ unsigned long long lane_xor_u64(unsigned long long x, int lane) {
  if constexpr (LX == 1) return dpp_u64<0xB1>(x);   // quad_perm [1,0,3,2]
  if constexpr (LX == 2) return dpp_u64<0x4E>(x);   // quad_perm [2,3,0,1]
  if constexpr (LX == 8) return dpp_u64<0x128>(x);  // row_ror:8
  if constexpr (LX == 4) {                           // ds_swizzle, bit-mask mode: and 0x1f, or 0, xor 4
    const unsigned lo = __builtin_amdgcn_ds_swizzle((unsigned)x, 0x101F);
    const unsigned hi = __builtin_amdgcn_ds_swizzle((unsigned)(x >> 32), 0x101F);
    return ((unsigned long long)hi << 32) | lo;
  }
  if constexpr (LX == 16) {  // v_permlane16_swap: odd rows of the first operand <-> even rows of the second
    const auto a = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap((unsigned)(x >> 32), (unsigned)(x >> 32), false, false);
    const bool odd = (lane & 16) != 0;
    const unsigned lo = odd ? a[0] : a[1], hi = odd ? b[0] : b[1];
    return ((unsigned long long)hi << 32) | lo;
  }
  if constexpr (LX == 32) {  // v_permlane32_swap: upper half of the first operand <-> lower half of the second
    const auto a = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap((unsigned)(x >> 32), (unsigned)(x >> 32), false, false);
    const bool up = (lane & 32) != 0;
    const unsigned lo = up ? a[0] : a[1], hi = up ? b[0] : b[1];
    return ((unsigned long long)hi << 32) | lo;
  }
  return x;
}

template <int EPT, int LX>
__device__ __forceinline__ void wave_step(unsigned long long (&v)[EPT], int t, bool desc) {
  const bool keepmax = ((t & LX) == 0) == desc;
#pragma unroll
  for (int m = 0; m < EPT; m++) {
    const unsigned long long o = lane_xor_u64<LX>(v[m], t & 63);
    v[m] = ((o > v[m]) == keepmax) ? o : v[m];
  }
}

template <int EPT>
__device__ __forceinline__ void thread_steps(unsigned long long (&v)[EPT], int jmax, bool desc_thread, int k) {
  // compare-exchange distances jmax, jmax / 2 .. 1 inside the thread; k < EPT: direction from the slot index
#pragma unroll
  for (int J = EPT >> 1; J > 0; J >>= 1) {
    if (J > jmax) continue;
#pragma unroll
    for (int m = 0; m < EPT; m++) {
      if ((m & J) == 0) {
        const bool desc = k < EPT ? ((m & k) == 0) : desc_thread;
        const unsigned long long x = v[m], y = v[m | J];
        const bool sw = desc ? (x < y) : (x > y);
        v[m] = sw ? y : x;
        v[m | J] = sw ? x : y;
      }
    }
  }
}

// v[m] = element EPT * threadIdx.x + m; on return the same, sorted descending over the block.
// exch: LDS scratch of EPT * 1024 u64 (may be the array the keys were loaded from: the first access is behind a barrier)
template <int EPT, int T = 1024>
__device__ __forceinline__ void sort_desc_blocked(unsigned long long (&v)[EPT], unsigned long long* exch) {
  const int t = threadIdx.x;
  // stages k = 2 .. EPT: inside the thread
#pragma unroll
  for (int k = 2; k <= EPT; k <<= 1) thread_steps<EPT>(v, k >> 1, (t & 1) == 0, k);
  // stages k = 2 EPT .. EPT T, kt = k / EPT
  for (int kt = 2; kt <= T; kt <<= 1) {
    const bool desc = (t & kt) == 0;
    for (int jt = kt >> 1; jt >= 64; jt >>= 1) {
      __syncthreads();
#pragma unroll
      for (int m = 0; m < EPT; m++) exch[m * T + t] = v[m];
      __syncthreads();
      const bool keepmax = ((t & jt) == 0) == desc;
#pragma unroll
      for (int m = 0; m < EPT; m++) {
        const unsigned long long o = exch[m * T + (t ^ jt)];
        v[m] = ((o > v[m]) == keepmax) ? o : v[m];
      }
    }
    if (kt > 32) wave_step<EPT, 32>(v, t, desc);
    if (kt > 16) wave_step<EPT, 16>(v, t, desc);
    if (kt > 8) wave_step<EPT, 8>(v, t, desc);
    if (kt > 4) wave_step<EPT, 4>(v, t, desc);
    if (kt > 2) wave_step<EPT, 2>(v, t, desc);
    wave_step<EPT, 1>(v, t, desc);
    thread_steps<EPT>(v, EPT >> 1, desc, EPT * kt);
  }
  __syncthreads();  // exch is free again
}

}  // namespace blocksort
