// Shared by the Lucas-Kanade kernels (k_track.hip: one wavefront per point; k_lk8.hip: eight points per wavefront):
// pyramid level views, OpenCV's 14-bit bilinear weights, packed 16-bit dot products and DPP helpers.
// Included inside namespace kvfe.

struct LevelImg {
  const unsigned char* p;
  int w, h;
  size_t stride;
};

__device__ __forceinline__ LevelImg level_img(const KParams& P, const unsigned char* img0,
                                              size_t row_stride, const unsigned char* pyr, int l) {
  LevelImg L;
  if (l == 0) {
    L.p = img0;
    L.stride = row_stride;
  } else {
    L.p = pyr + P.loff[l];
    L.stride = (size_t)P.lw[l];
  }
  L.w = P.lw[l];
  L.h = P.lh[l];
  return L;
}

__device__ __forceinline__ int at101(const LevelImg& L, int x, int y) {
  return L.p[(size_t)reflect101(y, L.h) * L.stride + reflect101(x, L.w)];
}

__device__ __forceinline__ void lk_weights(float a, float b, int* w00, int* w01, int* w10,
                                           int* w11) {
  const int W_BITS = 14;
  *w00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << W_BITS));
  *w01 = __float2int_rn(a * (1.f - b) * (1 << W_BITS));
  *w10 = __float2int_rn((1.f - a) * b * (1 << W_BITS));
  *w11 = (1 << W_BITS) - *w00 - *w01 - *w10;
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dpp_row_shr1(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
// carry hand-over of a systolic chain fused with the next addition: (value of lane q-1, 0 for the first lane of a DPP row)
// + b, one v_add_f32_dpp.  Stage 0 of a chain passes 0 as carry: 0 + x is x in IEEE arithmetic (the terms are never -0
// sums that matter: +0 + -0 = +0 only changes the sign of a zero), exactly the `carry = 0` start of the plain form.
__device__ __forceinline__ float dpp_shr1_add(float carry_src, float b) {
  // plain form: hipcc fuses the DPP move + add into one v_add_f32_dpp where it can and inserts the wait states itself
  return dpp_row_shr1(carry_src) + b;
}
// The same hand-over for the two / three carries of a chain stage in ONE asm statement.  gfx9 needs 2 wait states
// between a VALU write of a VGPR and a DPP read of it, and hipcc cannot see a DPP operand inside inline asm (ADVICE round 3:
// 99 of 215 sites had 1 wait state).  The leading `s_nop 1` provides them for the first carry whatever instruction the
// compiler scheduled in front of the statement; it and the first hand-over provide them for the second and third.
// tools/check_dpp_hazard.py scans the compiled ISA for this hazard (tests/test_host_logic.py runs it).
__device__ __forceinline__ void dpp_shr1_add3(float& c0, float& c1, float& c2, float b0, float b1, float b2) {
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %2, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "+v"(c0), "+v"(c1), "+v"(c2)
      : "v"(b0), "v"(b1), "v"(b2));
}
__device__ __forceinline__ void dpp_shr1_add2(float& c0, float& c1, float b0, float b1) {
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %1, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "+v"(c0), "+v"(c1)
      : "v"(b0), "v"(b1));
}
__device__ __forceinline__ int dot2_i16(int a, int b, int c) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b), c, false);
}
// c + a1 . b1 + a0 . b0 and (a . b0, a . b1) as three-operand dot products.  hipcc only emits the accumulate-in-place form
// (v_dot2c), which costs a v_mov per start value -- 36 of the 320 vector instructions of an LK iteration.  The hazards
// the compiler would otherwise handle are closed inside the block: a dot product feeding the SAME opcode's accumulator
// needs no wait state, any other reader of a dot product's result needs three (s_nop 2).
__device__ __forceinline__ int dot2x2_i16(int a1, int b1, int c, int a0, int b0) {
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, %3\n\tv_dot2_i32_i16 %0, %4, %5, %0\n\ts_nop 2"
      : "=&v"(d)
      : "v"(a1), "v"(b1), "v"(c), "v"(a0), "v"(b0));
  return d;
}
__device__ __forceinline__ void dot2_pair0_i16(int a, int b0, int b1, int& d0, int& d1) {
  asm("v_dot2_i32_i16 %0, %2, %3, 0\n\tv_dot2_i32_i16 %1, %2, %4, 0\n\ts_nop 2"
      : "=&v"(d0), "=&v"(d1)
      : "v"(a), "v"(b0), "v"(b1));
}
__device__ __forceinline__ int pack_lo16(int lo, int hi) {  // (lo & 0xffff) | (hi << 16)
  return (int)__builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x05040100u);
}
__device__ __forceinline__ int pack_hi16(int lo, int hi) {  // (lo >> 16) | (hi & 0xffff0000)
  return (int)__builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x07060302u);
}
__device__ __forceinline__ float lane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ int sat16(int v) { return max(-32768, min(32767, v)); }

typedef int int_u __attribute__((aligned(1)));              // dword at any byte address
typedef unsigned short ushort_u __attribute__((aligned(1)));
typedef unsigned short v2us __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2us as_v2us(int v) { return __builtin_bit_cast(v2us, v); }
__device__ __forceinline__ int as_i32(v2us v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ int perm_b32(int hi, int lo, unsigned sel) {
  return (int)__builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, sel);
}
