// The two-piece f16 split of an fp32 operand, one instruction per function (gfx950):
//     v = v1 + v2' / 4096 + e,   v1 = rn16(v),   v2' = rn16(4096 (v - v1)),   |e| <= 2^-23 |v|
// (v - v1 is exact in fp32; the residual is kept scaled by 2^12 so that it stays a normal f16 number), and
//     a b ~ a1 b1 + 2^-12 (a1 b2' + a2' b1)        three v_mfma_f32_16x16x32_f16, two fp32 accumulators.
// Dropped: a2 b2 <= 2^-22 |a b|. Against a float64 product (tools/microbench/f16_split.hip, K = 128) the max
// error is 0.35 x that of an fp32 fmaf chain.
//
// RANGE. f16 has 5 exponent bits, so the identity above holds only inside a window:
//   * top: |v| >= 32768 — the head fits up to 65504, but in the top binade the residual reaches 16 and its scaled
//     form 16 * 4096 = 65536 overflows to Inf (v = 32784 does it);
//   * bottom: below 2^-14 the head is an f16 denormal and below 2^-25 both pieces lose bits: the ABSOLUTE error
//     floor is 2^-37 (the matrix core honours f16 denormals, so this is all that happens). That is harmless
//     next to larger neighbours, and wrong for operands that are small as a whole (a gradient row of 1e-9).
// Callers therefore pass what they split through a RangeTrack — the running max |v| and the running min, over the
// aligned 4-element segments a lane splits, of the segment's max |v| (all-zero segments aside) — and re-evaluate in
// fp32 whatever saw |v| >= 32768 or a segment whose max is below 2^-18 (range_bad). The segment stands in for the
// ROW (a row maximum would cost a cross-lane reduction per row): right for operands without structure inside a row
// (embedding sums, layer-normed rows, weights), wrong for a recurrent h, where saturated gates leave 1e-14 next to
// 0.9 — h therefore takes the top check only (range_seg4_hi). What stays on the fast path
// has every element within max(2^-23 |v|, 2^-37) <= 2^-19 x its segment's max of the fp32 value (2^-23 when the
// segment's max is >= 2^-14). Operands whose scale is arbitrary by nature (gradients: attn_bwd_tail_f16.hip) are
// brought into the window by an exact power-of-two scale per row instead.
#pragma once
#include <stdint.h>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float kF16Lim = 32768.f;          // |v| >= this does not split (see RANGE)
constexpr uint32_t kF16LowBits = 0x36800000u;  // bits of 2^-18: a segment whose max is in (0, 2^-18) does not either
constexpr float kLoInv = 1.f / 4096.f;

// heads of two floats (round to nearest), packed with the first in the low half
__device__ __forceinline__ int head2(float a, float b) {
  return __builtin_bit_cast(int, __builtin_convertvector((f32x2{a, b}), f16x2));   // v_cvt_pk_f16_f32
}
// a - (the low / high half of pk), exact
template <int HI>
__device__ __forceinline__ float resid(int pk, float a) {
  float r;
  if constexpr (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(a));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(a));
  return r;
}
// rn16(r * k) into the low / high half of a pair (k = 4096 in a register: VOP3P takes no literal)
__device__ __forceinline__ int tail_lo(float r, float k) {
  int s;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(s) : "v"(r), "v"(k));
  return s;
}
__device__ __forceinline__ int tail_hi(int s, float r, float k) {
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(s) : "v"(r), "v"(k));
  return s;
}
// scaled residuals of two floats whose packed heads are hd
__device__ __forceinline__ int tail2(int hd, float a, float b, float k) {
  return tail_hi(tail_lo(resid<0>(hd, a), k), resid<1>(hd, b), k);
}
__device__ __forceinline__ float max3abs(float m, float a, float b) {
  float r;   // one instruction (fmaxf(fabsf(.)) costs a canonicalising v_max per operand); a NaN operand is ignored
  asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
  return r;
}

// max(|a|, |b|, |c|) in one instruction
__device__ __forceinline__ float maxabs3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float maxabs_acc(float m, float a) {   // max(m, |a|), m >= 0
  float r;
  asm("v_max_f32 %0, %1, |%2|" : "=v"(r) : "v"(m), "v"(a));
  return r;
}

// The range of everything a lane has split so far (see RANGE above).
struct RangeTrack {
  float hi;        // max |v|
  uint32_t lo;     // min over the segments of bits(segment max) - 1: an all-zero segment wraps to 0xFFFFFFFF and never lowers it
};
__device__ __forceinline__ RangeTrack range_init() { return RangeTrack{0.f, 0xFFFFFFFFu}; }
// one segment whose max |v| is s (>= 0)
__device__ __forceinline__ void range_seg(RangeTrack& r, float s) {
  r.hi = maxabs_acc(r.hi, s);
  const uint32_t t = __builtin_bit_cast(uint32_t, s) - 1u;
  r.lo = t < r.lo ? t : r.lo;
}
__device__ __forceinline__ void range_seg4(RangeTrack& r, float a, float b, float c, float d) {
  range_seg(r, maxabs_acc(maxabs3(a, b, c), d));
}
// the top of the window only — for operands that are bounded below by construction or whose small rows are caught
// through another operand: a recurrent h (saturated gates leave 1e-14 next to 0.9 in one row; a row of h that is tiny
// as a whole comes from an x row that is, and that one is caught)
__device__ __forceinline__ void range_seg4_hi(RangeTrack& r, float a, float b, float c, float d) {
  r.hi = maxabs_acc(r.hi, maxabs_acc(maxabs3(a, b, c), d));
}
__device__ __forceinline__ bool range_bad(const RangeTrack& r) { return r.hi >= kF16Lim || r.lo < kF16LowBits - 1u; }

// ---- the same split of v * s for a power of two s (v * s is exact), without forming v * s:
// rn16(a s) into the low / high half of a pair
__device__ __forceinline__ int head_lo_scaled(float a, float s) {
  int h;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "v"(s));
  return h;
}
__device__ __forceinline__ int head_hi_scaled(int h, float a, float s) {
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a), "v"(s));
  return h;
}
// a s - (the low / high half of pk), exact
template <int HI>
__device__ __forceinline__ float resid_scaled(int pk, float a, float s) {
  float r;
  if constexpr (HI) asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(s), "v"(pk));
  else asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(s), "v"(pk));
  return r;
}
// heads and scaled residuals of (a s, b s)
__device__ __forceinline__ void split2_scaled(float a, float b, float s, float k, int& hd, int& tl) {
  hd = head_hi_scaled(head_lo_scaled(a, s), b, s);
  tl = tail_hi(tail_lo(resid_scaled<0>(hd, a, s), k), resid_scaled<1>(hd, b, s), k);
}

}  // namespace
