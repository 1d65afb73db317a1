// The two-piece f16 split of an fp32 operand, one instruction per function (gfx950):
//     v = v1 + v2' / 4096 + e,   v1 = rn16(v),   v2' = rn16(4096 (v - v1)),   |e| <= 2^-23 |v|
// (v - v1 is exact in fp32; the residual is kept scaled by 2^12 so that it stays a normal f16 number), and
//     a b ~ a1 b1 + 2^-12 (a1 b2' + a2' b1)        three v_mfma_f32_16x16x32_f16, two fp32 accumulators.
// Dropped: a2 b2 <= 2^-22 |a b|. Against a float64 product (tools/microbench/f16_split.hip, K = 128) the max
// error is 0.35 x that of an fp32 fmaf chain. |v| > 65504 does not fit a piece: callers keep a running
// max of what they split (max3abs) and re-evaluate in fp32 what exceeded it. The matrix core honours f16
// denormals (same microbenchmark), so small values only move bits into the scaled residual.
#pragma once

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float kF16Max = 65504.f;
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

}  // namespace
