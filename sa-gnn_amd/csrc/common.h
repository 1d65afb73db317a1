// Shared host-side helpers for libsagnn.so (error reporting, argument checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "sagnn.h"

namespace sagnn {

std::string& last_error_slot();
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int hip_fail(hipError_t e, const char* what);

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define SAGNN_HIP_TRY(expr)                                   \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) return ::sagnn::hip_fail(_e, #expr); \
  } while (0)

// Lanes per feature row: each lane owns one float4, so a row of d floats needs d/4 lanes,
// rounded up to a power of two that divides the 64-lane wavefront.
inline int lanes_per_row(int d) {
  int need = (d + 3) / 4;
  int lpr = 8;
  while (lpr < need) lpr <<= 1;
  return lpr;  // 8, 16, 32 or 64 for d <= 256
}

// Raises a kernel's dynamic-LDS limit to `bytes` on the CURRENT device if it is not already that
// high there. State is per (device, kernel) behind a mutex: the library may be driven from several
// threads and devices (include/sagnn.h: reentrant, no global mutable state visible to callers).
int ensure_dynamic_lds(const void* kernel, size_t bytes);
int cu_count_current();   // compute units of the current device (256 on MI355X), cached per device
// Device counter (one per device, allocated on first use) that the f16 x 2 kernels bump once per tile / chunk they
// re-evaluate in fp32 because an operand left the split's window (f16_split.h, RANGE): sagnn_range_redo_count reads it.
unsigned int* redo_counter();

// fusion_mfma.hip
bool lstm_mfma_supported(int d);
int lstm_fwd_mfma(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, const float* W,
                  const float* b, float forget_bias, const float* drop, float* h, int64_t ld_h,
                  float* gates_out, float* c_out, const float* h_init, int64_t ld_hi, const float* c_init,
                  float* c_final, hipStream_t s);
// lstm_f16.hip: the same LSTM on the f16 matrix cores, operands split in two round-to-nearest pieces (default engine)
bool lstm_f16_supported(int d);
int lstm_fwd_f16(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, const float* W, const float* b,
                 float forget_bias, const float* drop, float* h, int64_t ld_h, float* gates_out, float* c_out,
                 const float* h_init, int64_t ld_hi, const float* c_init, float* c_final, hipStream_t s);
// attn_split.hip: layer norm + Q|K|V + attention + mean with the products on the f16 matrix cores (split operands)
bool mhsa_split_supported(int d, int t, int heads);
int ln_mhsa_mean_split(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads,
                       const float* gamma, const float* beta, float eps, int apply_ln, const float* Wq,
                       const float* bq, const float* Wk, const float* bk, const float* Wv, const float* bv,
                       float* out, int64_t ld_out, hipStream_t s);
// lstm_split128.hip: d = 128, one launch per step, hidden slices across workgroups (drop: training form only)
bool lstm_split128_supported(int d);
int lstm_fwd_split128(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, const float* W, const float* b,
                      float forget_bias, const float* drop, float* h, int64_t ld_h, float* gates_out, float* c_out,
                      const float* h_init, int64_t ld_hi, const float* c_init, float* c_final, hipStream_t s);
bool force_f32_mfma();  // the calling thread chose SAGNN_ENGINE_F32 (sagnn_set_engine): the exact-fp32 MFMA kernels
bool mhsa_mfma_supported(int d, int t, int heads);
int ln_mhsa_mean_mfma(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads,
                      const float* gamma, const float* beta, float eps, int apply_ln, const float* Wq,
                      const float* bq, const float* Wk, const float* bk, const float* Wv,
                      const float* bv, float* out, int64_t ld_out, hipStream_t s);
// attn_bwd_tail_f16.hip: dy / dW / db of the three dense layers on the f16 engine (d = 32 / 64)
int attn_bwd_tail_f16(float* y, const float* dqkv, int64_t rows, int d, const float* Wqkv, float* dWqkv, float* dbqkv,
                      hipStream_t s);
// same kernel, LSTM form: dW [2d, 4d] += sum_s [x_s | h_{s-1}]^T dG_s from gate gradients stored time-major [t, n, 4d]
int lstm_dw_f16(const float* x, int64_t ld_n, int64_t ld_t, const float* h, const float* dg, int64_t n, int t, int d, float* dW,
                hipStream_t s);
bool attn_bwd_front_split_supported(int d, int t, int heads);   // attn_split.hip: the same on the f16 engine (t <= 6)
int attn_bwd_front_split(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads, const float* gamma,
                         const float* beta, float eps, int apply_ln, const float* Wq, const float* bq, const float* Wk,
                         const float* bk, const float* Wv, const float* bv, const float* g_out, int64_t ld_g, float* dqkv,
                         float* y, hipStream_t s);
bool attn_bwd_front_supported(int d, int t, int heads);
int attn_bwd_front_mfma(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads,
                        const float* gamma, const float* beta, float eps, int apply_ln, const float* Wq,
                        const float* bq, const float* Wk, const float* bk, const float* Wv, const float* bv,
                        const float* g_out, int64_t ld_g, float* dqkv, float* y, hipStream_t s);
// dense.hip: products of any size (multiples of 32), cut into LDS-fitting pieces
int dense_nn_any(const float* X, int64_t ldx, int64_t n, int din, int dout, const float* W, int64_t ldw,
                 const float* bias, float* Y, int64_t ldy, int accumulate, hipStream_t s);
int dense_tn_any(const float* X, int64_t ldx, const float* G, int64_t ldg, int64_t n, int din, int dout, float* dW,
                 int64_t lddw, float* db, hipStream_t s);
// dense_gemm.hip: tiled fp32-MFMA GEMMs for weight blocks that do not fit LDS (the d = 128 BPTT products)
int gemm_nn(const float* X, int64_t ldx, int64_t n, int din, int dout, const float* W, int64_t ldw, const float* bias, float* Y,
            int64_t ldy, int accumulate, hipStream_t s);
int gemm_tn(const float* X, int64_t ldx, int64_t xseg, const float* G, int64_t ldg, int64_t gseg, int64_t seg_rows, int64_t n_seg,
            int din, int dout, float* dW, int64_t lddw, float* db, hipStream_t s);
bool force_valu();  // the calling thread chose SAGNN_ENGINE_VALU

// ---- optional per-launch timing (sagnn_profile_*) -------------------------------------------
enum ProfileKind { kProfSpmmRows = 0, kProfSpmmFixup = 1, kProfLstm = 2, kProfLayerNorm = 3, kProfMhsa = 4 };

// Records a hipEvent pair around the launches issued during its lifetime, on the launch stream,
// when profiling is enabled; otherwise does nothing.
class ProfileScope {
 public:
  ProfileScope(int kind, hipStream_t stream, int64_t units_a, int64_t units_b);
  ~ProfileScope();
  ProfileScope(const ProfileScope&) = delete;
  ProfileScope& operator=(const ProfileScope&) = delete;

 private:
  int slot_;
  hipStream_t stream_;
};

}  // namespace sagnn
