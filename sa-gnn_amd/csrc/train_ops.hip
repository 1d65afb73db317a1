// Training-side operators around the hot path: the backward of the
// prediction head's pair scores, the SSL pair scores and meta-weights, the hinge losses.
// Reference: model.py:169-205 (forward graph), model.py:241-250 (losses); the gradients are what
// tf.gradients derives. Row gathers / scatter-adds move whole 16-byte-aligned feature rows; the
// scatter side uses float atomics on contiguous row segments (the shape the chip runs fastest).
#include "common.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float leakyf(float x, float a) { return fmaxf(a * x, x); }
// slope of tf.maximum(a*x, x): the gradient goes to the first argument on ties
__device__ __forceinline__ float slopef(float x, float a) { return (x > a * x) ? 1.f : a; }

struct PairGeom {
  int lpr;   // lanes per pair = d/4 (power of two)
  int ppw;   // pairs per wave
};

// ---- head pair score, backward (forward: fusion_bwd.hip pair_score_kernel) --------------------
// preds[e] = <U[u], I[i]> + <leaky(S[l]), A[i]>;  given g[e]:
//   dU[u] += g I[i];  dI[i] += g U[u];  dA[i] += g leaky(S[l]);  dS[l] += g slope(S[l]) A[i]
__global__ void pair_score_bwd_kernel(const float* __restrict__ U, int64_t ldu, const float* __restrict__ I,
                                      int64_t ldi, const float* __restrict__ S, int64_t lds_,
                                      const float* __restrict__ A, int64_t lda, const int32_t* __restrict__ uids,
                                      const int32_t* __restrict__ iids, const int32_t* __restrict__ locs,
                                      float leaky, const float* __restrict__ g, float* __restrict__ dU,
                                      float* __restrict__ dI, float* __restrict__ dS, float* __restrict__ dA,
                                      int64_t n_pairs, int d) {
  const int lpr = d >> 2, lane = threadIdx.x & 63, ppw = 64 / lpr;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t e = wave * ppw + lane / lpr;
  if (e >= n_pairs) return;
  const int col = (lane % lpr) * 4;
  const int64_t u = uids[e], it = iids[e];
  const float ge = g[e];
  const float4 a = *reinterpret_cast<const float4*>(U + u * ldu + col);
  const float4 b = *reinterpret_cast<const float4*>(I + it * ldi + col);
  float* du = dU + u * (int64_t)d + col;
  float* di = dI + it * (int64_t)d + col;
  atomicAdd(du + 0, ge * b.x);
  atomicAdd(du + 1, ge * b.y);
  atomicAdd(du + 2, ge * b.z);
  atomicAdd(du + 3, ge * b.w);
  atomicAdd(di + 0, ge * a.x);
  atomicAdd(di + 1, ge * a.y);
  atomicAdd(di + 2, ge * a.z);
  atomicAdd(di + 3, ge * a.w);
  if (S) {
    const int64_t l = locs[e];
    const float4 s = *reinterpret_cast<const float4*>(S + l * lds_ + col);
    const float4 c = *reinterpret_cast<const float4*>(A + it * lda + col);
    float* da = dA + it * (int64_t)d + col;
    float* dsp = dS + l * (int64_t)d + col;
    atomicAdd(da + 0, ge * leakyf(s.x, leaky));
    atomicAdd(da + 1, ge * leakyf(s.y, leaky));
    atomicAdd(da + 2, ge * leakyf(s.z, leaky));
    atomicAdd(da + 3, ge * leakyf(s.w, leaky));
    atomicAdd(dsp + 0, ge * slopef(s.x, leaky) * c.x);
    atomicAdd(dsp + 1, ge * slopef(s.y, leaky) * c.y);
    atomicAdd(dsp + 2, ge * slopef(s.z, leaky) * c.z);
    atomicAdd(dsp + 3, ge * slopef(s.w, leaky) * c.w);
  }
}

// ---- SSL pair score: s[e] = sum_j leaky(X[u][j] * Y[i][j])   (model.py:191, :199) -------------
__global__ void prod_leaky_sum_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ Y,
                                      int64_t ldy, const int32_t* __restrict__ uids,
                                      const int32_t* __restrict__ iids, float leaky, float* __restrict__ out,
                                      int64_t n_pairs, int d) {
  const int lpr = d >> 2, lane = threadIdx.x & 63, ppw = 64 / lpr;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t e = wave * ppw + lane / lpr;
  const int col = (lane % lpr) * 4;
  float acc = 0.f;
  if (e < n_pairs) {
    const float4 a = *reinterpret_cast<const float4*>(X + (int64_t)uids[e] * ldx + col);
    const float4 b = *reinterpret_cast<const float4*>(Y + (int64_t)iids[e] * ldy + col);
    acc = leakyf(a.x * b.x, leaky) + leakyf(a.y * b.y, leaky) + leakyf(a.z * b.z, leaky) + leakyf(a.w * b.w, leaky);
  }
  for (int off = 1; off < lpr; off <<= 1) acc += __shfl_xor(acc, off);
  if (e < n_pairs && (lane % lpr) == 0) out[e] = acc;
}

// dX[u] += g slope(ab) b;  dY[i] += g slope(ab) a
__global__ void prod_leaky_sum_bwd_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ Y,
                                          int64_t ldy, const int32_t* __restrict__ uids,
                                          const int32_t* __restrict__ iids, float leaky,
                                          const float* __restrict__ g, float* __restrict__ dX,
                                          float* __restrict__ dY, int64_t n_pairs, int d) {
  const int lpr = d >> 2, lane = threadIdx.x & 63, ppw = 64 / lpr;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t e = wave * ppw + lane / lpr;
  if (e >= n_pairs) return;
  const int col = (lane % lpr) * 4;
  const int64_t u = uids[e], it = iids[e];
  const float ge = g[e];
  const float4 a = *reinterpret_cast<const float4*>(X + u * ldx + col);
  const float4 b = *reinterpret_cast<const float4*>(Y + it * ldy + col);
  float* dx = dX + u * (int64_t)d + col;
  float* dy = dY + it * (int64_t)d + col;
  const float sx = ge * slopef(a.x * b.x, leaky), sy = ge * slopef(a.y * b.y, leaky);
  const float sz = ge * slopef(a.z * b.z, leaky), sw = ge * slopef(a.w * b.w, leaky);
  atomicAdd(dx + 0, sx * b.x);
  atomicAdd(dx + 1, sy * b.y);
  atomicAdd(dx + 2, sz * b.z);
  atomicAdd(dx + 3, sw * b.w);
  atomicAdd(dy + 0, sx * a.x);
  atomicAdd(dy + 1, sy * a.y);
  atomicAdd(dy + 2, sz * a.z);
  atomicAdd(dy + 3, sw * a.w);
}

// ---- meta-net input: m[e] = [F[u]*V[u] | F[u] | V[u]]   (model.py:179) -----------------------
__global__ void meta_features_kernel(const float* __restrict__ F, int64_t ldf, const float* __restrict__ V,
                                     int64_t ldv, const int32_t* __restrict__ uids, float* __restrict__ out,
                                     int64_t n, int d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int d4 = d >> 2;
  if (i >= n * d4) return;
  const int64_t e = i / d4;
  const int col = (int)(i - e * d4) * 4;
  const int64_t u = uids[e];
  const float4 f = *reinterpret_cast<const float4*>(F + u * ldf + col);
  const float4 v = *reinterpret_cast<const float4*>(V + u * ldv + col);
  float* o = out + e * (int64_t)(3 * d) + col;
  *reinterpret_cast<float4*>(o) = make_float4(f.x * v.x, f.y * v.y, f.z * v.z, f.w * v.w);
  *reinterpret_cast<float4*>(o + d) = f;
  *reinterpret_cast<float4*>(o + 2 * d) = v;
}

// dF[u] += dm0*V + dm1 ; dV[u] += dm0*F + dm2
__global__ void meta_features_bwd_kernel(const float* __restrict__ F, int64_t ldf, const float* __restrict__ V,
                                         int64_t ldv, const int32_t* __restrict__ uids,
                                         const float* __restrict__ dm, float* __restrict__ dF,
                                         float* __restrict__ dV, int64_t n, int d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int d4 = d >> 2;
  if (i >= n * d4) return;
  const int64_t e = i / d4;
  const int col = (int)(i - e * d4) * 4;
  const int64_t u = uids[e];
  const float4 f = *reinterpret_cast<const float4*>(F + u * ldf + col);
  const float4 v = *reinterpret_cast<const float4*>(V + u * ldv + col);
  const float* g = dm + e * (int64_t)(3 * d) + col;
  const float4 g0 = *reinterpret_cast<const float4*>(g);
  const float4 g1 = *reinterpret_cast<const float4*>(g + d);
  const float4 g2 = *reinterpret_cast<const float4*>(g + 2 * d);
  float* df = dF + u * (int64_t)d + col;
  float* dv = dV + u * (int64_t)d + col;
  atomicAdd(df + 0, g0.x * v.x + g1.x);
  atomicAdd(df + 1, g0.y * v.y + g1.y);
  atomicAdd(df + 2, g0.z * v.z + g1.z);
  atomicAdd(df + 3, g0.w * v.w + g1.w);
  atomicAdd(dv + 0, g0.x * f.x + g2.x);
  atomicAdd(dv + 1, g0.y * f.y + g2.y);
  atomicAdd(dv + 2, g0.z * f.z + g2.z);
  atomicAdd(dv + 3, g0.w * f.w + g2.w);
}

// ---- element-wise pieces ------------------------------------------------------------------------
// mode 0: out = max(leaky*a, a); mode 1: out = g * slope(a)   (a = pre-activation)
__global__ void leaky_kernel(const float* __restrict__ a, const float* __restrict__ g, float* __restrict__ out,
                             float leaky, int64_t count, int mode) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  out[i] = mode == 0 ? leakyf(a[i], leaky) : g[i] * slopef(a[i], leaky);
}

// w[e] = sigmoid(<A[e, :k], w3> + b3)   (FC(meta2, 1, sigmoid), model.py:182). One thread per row.
__global__ void rowdot_sigmoid_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ w3,
                                      const float* __restrict__ b3, float* __restrict__ out, int64_t n, int k) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float z = b3[0];
  for (int j = 0; j < k; ++j) z = fmaf(A[e * lda + j], w3[j], z);
  out[e] = 1.f / (1.f + expf(-z));
}

// given dw[e]: dz = dw w (1-w); dA[e, :] = dz w3; dw3 += sum_e dz A[e, :]; db3 += sum_e dz
__global__ void rowdot_sigmoid_bwd_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ w3,
                                          const float* __restrict__ w, const float* __restrict__ dw,
                                          float* __restrict__ dA, int64_t ldda, float* __restrict__ dw3,
                                          float* __restrict__ db3, int64_t n, int k) {
  extern __shared__ float red[];  // [k + 1] block partials
  for (int j = threadIdx.x; j <= k; j += blockDim.x) red[j] = 0.f;
  __syncthreads();
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) {
    const float wv = w[e];
    const float dz = dw[e] * wv * (1.f - wv);
    for (int j = 0; j < k; ++j) {
      dA[e * ldda + j] = dz * w3[j];
      atomicAdd(&red[j], dz * A[e * lda + j]);
    }
    atomicAdd(&red[k], dz);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < k; j += blockDim.x) atomicAdd(dw3 + j, red[j]);
  if (threadIdx.x == 0) atomicAdd(db3, red[k]);
}

// ---- hinge losses (model.py:202, :244). pos/neg are the two halves of one score vector ----------
//   plain (wp == NULL): loss += scale * max(0, 1 - (pos - neg));         dpos = -scale, dneg = +scale
//   weighted: S = wp*sp - wn*sn (sp, sn constants), loss += scale * max(0, 1 - S*(pos - neg));
//             dpos = -scale*S, dneg = +scale*S, dwp = -scale*(pos-neg)*sp, dwn = +scale*(pos-neg)*sn
__global__ void hinge_kernel(const float* __restrict__ pos, const float* __restrict__ neg,
                             const float* __restrict__ wp, const float* __restrict__ wn,
                             const float* __restrict__ sp, const float* __restrict__ sn, float scale,
                             float* __restrict__ loss, float* __restrict__ dpos, float* __restrict__ dneg,
                             float* __restrict__ dwp, float* __restrict__ dwn, int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float term = 0.f;
  if (e < n) {
    const float delta = pos[e] - neg[e];
    const float S = wp ? wp[e] * sp[e] - wn[e] * sn[e] : 1.f;
    const float h = 1.f - S * delta;
    const bool active = h > 0.f;
    term = active ? scale * h : 0.f;
    const float k = active ? scale : 0.f;
    if (dpos) {
      dpos[e] = -k * S;
      dneg[e] = k * S;
    }
    if (wp && dwp) {
      dwp[e] = -k * delta * sp[e];
      dwn[e] = k * delta * sn[e];
    }
  }
  for (int off = 32; off > 0; off >>= 1) term += __shfl_xor(term, off);
  if ((threadIdx.x & 63) == 0 && term != 0.f) atomicAdd(loss, term);
}

int64_t pair_blocks(int64_t n_pairs, int d) {
  const int ppw = 64 / (d / 4);
  const int64_t waves = (n_pairs + ppw - 1) / ppw;
  return (waves + 3) / 4;
}

int check_pair_dims(int d) {
  const int lpr = d / 4;
  if (d < 4 || d > 256 || (d & 3) || (lpr & (lpr - 1)))
    return sagnn::fail(SAGNN_ERR_DIM, "d = %d: need 4 * a power of two, <= 256", d);
  return SAGNN_OK;
}

}  // namespace

extern "C" int sagnn_pair_score_bwd_f32(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* S,
                                        int64_t lds, const float* A, int64_t lda, const int32_t* uids,
                                        const int32_t* iids, const int32_t* locs, float leaky, const float* g,
                                        float* dU, float* dI, float* dS, float* dA, int64_t n_pairs, int d,
                                        void* stream) {
  if (int rc = check_pair_dims(d)) return rc;
  if (!U || !I || !uids || !iids || !g || !dU || !dI) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (S && (!A || !locs || !dS || !dA)) return sagnn::fail(SAGNN_ERR_NULL, "S needs A, locs, dS, dA");
  if (n_pairs <= 0) return SAGNN_OK;
  hipLaunchKernelGGL(pair_score_bwd_kernel, dim3((unsigned)pair_blocks(n_pairs, d)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), U, ldu, I, ldi, S, lds, A, lda, uids, iids, locs, leaky, g, dU,
                     dI, dS, dA, n_pairs, d);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_prod_leaky_sum_f32(const float* X, int64_t ldx, const float* Y, int64_t ldy,
                                        const int32_t* uids, const int32_t* iids, float leaky, float* out,
                                        int64_t n_pairs, int d, void* stream) {
  if (int rc = check_pair_dims(d)) return rc;
  if (!X || !Y || !uids || !iids || !out) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (n_pairs <= 0) return SAGNN_OK;
  hipLaunchKernelGGL(prod_leaky_sum_kernel, dim3((unsigned)pair_blocks(n_pairs, d)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), X, ldx, Y, ldy, uids, iids, leaky, out, n_pairs, d);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_prod_leaky_sum_bwd_f32(const float* X, int64_t ldx, const float* Y, int64_t ldy,
                                            const int32_t* uids, const int32_t* iids, float leaky, const float* g,
                                            float* dX, float* dY, int64_t n_pairs, int d, void* stream) {
  if (int rc = check_pair_dims(d)) return rc;
  if (!X || !Y || !uids || !iids || !g || !dX || !dY) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (n_pairs <= 0) return SAGNN_OK;
  hipLaunchKernelGGL(prod_leaky_sum_bwd_kernel, dim3((unsigned)pair_blocks(n_pairs, d)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), X, ldx, Y, ldy, uids, iids, leaky, g, dX, dY, n_pairs, d);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_meta_features_f32(const float* F, int64_t ldf, const float* V, int64_t ldv, const int32_t* uids,
                                       float* out, int64_t n, int d, void* stream) {
  if (d < 4 || (d & 3)) return sagnn::fail(SAGNN_ERR_DIM, "d = %d", d);
  if (!F || !V || !uids || !out) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (n <= 0) return SAGNN_OK;
  const int64_t total = n * (d / 4);
  hipLaunchKernelGGL(meta_features_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), F, ldf, V, ldv, uids, out, n, d);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_meta_features_bwd_f32(const float* F, int64_t ldf, const float* V, int64_t ldv,
                                           const int32_t* uids, const float* dm, float* dF, float* dV, int64_t n,
                                           int d, void* stream) {
  if (d < 4 || (d & 3)) return sagnn::fail(SAGNN_ERR_DIM, "d = %d", d);
  if (!F || !V || !uids || !dm || !dF || !dV) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (n <= 0) return SAGNN_OK;
  const int64_t total = n * (d / 4);
  hipLaunchKernelGGL(meta_features_bwd_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), F, ldf, V, ldv, uids, dm, dF, dV, n, d);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_leaky_f32(const float* a, const float* g, float* out, float leaky, int64_t count, int backward,
                               void* stream) {
  if (!a || !out || (backward && !g)) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (count <= 0) return SAGNN_OK;
  hipLaunchKernelGGL(leaky_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), a, g, out, leaky, count, backward ? 1 : 0);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_rowdot_sigmoid_f32(const float* A, int64_t lda, const float* w3, const float* b3, float* out,
                                        int64_t n, int k, void* stream) {
  if (!A || !w3 || !b3 || !out) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (n <= 0) return SAGNN_OK;
  hipLaunchKernelGGL(rowdot_sigmoid_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), A, lda, w3, b3, out, n, k);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_rowdot_sigmoid_bwd_f32(const float* A, int64_t lda, const float* w3, const float* w,
                                            const float* dw, float* dA, int64_t ldda, float* dw3, float* db3,
                                            int64_t n, int k, void* stream) {
  if (!A || !w3 || !w || !dw || !dA || !dw3 || !db3) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (n <= 0) return SAGNN_OK;
  hipLaunchKernelGGL(rowdot_sigmoid_bwd_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock),
                     (size_t)(k + 1) * sizeof(float), static_cast<hipStream_t>(stream), A, lda, w3, w, dw, dA, ldda,
                     dw3, db3, n, k);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_hinge_f32(const float* pos, const float* neg, const float* wp, const float* wn, const float* sp,
                               const float* sn, float scale, float* loss, float* dpos, float* dneg, float* dwp,
                               float* dwn, int64_t n, void* stream) {
  if (!pos || !neg || !loss) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (wp && (!wn || !sp || !sn)) return sagnn::fail(SAGNN_ERR_NULL, "weighted hinge needs wn, sp, sn");
  if (n <= 0) return SAGNN_OK;
  hipLaunchKernelGGL(hinge_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), pos, neg, wp, wn, sp, sn, scale, loss, dpos, dneg, dwp, dwn, n);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}
