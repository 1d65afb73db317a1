// Interval fusion for gfx950 (reference model.py:135-155): BasicLSTMCell over the T interval
// embeddings of every node, layer-norm over (T, d), multi-head self-attention over T with the
// reference's exp/(sum + 1e-8) normalisation, mean over T.
//
// This file holds the VALU (fp32 FMA) formulation: one thread per (node, hidden unit), weights
// streamed through L2, activations staged in LDS. fusion_mfma.hip holds the MFMA formulation
// of the two GEMM-shaped stages; both are checked against the same oracle.
#include "common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kLstmRowsPerThread = 4;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// TF 1.14 BasicLSTMCell.call (rnn_cell_impl.py): gate_inputs = [x, h] @ W + b;
// i, j, f, o = split(gate_inputs, 4); c' = c*sigmoid(f + forget_bias) + sigmoid(i)*tanh(j);
// h' = tanh(c')*sigmoid(o).
__global__ void lstm_fwd_valu_kernel(const float* __restrict__ x, int64_t ld_n, int64_t ld_t,
                                     int64_t n, int t, int d, const float* __restrict__ W,
                                     const float* __restrict__ b, float forget_bias,
                                     const float* __restrict__ drop, float* __restrict__ h_out,
                                     int64_t ld_h, float* __restrict__ gates_out,
                                     float* __restrict__ c_out, const float* __restrict__ h_init, int64_t ld_hi,
                                     const float* __restrict__ c_init, float* __restrict__ c_final) {
  extern __shared__ float sm[];  // [rows_per_block][2d]: x_t | h
  constexpr int RPT = kLstmRowsPerThread;
  const int j = threadIdx.x % d;
  const int rs = threadIdx.x / d;
  const int slots = blockDim.x / d;
  const int rows_pb = slots * RPT;
  const int64_t row_base = (int64_t)blockIdx.x * rows_pb;
  const int d2 = 2 * d, d4 = 4 * d;

  float c[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int64_t row = row_base + rs * RPT + r;
    c[r] = (c_init && row < n) ? c_init[row * d + j] : 0.f;   // state to continue from (NULL = zero)
  }
  for (int i = threadIdx.x; i < rows_pb * d; i += blockDim.x) {
    const int64_t row = row_base + i / d;
    sm[(i / d) * d2 + d + (i % d)] = (h_init && row < n) ? h_init[row * ld_hi + (i % d)] : 0.f;
  }

  const float b_i = b[j], b_j = b[d + j], b_f = b[2 * d + j], b_o = b[3 * d + j];
  for (int ts = 0; ts < t; ++ts) {
    for (int i = threadIdx.x; i < rows_pb * d; i += blockDim.x) {
      const int r = i / d, k = i % d;
      const int64_t row = row_base + r;
      sm[r * d2 + k] = row < n ? x[row * ld_n + (int64_t)ts * ld_t + k] : 0.f;
    }
    __syncthreads();
    float gi[RPT], gj[RPT], gf[RPT], go[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      gi[r] = 0.f;
      gj[r] = 0.f;
      gf[r] = 0.f;
      go[r] = 0.f;
    }
    for (int k = 0; k < d2; ++k) {
      const float* wr = W + (int64_t)k * d4 + j;
      const float wi = wr[0], wj = wr[d], wf = wr[2 * d], wo = wr[3 * d];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const float a = sm[(rs * RPT + r) * d2 + k];
        gi[r] = fmaf(a, wi, gi[r]);
        gj[r] = fmaf(a, wj, gj[r]);
        gf[r] = fmaf(a, wf, gf[r]);
        go[r] = fmaf(a, wo, go[r]);
      }
    }
    __syncthreads();  // every thread has finished reading x_t | h
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int lr = rs * RPT + r;
      const int64_t row = row_base + lr;
      const float si = sigmoidf_(gi[r] + b_i), tj = tanhf(gj[r] + b_j);
      const float sf = sigmoidf_(gf[r] + b_f + forget_bias), so = sigmoidf_(go[r] + b_o);
      const float cn = c[r] * sf + si * tj;
      const float hn = tanhf(cn) * so;
      c[r] = cn;
      sm[lr * d2 + d + j] = hn;
      if (gates_out && row < n) {  // training forward: activations + cell state for the backward pass
        float* gp = gates_out + (row * t + ts) * (int64_t)d4 + j;
        gp[0] = si;
        gp[d] = tj;
        gp[2 * d] = sf;
        gp[3 * d] = so;
        c_out[(row * t + ts) * (int64_t)d + j] = cn;
      }
      if (row < n) {
        const int64_t o = row * ld_h + (int64_t)ts * d + j;
        h_out[o] = drop ? hn * drop[row * (int64_t)t * d + (int64_t)ts * d + j] : hn;
      }
    }
  }
  if (c_final) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int64_t row = row_base + rs * RPT + r;
      if (row < n) c_final[row * d + j] = c[r];
    }
  }
}

// tf.contrib.layers.layer_norm(begin_norm_axis=1, begin_params_axis=-1): moments over (t, d),
// then tf.nn.batch_normalization: inv = rsqrt(var + eps) * gamma; y = x*inv + (beta - mean*inv).
// One wavefront per node.
// x and y may alias (in-place): each element is read and written by the same lane.
__global__ void layernorm_td_kernel(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d,
                                    const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float eps, float* y,
                                    int64_t ld_y) {
  const int lane = threadIdx.x & 63;
  const int64_t node = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (node >= n) return;
  const int td = t * d;
  const float* xr = x + node * ld_n;
  // element i = (ts, k) of the node sits at ts*ld_t + k in x and at i in the dense output
  float s = 0.f;
  for (int i = lane; i < td; i += 64) s += xr[(int64_t)(i / d) * ld_t + (i % d)];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)td;
  float v = 0.f;
  for (int i = lane; i < td; i += 64) {
    const float dlt = xr[(int64_t)(i / d) * ld_t + (i % d)] - mean;
    v = fmaf(dlt, dlt, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const float rstd = rsqrtf(v / (float)td + eps);
  float* yr = y + node * ld_y;
  for (int i = lane; i < td; i += 64) {
    const int k = i % d;
    const float inv = rstd * gamma[k];
    yr[i] = xr[(int64_t)(i / d) * ld_t + k] * inv + (beta[k] - mean * inv);
  }
}

// MultiHeadSelfAttention.attention + reduce_mean over the query axis. d threads per node.
__global__ void mhsa_mean_valu_kernel(const float* __restrict__ x, int64_t ld_n, int64_t ld_t,
                                      int64_t n, int t, int d, int heads,
                                      const float* __restrict__ Wq,
                                      const float* __restrict__ bq, const float* __restrict__ Wk,
                                      const float* __restrict__ bk, const float* __restrict__ Wv,
                                      const float* __restrict__ bv, float* __restrict__ out,
                                      int64_t ld_out) {
  extern __shared__ float sm[];  // per slot: y | q | k | v, each [t][d]
  const int j = threadIdx.x % d;
  const int slot = threadIdx.x / d;
  const int slots = blockDim.x / d;
  const int td = t * d;
  float* ys = sm + (size_t)slot * 4 * td;
  float* qs = ys + td;
  float* ks = qs + td;
  float* vs = ks + td;
  const int dk = d / heads;
  const int hoff = (j / dk) * dk;
  const float scale = 1.f / sqrtf((float)dk);

  for (int64_t node0 = (int64_t)blockIdx.x * slots; node0 < n; node0 += (int64_t)gridDim.x * slots) {
    const int64_t node = node0 + slot;
    const bool valid = node < n;
    for (int ts = 0; ts < t; ++ts) ys[ts * d + j] = valid ? x[node * ld_n + (int64_t)ts * ld_t + j] : 0.f;
    __syncthreads();
    for (int ts = 0; ts < t; ++ts) {
      float q = 0.f, k = 0.f, v = 0.f;
      for (int kk = 0; kk < d; ++kk) {
        const float a = ys[ts * d + kk];
        q = fmaf(a, Wq[kk * d + j], q);
        k = fmaf(a, Wk[kk * d + j], k);
        v = fmaf(a, Wv[kk * d + j], v);
      }
      qs[ts * d + j] = q + bq[j];
      ks[ts * d + j] = k + bk[j];
      vs[ts * d + j] = v + bv[j];
    }
    __syncthreads();
    float o = 0.f;
    for (int tq = 0; tq < t; ++tq) {
      float rowsum = 0.f, ctx = 0.f;
      for (int s = 0; s < t; ++s) {
        float dot = 0.f;
        for (int c = 0; c < dk; ++c) dot = fmaf(qs[tq * d + hoff + c], ks[s * d + hoff + c], dot);
        const float e = expf(dot * scale);
        rowsum += e;
        ctx = fmaf(e, vs[s * d + j], ctx);
      }
      o += ctx / (rowsum + 1e-8f);
    }
    if (valid) out[node * ld_out + j] = o / (float)t;
    __syncthreads();
  }
}


// ---- "wide" attention: any d that is a multiple of 32 (e.g. 128): Q|K|V by the MFMA dense products
// of dense.hip into a [n, t, 3d] buffer, then this per-node kernel. (The LSTM keeps its VALU form
// for such d: an un-fused MFMA composition measured no faster, 37.5 vs 37.2 ms at n = 1M, d = 128,
// T = 6 — the gate round trip through HBM eats the MFMA gain.) ---------------------------------------

// Attention + mean over queries from a Q|K|V buffer [n, t, 3d] (rows as x@W+b produced them).
__global__ void attn_mean_from_qkv_kernel(const float* __restrict__ qkv, int64_t n, int t, int d, int heads,
                                          float* __restrict__ out, int64_t ld_out) {
  extern __shared__ float sm[];
  const int j = threadIdx.x % d;
  const int slot = threadIdx.x / d;
  const int slots = blockDim.x / d;
  const int td = t * d;
  float* qs = sm + (size_t)slot * 3 * td;
  float* ks = qs + td;
  float* vs = ks + td;
  const int dk = d / heads;
  const int hoff = (j / dk) * dk;
  const float scale = 1.f / sqrtf((float)dk);
  for (int64_t node0 = (int64_t)blockIdx.x * slots; node0 < n; node0 += (int64_t)gridDim.x * slots) {
    const int64_t node = node0 + slot;
    const bool valid = node < n;
    const float* row = qkv + (valid ? node : 0) * (int64_t)(3 * td);
    for (int ts = 0; ts < t; ++ts) {
      qs[ts * d + j] = row[ts * 3 * d + j];
      ks[ts * d + j] = row[ts * 3 * d + d + j];
      vs[ts * d + j] = row[ts * 3 * d + 2 * d + j];
    }
    __syncthreads();
    float o = 0.f;
    for (int tq = 0; tq < t; ++tq) {
      float rowsum = 0.f, ctx = 0.f;
      for (int s2 = 0; s2 < t; ++s2) {
        float dot = 0.f;
        for (int c = 0; c < dk; ++c) dot = fmaf(qs[tq * d + hoff + c], ks[s2 * d + hoff + c], dot);
        const float e = expf(dot * scale);
        rowsum += e;
        ctx = fmaf(e, vs[s2 * d + j], ctx);
      }
      o += ctx / (rowsum + 1e-8f);
    }
    if (valid) out[node * ld_out + j] = o / (float)t;
    __syncthreads();
  }
}

int check_dims(int64_t n, int t, int d) {
  if (n < 0) return sagnn::fail(SAGNN_ERR_ARG, "n = %lld < 0", (long long)n);
  if (t < 1 || t > 64) return sagnn::fail(SAGNN_ERR_DIM, "t = %d: need 1..64", t);
  if (d < 4 || d > 256 || (d & 3)) return sagnn::fail(SAGNN_ERR_DIM, "d = %d: need a multiple of 4 in [4, 256]", d);
  return SAGNN_OK;
}

// x[node, ts, :] lives at node*ld_n + ts*ld_t: either [n, t, d] (ld_t = d, ld_n >= t*d) or
// [t, n, d] (ld_n = d, ld_t >= n*d) or any other non-overlapping pair of strides.
int check_strides(int64_t ld_n, int64_t ld_t, int64_t n, int t, int d) {
  if (ld_n < d || ld_t < d) return sagnn::fail(SAGNN_ERR_ARG, "ld_n / ld_t smaller than d");
  const bool node_major = ld_n >= (int64_t)(t - 1) * ld_t + d;
  const bool time_major = ld_t >= (n - 1) * ld_n + d;
  if (!node_major && !time_major && n > 1 && t > 1)
    return sagnn::fail(SAGNN_ERR_ARG, "ld_n = %lld, ld_t = %lld overlap for n = %lld, t = %d",
                       (long long)ld_n, (long long)ld_t, (long long)n, t);
  return SAGNN_OK;
}

}  // namespace

namespace sagnn {

int lstm_fwd_valu(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, const float* W,
                  const float* b, float forget_bias, const float* drop, float* h, int64_t ld_h,
                  float* gates_out, float* c_out, const float* h_init, int64_t ld_hi, const float* c_init,
                  float* c_final, hipStream_t s) {
  const int slots = kBlock / d > 0 ? kBlock / d : 1;
  const int threads = slots * d;
  const int rows_pb = slots * kLstmRowsPerThread;
  const int64_t blocks = (n + rows_pb - 1) / rows_pb;
  if (blocks > INT32_MAX) return fail(SAGNN_ERR_ARG, "grid too large");
  const size_t lds = (size_t)rows_pb * 2 * d * sizeof(float);
  ProfileScope prof(kProfLstm, s, n, t);
  hipLaunchKernelGGL(lstm_fwd_valu_kernel, dim3((unsigned)blocks), dim3(threads), lds, s, x, ld_n, ld_t,
                     n, t, d, W, b, forget_bias, drop, h, ld_h, gates_out, c_out, h_init, ld_hi, c_init, c_final);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

int mhsa_mean_valu(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads,
                   const float* Wq, const float* bq, const float* Wk, const float* bk,
                   const float* Wv, const float* bv, float* out, int64_t ld_out, hipStream_t s) {
  int slots = kBlock / d > 0 ? kBlock / d : 1;
  while (slots > 1 && (size_t)slots * 4 * t * d * sizeof(float) > 64 * 1024) slots >>= 1;
  const size_t lds = (size_t)slots * 4 * t * d * sizeof(float);
  if (lds > 160 * 1024) return fail(SAGNN_ERR_DIM, "t*d = %d too large for LDS", t * d);
  int64_t blocks = (n + slots - 1) / slots;
  if (blocks > 8192) blocks = 8192;
  ProfileScope prof(kProfMhsa, s, n, t);
  hipLaunchKernelGGL(mhsa_mean_valu_kernel, dim3((unsigned)blocks), dim3(slots * d), lds, s, x, ld_n,
                     ld_t, n, t, d, heads, Wq, bq, Wk, bk, Wv, bv, out, ld_out);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

}  // namespace sagnn


namespace sagnn {

bool wide_supported(int d) { return d % 32 == 0 && d >= 32 && d <= 256; }

// MHSA + mean for any d % 32 == 0: Q|K|V by three MFMA products per interval into ws [n, t, 3d],
// then the per-node attention kernel.
int mhsa_mean_wide(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads, const float* Wq,
                   const float* bq, const float* Wk, const float* bk, const float* Wv, const float* bv, float* out,
                   int64_t ld_out, float* ws, hipStream_t s) {
  ProfileScope prof(kProfMhsa, s, n, t);
  const int64_t ldq = (int64_t)t * 3 * d;
  for (int ts = 0; ts < t; ++ts) {
    const float* xt = x + (int64_t)ts * ld_t;
    float* qt = ws + (int64_t)ts * 3 * d;
    if (int rc = dense_nn_any(xt, ld_n, n, d, d, Wq, d, bq, qt, ldq, 0, s)) return rc;
    if (int rc = dense_nn_any(xt, ld_n, n, d, d, Wk, d, bk, qt + d, ldq, 0, s)) return rc;
    if (int rc = dense_nn_any(xt, ld_n, n, d, d, Wv, d, bv, qt + 2 * d, ldq, 0, s)) return rc;
  }
  int slots = kBlock / d > 0 ? kBlock / d : 1;
  while (slots > 1 && (size_t)slots * 3 * t * d * sizeof(float) > 64 * 1024) slots >>= 1;
  const size_t lds = (size_t)slots * 3 * t * d * sizeof(float);
  if (lds > 160 * 1024) return fail(SAGNN_ERR_DIM, "t*d = %d too large for LDS", t * d);
  if (int rc = sagnn::ensure_dynamic_lds(reinterpret_cast<const void*>(&attn_mean_from_qkv_kernel), lds)) return rc;
  int64_t blocks = (n + slots - 1) / slots;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(attn_mean_from_qkv_kernel, dim3((unsigned)blocks), dim3(slots * d), lds, s, ws, n, t, d, heads,
                     out, ld_out);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

}  // namespace sagnn

extern "C" size_t sagnn_mhsa_wide_workspace_bytes(int64_t n, int t, int d) {
  return (n <= 0 || d <= 0 || t <= 0) ? 0 : (size_t)n * (size_t)t * 3 * (size_t)d * sizeof(float);
}

extern "C" int sagnn_mhsa_mean_wide_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d,
                                        int heads, const float* Wq, const float* bq, const float* Wk,
                                        const float* bk, const float* Wv, const float* bv, float* out,
                                        int64_t ld_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_dims(n, t, d)) return rc;
  if (!sagnn::wide_supported(d)) return sagnn::fail(SAGNN_ERR_DIM, "d = %d: the wide path needs a multiple of 32", d);
  if (heads < 1 || d % heads) return sagnn::fail(SAGNN_ERR_DIM, "heads = %d does not divide d = %d", heads, d);
  if (!x || !Wq || !bq || !Wk || !bk || !Wv || !bv || !out) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (int rc = check_strides(ld_n, ld_t, n, t, d)) return rc;
  if ((ld_n & 3) || (ld_t & 3) || !sagnn::aligned16(x)) return sagnn::fail(SAGNN_ERR_ALIGN, "x rows must be 16-byte aligned");
  if (n == 0) return SAGNN_OK;
  if (!workspace || workspace_bytes < sagnn_mhsa_wide_workspace_bytes(n, t, d))
    return sagnn::fail(SAGNN_ERR_WORKSPACE, "mhsa wide workspace needs %zu bytes", sagnn_mhsa_wide_workspace_bytes(n, t, d));
  return sagnn::mhsa_mean_wide(x, ld_n, ld_t, n, t, d, heads, Wq, bq, Wk, bk, Wv, bv, out, ld_out,
                               static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
}

extern "C" int sagnn_lstm_fwd_state_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d,
                                        const float* W, const float* b, float forget_bias,
                                        const float* drop_scale, const float* h_init, int64_t ld_hi,
                                        const float* c_init, float* h, int64_t ld_h, float* c_final, void* stream) {
  if (int rc = check_dims(n, t, d)) return rc;
  if (!x || !W || !b || !h) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if ((h_init == nullptr) != (c_init == nullptr)) return sagnn::fail(SAGNN_ERR_NULL, "give both h_init and c_init or neither");
  if (int rc = check_strides(ld_n, ld_t, n, t, d)) return rc;
  if (ld_h < (int64_t)t * d) return sagnn::fail(SAGNN_ERR_ARG, "ld_h smaller than t*d");
  if (h_init && ld_hi < d) return sagnn::fail(SAGNN_ERR_ARG, "ld_hi smaller than d");
  if (n == 0) return SAGNN_OK;
  // matrix-core path: d = 32 / 64 with 16-byte aligned rows; SAGNN_FUSION=valu forces the
  // VALU formulation (A/B runs)
  const bool vec_ok = sagnn::aligned16(x) && (ld_n & 3) == 0 && (ld_t & 3) == 0 &&
                      (!h_init || (sagnn::aligned16(h_init) && (ld_hi & 3) == 0));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (sagnn::lstm_mfma_supported(d) && vec_ok && !sagnn::force_valu())
    return sagnn::lstm_fwd_mfma(x, ld_n, ld_t, n, t, d, W, b, forget_bias, drop_scale, h, ld_h, nullptr, nullptr, h_init,
                                ld_hi, c_init, c_final, s);
  if (sagnn::lstm_split128_supported(d) && vec_ok && !drop_scale && (ld_h & 3) == 0 && sagnn::aligned16(h) &&
      !sagnn::force_valu() && !sagnn::force_f32_mfma())
    return sagnn::lstm_fwd_split128(x, ld_n, ld_t, n, t, W, b, forget_bias, nullptr, h, ld_h, nullptr, nullptr, h_init, ld_hi,
                                    c_init, c_final, s);
  return sagnn::lstm_fwd_valu(x, ld_n, ld_t, n, t, d, W, b, forget_bias, drop_scale, h, ld_h, nullptr, nullptr, h_init,
                              ld_hi, c_init, c_final, s);
}

extern "C" int sagnn_lstm_fwd_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d,
                                  const float* W, const float* b, float forget_bias,
                                  const float* drop_scale, float* h, int64_t ld_h, void* stream) {
  return sagnn_lstm_fwd_state_f32(x, ld_n, ld_t, n, t, d, W, b, forget_bias, drop_scale, nullptr, 0, nullptr, h, ld_h,
                                  nullptr, stream);
}

extern "C" int sagnn_layernorm_td_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d,
                                      const float* gamma, const float* beta, float eps, float* y,
                                      int64_t ld_y, void* stream) {
  if (int rc = check_dims(n, t, d)) return rc;
  if (!x || !gamma || !beta || !y) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (int rc = check_strides(ld_n, ld_t, n, t, d)) return rc;
  if (ld_y < (int64_t)t * d) return sagnn::fail(SAGNN_ERR_ARG, "ld_y smaller than t*d");
  if (n == 0) return SAGNN_OK;
  const int64_t blocks = (n + 3) / 4;
  if (blocks > INT32_MAX) return sagnn::fail(SAGNN_ERR_ARG, "grid too large");
  sagnn::ProfileScope prof(sagnn::kProfLayerNorm, static_cast<hipStream_t>(stream), n, t);
  hipLaunchKernelGGL(layernorm_td_kernel, dim3((unsigned)blocks), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), x, ld_n, ld_t, n, t, d, gamma, beta, eps, y, ld_y);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

// layer norm + Q|K|V + attention + mean in ONE kernel: the f32-MFMA kernel (d = 32 / 64) or the split-bf16 one
// (d = 32 / 64 / 128, 16 heads, specialised interval counts)
static bool fused_attention(int d, int t, int heads) {
  if (sagnn::force_valu()) return false;
  return sagnn::mhsa_mfma_supported(d, t, heads) || (sagnn::mhsa_split_supported(d, t, heads) && !sagnn::force_f32_mfma());
}

extern "C" int sagnn_mhsa_mean_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads,
                                   const float* Wq, const float* bq, const float* Wk,
                                   const float* bk, const float* Wv, const float* bv, float* out,
                                   int64_t ld_out, void* stream) {
  if (int rc = check_dims(n, t, d)) return rc;
  if (heads < 1 || d % heads) return sagnn::fail(SAGNN_ERR_DIM, "heads = %d does not divide d = %d", heads, d);
  if (!x || !Wq || !bq || !Wk || !bk || !Wv || !bv || !out)
    return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (int rc = check_strides(ld_n, ld_t, n, t, d)) return rc;
  if (ld_out < d) return sagnn::fail(SAGNN_ERR_ARG, "ld_out smaller than d");
  if (n == 0) return SAGNN_OK;
  const bool vec_ok = sagnn::aligned16(x) && (ld_n & 3) == 0 && (ld_t & 3) == 0;
  if (fused_attention(d, t, heads) && vec_ok)
    return sagnn::ln_mhsa_mean_mfma(x, ld_n, ld_t, n, t, d, heads, nullptr, nullptr, 0.f, 0, Wq, bq, Wk,
                                    bk, Wv, bv, out, ld_out, static_cast<hipStream_t>(stream));
  return sagnn::mhsa_mean_valu(x, ld_n, ld_t, n, t, d, heads, Wq, bq, Wk, bk, Wv, bv, out, ld_out,
                               static_cast<hipStream_t>(stream));
}

static bool use_wide(int d) { return !sagnn::lstm_mfma_supported(d) && sagnn::wide_supported(d) && !sagnn::force_valu(); }

extern "C" size_t sagnn_interval_fusion_workspace_bytes(int64_t n, int t, int d) {
  if (n <= 0 || t <= 0 || d <= 0) return 0;
  size_t bytes = (size_t)n * (size_t)t * (size_t)d * sizeof(float);  // h, normalised in place
  if (use_wide(d)) bytes += sagnn_mhsa_wide_workspace_bytes(n, t, d);  // Q|K|V of the wide attention (when it is taken)
  return bytes;
}

extern "C" int sagnn_interval_fusion_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t,
                                         int d, int heads, const float* lstm_W, const float* lstm_b,
                                         float forget_bias, const float* ln_gamma,
                                         const float* ln_beta, float ln_eps, const float* Wq,
                                         const float* bq, const float* Wk, const float* bk,
                                         const float* Wv, const float* bv, float* out,
                                         int64_t ld_out, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  if (int rc = check_dims(n, t, d)) return rc;
  const size_t need = sagnn_interval_fusion_workspace_bytes(n, t, d);
  if (n > 0 && (!workspace || workspace_bytes < need))
    return sagnn::fail(SAGNN_ERR_WORKSPACE, "fusion workspace needs %zu bytes", need);
  float* h = static_cast<float*>(workspace);
  const int64_t ldw = (int64_t)t * d;
  if (heads < 1 || d % heads) return sagnn::fail(SAGNN_ERR_DIM, "heads = %d does not divide d = %d", heads, d);
  if (!ln_gamma || !ln_beta || !Wq || !bq || !Wk || !bk || !Wv || !bv || !out)
    return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (n == 0) return SAGNN_OK;
  if (use_wide(d) && !fused_attention(d, t, heads)) {  // d = 96, 160, ...: MFMA products + per-node kernel for the attention
    float* scratch = h + n * ldw;
    const size_t sbytes = workspace_bytes - (size_t)n * ldw * sizeof(float);
    if (int rc = sagnn_lstm_fwd_f32(x, ld_n, ld_t, n, t, d, lstm_W, lstm_b, forget_bias, nullptr, h, ldw, stream)) return rc;
    if (int rc = sagnn_layernorm_td_f32(h, ldw, d, n, t, d, ln_gamma, ln_beta, ln_eps, h, ldw, stream)) return rc;
    return sagnn_mhsa_mean_wide_f32(h, ldw, d, n, t, d, heads, Wq, bq, Wk, bk, Wv, bv, out, ld_out, scratch, sbytes, stream);
  }
  if (int rc = sagnn_lstm_fwd_f32(x, ld_n, ld_t, n, t, d, lstm_W, lstm_b, forget_bias, nullptr, h, ldw, stream)) return rc;
  // layer norm rides on the attention kernel's A operand when the matrix-core path applies:
  // h is read once and never rewritten
  if (fused_attention(d, t, heads))
    return sagnn::ln_mhsa_mean_mfma(h, ldw, d, n, t, d, heads, ln_gamma, ln_beta, ln_eps, 1, Wq, bq, Wk, bk,
                                    Wv, bv, out, ld_out, static_cast<hipStream_t>(stream));
  if (int rc = sagnn_layernorm_td_f32(h, ldw, d, n, t, d, ln_gamma, ln_beta, ln_eps, h, ldw, stream)) return rc;
  return sagnn_mhsa_mean_f32(h, ldw, d, n, t, d, heads, Wq, bq, Wk, bk, Wv, bv, out, ld_out, stream);
}


// layer_norm over (T, d) + attention + mean (model.py:152-155) without the LSTM: the training
// forward runs the LSTM separately (it stores gates / cell) and comes here with the emitted h.
extern "C" size_t sagnn_ln_mhsa_mean_workspace_bytes(int64_t n, int t, int d, int heads) {
  if (n <= 0 || t <= 0 || d <= 0) return 0;
  if (fused_attention(d, t, heads)) return 0;  // normalised on the way into the product
  size_t bytes = (size_t)n * (size_t)t * (size_t)d * sizeof(float);
  if (use_wide(d)) bytes += sagnn_mhsa_wide_workspace_bytes(n, t, d);
  return bytes;
}

extern "C" int sagnn_ln_mhsa_mean_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads,
                                      const float* ln_gamma, const float* ln_beta, float ln_eps, const float* Wq,
                                      const float* bq, const float* Wk, const float* bk, const float* Wv,
                                      const float* bv, float* out, int64_t ld_out, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  if (int rc = check_dims(n, t, d)) return rc;
  if (heads < 1 || d % heads) return sagnn::fail(SAGNN_ERR_DIM, "heads = %d does not divide d = %d", heads, d);
  if (!x || !ln_gamma || !ln_beta || !Wq || !bq || !Wk || !bk || !Wv || !bv || !out)
    return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (int rc = check_strides(ld_n, ld_t, n, t, d)) return rc;
  if (ld_out < d) return sagnn::fail(SAGNN_ERR_ARG, "ld_out smaller than d");
  if (n == 0) return SAGNN_OK;
  const bool vec_ok = sagnn::aligned16(x) && (ld_n & 3) == 0 && (ld_t & 3) == 0;
  if (fused_attention(d, t, heads) && vec_ok)
    return sagnn::ln_mhsa_mean_mfma(x, ld_n, ld_t, n, t, d, heads, ln_gamma, ln_beta, ln_eps, 1, Wq, bq, Wk, bk,
                                    Wv, bv, out, ld_out, static_cast<hipStream_t>(stream));
  const size_t ybytes = (size_t)n * t * d * sizeof(float);
  size_t need = ybytes + (use_wide(d) ? sagnn_mhsa_wide_workspace_bytes(n, t, d) : 0);
  if (!workspace || workspace_bytes < need)
    return sagnn::fail(SAGNN_ERR_WORKSPACE, "ln_mhsa_mean workspace needs %zu bytes", need);
  float* y = static_cast<float*>(workspace);
  const int64_t ldw = (int64_t)t * d;
  if (int rc = sagnn_layernorm_td_f32(x, ld_n, ld_t, n, t, d, ln_gamma, ln_beta, ln_eps, y, ldw, stream)) return rc;
  if (use_wide(d))
    return sagnn_mhsa_mean_wide_f32(y, ldw, d, n, t, d, heads, Wq, bq, Wk, bk, Wv, bv, out, ld_out, y + n * ldw,
                                    workspace_bytes - ybytes, stream);
  return sagnn_mhsa_mean_f32(y, ldw, d, n, t, d, heads, Wq, bq, Wk, bk, Wv, bv, out, ld_out, stream);
}

// Front of the attention backward pass in one launch (d in {32, 64}, d_k in {2, 4}, t in {1..6, 8}):
// y = LN(x) (apply_ln) or x, Q|K|V = y W + b on the matrix cores, attention backward per (node, head)
// in registers -> dqkv [n*t, 3d]; y [n*t, d] is written when y_out is given (operand of dW = y^T dQKV).
extern "C" int sagnn_attn_bwd_front_supported(int d, int t, int heads) {
  return sagnn::attn_bwd_front_supported(d, t, heads) && !sagnn::force_valu();
}

extern "C" int sagnn_attn_bwd_front_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads,
                                        const float* ln_gamma, const float* ln_beta, float ln_eps, int apply_ln,
                                        const float* Wq, const float* bq, const float* Wk, const float* bk,
                                        const float* Wv, const float* bv, const float* g_out, int64_t ld_g,
                                        float* dqkv, float* y_out, void* stream) {
  if (int rc = check_dims(n, t, d)) return rc;
  if (!sagnn::attn_bwd_front_supported(d, t, heads))
    return sagnn::fail(SAGNN_ERR_DIM, "attn_bwd_front: unsupported d = %d, t = %d, heads = %d", d, t, heads);
  if (!x || !Wq || !bq || !Wk || !bk || !Wv || !bv || !g_out || !dqkv || (apply_ln && (!ln_gamma || !ln_beta)))
    return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (int rc = check_strides(ld_n, ld_t, n, t, d)) return rc;
  if (ld_g < d) return sagnn::fail(SAGNN_ERR_ARG, "ld_g smaller than d");
  if (!sagnn::aligned16(x) || (ld_n & 3) || (ld_t & 3) || !sagnn::aligned16(g_out) || (ld_g & 3) ||
      !sagnn::aligned16(dqkv) || (y_out && !sagnn::aligned16(y_out)))
    return sagnn::fail(SAGNN_ERR_ALIGN, "attn_bwd_front: need 16-byte aligned rows");
  if (n == 0) return SAGNN_OK;
  return sagnn::attn_bwd_front_mfma(x, ld_n, ld_t, n, t, d, heads, ln_gamma, ln_beta, ln_eps, apply_ln, Wq, bq, Wk, bk,
                                    Wv, bv, g_out, ld_g, dqkv, y_out, static_cast<hipStream_t>(stream));
}

extern "C" int sagnn_lstm_fwd_train_f32(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d,
                                        const float* W, const float* b, float forget_bias,
                                        const float* drop_scale, float* h, int64_t ld_h, float* gates,
                                        float* cell, void* stream) {
  if (int rc = check_dims(n, t, d)) return rc;
  if (!x || !W || !b || !h || !gates || !cell) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (int rc = check_strides(ld_n, ld_t, n, t, d)) return rc;
  if (ld_h < (int64_t)t * d) return sagnn::fail(SAGNN_ERR_ARG, "ld_h smaller than t*d");
  if (n == 0) return SAGNN_OK;
  const bool vec_ok = sagnn::aligned16(x) && (ld_n & 3) == 0 && (ld_t & 3) == 0;
  if (sagnn::lstm_mfma_supported(d) && vec_ok && !sagnn::force_valu())
    return sagnn::lstm_fwd_mfma(x, ld_n, ld_t, n, t, d, W, b, forget_bias, drop_scale, h, ld_h, gates, cell, nullptr, 0,
                                nullptr, nullptr, static_cast<hipStream_t>(stream));
  if (sagnn::lstm_split128_supported(d) && vec_ok && (ld_h & 3) == 0 && sagnn::aligned16(h) &&
      (!drop_scale || sagnn::aligned16(drop_scale)) && !sagnn::force_valu() && !sagnn::force_f32_mfma())
    return sagnn::lstm_fwd_split128(x, ld_n, ld_t, n, t, W, b, forget_bias, drop_scale, h, ld_h, gates, cell, nullptr, 0, nullptr,
                                    nullptr, static_cast<hipStream_t>(stream));
  return sagnn::lstm_fwd_valu(x, ld_n, ld_t, n, t, d, W, b, forget_bias, drop_scale, h, ld_h, gates, cell, nullptr, 0,
                              nullptr, nullptr, static_cast<hipStream_t>(stream));
}
