// Backward pieces of the interval fusion (reference: what tf.gradients derives for
// model.py:135-155 when model.py:250 minimises the loss). The GEMM-shaped parts go through
// dense.hip; this file holds the per-node / element-wise parts:
//   attn_bwd:      Q|K|V and dL/d(mean context) -> dQ|dK|dV              (Utils/attention.py:35-45)
//   layernorm_bwd: dL/dy -> dL/dh, dgamma, dbeta                        (model.py:152-153)
//   lstm_bwd_step: one step of BPTT on saved gate activations            (model.py:135-146)
#include <math.h>

#include "common.h"

namespace {

constexpr int kBlock = 256;

// One node per group of d threads (thread = feature column j, head h = j / dk). qkv [n, t, 3d]
// (Q | K | V per row) is overwritten in place with dQ | dK | dV.
// out = mean_q sum_s a_qs V[s], a_qs = e_qs / (R_q + 1e-8), e = exp(Q K^T * scale). With
// g = dL/dout / t (the same for every query position):
//   p_s   = sum_{j in head} g_j V[s][j]
//   dz_qs = a_qs (p_s - sum_s' a_qs' p_s')
//   dQ[q] = scale * sum_s dz_qs K[s];  dK[s] = scale * sum_q dz_qs Q[q];  dV[s] = g * sum_q a_qs
__global__ void attn_bwd_kernel(float* __restrict__ qkv, const float* __restrict__ g_out, int64_t ld_g,
                                int64_t n, int t, int d, int heads) {
  extern __shared__ float sm[];
  const int j = threadIdx.x % d;
  const int slot = threadIdx.x / d;
  const int slots = blockDim.x / d;
  const int dk = d / heads;
  const int h = j / dk, c = j % dk, h0 = h * dk;
  const int td = t * d, tt = t * t;
  float* qs = sm + (size_t)slot * (3 * td + heads * (tt + 2 * t));
  float* ks = qs + td;
  float* vs = ks + td;
  float* as = vs + td;            // [heads][t][t]: e, then a
  float* ps = as + heads * tt;    // [heads][t]: p_s
  float* ds = ps + heads * t;     // [heads][t]: sum_s' a_qs' p_s' per query
  const float scale = 1.f / sqrtf((float)dk);
  const float inv_t = 1.f / (float)t;

  for (int64_t node0 = (int64_t)blockIdx.x * slots; node0 < n; node0 += (int64_t)gridDim.x * slots) {
    const int64_t node = node0 + slot;
    const bool valid = node < n;
    float* row = qkv + (valid ? node : 0) * (int64_t)(3 * td);
    for (int ts = 0; ts < t; ++ts) {
      qs[ts * d + j] = valid ? row[ts * 3 * d + j] : 0.f;
      ks[ts * d + j] = valid ? row[ts * 3 * d + d + j] : 0.f;
      vs[ts * d + j] = valid ? row[ts * 3 * d + 2 * d + j] : 0.f;
    }
    const float g = valid ? g_out[node * ld_g + j] * inv_t : 0.f;
    __syncthreads();
    // e_qs of this head: its dk lanes split the (q, s) pairs
    for (int pr = c; pr < tt; pr += dk) {
      const int q = pr / t, s = pr - q * t;
      float z = 0.f;
      for (int cc = 0; cc < dk; ++cc) z = fmaf(qs[q * d + h0 + cc], ks[s * d + h0 + cc], z);
      as[h * tt + pr] = expf(z * scale);
    }
    // p_s: head sum of g_j V[s][j] (the dk lanes of a head are adjacent lanes of one wave)
    for (int s = 0; s < t; ++s) {
      float part = g * vs[s * d + j];
      for (int off = 1; off < dk; off <<= 1) part += __shfl_xor(part, off);
      if (c == 0) ps[h * t + s] = part;
    }
    __syncthreads();
    // normalise the rows: lane c takes queries q = c, c + dk, ...
    for (int q = c; q < t; q += dk) {
      float* ar = as + h * tt + q * t;
      float rsum = 0.f;
      for (int s = 0; s < t; ++s) rsum += ar[s];
      const float inv = 1.f / (rsum + 1e-8f);
      float dot = 0.f;
      for (int s = 0; s < t; ++s) {
        const float a = ar[s] * inv;
        ar[s] = a;
        dot = fmaf(a, ps[h * t + s], dot);
      }
      ds[h * t + q] = dot;
    }
    __syncthreads();
    // per-column outputs, in place
    if (valid) {
      for (int q = 0; q < t; ++q) {
        const float* ar = as + h * tt + q * t;
        const float dot = ds[h * t + q];
        float dq = 0.f;
        for (int s = 0; s < t; ++s) dq = fmaf(ar[s] * (ps[h * t + s] - dot), ks[s * d + j], dq);
        row[q * 3 * d + j] = dq * scale;
      }
      for (int s = 0; s < t; ++s) {
        const float p = ps[h * t + s];
        float dkv = 0.f, asum = 0.f;
        for (int q = 0; q < t; ++q) {
          const float a = as[h * tt + q * t + s];
          dkv = fmaf(a * (p - ds[h * t + q]), qs[q * d + j], dkv);
          asum += a;
        }
        row[s * 3 * d + d + j] = dkv * scale;
        row[s * 3 * d + 2 * d + j] = g * asum;
      }
    }
    __syncthreads();
  }
}

// layer_norm over (t, d) per node, backward. y = (h - mean) * rstd * gamma + beta.
//   dhat = dy * gamma;  dh = rstd * (dhat - mean(dhat) - hhat * mean(dhat * hhat))
//   dgamma[k] += sum dy * hhat;  dbeta[k] += sum dy        (atomics once per thread at the end)
// One wavefront per node, persistent grid. Needs 64 % d == 0 or d % 64 == 0 so that a lane's
// elements fall in a fixed set of at most 4 columns.
__global__ void layernorm_td_bwd_kernel(const float* __restrict__ h, int64_t ld_h, const float* dy,
                                        int64_t ld_dy, int64_t n, int t, int d,
                                        const float* __restrict__ gamma, float eps, float* dh,
                                        int64_t ld_dh, float* __restrict__ dgamma,
                                        float* __restrict__ dbeta) {
  const int lane = threadIdx.x & 63;
  const int waves = (gridDim.x * blockDim.x) >> 6;
  const int td = t * d;
  const float inv_m = 1.f / (float)td;
  const int ncol = d >= 64 ? d / 64 : 1;
  float gacc[4] = {0.f, 0.f, 0.f, 0.f}, bacc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t node = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; node < n; node += waves) {
    const float* hr = h + node * ld_h;
    const float* dyr = dy + node * ld_dy;
    float s = 0.f;
    for (int i = lane; i < td; i += 64) s += hr[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s * inv_m;
    float v = 0.f;
    for (int i = lane; i < td; i += 64) {
      const float dl = hr[i] - mean;
      v = fmaf(dl, dl, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const float rstd = rsqrtf(v * inv_m + eps);
    float s1 = 0.f, s2 = 0.f;
    int m = 0;
    for (int i = lane; i < td; i += 64, ++m) {
      const int k = i % d;
      const float hhat = (hr[i] - mean) * rstd;
      const float g = dyr[i];
      const float dhat = g * gamma[k];
      s1 += dhat;
      s2 = fmaf(dhat, hhat, s2);
      const int slot = m % ncol;
      gacc[slot] = fmaf(g, hhat, gacc[slot]);
      bacc[slot] += g;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s1 += __shfl_xor(s1, off);
      s2 += __shfl_xor(s2, off);
    }
    s1 *= inv_m;
    s2 *= inv_m;
    float* dhr = dh + node * ld_dh;
    for (int i = lane; i < td; i += 64) {
      const int k = i % d;
      const float hhat = (hr[i] - mean) * rstd;
      dhr[i] = rstd * (dyr[i] * gamma[k] - s1 - hhat * s2);
    }
  }
  for (int slot = 0; slot < ncol; ++slot) {
    const int k = (lane + 64 * slot) % d;
    atomicAdd(dgamma + k, gacc[slot]);
    atomicAdd(dbeta + k, bacc[slot]);
  }
}

// Vector form of the above for d a power of two: a node is held in registers by a group of LPN
// lanes (VPL float4 each: t*d <= 4*LPN*VPL), 64/LPN nodes per wavefront at a time; h and dy are
// read once, every lane owns 4 fixed columns (4*LPN is a multiple of d), group sums run as xor
// shuffles inside the group, and dgamma / dbeta are reduced over the block in LDS before one
// atomic per column and block.
template <int VPL>
__global__ __launch_bounds__(kBlock) void layernorm_td_bwd_vec_kernel(
    const float* __restrict__ h, int64_t ld_h, const float* dy, int64_t ld_dy, int64_t n, int t, int d,
    const float* __restrict__ gamma, float eps, float* dh, int64_t ld_dh, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int lpn) {
  __shared__ float red[2 * 256];  // dgamma | dbeta partials by column (d <= 256)
  const int lane = threadIdx.x & 63;
  const int gl = lane & (lpn - 1);          // lane within the node's group
  const int grp = lane / lpn;               // node slot within the wave
  const int npw = 64 / lpn;                 // nodes per wave iteration
  const int td = t * d;
  const float inv_m = 1.f / (float)td;
  const int col = (4 * gl) % d;             // my 4 columns, the same for every j (d | 4*lpn)
  const float4 gm = *reinterpret_cast<const float4*>(gamma + col);
  float4 gacc = make_float4(0.f, 0.f, 0.f, 0.f), bacc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = threadIdx.x; i < 2 * 256; i += blockDim.x) red[i] = 0.f;
  __syncthreads();
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t base = wave0 * npw; base < n; base += waves * npw) {
    const int64_t node = base + grp;
    const bool live = node < n;
    const int64_t nd = live ? node : n - 1;
    float4 hv[VPL], gv[VPL];
    bool on[VPL];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int i = 4 * (gl + lpn * j);
      on[j] = i < td;
      hv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      gv[j] = hv[j];
      if (on[j]) {
        hv[j] = *reinterpret_cast<const float4*>(h + nd * ld_h + i);
        gv[j] = *reinterpret_cast<const float4*>(dy + nd * ld_dy + i);
      }
      s += (hv[j].x + hv[j].y) + (hv[j].z + hv[j].w);
    }
    for (int off = lpn >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s * inv_m;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      if (on[j]) {
        const float a = hv[j].x - mean, b = hv[j].y - mean, c = hv[j].z - mean, e = hv[j].w - mean;
        v += (a * a + b * b) + (c * c + e * e);
      }
    }
    for (int off = lpn >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const float rstd = rsqrtf(v * inv_m + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      // hv becomes hhat, gv stays dy (zero where the slot is off)
      hv[j].x = on[j] ? (hv[j].x - mean) * rstd : 0.f;
      hv[j].y = on[j] ? (hv[j].y - mean) * rstd : 0.f;
      hv[j].z = on[j] ? (hv[j].z - mean) * rstd : 0.f;
      hv[j].w = on[j] ? (hv[j].w - mean) * rstd : 0.f;
      const float dx_ = gv[j].x * gm.x, dy_ = gv[j].y * gm.y, dz_ = gv[j].z * gm.z, dw_ = gv[j].w * gm.w;
      s1 += (dx_ + dy_) + (dz_ + dw_);
      s2 += (dx_ * hv[j].x + dy_ * hv[j].y) + (dz_ * hv[j].z + dw_ * hv[j].w);
      if (live) {
        gacc.x = fmaf(gv[j].x, hv[j].x, gacc.x);
        gacc.y = fmaf(gv[j].y, hv[j].y, gacc.y);
        gacc.z = fmaf(gv[j].z, hv[j].z, gacc.z);
        gacc.w = fmaf(gv[j].w, hv[j].w, gacc.w);
        bacc.x += gv[j].x;
        bacc.y += gv[j].y;
        bacc.z += gv[j].z;
        bacc.w += gv[j].w;
      }
    }
    for (int off = lpn >> 1; off > 0; off >>= 1) {
      s1 += __shfl_xor(s1, off);
      s2 += __shfl_xor(s2, off);
    }
    s1 *= inv_m;
    s2 *= inv_m;
    if (live) {
#pragma unroll
      for (int j = 0; j < VPL; ++j) {
        if (on[j]) {
          float4 o;
          o.x = rstd * (gv[j].x * gm.x - s1 - hv[j].x * s2);
          o.y = rstd * (gv[j].y * gm.y - s1 - hv[j].y * s2);
          o.z = rstd * (gv[j].z * gm.z - s1 - hv[j].z * s2);
          o.w = rstd * (gv[j].w * gm.w - s1 - hv[j].w * s2);
          *reinterpret_cast<float4*>(dh + node * ld_dh + 4 * (gl + lpn * j)) = o;
        }
      }
    }
  }
  atomicAdd(red + col + 0, gacc.x);
  atomicAdd(red + col + 1, gacc.y);
  atomicAdd(red + col + 2, gacc.z);
  atomicAdd(red + col + 3, gacc.w);
  atomicAdd(red + 256 + col + 0, bacc.x);
  atomicAdd(red + 256 + col + 1, bacc.y);
  atomicAdd(red + 256 + col + 2, bacc.z);
  atomicAdd(red + 256 + col + 3, bacc.w);
  __syncthreads();
  for (int k = threadIdx.x; k < d; k += blockDim.x) {
    atomicAdd(dgamma + k, red[k]);
    atomicAdd(dbeta + k, red[256 + k]);
  }
}

// One BPTT step on saved activations. gates [n, t, 4d] = sigmoid(i) | tanh(j) | sigmoid(f + fb) |
// sigmoid(o); cell [n, t, d]. h = tanh(c) * o, c = c_prev * f + i * j.
__global__ void lstm_bwd_step_kernel(const float* __restrict__ gates, const float* __restrict__ cell,
                                     const float* __restrict__ dh_ext, int64_t ld_dhe,
                                     const float* __restrict__ drop, const float* __restrict__ dh_rec,
                                     int64_t ld_dhr, const float* __restrict__ dc_in,
                                     float* __restrict__ dgates, float* __restrict__ dc_out, int64_t n,
                                     int t, int d, int ts) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int d4 = d >> 2;
  if (idx >= n * d4) return;
  const int64_t node = idx / d4;
  const int col = (int)(idx - node * d4) * 4;
  const int64_t gbase = (node * t + ts) * (int64_t)(4 * d) + col;
  const float4 gi = *reinterpret_cast<const float4*>(gates + gbase);
  const float4 gj = *reinterpret_cast<const float4*>(gates + gbase + d);
  const float4 gf = *reinterpret_cast<const float4*>(gates + gbase + 2 * d);
  const float4 go = *reinterpret_cast<const float4*>(gates + gbase + 3 * d);
  const float4 c = *reinterpret_cast<const float4*>(cell + (node * t + ts) * (int64_t)d + col);
  float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ts > 0) cp = *reinterpret_cast<const float4*>(cell + (node * t + ts - 1) * (int64_t)d + col);
  float4 dh = *reinterpret_cast<const float4*>(dh_ext + node * ld_dhe + (int64_t)ts * d + col);
  if (drop) {
    const float4 sc = *reinterpret_cast<const float4*>(drop + (node * t + ts) * (int64_t)d + col);
    dh.x *= sc.x;
    dh.y *= sc.y;
    dh.z *= sc.z;
    dh.w *= sc.w;
  }
  if (dh_rec) {
    const float4 r = *reinterpret_cast<const float4*>(dh_rec + node * ld_dhr + col);
    dh.x += r.x;
    dh.y += r.y;
    dh.z += r.z;
    dh.w += r.w;
  }
  float4 dc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (dc_in) dc = *reinterpret_cast<const float4*>(dc_in + node * (int64_t)d + col);
  float4 o_i, o_j, o_f, o_o, o_c;
#define SAGNN_LSTM_BWD(e)                                 \
  {                                                       \
    const float tc = tanhf(c.e);                          \
    const float dcv = dc.e + dh.e * go.e * (1.f - tc * tc); \
    o_o.e = dh.e * tc * go.e * (1.f - go.e);              \
    o_i.e = dcv * gj.e * gi.e * (1.f - gi.e);             \
    o_j.e = dcv * gi.e * (1.f - gj.e * gj.e);             \
    o_f.e = dcv * cp.e * gf.e * (1.f - gf.e);             \
    o_c.e = dcv * gf.e;                                   \
  }
  SAGNN_LSTM_BWD(x)
  SAGNN_LSTM_BWD(y)
  SAGNN_LSTM_BWD(z)
  SAGNN_LSTM_BWD(w)
#undef SAGNN_LSTM_BWD
  float* dg = dgates + node * (int64_t)(4 * d) + col;
  *reinterpret_cast<float4*>(dg) = o_i;
  *reinterpret_cast<float4*>(dg + d) = o_j;
  *reinterpret_cast<float4*>(dg + 2 * d) = o_f;
  *reinterpret_cast<float4*>(dg + 3 * d) = o_o;
  *reinterpret_cast<float4*>(dc_out + node * (int64_t)d + col) = o_c;
}

// out[i] = a[i] * b[i]
__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                           int64_t count) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < count) {
    const float4 x = *reinterpret_cast<const float4*>(a + i), y = *reinterpret_cast<const float4*>(b + i);
    *reinterpret_cast<float4*>(out + i) = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
  } else {
    for (int64_t k = i; k < count; ++k) out[k] = a[k] * b[k];
  }
}

// TF1 AdamOptimizer step with the L2 term of the reference's regLoss folded into the gradient:
//   g' = g + 2*l2*p;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  p -= lr_t * m / (sqrt(v) + eps)
// lr_t = lr * sqrt(1 - b2^step) / (1 - b1^step) is formed by the caller.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t count, float lr_t, float b1, float b2, float eps,
                            float l2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float pv = p[i];
  const float gv = g[i] + 2.f * l2 * pv;
  const float mv = b1 * m[i] + (1.f - b1) * gv;
  const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
  m[i] = mv;
  v[i] = vv;
  p[i] = pv - lr_t * mv / (sqrtf(vv) + eps);
}

// One launch for up to kAdamMaxTensors parameter tensors: block b works on elements
// [b*kAdamChunk, (b+1)*kAdamChunk) of the concatenation; the owning tensor is found by a binary
// search over the (chunk-aligned) block prefix held in the kernel arguments. g == NULL: zero
// gradient (a registered tensor that only the L2 term reaches).
constexpr int kAdamMaxTensors = 48;
constexpr int kAdamChunk = 1024;   // elements per block (256 threads x float4)
struct AdamTable {
  float* p[kAdamMaxTensors];
  const float* g[kAdamMaxTensors];
  float* m[kAdamMaxTensors];
  float* v[kAdamMaxTensors];
  int64_t block0[kAdamMaxTensors + 1];   // first block of tensor i; [n] = total blocks
  int64_t count[kAdamMaxTensors];
  float l2[kAdamMaxTensors];
  int n;
};

__device__ __forceinline__ void adam_elem(float& pv, float gv, float& mv, float& vv, float lr_t, float b1, float b2,
                                          float eps, float l2) {
  gv += 2.f * l2 * pv;
  mv = b1 * mv + (1.f - b1) * gv;
  vv = b2 * vv + (1.f - b2) * gv * gv;
  pv -= lr_t * mv / (sqrtf(vv) + eps);
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamTable tb, float lr_t, float b1, float b2, float eps) {
  const int64_t b = blockIdx.x;
  int lo = 0, hi = tb.n - 1;          // largest i with block0[i] <= b (wave-uniform: stays on the SALU)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tb.block0[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const int64_t base = (b - tb.block0[lo]) * kAdamChunk + (int64_t)threadIdx.x * 4;
  const int64_t count = tb.count[lo];
  if (base >= count) return;
  float* __restrict__ p = tb.p[lo] + base;
  float* __restrict__ m = tb.m[lo] + base;
  float* __restrict__ v = tb.v[lo] + base;
  const float* __restrict__ g = tb.g[lo] ? tb.g[lo] + base : nullptr;
  const float l2 = tb.l2[lo];
  if (base + 4 <= count) {            // tensors start 16-byte aligned (checked on the host), chunks keep it
    float4 pv = *reinterpret_cast<float4*>(p), mv = *reinterpret_cast<float4*>(m), vv = *reinterpret_cast<float4*>(v);
    const float4 gv = g ? *reinterpret_cast<const float4*>(g) : make_float4(0.f, 0.f, 0.f, 0.f);
    adam_elem(pv.x, gv.x, mv.x, vv.x, lr_t, b1, b2, eps, l2);
    adam_elem(pv.y, gv.y, mv.y, vv.y, lr_t, b1, b2, eps, l2);
    adam_elem(pv.z, gv.z, mv.z, vv.z, lr_t, b1, b2, eps, l2);
    adam_elem(pv.w, gv.w, mv.w, vv.w, lr_t, b1, b2, eps, l2);
    *reinterpret_cast<float4*>(p) = pv;
    *reinterpret_cast<float4*>(m) = mv;
    *reinterpret_cast<float4*>(v) = vv;
  } else {
    for (int e = 0; base + e < count; ++e) {
      float pv = p[e], mv = m[e], vv = v[e];
      adam_elem(pv, g ? g[e] : 0.f, mv, vv, lr_t, b1, b2, eps, l2);
      p[e] = pv;
      m[e] = mv;
      v[e] = vv;
    }
  }
}

// out[i] = max(leaky*a[i], a[i]) + b[i]   (b nullable)
__global__ void leaky_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                 float leaky, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float x = a[i];
  out[i] = fmaxf(leaky * x, x) + (b ? b[i] : 0.f);
}

// preds[e] = <U[uid_e], I[iid_e]> + <leaky(S[loc_e]), A[iid_e]>   (second term optional)
// One wavefront per pair group: LPR = d/4 lanes per pair, float4 per lane.
__global__ void pair_score_kernel(const float* __restrict__ U, int64_t ldu, const float* __restrict__ I, int64_t ldi,
                                  const float* __restrict__ S, int64_t lds_, const float* __restrict__ A, int64_t lda,
                                  const int32_t* __restrict__ uids, const int32_t* __restrict__ iids,
                                  const int32_t* __restrict__ locs, float leaky, float* __restrict__ out,
                                  int64_t n_pairs, int d) {
  const int lpr = d >> 2;                       // lanes per pair (power of two <= 64)
  const int lane = threadIdx.x & 63;
  const int ppw = 64 / lpr;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t e = wave * ppw + lane / lpr;
  const int col = (lane % lpr) * 4;
  float acc = 0.f;
  if (e < n_pairs) {
    const int64_t u = uids[e], it = iids[e];
    const float4 a = *reinterpret_cast<const float4*>(U + u * ldu + col);
    const float4 b = *reinterpret_cast<const float4*>(I + it * ldi + col);
    acc = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    if (S) {
      const float4 s = *reinterpret_cast<const float4*>(S + (int64_t)locs[e] * lds_ + col);
      const float4 c = *reinterpret_cast<const float4*>(A + it * lda + col);
      acc += fmaxf(leaky * s.x, s.x) * c.x + fmaxf(leaky * s.y, s.y) * c.y + fmaxf(leaky * s.z, s.z) * c.z +
             fmaxf(leaky * s.w, s.w) * c.w;
    }
  }
  for (int off = 1; off < lpr; off <<= 1) acc += __shfl_xor(acc, off);
  if (e < n_pairs && (lane % lpr) == 0) out[e] = acc;
}

}  // namespace

extern "C" int sagnn_leaky_add_f32(const float* a, const float* b, float* out, float leaky, int64_t count,
                                   void* stream) {
  if (!a || !out) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (count <= 0) return SAGNN_OK;
  const int64_t blocks = (count + kBlock - 1) / kBlock;
  if (blocks > INT32_MAX) return sagnn::fail(SAGNN_ERR_ARG, "grid too large");
  hipLaunchKernelGGL(leaky_add_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), a,
                     b, out, leaky, count);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_pair_score_f32(const float* U, int64_t ldu, const float* I, int64_t ldi, const float* S,
                                    int64_t lds, const float* A, int64_t lda, const int32_t* uids,
                                    const int32_t* iids, const int32_t* locs, float leaky, float* out,
                                    int64_t n_pairs, int d, void* stream) {
  if (!U || !I || !uids || !iids || !out) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if ((S == nullptr) != (A == nullptr) || (S && !locs)) return sagnn::fail(SAGNN_ERR_NULL, "S, A and locs go together");
  const int lpr = d / 4;
  if (d < 4 || d > 256 || (d & 3) || (lpr & (lpr - 1)))
    return sagnn::fail(SAGNN_ERR_DIM, "d = %d: need 4 * a power of two, <= 256", d);
  if ((ldu & 3) || (ldi & 3) || (S && ((lds & 3) || (lda & 3)))) return sagnn::fail(SAGNN_ERR_ALIGN, "strides must be multiples of 4");
  if (n_pairs <= 0) return SAGNN_OK;
  const int ppw = 64 / lpr;
  const int64_t waves = (n_pairs + ppw - 1) / ppw;
  const int64_t blocks = (waves + 3) / 4;
  if (blocks > INT32_MAX) return sagnn::fail(SAGNN_ERR_ARG, "grid too large");
  hipLaunchKernelGGL(pair_score_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), U,
                     ldu, I, ldi, S, lds, A, lda, uids, iids, locs, leaky, out, n_pairs, d);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_mul_f32(const float* a, const float* b, float* out, int64_t count, void* stream) {
  if (!a || !b || !out) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (count <= 0) return SAGNN_OK;
  if (!sagnn::aligned16(a) || !sagnn::aligned16(b) || !sagnn::aligned16(out))
    return sagnn::fail(SAGNN_ERR_ALIGN, "operands must be 16-byte aligned");
  const int64_t blocks = ((count + 3) / 4 + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(mul_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), a, b, out, count);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_adam_step_f32(float* param, const float* grad, float* m, float* v, int64_t count, float lr,
                                   float beta1, float beta2, float eps, float l2, int64_t step, void* stream) {
  if (!param || !grad || !m || !v) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (step < 1) return sagnn::fail(SAGNN_ERR_ARG, "step counts from 1");
  if (count <= 0) return SAGNN_OK;
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
  const int64_t blocks = (count + kBlock - 1) / kBlock;
  if (blocks > INT32_MAX) return sagnn::fail(SAGNN_ERR_ARG, "grid too large");
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream), param,
                     grad, m, v, count, (float)lr_t, beta1, beta2, eps, l2);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_adam_multi_f32(int n_tensors, float* const* params, const float* const* grads, float* const* m,
                                    float* const* v, const int64_t* counts, const float* l2, float lr, float beta1,
                                    float beta2, float eps, int64_t step, void* stream) {
  if (n_tensors < 0) return sagnn::fail(SAGNN_ERR_ARG, "n_tensors = %d", n_tensors);
  if (n_tensors == 0) return SAGNN_OK;
  if (!params || !grads || !m || !v || !counts || !l2) return sagnn::fail(SAGNN_ERR_NULL, "null table pointer");
  if (step < 1) return sagnn::fail(SAGNN_ERR_ARG, "step counts from 1");
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
  for (int i = 0; i < n_tensors;) {   // one launch per kAdamMaxTensors NON-EMPTY tensors; `i` carries over between launches
    AdamTable tb{};
    int64_t blocks = 0;
    int k = 0;
    for (; i < n_tensors && k < kAdamMaxTensors; ++i) {
      if (counts[i] < 0) return sagnn::fail(SAGNN_ERR_ARG, "tensor %d: count = %lld", i, (long long)counts[i]);
      if (counts[i] == 0) continue;
      if (!params[i] || !m[i] || !v[i]) return sagnn::fail(SAGNN_ERR_NULL, "tensor %d: null param / m / v", i);
      if (!sagnn::aligned16(params[i]) || !sagnn::aligned16(m[i]) || !sagnn::aligned16(v[i]) ||
          (grads[i] && !sagnn::aligned16(grads[i])))
        return sagnn::fail(SAGNN_ERR_ALIGN, "tensor %d: 16-byte alignment required", i);
      tb.p[k] = params[i];
      tb.g[k] = grads[i];
      tb.m[k] = m[i];
      tb.v[k] = v[i];
      tb.count[k] = counts[i];
      tb.l2[k] = l2[i];
      tb.block0[k] = blocks;
      blocks += (counts[i] + kAdamChunk - 1) / kAdamChunk;
      ++k;
    }
    if (k == 0) continue;
    tb.n = k;
    tb.block0[k] = blocks;
    if (blocks > INT32_MAX) return sagnn::fail(SAGNN_ERR_ARG, "grid too large");
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), tb,
                       (float)lr_t, beta1, beta2, eps);
    SAGNN_HIP_TRY(hipGetLastError());
  }
  return SAGNN_OK;
}

extern "C" int sagnn_attn_bwd_f32(float* qkv, const float* g_out, int64_t ld_g, int64_t n, int t, int d,
                                  int heads, void* stream) {
  if (n < 0 || t < 1 || t > 64 || d < 4 || d > 256 || (d & 3)) return sagnn::fail(SAGNN_ERR_DIM, "bad n/t/d");
  if (heads < 1 || d % heads) return sagnn::fail(SAGNN_ERR_DIM, "heads = %d does not divide d = %d", heads, d);
  const int dk = d / heads;
  if (dk & (dk - 1)) return sagnn::fail(SAGNN_ERR_DIM, "d_k = %d must be a power of two", dk);
  if (64 % d != 0 && d % 64 != 0) return sagnn::fail(SAGNN_ERR_DIM, "d = %d: need 64 %% d == 0 or d %% 64 == 0", d);
  if (!qkv || !g_out) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (n == 0) return SAGNN_OK;
  int slots = kBlock / d > 0 ? kBlock / d : 1;
  const size_t per_slot = (size_t)(3 * t * d + heads * (t * t + 2 * t)) * sizeof(float);
  while (slots > 1 && slots * per_slot > 64 * 1024) slots >>= 1;
  const size_t lds = slots * per_slot;
  if (lds > 160 * 1024) return sagnn::fail(SAGNN_ERR_DIM, "t*d too large for LDS");
  if (int rc = sagnn::ensure_dynamic_lds(reinterpret_cast<const void*>(&attn_bwd_kernel), lds)) return rc;
  int64_t blocks = (n + slots - 1) / slots;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)blocks), dim3(slots * d), lds, static_cast<hipStream_t>(stream),
                     qkv, g_out, ld_g, n, t, d, heads);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_layernorm_td_bwd_f32(const float* h, int64_t ld_h, const float* dy, int64_t ld_dy, int64_t n,
                                          int t, int d, const float* gamma, float eps, float* dh, int64_t ld_dh,
                                          float* dgamma, float* dbeta, void* stream) {
  if (n < 0 || t < 1 || t > 64 || d < 4 || d > 256 || (d & 3)) return sagnn::fail(SAGNN_ERR_DIM, "bad n/t/d");
  if (64 % d != 0 && d % 64 != 0) return sagnn::fail(SAGNN_ERR_DIM, "d = %d: need 64 %% d == 0 or d %% 64 == 0", d);
  if (!h || !dy || !gamma || !dh || !dgamma || !dbeta) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (ld_h < (int64_t)t * d || ld_dy < (int64_t)t * d || ld_dh < (int64_t)t * d)
    return sagnn::fail(SAGNN_ERR_ARG, "node stride smaller than t*d");
  if (n == 0) return SAGNN_OK;
  // vector form: d a power of two, 16-byte aligned rows, at most 8 float4 per lane
  const int td4 = t * d / 4;
  int lpn = 1;
  while (lpn < td4 && lpn < 64) lpn <<= 1;
  if (lpn < d / 4) lpn = d / 4;
  const int vpl = (td4 + lpn - 1) / lpn;
  const bool vec_ok = (d & (d - 1)) == 0 && lpn <= 64 && vpl <= 8 && sagnn::aligned16(h) && sagnn::aligned16(dy) &&
                      sagnn::aligned16(dh) && sagnn::aligned16(gamma) && !((ld_h | ld_dy | ld_dh) & 3);
  if (vec_ok) {
    const int npw = 64 / lpn;
    int64_t vblocks = (n + 4 * npw - 1) / (4 * npw);
    if (vblocks > 4096) vblocks = 4096;
    hipStream_t s = static_cast<hipStream_t>(stream);
#define SAGNN_LNB(V)                                                                                              \
  hipLaunchKernelGGL(layernorm_td_bwd_vec_kernel<V>, dim3((unsigned)vblocks), dim3(kBlock), 0, s, h, ld_h, dy, ld_dy, \
                     n, t, d, gamma, eps, dh, ld_dh, dgamma, dbeta, lpn)
    if (vpl <= 1) SAGNN_LNB(1);
    else if (vpl <= 2) SAGNN_LNB(2);
    else if (vpl <= 4) SAGNN_LNB(4);
    else SAGNN_LNB(8);
#undef SAGNN_LNB
    SAGNN_HIP_TRY(hipGetLastError());
    return SAGNN_OK;
  }
  int64_t blocks = (n + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(layernorm_td_bwd_kernel, dim3((unsigned)blocks), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), h, ld_h, dy, ld_dy, n, t, d, gamma, eps, dh, ld_dh, dgamma,
                     dbeta);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

extern "C" int sagnn_lstm_bwd_step_f32(const float* gates, const float* cell, const float* dh_ext, int64_t ld_dhe,
                                       const float* drop_scale, const float* dh_rec, int64_t ld_dhr,
                                       const float* dc_in, float* dgates, float* dc_out, int64_t n, int t, int d,
                                       int ts, void* stream) {
  if (n < 0 || t < 1 || d < 4 || (d & 3) || ts < 0 || ts >= t) return sagnn::fail(SAGNN_ERR_DIM, "bad n/t/d/ts");
  if (!gates || !cell || !dh_ext || !dgates || !dc_out) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if ((ld_dhe & 3) || (dh_rec && (ld_dhr & 3))) return sagnn::fail(SAGNN_ERR_ALIGN, "strides must be multiples of 4");
  if (n == 0) return SAGNN_OK;
  const int64_t total = n * (d / 4);
  const int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > INT32_MAX) return sagnn::fail(SAGNN_ERR_ARG, "grid too large");
  hipLaunchKernelGGL(lstm_bwd_step_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                     gates, cell, dh_ext, ld_dhe, drop_scale, dh_rec, ld_dhr, dc_in, dgates, dc_out, n, t, d, ts);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}
