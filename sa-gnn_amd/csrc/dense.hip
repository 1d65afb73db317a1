// Tall-skinny dense products on the matrix cores (gfx950, exact-fp32 v_mfma_f32_32x32x2_f32):
// the building blocks of the backward pass of the interval fusion and of NNLayers.FC
// (reference Utils/NNLayers.py:98-115: `inp @ W`; tf.layers.dense in Utils/attention.py:66-72;
// their gradients as tf.gradients builds them for model.py:250).
//
//   nn:  Y[n, DOUT]   = X[n, DIN] @ W[DIN, DOUT] (+ bias)        n huge, W small (LDS-resident)
//   tn:  dW[DIN,DOUT] += X[n, DIN]^T @ G[n, DOUT], db += colsum(G)   reduction over the n rows
//
// DIN, DOUT are multiples of 32; W (DIN*DOUT*4 bytes) must fit LDS next to the staging tiles.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kBlock = 256;
constexpr int kRowsPerWave = 32;

__device__ __forceinline__ int crow(int r, int rh) { return (r & 3) + 8 * (r >> 2) + 4 * rh; }

// ---------------------------------------------------------------------------------------------
// nn: one wavefront = 32 rows x all DOUT columns (CT tiles); W in LDS as per-(k-step, tile)
// fragments Wf[(kk*CT + ct)*64 + l] = W[2kk + (l>>5)][ct*32 + (l&31)]; X staged 32 columns at a
// time through a per-wave XOR-swizzled [32][32] LDS tile (next chunk prefetched in registers).
// ---------------------------------------------------------------------------------------------
template <int CT>
__global__ __launch_bounds__(kBlock, 1) void dense_nn_kernel(const float* __restrict__ X, int64_t ldx,
                                                             int64_t n, int din,
                                                             const float* __restrict__ W, int64_t ldw,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ Y, int64_t ldy,
                                                             int accumulate, int64_t n_tiles) {
  constexpr int DOUT = CT * 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Wf = lds;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* stage = lds + (size_t)din * DOUT + wave * (32 * 32);
  const int ai = lane & 31, kh = lane >> 5;
  const int cj = lane & 31, rh = lane >> 5;
  const int fr = lane >> 3, fc4 = (lane & 7) * 4;  // fill: 8 lanes x float4 per row, 8 rows per instr

  const int ksteps = din / 2;
  for (int idx = threadIdx.x; idx < din * DOUT; idx += blockDim.x) {
    const int l = idx & 63;
    const int rest = idx >> 6;
    const int ct = rest % CT;
    const int kk = rest / CT;
    Wf[idx] = W[(size_t)(2 * kk + (l >> 5)) * ldw + ct * 32 + (l & 31)];
  }
  __syncthreads();
  float bcol[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) bcol[ct] = bias ? bias[ct * 32 + cj] : 0.f;
  const int nchunks = din / 32;

  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * 128 + (int64_t)wave * kRowsPerWave;
    if (row0 >= n) continue;  // wave-uniform
    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    float4 xr[4];
    auto fetch = [&](int kc) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t row = row0 + q * 8 + fr;
        xr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < n) xr[q] = *reinterpret_cast<const float4*>(X + row * ldx + kc * 32 + fc4);
      }
    };
    fetch(0);
    for (int kc = 0; kc < nchunks; ++kc) {
      int ai_ = ai, kh_ = kh, fr_ = fr, fc4_ = fc4;
      asm volatile("" : "+v"(ai_), "+v"(kh_), "+v"(fr_), "+v"(fc4_));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = q * 8 + fr_;
        float* dst = stage + r * 32;
        dst[(fc4_ + 0) ^ r] = xr[q].x;
        dst[(fc4_ + 1) ^ r] = xr[q].y;
        dst[(fc4_ + 2) ^ r] = xr[q].z;
        dst[(fc4_ + 3) ^ r] = xr[q].w;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float a[16];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) a[kk] = stage[ai_ * 32 + ((2 * kk + kh_) ^ ai_)];
      __builtin_amdgcn_wave_barrier();
      if (kc + 1 < nchunks) fetch(kc + 1);
      const float* wf = Wf + (size_t)(kc * 16) * CT * 64 + lane;
      float bcur[CT], bnxt[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) bcur[ct] = wf[ct * 64];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        if (kk + 1 < 16) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) bnxt[ct] = wf[((kk + 1) * CT + ct) * 64];
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bcur[ct], acc[ct], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) bcur[ct] = bnxt[ct];
      }
    }
    (void)ksteps;
    int cj_ = cj, rh_ = rh;
    asm volatile("" : "+v"(cj_), "+v"(rh_));
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + crow(r, rh_);
        if (row < n) {
          float* y = Y + row * ldy + ct * 32 + cj_;
          const float v = acc[ct][r] + bcol[ct];
          *y = accumulate ? (*y + v) : v;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// tn: the block walks RC-row chunks; X and G chunks sit row-major in a DOUBLE-BUFFERED LDS tile
// (the next chunk is in flight in registers while the MFMAs of the current one run; one barrier
// per chunk); the (DIN/32)x(DOUT/32) output tiles are dealt to the 4 waves (TPW each), accumulated
// over every chunk of the block and flushed once with float atomics (<= a few MB per launch).
// Column sums of G (bias gradient) ride along, one column per thread.
// ---------------------------------------------------------------------------------------------
template <int TPW, int RC, int NV>
__global__ __launch_bounds__(kBlock, TPW <= 4 ? 2 : 1) void dense_tn_kernel(const float* __restrict__ X, int64_t ldx,
                                                             const float* __restrict__ G, int64_t ldg,
                                                             int64_t n, int din, int dout,
                                                             float* __restrict__ dW, int64_t lddw,
                                                             float* __restrict__ db, int64_t n_chunks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int width = din + dout;       // one LDS row: X part | G part
  const int tile = RC * width;        // floats per buffer
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int nb = dout / 32;
  const int n_out_tiles = (din / 32) * nb;
  const int w4 = width / 4;           // float4 per row
  const int nvec = RC * w4;           // float4 per chunk (<= NV * kBlock)
  f32x16 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float colsum = 0.f;
  int aoff[TPW], boff[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    int tt = wave * TPW + j;
    if (tt >= n_out_tiles) tt = n_out_tiles - 1;
    const int ta = tt / nb, tb = tt - ta * nb;
    aoff[j] = ta * 32 + li;
    boff[j] = tb * 32 + li;
  }

  float4 stage[NV];
  auto fetch = [&](int64_t ch) {
    const int64_t row0 = ch * RC;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int i = threadIdx.x + v * kBlock;
      const int r = i / w4, c = (i - r * w4) * 4;
      // branch-free: out-of-range slots read a clamped address and are zeroed by a select
      const bool ok = i < nvec && row0 + r < n;
      const int64_t gr = row0 + r < n ? row0 + r : n - 1;
      const int rr = i < nvec ? c : 0;
      const float* src = rr < din ? X + gr * ldx + rr : G + gr * ldg + (rr - din);
      const float4 val = *reinterpret_cast<const float4*>(src);
      stage[v] = make_float4(ok ? val.x : 0.f, ok ? val.y : 0.f, ok ? val.z : 0.f, ok ? val.w : 0.f);
    }
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int i = threadIdx.x + v * kBlock;
      if (i < nvec) reinterpret_cast<float4*>(buf)[i] = stage[v];
    }
  };

  int64_t ch = blockIdx.x;
  int cur = 0;
  if (ch < n_chunks) {
    fetch(ch);
    commit(lds);
  }
  __syncthreads();
  for (; ch < n_chunks; ch += gridDim.x) {
    const int64_t nxt = ch + gridDim.x;
    if (nxt < n_chunks) fetch(nxt);  // lands under the MFMAs below
    const float* xs = lds + cur * tile;
    const float* gs = xs + din;
    if (db && (int)threadIdx.x < dout) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < RC; ++r) s += gs[r * width + threadIdx.x];
      colsum += s;
    }
    // No validity branch in here: a wave whose tile index runs past the last tile recomputes the
    // last one and drops it at the flush (a branch per MFMA would fence every LDS read). Operands
    // are read one GROUP of k-steps ahead of the MFMAs that use them: one MFMA (64 cycles) does
    // not cover an LDS round trip, so reading "just in time" halves the matrix-core rate.
    {
      constexpr int GK = TPW >= 4 ? 2 : 4;  // k-steps per group: 2*GK*TPW operand registers, twice
      constexpr int NG = RC / 2 / GK;
      float ac[GK][TPW], bc[GK][TPW], an[GK][TPW], bn[GK][TPW];
      auto read_group = [&](float (&a)[GK][TPW], float (&b)[GK][TPW], int g) {
#pragma unroll
        for (int u = 0; u < GK; ++u) {
          const int row = 2 * (g * GK + u) + kh;
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            a[u][j] = xs[row * width + aoff[j]];  // A[i = din index][k = row]
            b[u][j] = gs[row * width + boff[j]];  // B[k = row][j = dout index]
          }
        }
      };
      read_group(ac, bc, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) read_group(an, bn, g + 1);
#pragma unroll
        for (int u = 0; u < GK; ++u)
#pragma unroll
          for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u][j], bc[u][j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < GK; ++u)
#pragma unroll
          for (int j = 0; j < TPW; ++j) ac[u][j] = an[u][j], bc[u][j] = bn[u][j];
      }
    }
    if (nxt < n_chunks) commit(lds + (cur ^ 1) * tile);
    __syncthreads();  // next buffer complete; this one free to be overwritten next time round
    cur ^= 1;
  }
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int tt = wave * TPW + j;
    if (tt < n_out_tiles) {
      const int ta = tt / nb, tb = tt - ta * nb;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        atomicAdd(dW + (size_t)(ta * 32 + crow(r, kh)) * lddw + tb * 32 + li, acc[j][r]);
    }
  }
  if (db && (int)threadIdx.x < dout) atomicAdd(db + threadIdx.x, colsum);
}

int cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
  }
  return cus;
}

template <int CT>
int launch_nn(const float* X, int64_t ldx, int64_t n, int din, const float* W, int64_t ldw, const float* bias,
              float* Y, int64_t ldy, int accumulate, hipStream_t s) {
  const size_t lds = ((size_t)din * CT * 32 + 4 * 32 * 32) * sizeof(float);
  if (lds > 160 * 1024) return sagnn::fail(SAGNN_ERR_DIM, "dense_nn: W %d x %d does not fit LDS", din, CT * 32);
  if (int rc = sagnn::ensure_dynamic_lds(reinterpret_cast<const void*>(&dense_nn_kernel<CT>), lds)) return rc;
  const int64_t n_tiles = (n + 127) / 128;
  const int64_t blocks = n_tiles < cu_count() ? n_tiles : cu_count();
  hipLaunchKernelGGL(dense_nn_kernel<CT>, dim3((unsigned)blocks), dim3(kBlock), lds, s, X, ldx, n, din, W, ldw, bias,
                     Y, ldy, accumulate, n_tiles);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

template <int TPW, int RC, int NV>
int launch_tn_rc(const float* X, int64_t ldx, const float* G, int64_t ldg, int64_t n, int din, int dout, float* dW,
                 int64_t lddw, float* db, hipStream_t s) {
  const size_t lds = (size_t)2 * RC * (din + dout) * sizeof(float);
  if (int rc = sagnn::ensure_dynamic_lds(reinterpret_cast<const void*>(&dense_tn_kernel<TPW, RC, NV>), lds)) return rc;
  const int64_t n_chunks = (n + RC - 1) / RC;
  const int64_t want = (int64_t)cu_count() * ((TPW <= 4 && lds <= 72 * 1024) ? 2 : 1);
  const int64_t blocks = n_chunks < want ? n_chunks : want;
  hipLaunchKernelGGL((dense_tn_kernel<TPW, RC, NV>), dim3((unsigned)blocks), dim3(kBlock), lds, s, X, ldx, G, ldg, n,
                     din, dout, dW, lddw, db, n_chunks);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

// din + dout <= 384 (dense_tn_any slices to 128 x 256): 32-row chunks, two buffers <= 96 KB. With
// <= 4 tiles per wave two blocks share a CU (their chunks' loads and MFMAs interleave).
template <int TPW>
int launch_tn(const float* X, int64_t ldx, const float* G, int64_t ldg, int64_t n, int din, int dout, float* dW,
              int64_t lddw, float* db, hipStream_t s) {
  return launch_tn_rc<TPW, 32, 12>(X, ldx, G, ldg, n, din, dout, dW, lddw, db, s);
}

int check_xy(const char* name, const void* p, int64_t ld, int cols) {
  if (!p) return sagnn::fail(SAGNN_ERR_NULL, "%s is NULL", name);
  if (!sagnn::aligned16(p) || (ld & 3)) return sagnn::fail(SAGNN_ERR_ALIGN, "%s: need 16-byte aligned rows", name);
  if (ld < cols) return sagnn::fail(SAGNN_ERR_ARG, "%s: ld %lld < %d columns", name, (long long)ld, cols);
  return SAGNN_OK;
}

}  // namespace

namespace {

int nn_piece(const float* X, int64_t ldx, int64_t n, int din, int dout, const float* W, int64_t ldw,
             const float* bias, float* Y, int64_t ldy, int accumulate, hipStream_t s) {
  switch (dout / 32) {
    case 1: return launch_nn<1>(X, ldx, n, din, W, ldw, bias, Y, ldy, accumulate, s);
    case 2: return launch_nn<2>(X, ldx, n, din, W, ldw, bias, Y, ldy, accumulate, s);
    case 3: return launch_nn<3>(X, ldx, n, din, W, ldw, bias, Y, ldy, accumulate, s);
    case 4: return launch_nn<4>(X, ldx, n, din, W, ldw, bias, Y, ldy, accumulate, s);
    case 5: return launch_nn<5>(X, ldx, n, din, W, ldw, bias, Y, ldy, accumulate, s);
    case 6: return launch_nn<6>(X, ldx, n, din, W, ldw, bias, Y, ldy, accumulate, s);
    case 7: return launch_nn<7>(X, ldx, n, din, W, ldw, bias, Y, ldy, accumulate, s);
    default: return launch_nn<8>(X, ldx, n, din, W, ldw, bias, Y, ldy, accumulate, s);
  }
}

int tn_piece(const float* X, int64_t ldx, const float* G, int64_t ldg, int64_t n, int din, int dout, float* dW,
             int64_t lddw, float* db, hipStream_t s) {
  const int tiles = (din / 32) * (dout / 32);
  const int tpw = (tiles + 3) / 4;
  if (tpw <= 1) return launch_tn<1>(X, ldx, G, ldg, n, din, dout, dW, lddw, db, s);
  if (tpw <= 2) return launch_tn<2>(X, ldx, G, ldg, n, din, dout, dW, lddw, db, s);
  if (tpw <= 3) return launch_tn<3>(X, ldx, G, ldg, n, din, dout, dW, lddw, db, s);
  if (tpw <= 4) return launch_tn<4>(X, ldx, G, ldg, n, din, dout, dW, lddw, db, s);
  if (tpw <= 8) return launch_tn<8>(X, ldx, G, ldg, n, din, dout, dW, lddw, db, s);
  return sagnn::fail(SAGNN_ERR_DIM, "dense_tn piece %d x %d exceeds 128 x 256", din, dout);  // dense_tn_any slices
}

constexpr int kLdsFloatsForW = (160 * 1024 - 4 * 32 * 32 * 4) / 4;  // what dense_nn can give to W

}  // namespace

namespace sagnn {

// Y (+)= X @ W + bias for ANY din/dout that are multiples of 32: the product is cut into column
// slices of <= 256 and row (K) chunks whose W block fits LDS; later chunks accumulate.
int dense_nn_any(const float* X, int64_t ldx, int64_t n, int din, int dout, const float* W, int64_t ldw,
                 const float* bias, float* Y, int64_t ldy, int accumulate, hipStream_t s) {
  // a W that does not fit LDS next to the staging tiles would be cut into pieces, every piece re-reading X: the tiled
  // GEMM of dense_gemm.hip streams both operands instead (the d = 128 BPTT products)
  if ((int64_t)din * dout > kLdsFloatsForW) return gemm_nn(X, ldx, n, din, dout, W, ldw, bias, Y, ldy, accumulate, s);
  for (int c0 = 0; c0 < dout; c0 += 256) {
    const int cw = dout - c0 < 256 ? dout - c0 : 256;
    int kmax = (kLdsFloatsForW / cw) / 32 * 32;
    if (kmax > din) kmax = din;
    for (int k0 = 0; k0 < din; k0 += kmax) {
      const int kw = din - k0 < kmax ? din - k0 : kmax;
      if (int rc = nn_piece(X + k0, ldx, n, kw, cw, W + (size_t)k0 * ldw + c0, ldw, (k0 == 0 && bias) ? bias + c0 : nullptr,
                            Y + c0, ldy, (k0 > 0) ? 1 : accumulate, s))
        return rc;
    }
  }
  return SAGNN_OK;
}

// dW += X^T G, db += colsum(G) for any multiples of 32: 256 x 256 output blocks at most per launch.
int dense_tn_any(const float* X, int64_t ldx, const float* G, int64_t ldg, int64_t n, int din, int dout, float* dW,
                 int64_t lddw, float* db, hipStream_t s) {
  if (din > 128 || dout > 256) return gemm_tn(X, ldx, 0, G, ldg, 0, n, 1, din, dout, dW, lddw, db, s);   // more than one piece: dense_gemm.hip
  for (int r0 = 0; r0 < din; r0 += 128) {
    const int rw = din - r0 < 128 ? din - r0 : 128;
    for (int c0 = 0; c0 < dout; c0 += 256) {
      const int cw = dout - c0 < 256 ? dout - c0 : 256;
      if (int rc = tn_piece(X + r0, ldx, G + c0, ldg, n, rw, cw, dW + (size_t)r0 * lddw + c0, lddw,
                            (r0 == 0 && db) ? db + c0 : nullptr, s))
        return rc;
    }
  }
  return SAGNN_OK;
}

}  // namespace sagnn

extern "C" int sagnn_dense_nn_f32(const float* X, int64_t ldx, int64_t n, int din, int dout, const float* W,
                                  const float* bias, float* Y, int64_t ldy, int accumulate, void* stream) {
  if (n < 0 || din < 32 || dout < 32 || (din & 31) || (dout & 31))
    return sagnn::fail(SAGNN_ERR_DIM, "dense_nn: need din, dout multiples of 32 (got %d, %d)", din, dout);
  if (int rc = check_xy("X", X, ldx, din)) return rc;
  if (int rc = check_xy("Y", Y, ldy, dout)) return rc;
  if (!W) return sagnn::fail(SAGNN_ERR_NULL, "W is NULL");
  if (n == 0) return SAGNN_OK;
  return sagnn::dense_nn_any(X, ldx, n, din, dout, W, dout, bias, Y, ldy, accumulate, static_cast<hipStream_t>(stream));
}

extern "C" int sagnn_dense_tn_f32(const float* X, int64_t ldx, const float* G, int64_t ldg, int64_t n, int din,
                                  int dout, float* dW, float* db, void* stream) {
  if (n < 0 || din < 32 || dout < 32 || (din & 31) || (dout & 31))
    return sagnn::fail(SAGNN_ERR_DIM, "dense_tn: need din, dout multiples of 32 (got %d, %d)", din, dout);
  if (int rc = check_xy("X", X, ldx, din)) return rc;
  if (int rc = check_xy("G", G, ldg, dout)) return rc;
  if (!dW) return sagnn::fail(SAGNN_ERR_NULL, "dW is NULL");
  if (n == 0) return SAGNN_OK;
  return sagnn::dense_tn_any(X, ldx, G, ldg, n, din, dout, dW, dout, db, static_cast<hipStream_t>(stream));
}

extern "C" int sagnn_dense_tn_seg_f32(const float* X, int64_t ldx, int64_t x_seg, const float* G, int64_t ldg, int64_t g_seg,
                                      int64_t seg_rows, int n_seg, int din, int dout, float* dW, float* db, void* stream) {
  if (seg_rows < 0 || n_seg < 0 || din < 32 || dout < 32 || (din & 31) || (dout & 31))
    return sagnn::fail(SAGNN_ERR_DIM, "dense_tn_seg: need din, dout multiples of 32 (got %d, %d)", din, dout);
  if (int rc = check_xy("X", X, ldx, din)) return rc;
  if (int rc = check_xy("G", G, ldg, dout)) return rc;
  if ((x_seg & 3) || (g_seg & 3)) return sagnn::fail(SAGNN_ERR_ALIGN, "dense_tn_seg: segment strides must be multiples of 4 floats");
  if (!dW) return sagnn::fail(SAGNN_ERR_NULL, "dW is NULL");
  return sagnn::gemm_tn(X, ldx, x_seg, G, ldg, g_seg, seg_rows, n_seg, din, dout, dW, dout, db, static_cast<hipStream_t>(stream));
}
