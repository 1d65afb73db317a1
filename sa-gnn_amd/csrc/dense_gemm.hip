// Tiled fp32 GEMMs on v_mfma_f32_32x32x2_f32 for dense products whose weight block does not fit LDS (gfx950).
//
// dense.hip keeps W resident in LDS and streams the tall operand past it — right for [d, 3d] projections at d <= 64,
// wrong for the BPTT products of the d = 128 LSTM (reference model.py:135-146 differentiated; BASELINE config 3):
//   nn:  d[x|h] [n, 256]  = dG [n, 512] @ W^T [512, 256]
//   tn:  dW     [128, 512] += x_t^T [128, n] @ dG [n, 512]        (and the same with h_{t-1})
// whose W is 512 KB. dense_nn_any / dense_tn_any cut those into LDS-fitting pieces (4 launches per product, every piece
// re-reading the tall operand): 17 TFLOP/s on the MovieLens-shaped training step. Here: a 128 x 128 block tile, four waves
// of 64 x 64 (2 x 2 MFMA tiles, 64 accumulator registers), K in steps of 32 through double-buffered LDS tiles, one
// barrier per step, the next step's global loads in flight under the MFMAs.
//   nn  A [M, K] row-major: staged as [128][32 + 1] (a lane of the A operand reads one row: the odd stride spreads the
//       rows over the banks); B [K, N]: staged as [32][128], a lane reads consecutive columns.
//   tn  both operands are read along K as rows of [32][128] tiles (X [K, M], G [K, N]); the reduction dimension (the n
//       rows) is split across blockIdx.y and the partial tiles meet in dW by float atomics (dW is accumulated into, as
//       sagnn_dense_tn_f32 documents); db = column sums of G ride with the blocks of the first row tile.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 128, BK = 32, kThreads = 256;
constexpr int NLB = BK * BN / 4 / kThreads;   // float4 per thread and [BK][128] operand tile
constexpr int AS = BK + 1;

__device__ __forceinline__ int crow(int r, int rh) { return (r & 3) + 8 * (r >> 2) + 4 * rh; }   // C row of accumulator register r

// TM = 32-row MFMA tiles per wave: 2 (block tile 128 x 128) or 1 (64 x 128: twice the blocks when 128-row tiles would
// leave CUs with one block and others with two — the launches here are a single wave of blocks)
template <int TM>
__global__ __launch_bounds__(kThreads, 2) void gemm_nn_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                              int64_t ldb, const float* __restrict__ bias, float* __restrict__ C,
                                                              int64_t ldc, int64_t M, int N, int K, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BMT = 64 * TM;                     // rows of the block tile
  constexpr int NLA = BMT * BK / 4 / kThreads;
  float* const Bs0 = lds;                          // [2][BK * BN]
  float* const As0 = lds + 2 * BK * BN;            // [2][BMT * AS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * BMT;
  const int col0 = (int)blockIdx.y * BN;

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[NLA], rb[NLB];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int idx = tid + i * kThreads;
      const int r = idx / (BK / 4), k4 = idx % (BK / 4);           // A: 128 rows x BK / 4 float4
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < M) ra[i] = *reinterpret_cast<const float4*>(A + (row0 + r) * lda + k0 + 4 * k4);
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int idx = tid + i * kThreads;
      const int kr = idx >> 5, c4 = idx & 31;                       // B: BK rows x 32 float4
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col0 + 4 * c4 < N) rb[i] = *reinterpret_cast<const float4*>(B + (int64_t)(k0 + kr) * ldb + col0 + 4 * c4);
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int idx = tid + i * kThreads;
      const int r = idx / (BK / 4), k4 = idx % (BK / 4);
      float* a = As0 + buf * (BMT * AS) + r * AS + 4 * k4;
      a[0] = ra[i].x, a[1] = ra[i].y, a[2] = ra[i].z, a[3] = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int idx = tid + i * kThreads;
      const int kr = idx >> 5, c4 = idx & 31;
      *reinterpret_cast<float4*>(Bs0 + buf * (BK * BN) + kr * BN + 4 * c4) = rb[i];
    }
  };
  fetch(0);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += BK) {
    if (k0 + BK < K) fetch(k0 + BK);
    const float* as = As0 + buf * (BMT * AS) + (wm * 32 * TM + l31) * AS + kh;
    const float* bs = Bs0 + buf * (BK * BN) + kh * BN + wn * 64 + l31;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float b0 = bs[2 * kk * BN], b1 = bs[2 * kk * BN + 32];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float a = as[i * 32 * AS + 2 * kk];
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[i][1], 0, 0, 0);
      }
    }
    if (k0 + BK < K) stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col0 + wn * 64 + j * 32 + l31;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wm * 32 * TM + i * 32 + crow(r, kh);
        if (row < M) {
          float* c = C + row * ldc + col;
          const float v = acc[i][j][r] + bv;
          *c = accumulate ? *c + v : v;
        }
      }
    }
}

// C [M, N] += X^T G over rows [blockIdx.y * k_per, ...) of X [K, M] / G [K, N]; db [N] += column sums of G (blocks of row tile 0).
// The K rows come in segments of seg_rows: row k = (s, i) lives at X + s * xseg + i * ldx and G + s * gseg + i * ldg (one
// segment = a plain matrix; the BPTT weight gradient = one segment per step, x node-major or time-major, dG time-major).
__global__ __launch_bounds__(kThreads, 2) void gemm_tn_kernel(const float* __restrict__ X, int64_t ldx, int64_t xseg,
                                                              const float* __restrict__ G, int64_t ldg, int64_t gseg,
                                                              uint32_t seg_rows, float* __restrict__ C, int64_t ldc,
                                                              float* __restrict__ db, int M, int N, int64_t K, int64_t k_per,
                                                              int n_col_tiles) {
  __shared__ __attribute__((aligned(16))) float Xs[2][BK * BM];
  __shared__ __attribute__((aligned(16))) float Gs[2][BK * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5;
  const int mt = (int)blockIdx.x / n_col_tiles, nt = (int)blockIdx.x - mt * n_col_tiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int64_t kb = (int64_t)blockIdx.y * k_per;
  const int64_t ke = kb + k_per < K ? kb + k_per : K;
  if (kb >= ke) return;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 colsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool want_db = db != nullptr && mt == 0;

  float4 rx[NLB], rg[NLB];
  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int idx = tid + i * kThreads;
      const int kr = idx >> 5, c4 = idx & 31;
      rx[i] = rg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + kr < ke) {
        const uint32_t k = (uint32_t)(k0 + kr), sg = k / seg_rows, ri = k - sg * seg_rows;   // K < 2^31 (host)
        if (m0 + 4 * c4 < M) rx[i] = *reinterpret_cast<const float4*>(X + sg * xseg + ri * ldx + m0 + 4 * c4);
        if (n0 + 4 * c4 < N) rg[i] = *reinterpret_cast<const float4*>(G + sg * gseg + ri * ldg + n0 + 4 * c4);
      }
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int idx = tid + i * kThreads;
      const int kr = idx >> 5, c4 = idx & 31;
      *reinterpret_cast<float4*>(Xs[buf] + kr * BM + 4 * c4) = rx[i];
      *reinterpret_cast<float4*>(Gs[buf] + kr * BN + 4 * c4) = rg[i];
      if (want_db) colsum.x += rg[i].x, colsum.y += rg[i].y, colsum.z += rg[i].z, colsum.w += rg[i].w;
    }
  };
  fetch(kb);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (int64_t k0 = kb; k0 < ke; k0 += BK) {
    if (k0 + BK < ke) fetch(k0 + BK);
    const float* xs = Xs[buf] + kh * BM + wm * 64 + l31;
    const float* gs = Gs[buf] + kh * BN + wn * 64 + l31;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float a0 = xs[2 * kk * BM], a1 = xs[2 * kk * BM + 32];
      const float b0 = gs[2 * kk * BN], b1 = gs[2 * kk * BN + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (k0 + BK < ke) stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + crow(r, kh);
        if (row < M) atomicAdd(C + (int64_t)row * ldc + col, acc[i][j][r]);
      }
    }
  if (want_db) {   // the 8 threads that staged column group c4 (tid / 32 = 0 .. 7) meet in LDS; Xs is free after the loop's last barrier
    float4* part = reinterpret_cast<float4*>(Xs[0]);
    part[tid] = colsum;
    __syncthreads();
    if (tid < 32) {
      float4 sacc = part[tid];
#pragma unroll
      for (int g = 1; g < 8; ++g) {
        const float4 p = part[tid + 32 * g];
        sacc.x += p.x, sacc.y += p.y, sacc.z += p.z, sacc.w += p.w;
      }
      const int col = n0 + 4 * tid;
      if (col < N) {
        atomicAdd(db + col, sacc.x);
        atomicAdd(db + col + 1, sacc.y);
        atomicAdd(db + col + 2, sacc.z);
        atomicAdd(db + col + 3, sacc.w);
      }
    }
  }
}

}  // namespace

namespace sagnn {

// Y (+)= X [n, din] @ W [din, dout] (+ bias); din a multiple of 32, dout of 4, 16-byte aligned rows (checked by the callers)
int gemm_nn(const float* X, int64_t ldx, int64_t n, int din, int dout, const float* W, int64_t ldw, const float* bias, float* Y,
            int64_t ldy, int accumulate, hipStream_t s) {
  const int by = (dout + BN - 1) / BN;
  const int64_t slots = 2 * (int64_t)cu_count_current();
  const int64_t b128 = (n + 127) / 128 * by;
  // 128-row tiles when they fill the chip several times over; otherwise 64-row tiles (twice the blocks, better balance)
  const bool small = b128 < 4 * slots;
  const int64_t bx = small ? (n + 63) / 64 : (n + 127) / 128;
  if (bx > INT32_MAX) return fail(SAGNN_ERR_ARG, "gemm_nn: grid too large");
  const size_t lds = (size_t)2 * (BK * BN + (small ? 64 : 128) * AS) * sizeof(float);
  if (small) {
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_nn_kernel<1>), lds)) return rc;
    hipLaunchKernelGGL(gemm_nn_kernel<1>, dim3((unsigned)bx, (unsigned)by), dim3(kThreads), lds, s, X, ldx, W, ldw, bias, Y, ldy, n, dout,
                       din, accumulate);
  } else {
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_nn_kernel<2>), lds)) return rc;
    hipLaunchKernelGGL(gemm_nn_kernel<2>, dim3((unsigned)bx, (unsigned)by), dim3(kThreads), lds, s, X, ldx, W, ldw, bias, Y, ldy, n, dout,
                       din, accumulate);
  }
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

// dW [din, dout] += X^T @ G over n_seg segments of seg_rows rows (row i of segment s: X + s * xseg + i * ldx, G alike),
// db [dout] += column sums of G (db may be NULL)
int gemm_tn(const float* X, int64_t ldx, int64_t xseg, const float* G, int64_t ldg, int64_t gseg, int64_t seg_rows, int64_t n_seg,
            int din, int dout, float* dW, int64_t lddw, float* db, hipStream_t s) {
  const int64_t n = seg_rows * n_seg;
  if (n <= 0) return SAGNN_OK;
  if (n > INT32_MAX) return fail(SAGNN_ERR_ARG, "gemm_tn: %lld rows in one call (limit 2^31 - 1)", (long long)n);
  const int mt = (din + BM - 1) / BM, nt = (dout + BN - 1) / BN;
  // split the n rows so that the launch fills the chip twice over; every split a multiple of BK rows
  int64_t splits = (2 * (int64_t)cu_count_current() + mt * nt - 1) / (mt * nt);
  const int64_t max_splits = (n + 4 * BK - 1) / (4 * BK);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int64_t k_per = (n + splits - 1) / splits;
  k_per = (k_per + BK - 1) / BK * BK;
  splits = (n + k_per - 1) / k_per;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(mt * nt), (unsigned)splits), dim3(kThreads), 0, s, X, ldx, xseg, G, ldg, gseg,
                     (uint32_t)seg_rows, dW, lddw, db, din, dout, n, k_per, nt);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

}  // namespace sagnn
