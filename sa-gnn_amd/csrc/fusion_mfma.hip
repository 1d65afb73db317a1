// Interval fusion on the matrix cores (gfx950, exact-fp32 MFMA v_mfma_f32_32x32x2_f32).
//
// The two GEMM-shaped stages of reference model.py:135-155 — the BasicLSTMCell gate product
// [x_t | h] @ W[2d,4d] and the three dense layers of MultiHeadSelfAttention — run on MFMA tiles;
// fp32 in / fp32 accumulate is bit-for-bit an fmaf chain, so the 1e-4 parity bar holds (bf16
// would not). Each wavefront owns 32 rows and is independent of the others after the one-off
// weight load: one wave per SIMD, the whole 512-register budget, no block barriers in the loop.
//
//   LDS image of a weight matrix ("fragment order"): for k-step kk, half hf, lane l, e in 0..3:
//       Wf[((kk*HF + hf)*64 + l)*4 + e] = W[2kk + (l>>5)][(4hf + e)*32 + (l&31)]
//   so the B operands of 4 column tiles arrive with ONE conflict-free ds_read_b128 per lane.
//   A operands (x_t, h, layer-norm input) are staged per wave in a [32][D] tile whose column is
//   XOR-swizzled with the row, which makes both the row-major fill and the column-strided
//   A-layout read (lane = row) bank-conflict free.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 waves = one per SIMD
constexpr int kRowsPerWave = 32;
constexpr int kRowsPerBlock = 128;

__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __expf(-x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = 1 - 2/(1 + exp(2x)); saturates cleanly (exp -> inf gives 1, exp -> 0 gives -1)
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x));
}

// row of C/D register r in lane half rh for a 32x32 MFMA tile (cdna_hip_programming.md §3)
__device__ __forceinline__ int crow(int r, int rh) { return (r & 3) + 8 * (r >> 2) + 4 * rh; }

// Copies W [K][NC] (row-major, K even, NC a multiple of 128) into fragment order.
template <int NC>
__device__ __forceinline__ void load_weight_fragments(float* __restrict__ Wf,
                                                      const float* __restrict__ W, int K) {
  constexpr int HF = NC / 128;
  const int total = K * NC;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int e = idx & 3;
    const int l = (idx >> 2) & 63;
    const int rest = idx >> 8;
    const int hf = rest % HF;
    const int kk = rest / HF;
    Wf[idx] = W[(size_t)(2 * kk + (l >> 5)) * NC + (4 * hf + e) * 32 + (l & 31)];
  }
}

// acc[4*HF tiles] += A[32 x 2*KS] @ Wf[k-steps kbase .. kbase+KS), one 32x32x2 MFMA per tile and
// k-step. The B fragments of k-step kk+1 are requested before the MFMAs of k-step kk and a
// scheduling barrier closes every k-step, so exactly two fragment sets are live (left alone the
// scheduler hoists all 2*KS ds_read_b128 and spills).
template <int KS, int HF>
__device__ __forceinline__ void mfma_half(f32x16 (&acc)[4 * HF], const float (&a)[KS],
                                          const float* __restrict__ Wf, int kbase, int lane) {
  float4 cur[HF], nxt[HF];
#pragma unroll
  for (int hf = 0; hf < HF; ++hf)
    cur[hf] = *reinterpret_cast<const float4*>(Wf + ((kbase * HF + hf) * 64 + lane) * 4);
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    if (kk + 1 < KS) {
#pragma unroll
      for (int hf = 0; hf < HF; ++hf)
        nxt[hf] = *reinterpret_cast<const float4*>(Wf + (((kbase + kk + 1) * HF + hf) * 64 + lane) * 4);
    }
#pragma unroll
    for (int hf = 0; hf < HF; ++hf) {
      acc[4 * hf + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], cur[hf].x, acc[4 * hf + 0], 0, 0, 0);
      acc[4 * hf + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], cur[hf].y, acc[4 * hf + 1], 0, 0, 0);
      acc[4 * hf + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], cur[hf].z, acc[4 * hf + 2], 0, 0, 0);
      acc[4 * hf + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], cur[hf].w, acc[4 * hf + 3], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int hf = 0; hf < HF; ++hf) cur[hf] = nxt[hf];
  }
}

// ---------------------------------------------------------------------------------------------
// LSTM: TF 1.14 BasicLSTMCell over T steps, zero initial state (reference model.py:135-146).
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kBlock, 1) void lstm_fwd_mfma_kernel(
    const float* __restrict__ x, int64_t ld_n, int64_t ld_t, int64_t n, int t,
    const float* __restrict__ W, const float* __restrict__ bias, float forget_bias,
    const float* __restrict__ drop, float* __restrict__ h_out, int64_t ld_h, int64_t n_tiles) {
  constexpr int NC = 4 * D;        // gate columns
  constexpr int CT = NC / 32;      // column tiles (8 at D=64)
  constexpr int HF = CT / 4;       // b128 reads per k-step
  constexpr int KS = D / 2;        // k-steps per operand half (x or h)
  constexpr int HT = D / 32;       // hidden-unit tiles
  constexpr int LPR = D / 4;       // lanes per row in the coalesced fill
  constexpr int RPI = kWave / LPR; // rows per fill instruction
  constexpr int NFILL = kRowsPerWave / RPI;

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Wf = lds;                                            // 2D x 4D floats
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* stage = lds + 2 * D * NC + wave * (kRowsPerWave * D);  // private [32][D]
  const int ai = lane & 31, kh = lane >> 5;                   // A layout: row, k parity
  const int cj = lane & 31, rh = lane >> 5;                   // C layout: column, row half

  load_weight_fragments<NC>(Wf, W, 2 * D);
  __syncthreads();

  float bcol[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) bcol[ct] = bias[ct * 32 + cj];
  const int fr = lane / LPR, fc4 = (lane % LPR) * 4;           // fill: row-in-group, column

  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * kRowsPerBlock + (int64_t)wave * kRowsPerWave;
    if (row0 >= n) continue;  // wave-uniform
    float c[HT][16];
    float a_h[KS];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[ht][r] = 0.f;

    float4 xr[NFILL];
    auto fetch_x = [&](int ts) {
#pragma unroll
      for (int q = 0; q < NFILL; ++q) {
        const int64_t row = row0 + q * RPI + fr;
        xr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < n) xr[q] = *reinterpret_cast<const float4*>(x + row * ld_n + (int64_t)ts * ld_t + fc4);
      }
    };
    fetch_x(0);

    for (int ts = 0; ts < t; ++ts) {
      // Lane-derived LDS offsets are recomputed every step (a handful of VALU ops against 256+
      // MFMAs): left loop-invariant, the compiler hoists ~100 swizzled addresses out of the
      // step loop and spills them.
      int ai_ = ai, kh_ = kh, cj_ = cj, rh_ = rh, fr_ = fr, fc4_ = fc4;
      asm volatile("" : "+v"(ai_), "+v"(kh_), "+v"(cj_), "+v"(rh_), "+v"(fr_), "+v"(fc4_));

      // ---- stage x_t through LDS into the A layout --------------------------------------
      float a_x[KS];
#pragma unroll
      for (int q = 0; q < NFILL; ++q) {
        const int r = q * RPI + fr_;
        const int sw = r & 31;
        float* dst = stage + r * D;
        dst[(fc4_ + 0) ^ sw] = xr[q].x;
        dst[(fc4_ + 1) ^ sw] = xr[q].y;
        dst[(fc4_ + 2) ^ sw] = xr[q].z;
        dst[(fc4_ + 3) ^ sw] = xr[q].w;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) a_x[kk] = stage[ai_ * D + ((2 * kk + kh_) ^ ai_)];
      if (ts + 1 < t) fetch_x(ts + 1);  // in flight under the MFMAs below

      // ---- gates = x_t @ W[0:D] + h @ W[D:2D]  (bias joins in the gate math) -------------
      f32x16 acc[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
      mfma_half<KS, HF>(acc, a_x, Wf, 0, lane);
      if (ts > 0)  // h_0 = 0: the recurrent half contributes nothing at the first step
        mfma_half<KS, HF>(acc, a_h, Wf, KS, lane);

      // ---- gate math in the C layout (columns i | j | f | o, each D wide) ----------------
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float gi = acc[ht][r] + bcol[ht], gj = acc[HT + ht][r] + bcol[HT + ht];
          const float gf = acc[2 * HT + ht][r] + bcol[2 * HT + ht], go = acc[3 * HT + ht][r] + bcol[3 * HT + ht];
          const float cn = c[ht][r] * fast_sigmoid(gf + forget_bias) + fast_sigmoid(gi) * fast_tanh(gj);
          const float hn = fast_tanh(cn) * fast_sigmoid(go);
          c[ht][r] = cn;
          const int row = crow(r, rh_);
          const int col = ht * 32 + cj_;
          stage[row * D + (col ^ row)] = hn;  // for the next step's A operand
          const int64_t grow = row0 + row;
          if (grow < n) {
            float hv = hn;
            if (drop) hv *= drop[grow * (int64_t)t * D + (int64_t)ts * D + col];
            h_out[grow * ld_h + (int64_t)ts * D + col] = hv;
          }
          // keep the scheduler from interleaving all 16*HT chains (register pressure)
          if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (ts + 1 < t) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) a_h[kk] = stage[ai_ * D + ((2 * kk + kh_) ^ ai_)];
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

}  // namespace

namespace sagnn {

bool lstm_mfma_supported(int d) { return d == 32 || d == 64; }

template <int D>
static int launch_lstm_mfma(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t,
                            const float* W, const float* b, float forget_bias, const float* drop,
                            float* h, int64_t ld_h, hipStream_t s) {
  const size_t lds = (size_t)(2 * D * 4 * D + 4 * kRowsPerWave * D) * sizeof(float);
  static bool configured = false;
  if (!configured) {
    SAGNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_mfma_kernel<D>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = true;
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const int64_t n_tiles = (n + kRowsPerBlock - 1) / kRowsPerBlock;
  const int64_t blocks = n_tiles < cus ? n_tiles : cus;
  ProfileScope prof(kProfLstm, s, n, t);
  hipLaunchKernelGGL(lstm_fwd_mfma_kernel<D>, dim3((unsigned)blocks), dim3(kBlock), lds, s, x, ld_n, ld_t,
                     n, t, W, b, forget_bias, drop, h, ld_h, n_tiles);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

int lstm_fwd_mfma(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, const float* W,
                  const float* b, float forget_bias, const float* drop, float* h, int64_t ld_h,
                  hipStream_t s) {
  if (d == 64) return launch_lstm_mfma<64>(x, ld_n, ld_t, n, t, W, b, forget_bias, drop, h, ld_h, s);
  if (d == 32) return launch_lstm_mfma<32>(x, ld_n, ld_t, n, t, W, b, forget_bias, drop, h, ld_h, s);
  return fail(SAGNN_ERR_DIM, "MFMA LSTM supports d = 32 or 64, got %d", d);
}

}  // namespace sagnn
