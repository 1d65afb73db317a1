// Interval LSTM on the f16 matrix cores with fp32-grade operands (gfx950).
//
// TF 1.14 BasicLSTMCell over T steps (reference model.py:135-146): gates = [x_t | h] @ W[2d,4d] + b,
// i, j, f, o = split(gates); c' = c sigmoid(f + fb) + sigmoid(i) tanh(j); h' = tanh(c') sigmoid(o).
//
// Same decomposition as lstm_split_kernel.h (transposed product, W slices register-resident, x_t / h as
// 16-bit images in LDS, gate math hand-interleaved with the MFMAs) with HALF the matrix-core work: the
// two-piece f16 split of f16_split.h (three piece products into a head and a residual accumulator that the
// gate math joins with one fma) instead of three bf16 pieces and six products — the same accuracy class
// at 3/6 of the MFMAs and 2/3 of the LDS images and weight registers.
//
// f16 has 5 exponent bits, so the split holds inside a window (f16_split.h, RANGE): |v| < 32768, and no 4-element
// segment that is non-zero but below 2^-18 as a whole. Every x / initial-h value passes through a RangeTrack on
// its way into LDS (W when the fragments are built), and a workgroup that saw a value outside the window
// re-evaluates its 96-row tile after the fast pass with plain fp32 fmaf chains (the block after the step loop),
// from the tile's ORIGINAL initial state (c_final is written once, by whichever pass is the last): an input row of
// 1e-30 or of 1e30 gets the results of an fp32 evaluation. The fast path carries five extra VALU operations per
// 16 bytes of x for it.
#include <type_traits>
#include <utility>

#pragma once
#include "common.h"
#include "f16_split.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

#ifndef SAGNN_LSTM_F16_BT
#define SAGNN_LSTM_F16_BT 6
#endif
constexpr int kBT = SAGNN_LSTM_F16_BT;   // batch tiles of 16 rows per workgroup tile
static_assert(kBT % 2 == 0 && kBT >= 4, "x passes ride two per tile on the last kBT / 2 tiles");
constexpr int kRows = 16 * kBT;          // x and h both double-buffered in LDS: 8 images = 96 KB at d = 64, kBT = 6

// 16-byte slot swizzle of the [row][D] 16-bit images (as in lstm_split_kernel.h)
template <int D>
__device__ __forceinline__ int swz(int row) {
  if (D == 64) return (row >> 1) & 7;
  const int g = (row >> 2) & 3;
  return (0x78 >> (2 * g)) & 3;
}

template <int LO, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, LO + I>{}), ...);
}
template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (HI > LO) static_for_impl<LO>(static_cast<F&&>(f), std::make_integer_sequence<int, HI - LO>{});
}

// The interleaved operation list of `step`: kGateOps gate-math operations of one batch tile followed by
// 2 x kXOps operations that split two 16-row passes of the next step's x into their LDS images. The ORDER in
// which they are issued and the MFMA gap each one goes to come from a list scheduler over their dependency
// graph (tools/gen_lstm_schedule.py): at most one transcendental per gap, an even share of the issue cycles,
// and nothing reads a result of its own gap.
constexpr int kGateOps = 118, kXOps = 16;
#include "lstm_f16_schedule.inc"
template <int NM, bool XO>
__host__ __device__ constexpr int sched_start(int i) {
  if (NM == 48) return XO ? kStart48X[i] : kStart48[i];
  if (NM == 24) return XO ? kStart24X[i] : kStart24[i];
  return XO ? kStart12X[i] : kStart12[i];
}
template <int NM, bool XO>
__host__ __device__ constexpr int sched_op(int pos) {
  if (NM == 48) return XO ? kOrder48X[pos] : kOrder48[pos];
  if (NM == 24) return XO ? kOrder24X[pos] : kOrder24[pos];
  return XO ? kOrder12X[pos] : kOrder12[pos];
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float sigmoid_e2(float z) {   // the fast path's own formulas
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * z));
}

template <int D, bool SAVE, bool DROP>
__global__ __launch_bounds__(64 * (D / 16), 1) void lstm_fwd_f16_kernel(
    const float* __restrict__ x, int64_t ld_n, int64_t ld_t, int64_t n, int t, const float* __restrict__ W,
    const float* __restrict__ bias, float forget_bias, const float* __restrict__ drop, float* __restrict__ h_out,
    int64_t ld_h, float* __restrict__ gates_out, float* __restrict__ c_out, int64_t n_tiles,
    const float* __restrict__ h_init, int64_t ld_hi, const float* __restrict__ c_init, float* __restrict__ c_final,
    unsigned int* __restrict__ redo_ctr) {
  constexpr int NW = D / 16;          // waves per workgroup
  constexpr int NT = 64 * NW;         // threads
  constexpr int KSH = D / 32;         // k-steps (of 32) per operand half
  constexpr int KS = 2 * KSH;
  constexpr int NC = 4 * D;
  constexpr int PLANE = kRows * D * 2;          // bytes of one f16 image
  constexpr int LPR = D / 4;                    // threads per row in the fill (float4 each)
  constexpr int RPP = NT / LPR;                 // rows per fill pass (16)
  constexpr int NFILL = kRows / RPP;            // kBT
  constexpr float kL2E = 1.44269504088896340736f;

  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const Xp = lds;                         // 2 x 2 images
  char* const Hp = lds + 4 * PLANE;             // 2 x 2 images
  int* const flags = reinterpret_cast<int*>(lds + 8 * PLANE);   // [0], [1]: input outside the split's window seen in the tile of that parity; [2]: in W
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, q = lane >> 4;
  const int hid0 = 16 * wave + 4 * q;           // first of this lane's 4 hidden units (C rows 4q + r)
  const int fr = tid / LPR, fc4 = (tid % LPR) * 4;
  float k4096 = 4096.f;
  asm volatile("" : "+v"(k4096));               // one register for the whole kernel, not a literal per use

  if (tid < 3) flags[tid] = 0;
  __syncthreads();

  // ---- this wave's W slice as A fragments: A[mm = lane & 15][k = 32 ks + 8 q + j] = W[k][g D + 16 wave + mm],
  // heads in wf[..][0], scaled residuals in wf[..][1]
  i32x4 wf[4][KS][2];
  {
    RangeTrack wr = range_init();
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)  // column scale of the gate's non-linearity folded into W (see bc below)
          wv[j] = W[(size_t)(32 * ks + 8 * q + j) * NC + g * D + 16 * wave + m] * (g == 1 ? 2.f * kL2E : -kL2E);
        range_seg4(wr, wv[0], wv[1], wv[2], wv[3]);
        range_seg4(wr, wv[4], wv[5], wv[6], wv[7]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int hd = head2(wv[2 * e], wv[2 * e + 1]);
          wf[g][ks][0][e] = hd;
          wf[g][ks][1][e] = tail_hi(tail_lo(resid<0>(hd, wv[2 * e]), k4096), resid<1>(hd, wv[2 * e + 1]), k4096);
        }
      }
    if (range_bad(wr)) flags[2] = 1;      // ordered before its first reader by the tile loop's barriers
  }
  // Gate non-linearities are evaluated as exp2(t), t = k (pre-activation + bias), k = -log2(e) for the
  // sigmoids and 2 log2(e) for tanh(j): k is folded into this wave's columns of W and k * bias is what the
  // head accumulators start from, so t leaves the MFMAs ready (after the two accumulators are joined).
  f32x4 bc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float b = bias[g * D + hid0 + r] + (g == 2 ? forget_bias : 0.f);
      bc[g][r] = (g == 1 ? 2.f * kL2E : -kL2E) * b;
    }

  // one [row][4 floats] piece set -> two 8-byte LDS writes
  auto write_pieces = [&](char* img, int row, int col4, float4 v) {
    const int p0 = head2(v.x, v.y), p1 = head2(v.z, v.w);
    const int s0 = tail_hi(tail_lo(resid<0>(p0, v.x), k4096), resid<1>(p0, v.y), k4096);
    const int s1 = tail_hi(tail_lo(resid<0>(p1, v.z), k4096), resid<1>(p1, v.w), k4096);
    const int off = row * (D * 2) + ((((col4 >> 3)) ^ swz<D>(row)) << 4) + ((col4 >> 2) & 1) * 8;
    *reinterpret_cast<i32x2*>(img + off) = i32x2{p0, p1};
    *reinterpret_cast<i32x2*>(img + PLANE + off) = i32x2{s0, s1};
  };

  int par = 0;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, par ^= 1) {
    const int64_t row0 = tile * kRows;
    const int rows_valid = (int)(n - row0 < kRows ? n - row0 : kRows);
    if (tid == 0) flags[par] = 0;   // two tiles (and their barriers) after its last reader
    RangeTrack xrng = range_init();  // range of the x / h_init values this thread moved into LDS
    float4 xr[NFILL];
    auto fetch_x = [&](int ts) {
#pragma unroll
      for (int p = 0; p < NFILL; ++p) {
        const int r = p * RPP + fr;
        xr[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows_valid) xr[p] = *reinterpret_cast<const float4*>(x + (row0 + r) * ld_n + (int64_t)ts * ld_t + fc4);
      }
    };
    fetch_x(0);

    // rows past n are dropped by the descriptors' range check
    const auto rs_h = __builtin_amdgcn_make_buffer_rsrc(h_out + row0 * ld_h, 0, rows_valid * (int)ld_h * 4, 0x00020000);
    const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(SAVE ? gates_out + row0 * t * NC : h_out, 0,
                                                        SAVE ? rows_valid * t * NC * 4 : 0, 0x00020000);
    const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(SAVE ? c_out + row0 * t * D : h_out, 0,
                                                        SAVE ? rows_valid * t * D * 4 : 0, 0x00020000);
    const auto rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(drop ? drop + row0 * t * D : x), 0,
                                                        drop ? rows_valid * t * D * 4 : 0, 0x00020000);
    const auto rs_ci = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c_init ? c_init + row0 * D : x), 0,
                                                         c_init ? rows_valid * D * 4 : 0, 0x00020000);
    const auto rs_cf = __builtin_amdgcn_make_buffer_rsrc(c_final ? c_final + row0 * D : h_out, 0,
                                                         c_final ? rows_valid * D * 4 : 0, 0x00020000);

    f32x4 c[kBT];
#pragma unroll
    for (int bt = 0; bt < kBT; ++bt) {
      c[bt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (h_init) c[bt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_ci, ((bt * 16 + m) * D + hid0) * 4, 0, 0));
    }
    if (h_init) {  // continue from a given state: its pieces are step 0's recurrent operand (buffer 0)
#pragma unroll
      for (int p = 0; p < NFILL; ++p) {
        const int r = p * RPP + fr;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows_valid) v = *reinterpret_cast<const float4*>(h_init + (row0 + r) * ld_hi + fc4);
        range_seg4_hi(xrng, v.x, v.y, v.z, v.w);   // h: top of the window only (f16_split.h)
        write_pieces(Hp, r, fc4, v);
      }
    }
#pragma unroll
    for (int p = 0; p < NFILL; ++p) {
      range_seg4(xrng, xr[p].x, xr[p].y, xr[p].z, xr[p].w);
      write_pieces(Xp, p * RPP + fr, fc4, xr[p]);
    }
    lds_barrier();

    // One step, hand-scheduled: the MFMAs of batch tile bt (3 piece products x 4 gates x k-steps, a head
    // and a residual accumulator chain per gate) are issued one at a time, and after each one a slice of
    // the gate math of tile bt-1 — a fixed list of single-instruction operations — is emitted, closed by a
    // scheduling barrier so the order survives the compiler. B fragments are requested one k-step ahead.
    auto step = [&](auto recur_c, int ts) {
      constexpr bool RECUR = decltype(recur_c)::value;
      constexpr int KSN = RECUR ? KS : KSH;           // k-steps per tile
      constexpr int NM = 12 * KSN;                    // MFMAs per tile
      const char* const Xcur = Xp + (ts & 1) * 2 * PLANE;
      const char* const Hcur = Hp + (ts & 1) * 2 * PLANE;
      char* const Hnxt = Hp + ((ts & 1) ^ 1) * 2 * PLANE;
      char* const Xnxt = Xp + ((ts & 1) ^ 1) * 2 * PLANE;
      // lane-derived LDS offsets recomputed per step (left loop-invariant the compiler hoists and spills them)
      int m_ = m, q_ = q, fr_ = fr, fc4_ = fc4;
      asm volatile("" : "+v"(m_), "+v"(q_), "+v"(fr_), "+v"(fc4_));
      const int hid = 16 * wave + 4 * q_;

      f32x4 ahi[4], alo[4];            // tile in flight
      int xw0[2][2], xw1[2][2], xoff[2];   // pieces of the two x passes being written
      float xq[2][4], xs[2];
      uint32_t xsb[2];
      f32x4 ga[4], gl[4];              // pre-activations of the tile whose gate math is being interleaved
      f32x4 dv;                        // dropout scale of that tile
      float tt[4][4], pr[4], cn[4], u[4], hn[4], hv[4], r1[4];
      int w0[2], w1[2], hoff;
      i32x4 bf[2][2];                  // B fragments, double-buffered across k-steps

      auto read_b = [&](int bt, int ks, i32x4 (&dst)[2]) {
        const int row = bt * 16 + m_;
        const char* img = ks < KSH ? Xcur : Hcur;
        const int off = row * (D * 2) + ((((ks % KSH) * 4 + q_) ^ swz<D>(row)) << 4);
        dst[0] = *reinterpret_cast<const i32x4*>(img + off);
        dst[1] = *reinterpret_cast<const i32x4*>(img + PLANE + off);
      };
      // operation K of the gate math of tile PB (state in ga / tt / ...): one instruction each, more or less
      auto gate_op = [&](auto pb_c, auto k_c) {
        constexpr int PB = decltype(pb_c)::value, K = decltype(k_c)::value;
        const int row = PB * 16 + m_;
        if constexpr (K == 0) {
          const int e_td = (row * t + ts) * D + hid;
          if constexpr (DROP) dv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, e_td * 4, 0, 0));
        } else if constexpr (K < 17) {          // the two accumulators joined: t = head + 2^-12 residual
          constexpr int k = K - 1, g = k >> 2, r = k & 3;
          ga[g][r] = fmaf(gl[g][r], kLoInv, ga[g][r]);
        } else if constexpr (K < 33) {
          constexpr int k = K - 17, g = k >> 2, r = k & 3;
          tt[g][r] = __builtin_amdgcn_exp2f(ga[g][r]);
        } else if constexpr (K < 49) {
          constexpr int k = K - 33, g = k >> 2, r = k & 3;
          tt[g][r] = 1.f + tt[g][r];
        } else if constexpr (K < 65) {
          constexpr int k = K - 49, g = k >> 2, r = k & 3;
          tt[g][r] = __builtin_amdgcn_rcpf(tt[g][r]);       // sigmoid(i), 1/(1+e^2j), sigmoid(f), sigmoid(o)
        } else if constexpr (K < 69) {
          tt[1][K - 65] = fmaf(-2.f, tt[1][K - 65], 1.f);     // tanh(j)
        } else if constexpr (K < 73) {
          pr[K - 69] = tt[0][K - 69] * tt[1][K - 69];
        } else if constexpr (K < 77) {
          cn[K - 73] = fmaf(c[PB][K - 73], tt[2][K - 73], pr[K - 73]);
        } else if constexpr (K < 81) {
          u[K - 77] = cn[K - 77] * (2.f * kL2E);
        } else if constexpr (K < 85) {
          u[K - 81] = __builtin_amdgcn_exp2f(u[K - 81]);
        } else if constexpr (K < 89) {
          u[K - 85] = 1.f + u[K - 85];
        } else if constexpr (K < 93) {
          u[K - 89] = __builtin_amdgcn_rcpf(u[K - 89]);
        } else if constexpr (K < 97) {
          u[K - 93] = fmaf(-2.f, u[K - 93], 1.f);             // tanh(c')
        } else if constexpr (K < 101) {
          hn[K - 97] = u[K - 97] * tt[3][K - 97];
        } else if constexpr (K < 105) {
          if constexpr (DROP) hv[K - 101] = hn[K - 101] * dv[K - 101];
          else hv[K - 101] = hn[K - 101];
        } else if constexpr (K == 105) {
          w0[0] = head2(hn[0], hn[1]);
        } else if constexpr (K == 106) {
          w0[1] = head2(hn[2], hn[3]);
        } else if constexpr (K < 111) {
          constexpr int i = K - 107;
          r1[i] = resid<(i & 1)>(w0[i >> 1], hn[i]);
        } else if constexpr (K == 111) {
          w1[0] = tail_lo(r1[0], k4096);
        } else if constexpr (K == 112) {
          w1[0] = tail_hi(w1[0], r1[1], k4096);
        } else if constexpr (K == 113) {
          w1[1] = tail_lo(r1[2], k4096);
        } else if constexpr (K == 114) {
          w1[1] = tail_hi(w1[1], r1[3], k4096);
          hoff = row * (D * 2) + (((hid >> 3) ^ swz<D>(row)) << 4) + ((hid >> 2) & 1) * 8;
        } else if constexpr (K == 115) {
          // always written: after the last step nothing reads it (no branch in the interleaved stream)
          *reinterpret_cast<i32x2*>(Hnxt + hoff) = i32x2{w0[0], w0[1]};
          *reinterpret_cast<i32x2*>(Hnxt + PLANE + hoff) = i32x2{w1[0], w1[1]};
        } else if constexpr (K == 116) {
          const i32x4 hvv = {__builtin_bit_cast(int, hv[0]), __builtin_bit_cast(int, hv[1]), __builtin_bit_cast(int, hv[2]),
                             __builtin_bit_cast(int, hv[3])};
          __builtin_amdgcn_raw_buffer_store_b128(hvv, rs_h, (row * (int)ld_h + ts * D + hid) * 4, 0, 0);
          c[PB] = f32x4{cn[0], cn[1], cn[2], cn[3]};
        } else if constexpr (K == 117) {
          if constexpr (SAVE) {
            const int e_td = (row * t + ts) * D + hid;
            const int go_ = (row * t + ts) * NC + hid;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const i32x4 gv = {__builtin_bit_cast(int, tt[g][0]), __builtin_bit_cast(int, tt[g][1]),
                                __builtin_bit_cast(int, tt[g][2]), __builtin_bit_cast(int, tt[g][3])};
              __builtin_amdgcn_raw_buffer_store_b128(gv, rs_g, (go_ + g * D) * 4, 0, 0);
            }
            const i32x4 cv = {__builtin_bit_cast(int, cn[0]), __builtin_bit_cast(int, cn[1]), __builtin_bit_cast(int, cn[2]),
                              __builtin_bit_cast(int, cn[3])};
            __builtin_amdgcn_raw_buffer_store_b128(cv, rs_c, e_td * 4, 0, 0);
          }
        } else {
          // ---- K >= kGateOps: two 16-row passes of the next step's x, split and stored. They ride with the
          // LAST kBT / 2 tiles' gate math (tile PB + 1 carries passes 2 (PB - (kBT - 1 - kBT / 2)) and + 1), thousands of
          // cycles after their loads were issued at the top of the step. At the last step xr is stale and the
          // target buffer is never read: harmless, and branch-free.
          constexpr int PS = (K - kGateOps) / kXOps;                            // which of the tile's two passes
          constexpr int XP = 2 * (PB - (kBT - 1 - kBT / 2)) + PS;               // pass 0 .. NFILL-1
          constexpr int X = (K - kGateOps) % kXOps;
          const float xv[4] = {xr[XP].x, xr[XP].y, xr[XP].z, xr[XP].w};
          const int xrow = XP * RPP + fr_;
          if constexpr (X == 0) {
            xw0[PS][0] = head2(xv[0], xv[1]);
          } else if constexpr (X == 1) {
            xw0[PS][1] = head2(xv[2], xv[3]);
          } else if constexpr (X < 6) {
            constexpr int i = X - 2;
            xq[PS][i] = resid<(i & 1)>(xw0[PS][i >> 1], xv[i]);
          } else if constexpr (X == 6) {
            xw1[PS][0] = tail_lo(xq[PS][0], k4096);
          } else if constexpr (X == 7) {
            xw1[PS][0] = tail_hi(xw1[PS][0], xq[PS][1], k4096);
          } else if constexpr (X == 8) {
            xw1[PS][1] = tail_lo(xq[PS][2], k4096);
          } else if constexpr (X == 9) {
            xw1[PS][1] = tail_hi(xw1[PS][1], xq[PS][3], k4096);
          } else if constexpr (X == 10) {       // the segment's max |v| and what it does to the tile's range (f16_split.h, RANGE)
            xs[PS] = maxabs3(xv[0], xv[1], xv[2]);
          } else if constexpr (X == 11) {
            xs[PS] = maxabs_acc(xs[PS], xv[3]);
            xoff[PS] = xrow * (D * 2) + (((fc4_ >> 3) ^ swz<D>(xrow)) << 4) + ((fc4_ >> 2) & 1) * 8;
          } else if constexpr (X == 12) {
            xrng.hi = maxabs_acc(xrng.hi, xs[PS]);
          } else if constexpr (X == 13) {
            xsb[PS] = __builtin_bit_cast(uint32_t, xs[PS]) - 1u;
          } else if constexpr (X == 14) {
            xrng.lo = xsb[PS] < xrng.lo ? xsb[PS] : xrng.lo;
          } else {
            *reinterpret_cast<i32x2*>(Xnxt + xoff[PS]) = i32x2{xw0[PS][0], xw0[PS][1]};
            *reinterpret_cast<i32x2*>(Xnxt + PLANE + xoff[PS]) = i32x2{xw1[PS][0], xw1[PS][1]};
          }
        }
      };
      // MFMA number I of tile BT, then its share of tile BT-1's gate math
      auto slot = [&](auto bt_c, auto i_c) {
        constexpr int BT = decltype(bt_c)::value, I = decltype(i_c)::value;
        // gate index fastest: consecutive MFMAs go to four different accumulator chains
        constexpr int ks = I / 12, term = (I % 12) / 4, g = I % 4;
        constexpr int cur = (BT * KSN + ks) & 1;
        if constexpr (I % 12 == 0) {               // request the next k-step's (or the next tile's first) fragments
          if constexpr (ks + 1 < KSN) read_b(BT, ks + 1, bf[cur ^ 1]);
          else if constexpr (BT + 1 < kBT) read_b(BT + 1, 0, bf[cur ^ 1]);
        }
        if constexpr (term == 0) {          // a2' b1
          const f32x4 cin = ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : alo[g];
          alo[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[g][ks][1]),
                                                          __builtin_bit_cast(f16x8, bf[cur][0]), cin, 0, 0, 0);
        } else if constexpr (term == 1) {   // a1 b2'
          alo[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[g][ks][0]),
                                                          __builtin_bit_cast(f16x8, bf[cur][1]), alo[g], 0, 0, 0);
        } else {                            // a1 b1
          const f32x4 cin = ks == 0 ? bc[g] : ahi[g];
          ahi[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[g][ks][0]),
                                                          __builtin_bit_cast(f16x8, bf[cur][0]), cin, 0, 0, 0);
        }
        if constexpr (BT > 0) {
          constexpr bool XO = BT >= kBT - kBT / 2;     // this tile also carries two x passes
          constexpr int lo = sched_start<NM, XO>(I), hi = sched_start<NM, XO>(I + 1);
          static_for<lo, hi>([&](auto pos_c) {
            gate_op(std::integral_constant<int, BT - 1>{}, std::integral_constant<int, sched_op<NM, XO>(decltype(pos_c)::value)>{});
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      auto tile_fn = [&](auto bt_c) {
        static_for<0, NM>([&](auto i_c) { slot(bt_c, i_c); });
#pragma unroll
        for (int g = 0; g < 4; ++g) ga[g] = ahi[g], gl[g] = alo[g];
      };
      read_b(0, 0, bf[0]);
      static_for<0, kBT>(tile_fn);
      // the last tile's gates: the only part of the step the MFMAs do not cover
      static_for<0, kGateOps>([&](auto pos_c) {
        gate_op(std::integral_constant<int, kBT - 1>{}, std::integral_constant<int, kOrder48[decltype(pos_c)::value]>{});
      });
    };

    for (int ts = 0; ts < t; ++ts) {
      if (ts + 1 < t) fetch_x(ts + 1);               // in flight under this step
      if (ts > 0 || h_init != nullptr) step(std::true_type{}, ts);
      else step(std::false_type{}, ts);               // zero initial state: the h half contributes nothing
      // x_{ts+1} went into the other buffer inside the step (its loads were issued at the top of it)
      if (ts + 1 < t) lds_barrier();   // x_{ts+1} and every wave's columns of h_{ts+1} are in place; x_ts / h_ts are free
    }
    if (range_bad(xrng)) flags[par] = 1;
    lds_barrier();            // flags settled; the images are free (the next tile's fills overwrite them)
    const bool redo = (flags[par] | flags[2]) != 0;   // block-uniform
    // c_final may BE c_init (a continued state updated in place): it is written once, by the pass that counts, so the
    // redo below still finds the tile's initial cell state where the caller put it
    if (c_final && !redo) {
#pragma unroll
      for (int bt = 0; bt < kBT; ++bt)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, c[bt]), rs_cf, ((bt * 16 + m) * D + hid0) * 4, 0, 0);
    }
    if (redo) {
      if (tid == 0 && redo_ctr) atomicAdd(redo_ctr, 1u);
      __syncthreads();        // this tile's global stores have left: the slow pass rewrites the same addresses
      // ---- a value outside the split's window went into this tile (or sits in W): the whole tile again as plain fp32
      // fmaf chains, thread per (row, hidden unit), h and c of the tile in LDS. Never taken on sane data.
      float* const hs = reinterpret_cast<float*>(lds);     // [2][kRows][D]
      float* const cs = hs + 2 * kRows * D;                 // [kRows][D]
      float* const xs = cs + kRows * D;                     // [kRows][D]: x_t of the tile (the four tiles fill the image space exactly)
      for (int idx = tid; idx < kRows * D; idx += NT) {
        const int r = idx / D, uu = idx - r * D;
        const bool live = r < rows_valid;
        hs[idx] = (h_init && live) ? h_init[(row0 + r) * ld_hi + uu] : 0.f;
        cs[idx] = (c_init && live) ? c_init[(row0 + r) * D + uu] : 0.f;
      }
      // a thread owns ONE hidden unit (uu = tid % D: NT is a multiple of D) of RPT rows (r0, r0 + NT / D, ...): the four
      // weights of a k serve all its rows — one pass over W per step instead of one per (row, unit) pair, which made a
      // single redone tile cost 2 ms (Amazon-shaped training hits it once the L = 3 hub rows grow past 32768)
      constexpr int RSTEP = NT / D, RPT = kRows / RSTEP;
      static_assert(NT % D == 0 && kRows % RSTEP == 0, "one hidden unit per thread");
      const int uu = tid % D, r0 = tid / D;
      for (int ts = 0; ts < t; ++ts) {
        const float* const hc = hs + (ts & 1) * kRows * D;
        float* const hnx = hs + ((ts & 1) ^ 1) * kRows * D;
        __syncthreads();
        for (int idx = tid; idx < kRows * D; idx += NT) {
          const int r = idx / D, c_ = idx - r * D;
          xs[idx] = r < rows_valid ? x[(row0 + r) * ld_n + (int64_t)ts * ld_t + c_] : 0.f;
        }
        __syncthreads();
        float a[RPT][4];
#pragma unroll
        for (int j = 0; j < RPT; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) a[j][g] = bias[g * D + uu] + (g == 2 ? forget_bias : 0.f);
        for (int half = 0; half < 2; ++half) {
          const float* const src = half ? hc : xs;
          for (int k = 0; k < D; k += 4) {
            float w[4][4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
              for (int g = 0; g < 4; ++g) w[kk][g] = W[(size_t)(half * D + k + kk) * NC + g * D + uu];
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
              const float4 v = *reinterpret_cast<const float4*>(src + (r0 + j * RSTEP) * D + k);
              const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
              for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int g = 0; g < 4; ++g) a[j][g] = fmaf(vv[kk], w[kk][g], a[j][g]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
          const int r = r0 + j * RSTEP, idx = r * D + uu;
          // tanhf, not the fast path's 1 - 2 / (1 + 2^t): a tile comes here also because a row is tiny as a whole, and
          // such a row's h is as small as its x — only a tanh that is relatively accurate near zero keeps it
          const float gi = sigmoid_e2(a[j][0]), gj = tanhf(a[j][1]), gf = sigmoid_e2(a[j][2]), go = sigmoid_e2(a[j][3]);
          const float cnew = fmaf(cs[idx], gf, gi * gj);
          const float hnew = tanhf(cnew) * go;
          cs[idx] = cnew;
          hnx[idx] = hnew;
          if (r < rows_valid) {
            const int64_t e_td = ((row0 + r) * t + ts) * D + uu;
            h_out[(row0 + r) * ld_h + (int64_t)ts * D + uu] = DROP ? hnew * drop[e_td] : hnew;
            if constexpr (SAVE) {
              float* const gp = gates_out + ((row0 + r) * t + ts) * NC + uu;
              gp[0] = gi, gp[D] = gj, gp[2 * D] = gf, gp[3 * D] = go;
              c_out[e_td] = cnew;
            }
          }
        }
      }
      __syncthreads();
      if (c_final)
        for (int idx = tid; idx < rows_valid * D; idx += NT) c_final[row0 * D + idx] = cs[idx];
      __syncthreads();
    }
  }
}

}  // namespace

namespace sagnn {

template <int D, bool SAVE, bool DROP>
int launch_lstm_f16(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, const float* W, const float* b,
                    float forget_bias, const float* drop, float* h, int64_t ld_h, float* gates_out, float* c_out,
                    const float* h_init, int64_t ld_hi, const float* c_init, float* c_final, hipStream_t s) {
  // x and h: 2 x 2 images each (96 KB at D = 64) + the flags; the slow pass needs 3 [rows][D] fp32 tiles (72 KB)
  const size_t lds = (size_t)8 * kRows * D * 2 + 16;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&lstm_fwd_f16_kernel<D, SAVE, DROP>), lds)) return rc;
  const int per_cu = D == 64 ? 1 : 2;
  const int64_t n_tiles = (n + kRows - 1) / kRows;
  const int64_t want = (int64_t)cu_count_current() * per_cu;
  const int64_t blocks = n_tiles < want ? n_tiles : want;
  ProfileScope prof(kProfLstm, s, n, t);
  hipLaunchKernelGGL((lstm_fwd_f16_kernel<D, SAVE, DROP>), dim3((unsigned)blocks), dim3(64 * (D / 16)), lds, s, x, ld_n, ld_t,
                     n, t, W, b, forget_bias, drop, h, ld_h, gates_out, c_out, n_tiles, h_init, ld_hi, c_init, c_final, redo_counter());
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

// One translation unit per (d, training) pair, as for the bf16 form.
#define SAGNN_LSTM_F16_ARGS                                                                                       \
  const float *x, int64_t ld_n, int64_t ld_t, int64_t n, int t, const float *W, const float *b, float forget_bias, \
      const float *drop, float *h, int64_t ld_h, float *gates_out, float *c_out, const float *h_init, int64_t ld_hi, \
      const float *c_init, float *c_final, hipStream_t s
#define SAGNN_LSTM_F16_PASS x, ld_n, ld_t, n, t, W, b, forget_bias, drop, h, ld_h, gates_out, c_out, h_init, ld_hi, c_init, c_final, s
int lstm_f16_d64(SAGNN_LSTM_F16_ARGS);        // inference (drop optional)
int lstm_f16_d64_save(SAGNN_LSTM_F16_ARGS);   // training forward (stores gates / cell; no drop)
int lstm_f16_d32(SAGNN_LSTM_F16_ARGS);
int lstm_f16_d32_save(SAGNN_LSTM_F16_ARGS);

}  // namespace sagnn
