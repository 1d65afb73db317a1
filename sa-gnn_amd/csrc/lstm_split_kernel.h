// Interval LSTM on the bf16 matrix cores with fp32-exact operands (gfx950).
//
// TF 1.14 BasicLSTMCell over T steps (reference model.py:135-146): gates = [x_t | h] @ W[2d,4d] + b,
// i, j, f, o = split(gates); c' = c sigmoid(f + fb) + sigmoid(i) tanh(j); h' = tanh(c') sigmoid(o).
//
// Why not the f32 MFMA of fusion_mfma.hip: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate
// (64 FLOP/clk/SIMD, 1/16 of bf16) and the gate math is paid on top of it. Here every
// fp32 operand is cut EXACTLY into three bf16 pieces (x = x1 + x2 + x3: 8 + 8 + 8 significand
// bits, by masking, no rounding) and the product is evaluated as the six largest of the nine
// piece products,
//     a b ~ a3 b1 + a1 b3 + a2 b2 + a2 b1 + a1 b2 + a1 b1,
// each a v_mfma_f32_16x16x32_bf16 (bf16 x bf16 is exact in fp32, accumulation is fp32). Dropped:
// a2 b3 + a3 b2 + a3 b3 < 2^-20 |a b| — below the rounding an fp32 dot product of this length
// carries anyway (measured against a float64 product: max error 0.25x that of an fmaf chain).
// Six bf16 MFMAs cost 6/16 of one fp32 MFMA, and VALU work issues beside bf16 MFMAs.
//
// Decomposition (transposed product, gates^T = W^T [x|h]^T): a workgroup of NW = d/16 waves owns
// 96 rows; wave w owns hidden units 16w .. 16w+15 of all four gates. Its slice of W stays in
// REGISTERS for the whole kernel as ready-made A fragments (4 gates x 2d/32 k-steps x 3 pieces).
// x_t and h are shared through LDS as three bf16 images [96][d] each (B fragments: one
// ds_read_b128 per lane, 16-byte slots XOR-swizzled with the row so reads and writes are
// conflict-free). In the 16x16 C tile a lane holds 4 consecutive hidden units of ONE row, for all
// four gates: the gate math needs no cross-lane traffic, h leaves as 16-byte stores, and its three
// pieces go back to LDS as 8-byte writes. x and h are both double-buffered in LDS (12 images =
// 144 KB at d = 64; the next step's x is prefetched into registers and written at the end of the
// step): ONE workgroup barrier per step. The gate math of a batch tile is hand-interleaved with the
// MFMAs of the next one (see `step`).
#include <type_traits>
#include <utility>

#pragma once
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr int kBT = 6;             // batch tiles of 16 rows per workgroup tile
constexpr int kRows = 16 * kBT;    // 96 rows: x and h both double-buffered in LDS = 12 images = 144 KB at d = 64

// x = p1 + p2 + p3 exactly, each piece a bf16 value held in the top 16 bits of a float.
struct Pieces {
  float p1, p2, p3;
};
__device__ __forceinline__ Pieces split3(float x) {
  Pieces s;
  s.p1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u);
  const float r = x - s.p1;  // exact: the low 16 significand bits
  s.p2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r) & 0xFFFF0000u);
  s.p3 = r - s.p2;           // at most 8 significant bits: already a bf16 value
  return s;
}
// bf16 pair (lo element first) from the top halves of two floats
__device__ __forceinline__ int pack_hi(float lo, float hi) {
  return (int)__builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x07060302u);
}

// 16-byte slot swizzle of the [row][D] bf16 images. D = 64: 128-byte rows, two rows per 64-bank
// line -> xor with (row >> 1) & 7. D = 32: 64-byte rows, four rows per line -> xor with a
// permutation of (row >> 2) & 3 chosen so the mixed lane groups of ds_read_b128
// ({0-3, 12-15, 20-27}: rows 0-3 and 12-15 of k-group a, rows 4-11 of k-group a+1) stay disjoint.
template <int D>
__device__ __forceinline__ int swz(int row) {
  if (D == 64) return (row >> 1) & 7;
  const int g = (row >> 2) & 3;       // f = {0, 2, 3, 1}
  return (0x78 >> (2 * g)) & 3;       // 0b01'11'10'00 read from the low end: g=0 -> 0, 1 -> 2, 2 -> 3, 3 -> 1
}

// f(integral_constant<int, LO>), ..., f(integral_constant<int, HI-1>) in order
template <int LO, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, LO + I>{}), ...);
}
template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (HI > LO) static_for_impl<LO>(static_cast<F&&>(f), std::make_integer_sequence<int, HI - LO>{});
}

// The interleaved operation list of `step`: kGateOps gate-math operations of one batch tile followed by
// kXOps operations that split one 16-row pass of the next step's x into its LDS images. Transcendentals
// take two issue slots of four cycles, everything else one; operations are dealt to the MFMA slots by
// cumulative cost, so that no slot carries two transcendentals.
constexpr int kGateOps = 127, kXOps = 20;
__host__ __device__ constexpr int op_cost(int k) {
  if (k >= 1 && k < 17) return 0;   // retired operations (kept so the numbering of the stages stays put)
  return ((k >= 17 && k < 33) || (k >= 49 && k < 65) || (k >= 81 && k < 85) || (k >= 89 && k < 93)) ? 2 : 1;
}
__host__ __device__ constexpr int op_cum(int k) {
  int c = 0;
  for (int j = 0; j < k; ++j) c += op_cost(j);
  return c;
}
// first of `nops` operations that belongs to MFMA slot i or a later one (slot of k = cum(k) * nm / total)
__host__ __device__ constexpr int first_op_of_slot(int i, int nm, int nops) {
  const int tot = op_cum(nops);
  int k = 0;
  while (k < nops && op_cum(k) * nm / tot < i) ++k;
  return k;
}

__device__ __forceinline__ void lds_barrier() {
  // LDS traffic only: outstanding global loads (the x prefetch) and stores stay in flight across it
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int D, bool SAVE, bool DROP>
__global__ __launch_bounds__(64 * (D / 16), 1) void lstm_fwd_split_kernel(
    const float* __restrict__ x, int64_t ld_n, int64_t ld_t, int64_t n, int t, const float* __restrict__ W,
    const float* __restrict__ bias, float forget_bias, const float* __restrict__ drop, float* __restrict__ h_out,
    int64_t ld_h, float* __restrict__ gates_out, float* __restrict__ c_out, int64_t n_tiles,
    const float* __restrict__ h_init, int64_t ld_hi, const float* __restrict__ c_init, float* __restrict__ c_final) {
  constexpr int NW = D / 16;          // waves per workgroup
  constexpr int NT = 64 * NW;         // threads
  constexpr int KSH = D / 32;         // k-steps (of 32) per operand half
  constexpr int KS = 2 * KSH;
  constexpr int NC = 4 * D;
  constexpr int PLANE = kRows * D * 2;          // bytes of one bf16 image
  constexpr int LPR = D / 4;                    // threads per row in the fill (float4 each)
  constexpr int RPP = NT / LPR;                 // rows per fill pass (16)
  constexpr int NFILL = kRows / RPP;            // 6
  constexpr float kL2E = 1.44269504088896340736f;

  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const Xp = lds;                         // 2 x 3 images
  char* const Hp = lds + 6 * PLANE;             // 2 x 3 images
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, q = lane >> 4;
  const int hid0 = 16 * wave + 4 * q;           // first of this lane's 4 hidden units (C rows 4q + r)
  const int fr = tid / LPR, fc4 = (tid % LPR) * 4;

  // ---- this wave's W slice as A fragments: A[mm = lane & 15][k = 32 ks + 8 q + j] = W[k][g D + 16 wave + mm]
  i32x4 wf[4][KS][3];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Pieces pc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)   // column scale of the gate's non-linearity folded into W (see bc below)
        pc[j] = split3(W[(size_t)(32 * ks + 8 * q + j) * NC + g * D + 16 * wave + m] * (g == 1 ? 2.f * kL2E : -kL2E));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        wf[g][ks][0][e] = pack_hi(pc[2 * e].p1, pc[2 * e + 1].p1);
        wf[g][ks][1][e] = pack_hi(pc[2 * e].p2, pc[2 * e + 1].p2);
        wf[g][ks][2][e] = pack_hi(pc[2 * e].p3, pc[2 * e + 1].p3);
      }
    }
  // Gate non-linearities are evaluated as exp2(t), t = k (pre-activation + bias), k = -log2(e) for the
  // sigmoids and 2 log2(e) for tanh(j): k is folded into this wave's columns of W (one rounding of each
  // weight, 2^-24 relative) and k * bias is what the accumulators start from, so t leaves the MFMAs ready.
  f32x4 bc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float b = bias[g * D + hid0 + r] + (g == 2 ? forget_bias : 0.f);
      bc[g][r] = (g == 1 ? 2.f * kL2E : -kL2E) * b;
    }

  // one [row][4 floats] piece set -> three 8-byte LDS writes
  auto write_pieces = [&](char* img, int row, int col4, float4 v) {
    const Pieces a = split3(v.x), b = split3(v.y), c = split3(v.z), d = split3(v.w);
    const int off = row * (D * 2) + ((((col4 >> 3)) ^ swz<D>(row)) << 4) + ((col4 >> 2) & 1) * 8;
    *reinterpret_cast<i32x2*>(img + off) = i32x2{pack_hi(a.p1, b.p1), pack_hi(c.p1, d.p1)};
    *reinterpret_cast<i32x2*>(img + PLANE + off) = i32x2{pack_hi(a.p2, b.p2), pack_hi(c.p2, d.p2)};
    *reinterpret_cast<i32x2*>(img + 2 * PLANE + off) = i32x2{pack_hi(a.p3, b.p3), pack_hi(c.p3, d.p3)};
  };

  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * kRows;
    const int rows_valid = (int)(n - row0 < kRows ? n - row0 : kRows);
    float4 xr[NFILL];
    auto fetch_x = [&](int ts) {
#pragma unroll
      for (int p = 0; p < NFILL; ++p) {
        const int r = p * RPP + fr;
        xr[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows_valid) xr[p] = *reinterpret_cast<const float4*>(x + (row0 + r) * ld_n + (int64_t)ts * ld_t + fc4);
      }
    };
    auto write_x = [&](int buf) {
#pragma unroll
      for (int p = 0; p < NFILL; ++p) write_pieces(Xp + buf * 3 * PLANE, p * RPP + fr, fc4, xr[p]);
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
        write_pieces(Hp, r, fc4, v);
      }
    }
    write_x(0);
    lds_barrier();


    // One step, hand-scheduled. The MFMAs of batch tile bt (6 piece products x 4 gates x k-steps,
    // one accumulator chain per gate) are issued one at a time, and after each one a slice of the
    // gate math of tile bt-1 — a fixed list of kOps single-instruction operations — is emitted,
    // closed by a scheduling barrier so the order survives the compiler: VALU and transcendental
    // issue slots beside a bf16 MFMA are otherwise left empty (its scheduler keeps MFMAs and the
    // dependent-free gate math in two separate runs), and B fragments are requested one k-step
    // ahead of the MFMAs that read them.
    auto step = [&](auto recur_c, int ts) {
      constexpr bool RECUR = decltype(recur_c)::value;
      constexpr int KSN = RECUR ? KS : KSH;           // k-steps per tile
      constexpr int NM = 24 * KSN;                    // MFMAs per tile
      const char* const Xcur = Xp + (ts & 1) * 3 * PLANE;
      const char* const Hcur = Hp + (ts & 1) * 3 * PLANE;
      char* const Hnxt = Hp + ((ts & 1) ^ 1) * 3 * PLANE;
      char* const Xnxt = Xp + ((ts & 1) ^ 1) * 3 * PLANE;
      // lane-derived LDS offsets recomputed per step (left loop-invariant the compiler hoists and spills them)
      int m_ = m, q_ = q, fr_ = fr, fc4_ = fc4;
      asm volatile("" : "+v"(m_), "+v"(q_), "+v"(fr_), "+v"(fc4_));
      const int hid = 16 * wave + 4 * q_;

      f32x4 acc[4];                    // tile in flight
      float xp1[4], xr1[4], xp2[4], xp3[4];   // pieces of the x pass being written
      int xw0[2], xw1[2], xw2[2], xoff;
      f32x4 ga[4];                     // pre-activations of the tile whose gate math is being interleaved
      f32x4 dv;                        // dropout scale of that tile
      float tt[4][4], pr[4], cn[4], u[4], hn[4], hv[4], p1[4], r1[4], p2[4], p3[4];
      int w0[2], w1[2], w2[2], hoff;
      i32x4 bf[2][3];                  // B fragments, double-buffered across k-steps

      auto read_b = [&](int bt, int ks, i32x4 (&dst)[3]) {
        const int row = bt * 16 + m_;
        const char* img = ks < KSH ? Xcur : Hcur;
        const int off = row * (D * 2) + ((((ks % KSH) * 4 + q_) ^ swz<D>(row)) << 4);
        dst[0] = *reinterpret_cast<const i32x4*>(img + off);
        dst[1] = *reinterpret_cast<const i32x4*>(img + PLANE + off);
        dst[2] = *reinterpret_cast<const i32x4*>(img + 2 * PLANE + off);
      };
      // operation K of the gate math of tile PB (state in ga / tt / ...): one instruction each, more or less
      auto gate_op = [&](auto pb_c, auto k_c) {
        constexpr int PB = decltype(pb_c)::value, K = decltype(k_c)::value;
        constexpr int R = K & 3, G = (K >> 2) & 3;
        const int row = PB * 16 + m_;
        if constexpr (K == 0) {
          const int e_td = (row * t + ts) * D + hid;
          if constexpr (DROP) dv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, e_td * 4, 0, 0));
        } else if constexpr (K < 17) {          // 1..16: retired (scale and bias now ride in W and the accumulator)
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
        } else if constexpr (K < 109) {
          p1[K - 105] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, hn[K - 105]) & 0xFFFF0000u);
        } else if constexpr (K < 113) {
          r1[K - 109] = hn[K - 109] - p1[K - 109];
        } else if constexpr (K < 117) {
          p2[K - 113] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1[K - 113]) & 0xFFFF0000u);
        } else if constexpr (K < 121) {
          p3[K - 117] = r1[K - 117] - p2[K - 117];
        } else if constexpr (K == 121) {
          w0[0] = pack_hi(p1[0], p1[1]);
          w0[1] = pack_hi(p1[2], p1[3]);
        } else if constexpr (K == 122) {
          w1[0] = pack_hi(p2[0], p2[1]);
          w1[1] = pack_hi(p2[2], p2[3]);
        } else if constexpr (K == 123) {
          w2[0] = pack_hi(p3[0], p3[1]);
          w2[1] = pack_hi(p3[2], p3[3]);
          hoff = row * (D * 2) + (((hid >> 3) ^ swz<D>(row)) << 4) + ((hid >> 2) & 1) * 8;
        } else if constexpr (K == 124) {
          // always written: after the last step nothing reads it (no branch in the interleaved stream)
          *reinterpret_cast<i32x2*>(Hnxt + hoff) = i32x2{w0[0], w0[1]};
          *reinterpret_cast<i32x2*>(Hnxt + PLANE + hoff) = i32x2{w1[0], w1[1]};
          *reinterpret_cast<i32x2*>(Hnxt + 2 * PLANE + hoff) = i32x2{w2[0], w2[1]};
        } else if constexpr (K == 125) {
          const i32x4 hvv = {__builtin_bit_cast(int, hv[0]), __builtin_bit_cast(int, hv[1]), __builtin_bit_cast(int, hv[2]),
                             __builtin_bit_cast(int, hv[3])};
          __builtin_amdgcn_raw_buffer_store_b128(hvv, rs_h, (row * (int)ld_h + ts * D + hid) * 4, 0, 0);
          c[PB] = f32x4{cn[0], cn[1], cn[2], cn[3]};
        } else if constexpr (K == 126) {
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
          // LAST three tiles' gate math (tile PB + 1 carries passes 2 (PB - (kBT - 4)) and + 1), thousands of
          // cycles after their loads were issued at the top of the step: placed right behind the loads, the
          // first of them parked the whole MFMA stream on vmcnt. At the last step xr is stale and the target
          // buffer is never read: harmless, and branch-free.
          constexpr int XP = 2 * (PB - (kBT - 4)) + (K - kGateOps) / kXOps;     // pass 0 .. NFILL-1
          constexpr int X = (K - kGateOps) % kXOps;
          const float xv[4] = {xr[XP].x, xr[XP].y, xr[XP].z, xr[XP].w};
          const int xrow = XP * RPP + fr_;
          if constexpr (X < 4) {
            xp1[X] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, xv[X]) & 0xFFFF0000u);
          } else if constexpr (X < 8) {
            xr1[X - 4] = xv[X - 4] - xp1[X - 4];
          } else if constexpr (X < 12) {
            xp2[X - 8] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, xr1[X - 8]) & 0xFFFF0000u);
          } else if constexpr (X < 16) {
            xp3[X - 12] = xr1[X - 12] - xp2[X - 12];
          } else if constexpr (X == 16) {
            xw0[0] = pack_hi(xp1[0], xp1[1]);
            xw0[1] = pack_hi(xp1[2], xp1[3]);
          } else if constexpr (X == 17) {
            xw1[0] = pack_hi(xp2[0], xp2[1]);
            xw1[1] = pack_hi(xp2[2], xp2[3]);
          } else if constexpr (X == 18) {
            xw2[0] = pack_hi(xp3[0], xp3[1]);
            xw2[1] = pack_hi(xp3[2], xp3[3]);
            xoff = xrow * (D * 2) + (((fc4_ >> 3) ^ swz<D>(xrow)) << 4) + ((fc4_ >> 2) & 1) * 8;
          } else {
            *reinterpret_cast<i32x2*>(Xnxt + xoff) = i32x2{xw0[0], xw0[1]};
            *reinterpret_cast<i32x2*>(Xnxt + PLANE + xoff) = i32x2{xw1[0], xw1[1]};
            *reinterpret_cast<i32x2*>(Xnxt + 2 * PLANE + xoff) = i32x2{xw2[0], xw2[1]};
          }
        }
        (void)R;
        (void)G;
      };
      // MFMA number I of tile BT, then the slice [I kOps / NM, (I+1) kOps / NM) of tile BT-1's gate math
      auto slot = [&](auto bt_c, auto i_c) {
        constexpr int BT = decltype(bt_c)::value, I = decltype(i_c)::value;
        // gate index fastest: consecutive MFMAs go to four different accumulator chains
        constexpr int ks = I / 24, term = (I % 24) / 4, g = I % 4;
        constexpr int cur = (BT * KSN + ks) & 1;
        if constexpr (I % 24 == 0) {               // request the next k-step's (or the next tile's first) fragments
          if constexpr (ks + 1 < KSN) read_b(BT, ks + 1, bf[cur ^ 1]);
          else if constexpr (BT + 1 < kBT) read_b(BT + 1, 0, bf[cur ^ 1]);
        }
        // piece products, smallest first: a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1
        constexpr int ai = term == 0 ? 2 : (term == 2 || term == 3) ? 1 : 0;
        constexpr int bi = term == 1 ? 2 : (term == 2 || term == 4) ? 1 : 0;
        const f32x4 cin = (ks == 0 && term == 0) ? bc[g] : acc[g];
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[g][ks][ai]),
                                                         __builtin_bit_cast(bf16x8, bf[cur][bi]), cin, 0, 0, 0);
        if constexpr (BT > 0) {
          constexpr int NOPS = kGateOps + (BT >= kBT - 3 ? 2 * kXOps : 0);
          constexpr int lo = first_op_of_slot(I, NM, NOPS), hi = first_op_of_slot(I + 1, NM, NOPS);
          static_for<lo, hi>([&](auto k_c) { gate_op(std::integral_constant<int, BT - 1>{}, k_c); });
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      auto tile = [&](auto bt_c) {
        constexpr int BT = decltype(bt_c)::value;
        static_for<0, NM>([&](auto i_c) { slot(bt_c, i_c); });
#pragma unroll
        for (int g = 0; g < 4; ++g) ga[g] = acc[g];
      };
      read_b(0, 0, bf[0]);
      static_for<0, kBT>(tile);
      // the last tile's gates: the only part of the step the MFMAs do not cover
      static_for<0, kGateOps>([&](auto k_c) { gate_op(std::integral_constant<int, kBT - 1>{}, k_c); });
    };

    for (int ts = 0; ts < t; ++ts) {
      if (ts + 1 < t) fetch_x(ts + 1);               // in flight under this step
      if (ts > 0 || h_init != nullptr) step(std::true_type{}, ts);
      else step(std::false_type{}, ts);               // zero initial state: the h half contributes nothing
      // x_{ts+1} went into the other buffer inside the step (its loads were issued at the top of it)
      if (ts + 1 < t) lds_barrier();   // x_{ts+1} and every wave's columns of h_{ts+1} are in place; x_ts / h_ts are free
    }
    if (c_final) {
#pragma unroll
      for (int bt = 0; bt < kBT; ++bt)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, c[bt]), rs_cf, ((bt * 16 + m) * D + hid0) * 4, 0, 0);
    }
    lds_barrier();            // the next tile's fills overwrite what slower waves may still read
  }
}

}  // namespace

namespace sagnn {

template <int D, bool SAVE, bool DROP>
int launch_lstm_split(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, const float* W, const float* b,
                      float forget_bias, const float* drop, float* h, int64_t ld_h, float* gates_out, float* c_out,
                      const float* h_init, int64_t ld_hi, const float* c_init, float* c_final, hipStream_t s) {
  const size_t lds = (size_t)12 * kRows * D * 2;  // x and h: 2 x 3 images each (144 KB at D = 64)
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&lstm_fwd_split_kernel<D, SAVE, DROP>), lds)) return rc;
  const int per_cu = D == 64 ? 1 : 2;             // D = 32: 72 KB and 2 waves per workgroup
  const int64_t n_tiles = (n + kRows - 1) / kRows;
  const int64_t want = (int64_t)cu_count_current() * per_cu;
  const int64_t blocks = n_tiles < want ? n_tiles : want;
  ProfileScope prof(kProfLstm, s, n, t);
  hipLaunchKernelGGL((lstm_fwd_split_kernel<D, SAVE, DROP>), dim3((unsigned)blocks), dim3(64 * (D / 16)), lds, s, x, ld_n, ld_t,
                     n, t, W, b, forget_bias, drop, h, ld_h, gates_out, c_out, n_tiles, h_init, ld_hi, c_init, c_final);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

// One translation unit per (d, training) pair: the hand-scheduled step is ~3000 instructions per variant and the
// variants compile in parallel this way.
#define SAGNN_LSTM_SPLIT_ARGS                                                                                     \
  const float *x, int64_t ld_n, int64_t ld_t, int64_t n, int t, const float *W, const float *b, float forget_bias, \
      const float *drop, float *h, int64_t ld_h, float *gates_out, float *c_out, const float *h_init, int64_t ld_hi, \
      const float *c_init, float *c_final, hipStream_t s
#define SAGNN_LSTM_SPLIT_PASS x, ld_n, ld_t, n, t, W, b, forget_bias, drop, h, ld_h, gates_out, c_out, h_init, ld_hi, c_init, c_final, s
int lstm_split_d64(SAGNN_LSTM_SPLIT_ARGS);        // inference (drop optional)
int lstm_split_d64_save(SAGNN_LSTM_SPLIT_ARGS);   // training forward (stores gates / cell; no drop)
int lstm_split_d32(SAGNN_LSTM_SPLIT_ARGS);
int lstm_split_d32_save(SAGNN_LSTM_SPLIT_ARGS);

}  // namespace sagnn
