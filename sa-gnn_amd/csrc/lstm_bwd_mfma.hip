// Whole BPTT of the BasicLSTMCell (gradient of reference model.py:135-146) in one launch.
//
// The un-fused form (sagnn_lstm_bwd_step_f32 + two dense products per step) writes the gate
// gradients [n, 4d] to HBM and reads them back twice per step, and its products run at a third of
// the MFMA rate. Here a block (4 waves, one per SIMD) walks 32-row chunks through all T steps:
//
//   phase A (VALU, 256 threads): saved activations + dh -> gate gradients dG [32, 4D] into LDS
//   MFMA 1: d[x_t | h_{t-1}] [32, 2D] = dG @ W^T      one 32-column tile per wave; its slice of W^T
//           sits in LDS as ready-made B fragments (one ds_read_b128 per 4 MFMAs), except the
//           8 KB per block that do not fit the 160 KB next to the tiles: those stay in registers
//   MFMA 2: dW [2D, 4D] += [x_t | h_{t-1}]^T @ dG     wave w owns tile row w (D = 64: 8 tiles =
//           128 accumulator registers, kept across every chunk of the block, flushed once)
//
// dh_{t-1} goes back through an LDS tile to the next step's phase A; dc stays in registers; dG
// never leaves the CU. Global operands of the next step are requested before the MFMAs.
#include "common.h"

// Diagnostic build only (-DSAGNN_STAMPS): per-section cycle sums, read with sagnn_debug_read_stamps_bwd().
#ifdef SAGNN_STAMPS
__device__ unsigned long long g_bwd_stamps[8];
__device__ __forceinline__ unsigned long long bwd_stamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
__device__ __forceinline__ unsigned long long bwd_realtime_now() {  // constant 100 MHz
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define BSTAMP_DECL                                                                      \
  unsigned long long st_prev = bwd_stamp_now(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  \
  const unsigned long long st_t0 = st_prev, st_r0 = bwd_realtime_now()
#define BSTAMP(i)                                   \
  do {                                              \
    const unsigned long long _t = bwd_stamp_now();  \
    st_acc[i] += _t - st_prev;                      \
    st_prev = _t;                                   \
  } while (0)
#define BSTAMP_FLUSH                                                           \
  do {                                                                         \
    if (threadIdx.x == 0 && blockIdx.x == 0) {                                 \
      st_acc[6] = bwd_stamp_now() - st_t0;                                     \
      st_acc[7] = bwd_realtime_now() - st_r0;                                  \
    }                                                                          \
    if ((threadIdx.x & 63) == 0)                                               \
      for (int _i = 0; _i < 8; ++_i) atomicAdd(&g_bwd_stamps[_i], st_acc[_i]); \
  } while (0)
#else
#define BSTAMP_DECL
#define BSTAMP(i)
#define BSTAMP_FLUSH
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBlock = 256;
constexpr int kRows = 32;  // rows per chunk

__device__ __forceinline__ int crow(int r, int rh) { return (r & 3) + 8 * (r >> 2) + 4 * rh; }
__device__ __forceinline__ float fast_tanh(float x) {
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x));
}
// float4-slot swizzle of the dG tile: a bijection of the low 4 row bits whose bit 3 flips between
// rows 2k and 2k+1, so the row-per-lane b128 reads (MFMA 1), the two-rows-per-instruction b32
// reads (MFMA 2) and the row-major float4 fill are all bank-conflict free.
__device__ __forceinline__ int swz(int row) { return (row & 15) ^ ((row & 1) << 3); }

// DW = false: the weight-gradient product (MFMA 2: 63 % of the kernel's time on the fp32 matrix pipe) is left out and the
// gate gradients leave instead, time-major ([t, n, 4D] at dg_out): lstm_dw_f16.hip takes dW from them on the f16 x 2 engine.
template <int D, bool DW>
__global__ __launch_bounds__(kBlock, 1) void lstm_bwd_mfma_kernel(
    const float* __restrict__ x, int64_t ld_n, int64_t ld_t, const float* __restrict__ h,
    const float* __restrict__ gates, const float* __restrict__ cell, const float* __restrict__ dh_ext,
    int64_t ld_dhe, const float* __restrict__ drop, const float* __restrict__ W, float* __restrict__ dx,
    float* __restrict__ dW, float* __restrict__ db, int64_t n, int t, int64_t n_chunks, float* __restrict__ dg_out) {
  constexpr int NC = 4 * D;         // gate columns
  constexpr int S4 = NC / 4;        // float4 slots per dG row
  constexpr int XT = D / 32;        // column tiles of x (and of h)
  constexpr int TA = 2 * XT;        // tile rows of dW = column tiles of [x | h]
  constexpr int TB = NC / 32;       // tile columns of dW
  constexpr int WPT = 4 / TA;       // waves sharing one tile row (1 at D=64, 2 at D=32)
  constexpr int NTB = TB / WPT;     // dW tiles per wave
  constexpr int NQ = NC / 8;        // b128 A reads of MFMA 1
  constexpr int QR = D == 64 ? 2 : 0;  // of which W^T fragments held in registers (rest: LDS)
  constexpr int QL = NQ - QR;
  constexpr int LPR = D / 4;        // phase A: threads per row
  constexpr int RPP = kBlock / LPR; // rows per pass
  constexpr int NPASS = kRows / RPP;
  static_assert(TA <= 4 && 4 % TA == 0 && NPASS >= 1, "D must be 32 or 64");

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* dg = lds;                   // [32][NC], slot-swizzled
  float* dhr = lds + kRows * NC;     // [32][D] recurrent dh for the next (earlier) step
  float4* wl = reinterpret_cast<float4*>(dhr + kRows * D);  // [NW1][QL][64] W^T fragments
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: tile choices become SALU branches
  const int li = lane & 31, kh = lane >> 5;
  const int ta = wave % TA;          // my tile row of dW / my column tile of d[x|h]
  const int tb0 = wave / TA;         // my dW tile columns: tb0 + j*WPT
  const bool mfma1_wave = wave < TA; // D=32: waves 2,3 have no MFMA-1 tile
  const bool h_side = ta >= XT;      // my [x|h] columns belong to h

  // ---- W^T B fragments for output column tile `ta`: fragment q of lane (li, kh) =
  // W[32ta+li][8q+4kh .. +3], i.e. the B operands of MFMAs 4q .. 4q+3 (k order matches the b128
  // A reads of the dG tile)
  float wb[QR > 0 ? 4 * QR : 1];
  float wbp[QR > 0 ? 4 * QR : 1];  // h-side waves: the register fragments of x tile ta - XT (see MFMA 1)
  if (mfma1_wave && h_side) {
    const float* prow = W + (size_t)(32 * (ta - XT) + li) * NC + 4 * kh;
#pragma unroll
    for (int q = 0; q < QR; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(prow + 8 * q);
      wbp[4 * q + 0] = v.x;
      wbp[4 * q + 1] = v.y;
      wbp[4 * q + 2] = v.z;
      wbp[4 * q + 3] = v.w;
    }
  }
  if (mfma1_wave) {
    const float* wrow = W + (size_t)(32 * ta + li) * NC + 4 * kh;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(wrow + 8 * q);
      if (q < QR) {
        wb[4 * q + 0] = v.x;
        wb[4 * q + 1] = v.y;
        wb[4 * q + 2] = v.z;
        wb[4 * q + 3] = v.w;
      } else {
        wl[(wave * QL + (q - QR)) * 64 + lane] = v;
      }
    }
  }
  // (read back after the loop's first barrier)

  f32x16 accw[NTB];
#pragma unroll
  for (int j = 0; j < NTB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[j][r] = 0.f;
  float dbs[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) dbs[g][e] = 0.f;

  const int prow = tid / LPR, pc4 = tid % LPR;  // phase A: row within pass, float4 column

  // (chunk, step) items form ONE sequence, so the operands of a chunk's first step are requested
  // under the MFMAs of the previous chunk's last step.
  float4 dc[NPASS];
  float4 pg[NPASS][4], pcl[NPASS], pcp[NPASS], pdh[NPASS], pdr[NPASS];
  // Branch-free: rows past n read the last valid row and are zeroed in phase A (a zero dh makes every
  // gate gradient of the row zero); the missing c_{-1} of step 0 reads c_0 and is zeroed there too.
  // Addresses = wave-uniform 64-bit base (SALU) + a 32-bit lane offset: address VALU is not free
  // next to fp32 MFMAs (they share the SIMD's fp32 lanes).
  auto prefetch = [&](int64_t row0, int ts) {
    const int last = (int)(n - 1 - row0 < kRows - 1 ? n - 1 - row0 : kRows - 1);  // uniform
    const int64_t e0 = row0 * t + ts;
    const float* gbase = gates + e0 * NC;
    const float* cbase = cell + e0 * D;
    const float* pbase = cbase - (ts > 0 ? D : 0);
    const float* hbase = dh_ext + row0 * ld_dhe + (int64_t)ts * D;
    const float* dbase = drop ? drop + e0 * D : nullptr;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      // chunk-local row, clamped; UNSIGNED offsets so the loads take the SGPR-base + VGPR-offset form
      const uint32_t lr = p * RPP + prow < last ? p * RPP + prow : last;
      const uint32_t eo = lr * (uint32_t)t, c4 = 4 * pc4;
      pg[p][0] = *reinterpret_cast<const float4*>(gbase + (eo * NC + c4));
      pg[p][1] = *reinterpret_cast<const float4*>(gbase + (eo * NC + D + c4));
      pg[p][2] = *reinterpret_cast<const float4*>(gbase + (eo * NC + 2 * D + c4));
      pg[p][3] = *reinterpret_cast<const float4*>(gbase + (eo * NC + 3 * D + c4));
      pcl[p] = *reinterpret_cast<const float4*>(cbase + (eo * D + c4));
      pcp[p] = *reinterpret_cast<const float4*>(pbase + (eo * D + c4));
      pdh[p] = *reinterpret_cast<const float4*>(hbase + (lr * (uint32_t)ld_dhe + c4));
      // applied in phase A: multiplying here would wait for the loads right away
      if (drop) pdr[p] = *reinterpret_cast<const float4*>(dbase + (eo * D + c4));
    }
  };
  // A operand of MFMA 2 for item (row0, ts): my 32 columns of [x_t | h_{t-1}], rows 2kk + kh; loaded
  // one item ahead like the phase-A operands (rows past n: clamped here, zeroed when consumed)
  float a2n[16];
  auto prefetch_a2 = [&](int64_t row0, int ts) {
    const int last = (int)(n - 1 - row0 < kRows - 1 ? n - 1 - row0 : kRows - 1);
    const float* src = h_side ? h + (row0 * t + (ts > 0 ? ts - 1 : 0)) * D + 32 * (ta - XT)
                              : x + row0 * ld_n + (int64_t)ts * ld_t + 32 * ta;
    const uint32_t ldr = h_side ? (uint32_t)(t * D) : (uint32_t)ld_n;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const uint32_t lr = 2 * kk + kh < last ? 2 * kk + kh : last;
      a2n[kk] = src[lr * ldr + (uint32_t)li];
    }
  };

  int64_t ch = blockIdx.x;
  int ts = t - 1;
  if (ch < n_chunks) {
    prefetch(ch * kRows, ts);
    if constexpr (DW) prefetch_a2(ch * kRows, ts);
  }
  BSTAMP_DECL;
  while (ch < n_chunks) {
    const int64_t row0 = ch * kRows;
    const int64_t nch = ts > 0 ? ch : ch + gridDim.x;
    const int nts = ts > 0 ? ts - 1 : t - 1;
    // lane-derived offsets are re-materialised per step (cheap) instead of being hoisted
    int li_ = li, kh_ = kh, prow_ = prow, pc4_ = pc4;
    asm volatile("" : "+v"(li_), "+v"(kh_), "+v"(prow_), "+v"(pc4_));

    // ---- phase A: gate gradients of this step into the dG tile ----------------------------------
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int row = p * RPP + prow_;
      float4 dhv = pdh[p];
      const bool valid = row0 + row < n;
      const float4 sc = drop ? pdr[p] : make_float4(1.f, 1.f, 1.f, 1.f);
      dhv.x *= valid ? sc.x : 0.f;
      dhv.y *= valid ? sc.y : 0.f;
      dhv.z *= valid ? sc.z : 0.f;
      dhv.w *= valid ? sc.w : 0.f;
      const float cpm = ts > 0 ? 1.f : 0.f;  // c_{-1} = 0
      if (ts < t - 1) {
        const float4 r = *reinterpret_cast<const float4*>(dhr + row * D + 4 * pc4_);
        dhv.x += r.x;
        dhv.y += r.y;
        dhv.z += r.z;
        dhv.w += r.w;
      } else {
        dc[p] = make_float4(0.f, 0.f, 0.f, 0.f);  // first (= last in time) step of a chunk
      }
      float4 o_i, o_j, o_f, o_o;
#define SAGNN_GATE_BWD(e)                                                        \
  {                                                                              \
    const float gi = pg[p][0].e, gj = pg[p][1].e, gf = pg[p][2].e, go = pg[p][3].e; \
    const float tc = fast_tanh(pcl[p].e);                                        \
    const float dcv = dc[p].e + dhv.e * go * (1.f - tc * tc);                    \
    o_o.e = dhv.e * tc * go * (1.f - go);                                        \
    o_i.e = dcv * gj * gi * (1.f - gi);                                          \
    o_j.e = dcv * gi * (1.f - gj * gj);                                          \
    o_f.e = dcv * (pcp[p].e * cpm) * gf * (1.f - gf);                                  \
    dc[p].e = dcv * gf;                                                          \
  }
      SAGNN_GATE_BWD(x)
      SAGNN_GATE_BWD(y)
      SAGNN_GATE_BWD(z)
      SAGNN_GATE_BWD(w)
#undef SAGNN_GATE_BWD
      float4* drow = reinterpret_cast<float4*>(dg) + row * S4;
      const int sw = swz(row);
      drow[(pc4_) ^ sw] = o_i;
      drow[(LPR + pc4_) ^ sw] = o_j;
      drow[(2 * LPR + pc4_) ^ sw] = o_f;
      drow[(3 * LPR + pc4_) ^ sw] = o_o;
      dbs[0][0] += o_i.x, dbs[0][1] += o_i.y, dbs[0][2] += o_i.z, dbs[0][3] += o_i.w;
      dbs[1][0] += o_j.x, dbs[1][1] += o_j.y, dbs[1][2] += o_j.z, dbs[1][3] += o_j.w;
      dbs[2][0] += o_f.x, dbs[2][1] += o_f.y, dbs[2][2] += o_f.z, dbs[2][3] += o_f.w;
      dbs[3][0] += o_o.x, dbs[3][1] += o_o.y, dbs[3][2] += o_o.z, dbs[3][3] += o_o.w;
      if constexpr (!DW) {
        if (valid) {
          float* const go = dg_out + ((int64_t)ts * n + row0 + row) * NC + 4 * pc4_;
          *reinterpret_cast<float4*>(go) = o_i;
          *reinterpret_cast<float4*>(go + D) = o_j;
          *reinterpret_cast<float4*>(go + 2 * D) = o_f;
          *reinterpret_cast<float4*>(go + 3 * D) = o_o;
        }
      }
    }
    BSTAMP(0);
    __syncthreads();  // dG complete; dhr consumed
    BSTAMP(1);

    // ---- this item's MFMA-2 operand (requested one item ago); then the next item's global operands,
    // in flight under both MFMA phases ----------------------------------------------------------------
    const bool do_w = !h_side || ts > 0;  // h_{-1} = 0 contributes nothing
    float a2[16];
    if constexpr (DW) {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) a2[kk] = (row0 + 2 * kk + kh_ < n) ? a2n[kk] : 0.f;
    }
    if (nch < n_chunks) {
      prefetch(nch * kRows, nts);
      if constexpr (DW) prefetch_a2(nch * kRows, nts);
    }
    BSTAMP(2);

    // ---- MFMA 1: one 32-column tile of d[x_t | h_{t-1}] = dG @ W^T per wave. For ts > 0 wave w
    // takes tile w. At ts = 0 only dx is needed and the x-side waves still have their MFMA 2 to do,
    // so the otherwise idle h-side waves compute the dx tiles (the step then costs every wave 4D/2
    // MFMAs instead of 4D for half of them).
    const int m1 = ts > 0 ? ta : (h_side ? ta - XT : -1);
    if (mfma1_wave && m1 >= 0) {
      f32x16 c0, c1;
#pragma unroll
      for (int r = 0; r < 16; ++r) c0[r] = 0.f, c1[r] = 0.f;
      const float4* arow = reinterpret_cast<const float4*>(dg) + li_ * S4;
      const float4* wq = wl + m1 * QL * 64 + lane;
      const bool own = m1 == ta;
      const int sw = swz(li_);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float4 a = arow[(2 * q + kh_) ^ sw];
        float4 w;
        if (q < QR)
          w = make_float4(own ? wb[4 * q] : wbp[4 * q], own ? wb[4 * q + 1] : wbp[4 * q + 1],
                          own ? wb[4 * q + 2] : wbp[4 * q + 2], own ? wb[4 * q + 3] : wbp[4 * q + 3]);
        else
          w = wq[(q - QR) * 64];
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, c1, 0, 0, 0);
      }
      if (m1 < XT) {
        // per-chunk buffer descriptor: uniform base, 32-bit lane offsets, rows past n dropped by the
        // hardware range check (no 64-bit address arithmetic or predicates per element)
        const int rows_valid = (int)(n - row0 < kRows ? n - row0 : kRows);
        const auto rs_dx = __builtin_amdgcn_make_buffer_rsrc(dx + row0 * t * D, 0, rows_valid * t * D * 4, 0x00020000);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, c0[r] + c1[r]), rs_dx,
                                                ((crow(r, kh_) * t + ts) * D + 32 * m1 + li_) * 4, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) dhr[crow(r, kh_) * D + 32 * (m1 - XT) + li_] = c0[r] + c1[r];
      }
    }

    BSTAMP(3);
    // ---- MFMA 2: dW[32ta .. , :] += [x_t | h_{t-1}]^T dG. h_{-1} = 0: nothing to do for the h side at
    // ts = 0 — expressed as a zero trip count of a rolled loop, not as a branch around the block (with
    // a branch the compiler keeps two copies of the 128 accumulators and spills).
    if constexpr (DW) {
      // B operands are read one group of k-steps ahead (an MFMA does not cover an LDS round trip)
      constexpr int GK = NTB >= 8 ? 2 : 4;
      constexpr int NG = 16 / GK;
      float bc[GK][NTB], bn[GK][NTB];
      auto read_group = [&](float (&b)[GK][NTB], int g) {
#pragma unroll
        for (int u = 0; u < GK; ++u) {
          const int row = 2 * (g * GK + u) + kh_;
          const float* brow = dg + row * NC + (li_ & 3);
          const int sw = swz(row);
#pragma unroll
          for (int j = 0; j < NTB; ++j) b[u][j] = brow[(((tb0 + j * WPT) * 8 + (li_ >> 2)) ^ sw) * 4];
        }
      };
      const int ng = do_w ? NG : 0;
      read_group(bc, 0);
#pragma unroll 1
      for (int g = 0; g < ng; ++g) {
        read_group(bn, g + 1 < NG ? g + 1 : g);
        float av[GK];
#pragma unroll
        for (int u = 0; u < GK; ++u) {
          av[u] = a2[0];
#pragma unroll
          for (int k = 1; k < 16; ++k) av[u] = (g * GK + u == k) ? a2[k] : av[u];  // register select, no scratch
        }
#pragma unroll
        for (int u = 0; u < GK; ++u)
#pragma unroll
          for (int j = 0; j < NTB; ++j) accw[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bc[u][j], accw[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < GK; ++u)
#pragma unroll
          for (int j = 0; j < NTB; ++j) bc[u][j] = bn[u][j];
      }
    }
    BSTAMP(4);
    __syncthreads();  // dG consumed, dhr written
    BSTAMP(5);
    ch = nch;
    ts = nts;
  }

  BSTAMP_FLUSH;
  // ---- flush: dW tiles with float atomics, db through an LDS reduction over the row groups ------
  if constexpr (DW) {
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
      const int tb = tb0 + j * WPT;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        atomicAdd(dW + (size_t)(32 * ta + crow(r, kh)) * NC + 32 * tb + li, accw[j][r]);
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) dg[prow * NC + g * D + 4 * pc4 + e] = dbs[g][e];
  __syncthreads();
  if (tid < NC) {
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < RPP; ++r) s += dg[r * NC + tid];
    atomicAdd(db + tid, s);
  }
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

template <int D, bool DW = true>
int launch(const float* x, int64_t ld_n, int64_t ld_t, const float* h, const float* gates, const float* cell,
           const float* dh_ext, int64_t ld_dhe, const float* drop, const float* W, float* dx, float* dW, float* db,
           int64_t n, int t, hipStream_t s, float* dg_out = nullptr) {
  const int64_t n_chunks = (n + kRows - 1) / kRows;
  const int64_t blocks = n_chunks < cu_count() ? n_chunks : cu_count();
  const int nq = 4 * D / 8, ql = nq - (D == 64 ? 2 : 0), nw1 = 2 * D / 32;
  const size_t lds = (size_t)kRows * (4 * D + D) * sizeof(float) + (size_t)nw1 * ql * 64 * sizeof(float4);
  if (int rc = sagnn::ensure_dynamic_lds(reinterpret_cast<const void*>(&lstm_bwd_mfma_kernel<D, DW>), lds)) return rc;
  hipLaunchKernelGGL((lstm_bwd_mfma_kernel<D, DW>), dim3((unsigned)blocks), dim3(kBlock), lds, s, x, ld_n, ld_t, h, gates,
                     cell, dh_ext, ld_dhe, drop, W, dx, dW, db, n, t, n_chunks, dg_out);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

}  // namespace

extern "C" int sagnn_lstm_bwd_supported(int d) { return d == 32 || d == 64; }

extern "C" int sagnn_lstm_bwd_f32(const float* x, int64_t ld_n, int64_t ld_t, const float* h, const float* gates,
                                  const float* cell, const float* dh_ext, int64_t ld_dhe, const float* drop_scale,
                                  const float* W, float* dx, float* dW, float* db, int64_t n, int t, int d,
                                  void* stream) {
  if (n < 0 || t < 1) return sagnn::fail(SAGNN_ERR_DIM, "bad n/t");
  if (!sagnn_lstm_bwd_supported(d)) return sagnn::fail(SAGNN_ERR_DIM, "lstm_bwd: d must be 32 or 64 (got %d)", d);
  if (!x || !h || !gates || !cell || !dh_ext || !W || !dx || !dW || !db)
    return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (ld_n < d || (t > 1 && ld_t < d) || ld_dhe < (int64_t)t * d)
    return sagnn::fail(SAGNN_ERR_ARG, "strides smaller than the rows they address");
  if (ld_n >= (1 << 25) || ld_dhe >= (1 << 25) || (int64_t)t * d >= (1 << 20))
    return sagnn::fail(SAGNN_ERR_ARG, "lstm_bwd: row strides must stay below 2^25 floats (32-bit lane offsets)");
  if ((ld_dhe & 3) || !sagnn::aligned16(dh_ext) || !sagnn::aligned16(gates) || !sagnn::aligned16(cell) ||
      !sagnn::aligned16(W) || (drop_scale && !sagnn::aligned16(drop_scale)))
    return sagnn::fail(SAGNN_ERR_ALIGN, "lstm_bwd: need 16-byte aligned rows");
  if (n == 0) return SAGNN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (d == 64) return launch<64>(x, ld_n, ld_t, h, gates, cell, dh_ext, ld_dhe, drop_scale, W, dx, dW, db, n, t, s);
  return launch<32>(x, ld_n, ld_t, h, gates, cell, dh_ext, ld_dhe, drop_scale, W, dx, dW, db, n, t, s);
}

extern "C" size_t sagnn_lstm_bwd_workspace_bytes(int64_t n, int t, int d) {
  if (n <= 0 || t <= 0 || d <= 0) return 0;
  return (size_t)n * (size_t)t * 4 * (size_t)d * sizeof(float);
}

extern "C" int sagnn_lstm_bwd_ws_f32(const float* x, int64_t ld_n, int64_t ld_t, const float* h, const float* gates,
                                     const float* cell, const float* dh_ext, int64_t ld_dhe, const float* drop_scale,
                                     const float* W, float* dx, float* dW, float* db, int64_t n, int t, int d,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  // the exact-fp32 engines keep the one-launch form (its dW product IS the fp32 MFMA); so does a call without scratch
  if (!workspace || sagnn::force_f32_mfma() || sagnn::force_valu())
    return sagnn_lstm_bwd_f32(x, ld_n, ld_t, h, gates, cell, dh_ext, ld_dhe, drop_scale, W, dx, dW, db, n, t, d, stream);
  if (n < 0 || t < 1) return sagnn::fail(SAGNN_ERR_DIM, "bad n/t");
  if (!sagnn_lstm_bwd_supported(d)) return sagnn::fail(SAGNN_ERR_DIM, "lstm_bwd: d must be 32 or 64 (got %d)", d);
  if (!x || !h || !gates || !cell || !dh_ext || !W || !dx || !dW || !db)
    return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (ld_n < d || (t > 1 && ld_t < d) || ld_dhe < (int64_t)t * d)
    return sagnn::fail(SAGNN_ERR_ARG, "strides smaller than the rows they address");
  if (ld_n >= (1 << 25) || ld_dhe >= (1 << 25) || (int64_t)t * d >= (1 << 20))
    return sagnn::fail(SAGNN_ERR_ARG, "lstm_bwd: row strides must stay below 2^25 floats (32-bit lane offsets)");
  if ((ld_dhe & 3) || (ld_n & 3) || (ld_t & 3) || !sagnn::aligned16(x) || !sagnn::aligned16(h) || !sagnn::aligned16(dh_ext) ||
      !sagnn::aligned16(gates) || !sagnn::aligned16(cell) || !sagnn::aligned16(W) || !sagnn::aligned16(workspace) ||
      (drop_scale && !sagnn::aligned16(drop_scale)))
    return sagnn::fail(SAGNN_ERR_ALIGN, "lstm_bwd: need 16-byte aligned rows");
  if (workspace_bytes < sagnn_lstm_bwd_workspace_bytes(n, t, d))
    return sagnn::fail(SAGNN_ERR_WORKSPACE, "lstm_bwd: workspace of %zu bytes, need %zu", workspace_bytes,
                       sagnn_lstm_bwd_workspace_bytes(n, t, d));
  if (n == 0) return SAGNN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* const dg = static_cast<float*>(workspace);
  if (int rc = d == 64 ? launch<64, false>(x, ld_n, ld_t, h, gates, cell, dh_ext, ld_dhe, drop_scale, W, dx, dW, db, n, t, s, dg)
                       : launch<32, false>(x, ld_n, ld_t, h, gates, cell, dh_ext, ld_dhe, drop_scale, W, dx, dW, db, n, t, s, dg))
    return rc;
  return sagnn::lstm_dw_f16(x, ld_n, ld_t, h, dg, n, t, d, dW, s);
}

#ifdef SAGNN_STAMPS
extern "C" int sagnn_debug_read_stamps_bwd(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_stamps), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif
