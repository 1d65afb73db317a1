// layer_norm over (T, d) -> Q/K/V dense -> exp attention -> mean over queries, one pass over h
// (reference model.py:152-155, Utils/attention.py:35-45, 55-78), with the three dense products on
// the f16 matrix cores over split fp32 operands (f16_split.h: two round-to-nearest pieces per operand,
// three piece products, fp32 accumulation; not a half-precision GEMM — the result is as close to the
// float64 product as an fp32 fmaf chain is).
//
// A workgroup of NW = d/16 waves owns 64 GEMM rows = NB = 64/T nodes (row = nb*T + ts); it needs
// ~68 KB of LDS and <= 256 registers, so TWO workgroups share a CU (two waves per SIMD): one's
// attention phase (VALU + LDS latency) runs beside the other's products (matrix pipe).
//   fill + layer norm: 16 threads per row (float4 each). Moments per row around the row's own mean
//     (two passes in registers), combined over the T rows of a node with the exact pairwise
//     formula (M2 = sum M2_r + d sum (mean_r - mean)^2): two workgroup barriers, no cancellation — or, at d = 64 with T a
//     power of two, inside one wave through lane shuffles (WL below).
//     The normalised rows go to LDS as two f16 images [64][d] (B fragments).
//   Q|K|V: transposed product (W^T y^T). Wave w owns output columns 16w..16w+15 of EACH of Q, K, V
//     — with 16 heads these are whole heads — and keeps that slice of Wq/Wk/Wv in registers as A
//     fragments; Wq and bq carry the score scale log2(e) / sqrt(d_k), so the scores leave as exp2 arguments.
//     In the 16x16 C tile a lane holds 4 consecutive columns of one row: d_k = 4 -> the
//     q, k and v vectors of ONE head; they go to a wave-private LDS table [4][64 rows][q|k|v].
//   attention: LP lanes (1 for T <= 6, 2 for T = 8 / 12, 4 for T = 16) own one (node, head) pair and
//     split its queries; the pair's T q/k/v vectors are T contiguous 48-byte records. The mean over the
//     queries is taken BEFORE the values are touched: out = sum_s w_s v_s with w_s = sum_t e_ts / (sum_s' e_ts' + 1e-8)
//     / T — one multiply-add per (query, key) pair instead of d_k. Keys and values are taken in chunks of
//     <= 8, the partial results of the LP lanes meet through DPP quad butterflies, and the d_k outputs leave as one
//     vector store.
//   A y or W value outside the window of the split (f16_split.h, RANGE: |v| >= 32768, or a 4-element segment that is
//     non-zero but below 2^-18 as a whole — the normalised rows of an input of 1e-12) makes the workgroup redo that
//     tile's Q|K|V records with fp32 fmaf chains (slow_records).
#include "common.h"
#include "f16_split.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr int kRows = 64;
constexpr int kBT = 4;
constexpr int kRec = 48;   // bytes of one (row, 4-column group) record: q[4] | k[4] | v[4]

template <int D>
__device__ __forceinline__ int swz(int row) {
  if (D == 128) return row & 15;      // 256-byte rows all start on bank 0
  if (D == 64) return (row >> 1) & 7;
  const int g = (row >> 2) & 3;
  return (0x78 >> (2 * g)) & 3;
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Padding of a wave's q|k|v table, found by enumerating the bank sets of the attention phase's vector
// reads (lanes of a ds_read_b128 group {0-3, 12-15, 20-27}, ... belong to different (node, head)
// pairs): the 3072-byte head-group stride and, for T = 8 / 16, the T*48-byte node stride are
// multiples of 256 bytes, so without padding every pair of a lane group lands on the same banks
// (4-way). pad_node bytes after each node's T records, pad_group bytes after each 4-column group:
// conflict-free for every T below except T = 2 (2-way).
__host__ __device__ constexpr int pad_node(int t) { return t == 6 ? 16 : 0; }
__host__ __device__ constexpr int pad_group(int t) { return (t == 1 || t == 3 || t == 5) ? 64 : t == 6 ? 32 : 16; }
__host__ __device__ constexpr int lanes_per_pair(int t, int d) {
  if (d == 128) return (t % 4 == 0) ? 4 : (t % 2 == 0) ? 2 : 1;   // d_k = 8: two heads per wave, fewer pairs per tile
  return t == 16 ? 4 : (t == 8 || t == 12) ? 2 : 1;
}

// d_k floats of a q / k / v vector: one 4-column group of the table, or (d_k = 8) two neighbouring groups
template <int DK>
struct HeadVec {
  typedef float type __attribute__((ext_vector_type(DK)));
  static __device__ __forceinline__ type load(const char* p, int gs) { return *reinterpret_cast<const type*>(p); }
  static __device__ __forceinline__ void store(char* p, int gs, type v) { *reinterpret_cast<type*>(p) = v; }
};
template <>
struct HeadVec<8> {
  typedef float type __attribute__((ext_vector_type(8)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ type load(const char* p, int gs) {
    const f4 a = *reinterpret_cast<const f4*>(p), b = *reinterpret_cast<const f4*>(p + gs);
    return type{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  }
  static __device__ __forceinline__ void store(char* p, int gs, type v) {
    *reinterpret_cast<f4*>(p) = f4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f4*>(p + gs) = f4{v[4], v[5], v[6], v[7]};
  }
};

// Sum over the LPR lanes that hold one row (8, 16 or 32 consecutive lanes), result in every lane: DPP butterflies
// inside the 16-lane row (one VALU instruction each), a ds_bpermute only for the step across two rows (d = 128).
// The ds_bpermute form of every step was a chain of 8 dependent LDS round trips per row and pass — a tenth of the
// kernel's time at d = 64, a sixth at d = 128.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
  v = dpp_add<0xB1>(v);                            // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);                            // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);                           // row_half_mirror
  if constexpr (LPR >= 16) v = dpp_add<0x140>(v);  // row_mirror
  if constexpr (LPR == 32) v += __shfl_xor(v, 16);
  return v;
}

// T: intervals (compile time). LP = lanes_per_pair(T, D). d = 128: a workgroup (4 waves) covers 64 of the 128
// output columns of each of Q, K, V; blockIdx.y picks the half (the layer norm is evaluated by both).
// BWD (front of the attention backward pass; LP as in the forward, 1 at d = 128): `out` is the UPSTREAM gradient dL/d(mean context) [n, d] (read),
// and the kernel writes dQ|dK|dV [n*t, 3D] (dqkv_out) and, when y_out is not NULL, the normalised rows y [n*t, D].
template <int D, int T, int LP, bool BWD>
__global__ __launch_bounds__(64 * (D >= 64 ? 4 : D / 16), (D == 128 || (BWD && T >= 8)) ? 1 : 2) void ln_mhsa_split_kernel(
    const float* __restrict__ x, int64_t ld_n, int64_t ld_t, int64_t n, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, int apply_ln, const float* __restrict__ Wq,
    const float* __restrict__ bq, const float* __restrict__ Wk, const float* __restrict__ bk,
    const float* __restrict__ Wv, const float* __restrict__ bv, float* out, int64_t ld_out,
    int64_t n_tiles, float* __restrict__ dqkv_out, float* __restrict__ y_out, unsigned int* __restrict__ redo_ctr) {
  static_assert(!BWD || D <= 64 || LP == 1, "d = 128: the backward front keeps a pair in one lane");
  constexpr int NW = D >= 64 ? 4 : D / 16, NT = 64 * NW, KS = D / 32;
  constexpr int DK = D / 16;                    // 16 heads
  constexpr int HPW = 16 / DK;                  // heads per wave
  constexpr int NB = kRows / T;                 // nodes per tile
  constexpr int ROWS = NB * T;                  // rows in use
  constexpr int PLANE = kRows * D * 2;
  constexpr int LPR = D / 4, RPP = NT / LPR, NFILL = kRows / RPP;
  constexpr int PN = pad_node(T), PG = pad_group(T);
  constexpr int GS = kRows * kRec + NB * PN + PG;   // bytes of one 4-column group of the table
  constexpr int QKVW = 4 * GS;                  // bytes of one wave's q|k|v table
  typedef typename HeadVec<DK>::type vec;

  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const Yp = lds;                                     // 2 images (heads, scaled residuals)
  char* const QKV = lds + 2 * PLANE;                        // NW tables
  float2* const rstat = reinterpret_cast<float2*>(QKV + NW * QKVW);   // [64] (mean_r, M2_r)
  float2* const nstat = rstat + kRows;                      // [NB] (mean, rstd)
  int* const flags = reinterpret_cast<int*>(nstat + NB);    // [0], [1]: a y outside the split's window in the tile of that parity; [2]: a W
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, q = lane >> 4;
  const int cbase = 64 * (int)blockIdx.y + 16 * wave;   // this wave's 16 output columns of each of Q, K, V
  const int col0 = cbase + 4 * q;               // this lane's 4 of them
  const int fr = tid / LPR, fc4 = (tid % LPR) * 4;
  char* const tab = QKV + wave * QKVW;
  // WAVE-LOCAL layer-norm moments (d = 64, T a power of two): the fill deals rows so that all T rows of a node sit in ONE
  // wave (row 16 wave + 4 p + lane / 16 in pass p), and the node's moments meet through two lane shuffles instead of an
  // LDS table, two workgroup barriers and a stage where NB threads work and the rest wait.
  constexpr bool WL = D == 64 && (T & (T - 1)) == 0;
  constexpr int PP = T >= 4 ? T / 4 : 1;        // WL: passes per node
  constexpr int GP = T >= 4 ? 4 : T;            // WL: 16-lane row groups per node inside a pass
  auto fill_row = [&](int p, int fr_now) { return WL ? 16 * wave + 4 * p + (lane >> 4) : p * RPP + fr_now; };

  float k4096 = 4096.f;
  asm volatile("" : "+v"(k4096));               // one register for the whole kernel, not a literal per use
  if (tid < 3) flags[tid] = 0;
  __syncthreads();
  // exp(q.k / sqrt(d_k)) = exp2(q'.k) with q' = q log2(e) / sqrt(d_k): the factor rides in Wq and bq
  const float qscale = 1.44269504088896340736f * (DK == 4 ? 0.5f : DK == 2 ? 0.70710678118654752440f : 0.35355339059327376220f);

  // ---- this wave's slices of Wq / Wk / Wv as A fragments: A[mm][k = 32 ks + 8 q + j] = W[k][16 wave + mm],
  // heads in wf[..][0], scaled residuals in wf[..][1]
  i32x4 wf[3][KS][2];
  {
    const float* const Ws[3] = {Wq, Wk, Wv};
    RangeTrack wr = range_init();
#pragma unroll
    for (int mat = 0; mat < 3; ++mat)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = Ws[mat][(size_t)(32 * ks + 8 * q + j) * D + cbase + m] * (mat == 0 ? qscale : 1.f);
        range_seg4(wr, wv[0], wv[1], wv[2], wv[3]);
        range_seg4(wr, wv[4], wv[5], wv[6], wv[7]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int hd = head2(wv[2 * e], wv[2 * e + 1]);
          wf[mat][ks][0][e] = hd;
          wf[mat][ks][1][e] = tail2(hd, wv[2 * e], wv[2 * e + 1], k4096);
        }
      }
    if (range_bad(wr)) flags[2] = 1;      // ordered before its first reader by the tile loop's barriers
  }
  f32x4 bias3[3];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bias3[0][r] = bq[col0 + r] * qscale;
    bias3[1][r] = bk[col0 + r];
    bias3[2][r] = bv[col0 + r];
  }
  float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (apply_ln) {
    g4 = *reinterpret_cast<const float4*>(gamma + fc4);
    b4 = *reinterpret_cast<const float4*>(beta + fc4);
  }
  const float inv_t = 1.f / (float)T;

  float4 xr[NFILL];
  auto fetch_tile = [&](int64_t tile) {
    const int64_t node0 = tile * NB;
#pragma unroll
    for (int p = 0; p < NFILL; ++p) {
      const int r = fill_row(p, fr);
      const int nb = r / T, ts = r - nb * T;
      xr[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tile < n_tiles && r < ROWS && node0 + nb < n)
        xr[p] = *reinterpret_cast<const float4*>(x + (node0 + nb) * ld_n + (int64_t)ts * ld_t + fc4);
    }
  };
  fetch_tile(blockIdx.x);

  int par = 0;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, par ^= 1) {
    const int64_t node0 = tile * NB;
    if (tid == 0) flags[par] = 0;   // two tiles (and their barriers) after its last reader
    int fr_ = fr, fc4_ = fc4, m_ = m, q_ = q;
    asm volatile("" : "+v"(fr_), "+v"(fc4_), "+v"(m_), "+v"(q_));

    // ---- layer norm moments: per row around the row mean, then the exact combination over a node's rows
    float2 wst[NFILL];                  // WL: (mean, rstd) of the node of this thread's row in pass p
    if constexpr (WL) {
      if (apply_ln) {
        float mr[NFILL], m2[NFILL];
#pragma unroll
        for (int p = 0; p < NFILL; ++p) {
          mr[p] = row_sum<LPR>((xr[p].x + xr[p].y) + (xr[p].z + xr[p].w)) * (1.f / (float)D);
          const float dx = xr[p].x - mr[p], dy = xr[p].y - mr[p], dz = xr[p].z - mr[p], dw = xr[p].w - mr[p];
          m2[p] = row_sum<LPR>((dx * dx + dy * dy) + (dz * dz + dw * dw));
        }
#pragma unroll
        for (int j = 0; j < NFILL / PP; ++j) {          // node 16 wave / T + j ... of this wave (GP row groups x PP passes)
          float sm = 0.f;
#pragma unroll
          for (int i = 0; i < PP; ++i) sm += mr[j * PP + i];
          if constexpr (GP >= 2) sm += __shfl_xor(sm, 16);
          if constexpr (GP == 4) sm += __shfl_xor(sm, 32);
          const float mean = sm * inv_t;
          float q2 = 0.f;
#pragma unroll
          for (int i = 0; i < PP; ++i) {
            const float dm = mr[j * PP + i] - mean;
            q2 += m2[j * PP + i] + (float)D * dm * dm;
          }
          if constexpr (GP >= 2) q2 += __shfl_xor(q2, 16);
          if constexpr (GP == 4) q2 += __shfl_xor(q2, 32);
          const float2 st = make_float2(mean, rsqrtf(q2 * (1.f / (float)(T * D)) + eps));
#pragma unroll
          for (int i = 0; i < PP; ++i) wst[j * PP + i] = st;
          // the fp32 fallback (slow_records) reads the moments from LDS: one lane per node leaves them there
          const int r0 = fill_row(j * PP, fr_);
          if (fc4_ == 0 && r0 % T == 0) nstat[r0 / T] = st;
        }
      }
    } else if (apply_ln) {
#pragma unroll
      for (int p = 0; p < NFILL; ++p) {
        const float mr = row_sum<LPR>((xr[p].x + xr[p].y) + (xr[p].z + xr[p].w)) * (1.f / (float)D);
        const float dx = xr[p].x - mr, dy = xr[p].y - mr, dz = xr[p].z - mr, dw = xr[p].w - mr;
        const float m2 = row_sum<LPR>((dx * dx + dy * dy) + (dz * dz + dw * dw));
        if (fc4_ == 0) rstat[p * RPP + fr_] = make_float2(mr, m2);
      }
      lds_barrier();
      if (tid < NB) {
        float ms = 0.f;
#pragma unroll
        for (int ts = 0; ts < T; ++ts) ms += rstat[tid * T + ts].x;
        const float mean = ms * inv_t;
        float m2 = 0.f;
#pragma unroll
        for (int ts = 0; ts < T; ++ts) {
          const float2 st = rstat[tid * T + ts];
          const float dm = st.x - mean;
          m2 += st.y + (float)D * dm * dm;
        }
        nstat[tid] = make_float2(mean, rsqrtf(m2 * (1.f / (float)(T * D)) + eps));
      }
      lds_barrier();
    }
    // ---- normalise, split into f16 pieces, store the two images
    RangeTrack yr = range_init();
#pragma unroll
    for (int p = 0; p < NFILL; ++p) {
      const int r = fill_row(p, fr_);
      float4 y = xr[p];
      if (apply_ln) {
        const float2 st = WL ? wst[p] : nstat[r < ROWS ? r / T : 0];
        const float ix = st.y * g4.x, iy = st.y * g4.y, iz = st.y * g4.z, iw = st.y * g4.w;
        y.x = xr[p].x * ix + (b4.x - st.x * ix);
        y.y = xr[p].y * iy + (b4.y - st.x * iy);
        y.z = xr[p].z * iz + (b4.z - st.x * iz);
        y.w = xr[p].w * iw + (b4.w - st.x * iw);
      }
      range_seg4(yr, y.x, y.y, y.z, y.w);
      if constexpr (BWD) {
        if (y_out && blockIdx.y == 0 && r < ROWS && node0 + r / T < n)   // d = 128: both column halves normalise the rows, one stores them
          *reinterpret_cast<float4*>(y_out + (node0 * T + r) * D + fc4_) = y;
      }
      const int p0 = head2(y.x, y.y), p1 = head2(y.z, y.w);
      const int off = r * (D * 2) + (((fc4_ >> 3) ^ swz<D>(r)) << 4) + ((fc4_ >> 2) & 1) * 8;
      *reinterpret_cast<i32x2*>(Yp + off) = i32x2{p0, p1};
      *reinterpret_cast<i32x2*>(Yp + PLANE + off) = i32x2{tail2(p0, y.x, y.y, k4096), tail2(p1, y.z, y.w, k4096)};
    }
    fetch_tile(tile + gridDim.x);   // next tile's rows, in flight under this tile's products and attention
    lds_barrier();
    if (range_bad(yr)) flags[par] = 1;    // after the barrier that follows tid 0's reset; read after the next one

    // ---- Q | K | V columns of this wave for every row of the tile -> the wave's table
#pragma unroll
    for (int bt = 0; bt < kBT; ++bt) {
      const int row = bt * 16 + m_;
      const int sw = swz<D>(row);
      f32x4 hi[3] = {bias3[0], bias3[1], bias3[2]};
      f32x4 lo[3];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = row * (D * 2) + (((ks * 4 + q_) ^ sw) << 4);
        const f16x8 b1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(Yp + off));
        const f16x8 b2 = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(Yp + PLANE + off));
#pragma unroll
        for (int mat = 0; mat < 3; ++mat) {
          const f16x8 a1 = __builtin_bit_cast(f16x8, wf[mat][ks][0]);
          const f16x8 a2 = __builtin_bit_cast(f16x8, wf[mat][ks][1]);
          lo[mat] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : lo[mat], 0, 0, 0);
          lo[mat] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, lo[mat], 0, 0, 0);
          hi[mat] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, hi[mat], 0, 0, 0);
        }
      }
      f32x4 acc[3];
#pragma unroll
      for (int mat = 0; mat < 3; ++mat)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mat][r] = fmaf(lo[mat][r], kLoInv, hi[mat][r]);
      char* rec = tab + q_ * GS + row * kRec + (row / T) * PN;
      *reinterpret_cast<f32x4*>(rec) = acc[0];
      *reinterpret_cast<f32x4*>(rec + 16) = acc[1];
      *reinterpret_cast<f32x4*>(rec + 32) = acc[2];
    }
    lds_barrier();   // table complete (wave-private), and every wave is done reading the y images
    if (flags[par] | flags[2]) {
      if (tid == 0 && redo_ctr) atomicAdd(redo_ctr, 1u);
      // ---- slow_records: a y or W value does not fit an f16 piece. The tile's Q|K|V records again as fp32 fmaf
      // chains over y recomputed from x (the moments are still in LDS); thread per (row, matrix, column).
      constexpr int NCOL = 16 * NW;
      const float* const Ws[3] = {Wq, Wk, Wv};
      const float* const bs[3] = {bq, bk, bv};
      for (int idx = tid; idx < ROWS * 3 * NCOL; idx += NT) {
        const int r = idx / (3 * NCOL), rem = idx - r * (3 * NCOL), mat = rem / NCOL, cl = rem - mat * NCOL;
        const int nb = r / T, ts = r - nb * T;
        if (node0 + nb >= n) continue;
        const int col = 64 * (int)blockIdx.y + cl;
        const float sc = mat == 0 ? qscale : 1.f;
        const float2 st = apply_ln ? nstat[nb] : make_float2(0.f, 1.f);
        const float* const xrow = x + (node0 + nb) * ld_n + (int64_t)ts * ld_t;
        float acc = bs[mat][col] * sc;
        for (int k = 0; k < D; ++k) {
          float yv = xrow[k];
          if (apply_ln) {
            const float ik = st.y * gamma[k];
            yv = yv * ik + (beta[k] - st.x * ik);
          }
          acc = fmaf(yv, Ws[mat][(size_t)k * D + col] * sc, acc);
        }
        *reinterpret_cast<float*>(QKV + (cl >> 4) * QKVW + ((cl >> 2) & 3) * GS + r * kRec + nb * PN + mat * 16 + (cl & 3) * 4) = acc;
      }
      __syncthreads();
    }

    // ---- attention of each (node, head) pair of this wave's heads
    constexpr int PAIRS = NB * HPW;
    constexpr int LSTEP = 64 / LP;
    constexpr int TQ = T / LP;                       // queries per lane
    constexpr int KC = T <= 8 ? T : T / 2;           // keys per chunk
    static_assert(T % LP == 0 && T % KC == 0 && LP <= T, "query / key split");
    const int part = lane % LP;
    if constexpr (BWD) {
      // ---- backward of the attention of each pair, given g = dL/d(mean context) of its head (Utils/attention.py:55-78
      // differentiated): a_s = g . v_s / T, p_ts = e_ts / (sum_s' e_ts' + 1e-8), dz_ts = p_ts (a_s - sum_s' p_ts' a_s'),
      // dq_t = sum_s dz_ts k_s / sqrt(d_k), dk_s = sum_t dz_ts q_t / sqrt(d_k), dv_s = g / T sum_t p_ts. The table holds
      // q' = q log2(e) / sqrt(d_k): dk_s = sum_t dz_ts q'_t / log2(e).
      const float scale = qscale * 0.69314718055994530942f;
      // LP lanes of a quad share a pair (T = 8 / 12: 2, T = 16: 4, as in the forward): every lane holds all keys and
      // values, takes TQ of the queries, and the per-key sums over the queries (dk_s, sum_t p_ts) meet through quad DPP
      // butterflies; the dK | dV rows are then stored TQ per lane.
      auto quad_sum = [&](float v) {
        if constexpr (LP > 1) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // lane ^ 1
        if constexpr (LP > 2) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // lane ^ 2
        return v;
      };
#pragma unroll 1
      for (int p = lane / LP; p < PAIRS; p += LSTEP) {
        const int hl = p % HPW, nb = p / HPW;
        const char* base = tab + ((hl * DK) >> 2) * GS + nb * (T * kRec + PN) + ((hl * DK) & 3) * 4;
        const int64_t node = node0 + nb;
        vec g = (vec)(0.f);
        if (node < n) g = *reinterpret_cast<const vec*>(out + node * ld_out + cbase + hl * DK);
        g *= inv_t;
        vec kv[T], vv[T], dk[T];
        float ps[T], asum[T];
#pragma unroll
        for (int ts = 0; ts < T; ++ts) {
          kv[ts] = HeadVec<DK>::load(base + ts * kRec + 16, GS);
          vv[ts] = HeadVec<DK>::load(base + ts * kRec + 32, GS);
          dk[ts] = (vec)(0.f);
          asum[ts] = 0.f;
          float pd = g[0] * vv[ts][0];
#pragma unroll
          for (int c = 1; c < DK; ++c) pd = fmaf(g[c], vv[ts][c], pd);
          ps[ts] = pd;
        }
        float* const drow = dqkv_out + (node * T) * (3 * D) + cbase + hl * DK;
#pragma unroll
        for (int i = 0; i < TQ; ++i) {
          const int tq = part * TQ + i;
          const vec qv = HeadVec<DK>::load(base + tq * kRec, GS);
          float a[T];
          float rs = 0.f;
#pragma unroll
          for (int s_ = 0; s_ < T; ++s_) {
            float z = qv[0] * kv[s_][0];
#pragma unroll
            for (int c = 1; c < DK; ++c) z = fmaf(qv[c], kv[s_][c], z);
            a[s_] = __builtin_amdgcn_exp2f(z);
            rs += a[s_];
          }
          const float inv = __builtin_amdgcn_rcpf(rs + 1e-8f);
          float dot = 0.f;
#pragma unroll
          for (int s_ = 0; s_ < T; ++s_) {
            a[s_] *= inv;
            dot = fmaf(a[s_], ps[s_], dot);
          }
          vec dq = (vec)(0.f);
#pragma unroll
          for (int s_ = 0; s_ < T; ++s_) {
            const float dz = a[s_] * (ps[s_] - dot);
            dq += dz * kv[s_];
            dk[s_] += dz * qv;
            asum[s_] += a[s_];
          }
          if (node < n) *reinterpret_cast<vec*>(drow + tq * (3 * D)) = dq * scale;
        }
        if constexpr (LP > 1) {
          // through SCALAR copies: __builtin_bit_cast on an ext-vector element lvalue reads element 0 on this clang (see the forward's combine)
#pragma unroll
          for (int ts = 0; ts < T; ++ts) {
#pragma unroll
            for (int c = 0; c < DK; ++c) {
              float v = dk[ts][c];
              v = quad_sum(v);
              dk[ts][c] = v;
            }
            asum[ts] = quad_sum(asum[ts]);
          }
        }
        if (node < n) {
#pragma unroll
          for (int ts = 0; ts < T; ++ts) {
            if (LP > 1 && ts / TQ != part) continue;       // the pair's lanes share the T rows: TQ each
            *reinterpret_cast<vec*>(drow + ts * (3 * D) + D) = dk[ts] * 0.69314718055994530942f;
            *reinterpret_cast<vec*>(drow + ts * (3 * D) + 2 * D) = g * asum[ts];
          }
        }
      }
      continue;   // next tile
    }
#pragma unroll 1
    for (int p = lane / LP; p < PAIRS; p += LSTEP) {
      const int hl = p % HPW, nb = p / HPW;
      char* base = tab + ((hl * DK) >> 2) * GS + nb * (T * kRec + PN) + ((hl * DK) & 3) * 4;   // DK = 8: groups 2 hl, 2 hl + 1
      vec qv[TQ];
      float e[TQ][T], rs[TQ];
#pragma unroll
      for (int i = 0; i < TQ; ++i) qv[i] = HeadVec<DK>::load(base + (part * TQ + i) * kRec, GS);
      // e_ts = exp2(q'_t . k_s) for this lane's queries, all keys
#pragma unroll
      for (int ch = 0; ch < T / KC; ++ch) {
        vec k[KC];
#pragma unroll
        for (int s = 0; s < KC; ++s) k[s] = HeadVec<DK>::load(base + (ch * KC + s) * kRec + 16, GS);
#pragma unroll
        for (int i = 0; i < TQ; ++i)
#pragma unroll
          for (int s = 0; s < KC; ++s) {
            float pd = qv[i][0] * k[s][0];
#pragma unroll
            for (int c = 1; c < DK; ++c) pd = fmaf(qv[i][c], k[s][c], pd);
            const float ev = __builtin_amdgcn_exp2f(pd);
            e[i][ch * KC + s] = ev;
            rs[i] = (ch == 0 && s == 0) ? ev : rs[i] + ev;
          }
      }
      // the queries' mean taken on the weights: w_s = sum_t e_ts / (sum_s' e_ts' + 1e-8)
      float w[T];
#pragma unroll
      for (int i = 0; i < TQ; ++i) {
        const float inv = __builtin_amdgcn_rcpf(rs[i] + 1e-8f);
#pragma unroll
        for (int s = 0; s < T; ++s) w[s] = i == 0 ? e[0][s] * inv : fmaf(e[i][s], inv, w[s]);
      }
      vec o = (vec)(0.f);
#pragma unroll
      for (int ch = 0; ch < T / KC; ++ch) {
        vec v[KC];
#pragma unroll
        for (int s = 0; s < KC; ++s) v[s] = HeadVec<DK>::load(base + (ch * KC + s) * kRec + 32, GS);
#pragma unroll
        for (int s = 0; s < KC; ++s) o += w[ch * KC + s] * v[s];
      }
      if (LP > 1) {
        // The LP partial sums of a pair sit in LP neighbouring lanes of a quad: two DPP butterflies (quad_perm) and
        // every lane of the pair holds the total. Each component goes through a SCALAR copy: on this clang (ROCm 7.2)
        // __builtin_bit_cast applied to an ext-vector ELEMENT lvalue (o[c]) reads element 0 whatever c is — the IR of
        // round 2's form of this loop fed o[0] to all four update.dpp calls, which is what was recorded then as "the
        // DPP loop is miscompiled" (the ds_bpermute form recorded as "stale" is correct on the present
        // kernel: 0 of 9.2 M nodes differ, same time — tools/ab/combine_test.py).
#pragma unroll
        for (int c = 0; c < DK; ++c) {
          float oc = o[c];
          oc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, oc), 0xB1, 0xf, 0xf, true));   // lane ^ 1
          if (LP > 2) oc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, oc), 0x4E, 0xf, 0xf, true));   // lane ^ 2
          o[c] = oc;
        }
      }
      const int64_t node = node0 + nb;
      if (part == 0 && node < n) {
        const vec ov = o * inv_t;
        float* dst = out + node * ld_out + cbase + hl * DK;
        if constexpr (DK == 8) {
          *reinterpret_cast<float4*>(dst) = make_float4(ov[0], ov[1], ov[2], ov[3]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(ov[4], ov[5], ov[6], ov[7]);
        } else {
          *reinterpret_cast<vec*>(dst) = ov;
        }
      }
    }
    // the next tile's products overwrite the table only after its own barriers; its fill overwrites
    // the y images, which every wave finished reading before the barrier above
  }
}

}  // namespace

namespace sagnn {

bool mhsa_split_supported(int d, int t, int heads) {
  if (heads != 16 || !(d == 32 || d == 64 || d == 128)) return false;
  return (t >= 1 && t <= 6) || t == 8 || t == 12 || t == 16;
}

template <int D, int T, bool BWD = false>
static int launch_split(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, const float* gamma, const float* beta,
                        float eps, int apply_ln, const float* Wq, const float* bq, const float* Wk, const float* bk,
                        const float* Wv, const float* bv, float* out, int64_t ld_out, hipStream_t s,
                        float* dqkv = nullptr, float* y = nullptr) {
  constexpr int NW = D >= 64 ? 4 : D / 16, NB = kRows / T;
  constexpr int LP = (BWD && D == 128) ? 1 : lanes_per_pair(T, D);
  constexpr int GS = kRows * kRec + NB * pad_node(T) + pad_group(T);
  const size_t lds = (size_t)2 * kRows * D * 2 + (size_t)NW * 4 * GS + (size_t)(kRows + NB) * sizeof(float2) + 16;
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&ln_mhsa_split_kernel<D, T, LP, BWD>), lds)) return rc;
  // d = 64 / 32: ~68 KB / ~34 KB of LDS and <= 256 registers -> two waves per SIMD; d = 128: ~82 KB, one workgroup
  // per CU for each of the two column halves
  const int per_cu = (D == 128 || (BWD && T >= 8)) ? 1 : D == 64 ? 2 : 4;
  constexpr int CB = D == 128 ? 2 : 1;
  const int64_t n_tiles = (n + NB - 1) / NB;
  const int64_t want = (int64_t)cu_count_current() * per_cu / CB;
  const int64_t blocks = n_tiles < want ? n_tiles : (want > 0 ? want : 1);
  ProfileScope prof(kProfMhsa, s, n, T);
  hipLaunchKernelGGL((ln_mhsa_split_kernel<D, T, LP, BWD>), dim3((unsigned)blocks, CB), dim3(64 * NW), lds, s, x, ld_n, ld_t, n,
                     gamma, beta, eps, apply_ln, Wq, bq, Wk, bk, Wv, bv, out, ld_out, n_tiles, dqkv, y, redo_counter());
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

template <int D>
static int dispatch_t(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, const float* gamma,
                      const float* beta, float eps, int apply_ln, const float* Wq, const float* bq, const float* Wk,
                      const float* bk, const float* Wv, const float* bv, float* out, int64_t ld_out, hipStream_t s) {
#define SAGNN_T_CASE(TT) \
  case TT: return launch_split<D, TT>(x, ld_n, ld_t, n, gamma, beta, eps, apply_ln, Wq, bq, Wk, bk, Wv, bv, out, ld_out, s);
  switch (t) {
    SAGNN_T_CASE(1) SAGNN_T_CASE(2) SAGNN_T_CASE(3) SAGNN_T_CASE(4) SAGNN_T_CASE(5) SAGNN_T_CASE(6)
    SAGNN_T_CASE(8) SAGNN_T_CASE(12) SAGNN_T_CASE(16)
    default: return fail(SAGNN_ERR_DIM, "split attention: t = %d has no specialised kernel", t);
  }
#undef SAGNN_T_CASE
}

int ln_mhsa_mean_split(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads,
                       const float* gamma, const float* beta, float eps, int apply_ln, const float* Wq,
                       const float* bq, const float* Wk, const float* bk, const float* Wv, const float* bv,
                       float* out, int64_t ld_out, hipStream_t s) {
  if (!mhsa_split_supported(d, t, heads)) return fail(SAGNN_ERR_DIM, "split attention: unsupported d/t/heads");
  if (d == 128) return dispatch_t<128>(x, ld_n, ld_t, n, t, gamma, beta, eps, apply_ln, Wq, bq, Wk, bk, Wv, bv, out, ld_out, s);
  if (d == 64) return dispatch_t<64>(x, ld_n, ld_t, n, t, gamma, beta, eps, apply_ln, Wq, bq, Wk, bk, Wv, bv, out, ld_out, s);
  return dispatch_t<32>(x, ld_n, ld_t, n, t, gamma, beta, eps, apply_ln, Wq, bq, Wk, bk, Wv, bv, out, ld_out, s);
}

// Front of the attention backward pass on the same kernel (16 heads; d in {32, 64}: every T of the forward — one lane per
// pair up to t = 6, 2 lanes at t = 8 / 12 and 4 at t = 16, one workgroup per CU above t = 8 for the registers; d = 128,
// t <= 6: d_k = 8, one lane per pair, a pair's k / v / dk vectors take 144 of the 512 registers of a one-workgroup-per-CU wave):
// y = LN(x) (or x), Q|K|V, attention backward -> dqkv [n*t, 3d] and, when y is not NULL, y [n*t, d].
bool attn_bwd_front_split_supported(int d, int t, int heads) {
  if (heads != 16) return false;
  if (d == 128) return t >= 1 && t <= 6;
  return (d == 32 || d == 64) && ((t >= 1 && t <= 6) || t == 8 || t == 12 || t == 16);
}

template <int D>
static int dispatch_bwd_t(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, const float* gamma, const float* beta,
                          float eps, int apply_ln, const float* Wq, const float* bq, const float* Wk, const float* bk,
                          const float* Wv, const float* bv, const float* g_out, int64_t ld_g, float* dqkv, float* y,
                          hipStream_t s) {
#define SAGNN_T_CASE(TT)                                                                                              \
  case TT:                                                                                                            \
    return launch_split<D, TT, true>(x, ld_n, ld_t, n, gamma, beta, eps, apply_ln, Wq, bq, Wk, bk, Wv, bv,            \
                                     const_cast<float*>(g_out), ld_g, s, dqkv, y);
  switch (t) {
    SAGNN_T_CASE(1) SAGNN_T_CASE(2) SAGNN_T_CASE(3) SAGNN_T_CASE(4) SAGNN_T_CASE(5) SAGNN_T_CASE(6)
    case 8: case 12: case 16:
      if constexpr (D <= 64) {
        if (t == 8) return launch_split<D, 8, true>(x, ld_n, ld_t, n, gamma, beta, eps, apply_ln, Wq, bq, Wk, bk, Wv, bv, const_cast<float*>(g_out), ld_g, s, dqkv, y);
        if (t == 12) return launch_split<D, 12, true>(x, ld_n, ld_t, n, gamma, beta, eps, apply_ln, Wq, bq, Wk, bk, Wv, bv, const_cast<float*>(g_out), ld_g, s, dqkv, y);
        return launch_split<D, 16, true>(x, ld_n, ld_t, n, gamma, beta, eps, apply_ln, Wq, bq, Wk, bk, Wv, bv, const_cast<float*>(g_out), ld_g, s, dqkv, y);
      }
      [[fallthrough]];
    default: return fail(SAGNN_ERR_DIM, "split attention backward front: t = %d has no specialised kernel", t);
  }
#undef SAGNN_T_CASE
}

int attn_bwd_front_split(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, int d, int heads, const float* gamma,
                         const float* beta, float eps, int apply_ln, const float* Wq, const float* bq, const float* Wk,
                         const float* bk, const float* Wv, const float* bv, const float* g_out, int64_t ld_g, float* dqkv,
                         float* y, hipStream_t s) {
  if (!attn_bwd_front_split_supported(d, t, heads)) return fail(SAGNN_ERR_DIM, "split attention backward front: unsupported d/t/heads");
  if (d == 128)
    return dispatch_bwd_t<128>(x, ld_n, ld_t, n, t, gamma, beta, eps, apply_ln, Wq, bq, Wk, bk, Wv, bv, g_out, ld_g, dqkv, y, s);
  if (d == 64)
    return dispatch_bwd_t<64>(x, ld_n, ld_t, n, t, gamma, beta, eps, apply_ln, Wq, bq, Wk, bk, Wv, bv, g_out, ld_g, dqkv, y, s);
  return dispatch_bwd_t<32>(x, ld_n, ld_t, n, t, gamma, beta, eps, apply_ln, Wq, bq, Wk, bk, Wv, bv, g_out, ld_g, dqkv, y, s);
}

}  // namespace sagnn
