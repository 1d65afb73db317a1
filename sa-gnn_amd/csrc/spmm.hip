// Interval SpMM for gfx950: out = leaky(A·X) + residual, with the running add_n fused.
//
// Replaces the GatherV2 -> SegmentSum -> Pad -> GatherV2 -> Maximum chain of
// Recommender.messagePropagate (reference model.py:80-92) plus the residual add and add_n of
// model.py:124-127. The adjacency is a binary pattern (edge values are dropped by the
// reference, model.py:84-86), so the kernel reads rowptr + colidx only.
//
// Mapping (wave64): a feature row of d floats is covered by LPR = d/4 lanes holding one float4
// each, so one `global_load_dwordx4` wave-instruction gathers G = 64/LPR neighbour rows
// (1 KiB at d = 64). Rows are processed in three degree classes:
//   short  (deg <= short_thresh) one lane-group per row, G rows of the wave's row block at once,
//          a single accumulator added in edge order;
//   medium (<= long_thresh)      the whole wave on one row, G neighbours per instruction, UN
//          instructions in flight, then a cross-group shuffle reduction;
//   long   (> long_thresh)       pre-cut into chunks (plan), one wave per chunk writing a raw
//          partial sum, then a fix-up wave per row adds the partials in chunk order.
// Column indices are fetched coalesced (one per lane) and broadcast with ds_bpermute.
#include <algorithm>
#include <new>
#include <vector>

#include "common.h"

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;              // 4 waves
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kRowsPerWave = 16;         // row block per wavefront (rowptr slice fits one lane each)
// Dataset-sized graphs (tens of thousands of rows: Gowalla, Amazon, MovieLens) give 16-row blocks only a dozen
// waves per CU and every launch is a chain of three dependent L2 round trips: below kSmallRows rows a wave takes
// ONE row per lane group (4 rows at d = 64), four times the waves in flight.
constexpr int64_t kSmallRows = 262144;
constexpr int kUnroll = 8;               // gather instructions in flight per wave

struct Epilogue {
  const float* residual;
  int64_t ldr;
  const float* acc_in;
  int64_t ld_acc_in;
  float* out;
  int64_t ldo;
  float* acc_out;
  int64_t ld_acc_out;
  float leaky;
  // training extras (sagnn_spmm_ex_f32)
  uint8_t* mask_out;       // [n_rows, d/4]: bit j of byte l = 1 iff the activation passed s[4l+j] through
  const uint8_t* mask_in;  // same layout, for out2
  float* out2;             // out2 = v * (mask_in bit ? 1 : slope2), v = acc value if acc_out else y
  int64_t ldo2;
  float slope2;
  int mask_stride;         // bytes per mask row (= lanes per row)
  const float* acc_in2;    // second addend of the running sum (acc_out = acc_in + acc_in2 + y)
  int64_t ld_acc_in2;
};

__device__ __forceinline__ float4 ld4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Streaming accesses (each byte touched once per launch: row outputs, residual / running-sum
// reads, column indices) are marked non-temporal so they do not push gathered rows out of the
// caches: -3 % per launch (4.25 -> 4.10 ms at the roofline config). The gathers themselves stay
// plain loads: making the cold ones non-temporal (hot/cold split by column degree) was 17 %
// SLOWER. -DSAGNN_PLAIN_STREAMS restores plain accesses for A/B runs.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4s(const float* p) {
#ifndef SAGNN_PLAIN_STREAMS
  const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return ld4(p);
#endif
}
__device__ __forceinline__ void st4s(float* p, float4 v) {
#ifndef SAGNN_PLAIN_STREAMS
  const f32x4_t w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<f32x4_t*>(p));
#else
  st4(p, v);
#endif
}
__device__ __forceinline__ int ldi_s(const int32_t* p) {
#ifndef SAGNN_PLAIN_STREAMS
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void add4(float4& a, const float4& b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}

// y = max(leaky*s, s) + residual ; out = y ; acc_out = acc_in + y (+ acc_in2) ; training extras as above.
__device__ __forceinline__ void finish_row(const Epilogue& ep, int64_t row, int col, float4 s) {
  float4 y;
  y.x = fmaxf(ep.leaky * s.x, s.x);
  y.y = fmaxf(ep.leaky * s.y, s.y);
  y.z = fmaxf(ep.leaky * s.z, s.z);
  y.w = fmaxf(ep.leaky * s.w, s.w);
  if (ep.mask_out) {
    // tf.maximum(leaky*x, x) routes the gradient to its FIRST argument on ties (x = 0), so the
    // slope is 1 only where x is strictly the larger one
    const unsigned bits = (s.x > ep.leaky * s.x ? 1u : 0u) | (s.y > ep.leaky * s.y ? 2u : 0u) |
                          (s.z > ep.leaky * s.z ? 4u : 0u) | (s.w > ep.leaky * s.w ? 8u : 0u);
    ep.mask_out[row * ep.mask_stride + (col >> 2)] = (uint8_t)bits;
  }
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ep.residual) {
    r = ld4s(ep.residual + row * ep.ldr + col);
    add4(y, r);
  }
  if (ep.out) st4s(ep.out + row * ep.ldo + col, y);
  if (ep.acc_out) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ep.acc_in) {
      a = (ep.acc_in == ep.residual && ep.ld_acc_in == ep.ldr)
              ? r
              : ld4s(ep.acc_in + row * ep.ld_acc_in + col);
    }
    add4(a, y);
    if (ep.acc_in2) add4(a, ld4s(ep.acc_in2 + row * ep.ld_acc_in2 + col));
    st4s(ep.acc_out + row * ep.ld_acc_out + col, a);
    y = a;
  }
  if (ep.out2) {
    const unsigned bits = ep.mask_in ? ep.mask_in[row * ep.mask_stride + (col >> 2)] : 15u;
    float4 z;
    z.x = (bits & 1u) ? y.x : ep.slope2 * y.x;
    z.y = (bits & 2u) ? y.y : ep.slope2 * y.y;
    z.z = (bits & 4u) ? y.z : ep.slope2 * y.z;
    z.w = (bits & 8u) ? y.w : ep.slope2 * y.w;
    st4s(ep.out2 + row * ep.ldo2 + col, z);
  }
}

// gm[r, :] = g[r, :] * (mask bit ? 1 : slope): seeds the backward chain of the GNN stack.
__global__ void mask_scale_kernel(const float* __restrict__ g, int64_t ldg, const uint8_t* __restrict__ mask,
                                  int mask_stride, float slope, float* __restrict__ gm, int64_t ldgm,
                                  int64_t n_rows, int d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lpr = d >> 2;
  if (i >= n_rows * lpr) return;
  const int64_t row = i / lpr;
  const int l = (int)(i - row * lpr);
  const float4 v = ld4(g + row * ldg + 4 * l);
  const unsigned bits = mask[row * mask_stride + l];
  float4 z;
  z.x = (bits & 1u) ? v.x : slope * v.x;
  z.y = (bits & 2u) ? v.y : slope * v.y;
  z.z = (bits & 4u) ? v.z : slope * v.z;
  z.w = (bits & 8u) ? v.w : slope * v.w;
  st4(gm + row * ldgm + 4 * l, z);
}

// the same for interval blockIdx.y of a slab
__global__ void mask_scale_batch_kernel(const float* __restrict__ g, int64_t ldg, int64_t s_g, const uint8_t* __restrict__ mask,
                                        int64_t s_mask, int mask_stride, float slope, float* __restrict__ gm, int64_t ldgm,
                                        int64_t s_gm, int64_t n_rows, int d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lpr = d >> 2;
  if (i >= n_rows * lpr) return;
  const int64_t k = blockIdx.y;
  const int64_t row = i / lpr;
  const int l = (int)(i - row * lpr);
  const float4 v = ld4(g + k * s_g + row * ldg + 4 * l);
  const unsigned bits = mask[k * s_mask + row * mask_stride + l];
  float4 z;
  z.x = (bits & 1u) ? v.x : slope * v.x;
  z.y = (bits & 2u) ? v.y : slope * v.y;
  z.z = (bits & 4u) ? v.z : slope * v.z;
  z.w = (bits & 8u) ? v.w : slope * v.w;
  st4(gm + k * s_gm + row * ldgm + 4 * l, z);
}

// Whole wave sums X[idx[e], :] for e in [e0, e1): G neighbour rows per load instruction.
// IDENT: the "index" of edge e is e itself (fix-up pass over the partial-sum workspace).
// Returns the total in every lane-group (cross-group xor reduction).
template <int LPR, bool IDENT>
__device__ __forceinline__ float4 wave_row_sum(const int32_t* __restrict__ colidx, int e0, int e1,
                                               const float* __restrict__ X, int64_t ldx,
                                               int lane, int grp, int col, bool lane_on) {
  constexpr int G = kWave / LPR;
  constexpr int STEP = G * kUnroll;  // divides 64 for every LPR in {8,16,32,64}
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int idx_next = -1;
  if (e0 + lane < e1) idx_next = IDENT ? (e0 + lane) : ldi_s(colidx + e0 + lane);
  for (int e = e0; e < e1; e += kWave) {
    const int idx = idx_next;
    const int en = e + kWave;
    idx_next = -1;
    if (en + lane < e1) idx_next = IDENT ? (en + lane) : ldi_s(colidx + en + lane);
    const int cnt = min(kWave, e1 - e);
    for (int j = 0; j < cnt; j += STEP) {
      float4 v[kUnroll];
      int c[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) c[u] = __shfl(idx, j + u * G + grp);
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c[u] >= 0 && lane_on) v[u] = ld4(X + (int64_t)c[u] * ldx + col);
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) add4(acc, v[u]);
    }
  }
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off);
    acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off);
    acc.w += __shfl_xor(acc.w, off);
  }
  return acc;
}

// A wave's share of the row blocks: RPW consecutive rows starting at row0 (short rows by lane groups, medium rows by
// the whole wave; long rows belong to the chunk waves + fix-up).
template <int LPR, int RPW>
__device__ __forceinline__ void rows_wave(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                          const float* __restrict__ X, int64_t ldx, int d, int64_t n_rows, int64_t row0,
                                          int short_t, int long_t, const Epilogue& ep, int lane) {
  constexpr int G = kWave / LPR;
  const int grp = lane / LPR;
  const int sub = lane % LPR;
  const int col = 4 * sub;
  const bool lane_on = col < d;
  const int nr = (int)min((int64_t)RPW, n_rows - row0);
  // lane l (l <= nr) holds rowptr[row0 + l]; lanes beyond replicate the last entry (degree 0).
  const int rp = rowptr[row0 + min(lane, nr)];
  // the shuffle must run with every lane active: a lane masked off by the select below would
  // read 0 from ds_bpermute instead of its neighbour's rowptr entry
  const int rp_up = __shfl_down(rp, 1);
  const int deg_l = (lane < nr) ? (rp_up - rp) : 0;
  unsigned long long medium = __ballot(deg_l > short_t && deg_l <= long_t);

  // ---- short rows: one lane-group per row, G rows per iteration -------------------------
  // The column-index slice of the next iteration is requested before this iteration's
  // gathers so the two dependent HBM round trips overlap.
  int e0_n = __shfl(rp, grp);
  int dg_n = __shfl(deg_l, grp);
  int idx_n = -1;
  if (dg_n <= short_t && sub < dg_n) idx_n = ldi_s(colidx + e0_n + sub);
#pragma unroll 1
  for (int it = 0; it < RPW / G; ++it) {
    const int lr = it * G + grp;
    const int e0 = e0_n;
    const int dg = dg_n;
    int idx = idx_n;
    const bool mine = (lr < nr) && (dg <= short_t);
    const int my_deg = mine ? dg : 0;
    if (it + 1 < RPW / G) {
      e0_n = __shfl(rp, lr + G);
      dg_n = __shfl(deg_l, lr + G);
      idx_n = -1;
      if (dg_n <= short_t && sub < dg_n) idx_n = ldi_s(colidx + e0_n + sub);
    }
    int maxdeg = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int x = __builtin_amdgcn_readlane(deg_l, it * G + g);
      maxdeg = max(maxdeg, x <= short_t ? x : 0);
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int eo = 0; eo < maxdeg; eo += LPR) {
      if (eo > 0) {
        const int k = eo + sub;
        idx = (k < my_deg) ? ldi_s(colidx + e0 + k) : -1;
      }
      const int lim = min(LPR, maxdeg - eo);
      for (int j = 0; j < lim; j += kUnroll) {
        float4 v[kUnroll];
        int c[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) c[u] = __shfl(idx, grp * LPR + j + u);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c[u] >= 0 && lane_on) v[u] = ld4(X + (int64_t)c[u] * ldx + col);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) add4(acc, v[u]);
      }
    }
    if (mine && lane_on) finish_row(ep, row0 + lr, col, acc);
  }

  // ---- medium rows: the whole wave per row ----------------------------------------------
  while (medium) {
    const int lr = __builtin_ctzll(medium);
    medium &= medium - 1;
    const int e0 = __builtin_amdgcn_readlane(rp, lr);
    const int dg = __builtin_amdgcn_readlane(deg_l, lr);
    const float4 s = wave_row_sum<LPR, false>(colidx, e0, e0 + dg, X, ldx, lane, grp, col, lane_on);
    if (grp == 0 && lane_on) finish_row(ep, row0 + lr, col, s);
  }
}

// One launch covers the long-row chunks (first `chunk_blocks` blocks, heaviest work first)
// and the row blocks (remaining blocks).
template <int LPR, int RPW>
__global__ __launch_bounds__(kBlock) void spmm_rows_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
    const float* __restrict__ X, int64_t ldx, int d, int64_t n_rows, int short_t, int long_t,
    const int32_t* __restrict__ chunk_e0, const int32_t* __restrict__ chunk_e1, int64_t n_chunks,
    int chunk_blocks, float* __restrict__ partial, Epilogue ep) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;

  if ((int)blockIdx.x < chunk_blocks) {
    const int grp = lane / LPR;
    const int col = 4 * (lane % LPR);
    const bool lane_on = col < d;
    const int64_t ci = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (ci >= n_chunks) return;
    const float4 s =
        wave_row_sum<LPR, false>(colidx, chunk_e0[ci], chunk_e1[ci], X, ldx, lane, grp, col, lane_on);
    if (grp == 0 && lane_on) st4(partial + ci * (int64_t)d + col, s);
    return;
  }

  const int64_t row0 = ((int64_t)(blockIdx.x - chunk_blocks) * kWavesPerBlock + wave) * RPW;
  if (row0 >= n_rows) return;
  rows_wave<LPR, RPW>(rowptr, colidx, X, ldx, d, n_rows, row0, short_t, long_t, ep, lane);
}

// Fix-up for long rows: add the partial sums of a row in chunk order, then the epilogue.
template <int LPR>
__global__ __launch_bounds__(kBlock) void spmm_fixup_kernel(const int32_t* __restrict__ long_row,
                                                           const int32_t* __restrict__ long_slot,
                                                           int64_t n_long,
                                                           const float* __restrict__ partial, int d,
                                                           Epilogue ep) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int grp = lane / LPR;
  const int col = 4 * (lane % LPR);
  const bool lane_on = col < d;
  const int64_t li = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (li >= n_long) return;
  const float4 s = wave_row_sum<LPR, true>(nullptr, long_slot[li], long_slot[li + 1], partial, d,
                                           lane, grp, col, lane_on);
  if (grp == 0 && lane_on) finish_row(ep, long_row[li], col, s);
}


// ---- all T intervals of a layer, both directions, in ONE launch (dataset-sized graphs) ----------------------------
// The reference's loop over k (model.py:118-129) is independent per interval, so a layer of the stack is 2 T
// independent SpMMs: T with rows = users, T with rows = items. Segment s of a batch is (direction s / T, interval
// s % T); all segments of a direction have the same row count, so a row block finds its segment by division. The
// per-interval operands are slabs of [T, N, d] tensors (or columns of [N, T, d] ones): every pointer of the epilogue
// carries a slab stride, and segment (dir, k) works on base + k * stride — nothing per-call lives in device memory.
struct SegMeta {
  const int32_t* rowptr;
  const int32_t* colidx;
  int32_t short_t, long_t;
};
struct DirArgs {
  const float* X;
  int64_t ldx, s_X;
  Epilogue ep;
  // slab strides (elements; bytes for the masks) of ep's pointers
  int64_t s_res, s_acc_in, s_out, s_acc_out, s_mask_out, s_mask_in, s_out2, s_acc_in2;
};
struct BatchGeom {
  int32_t T;
  int32_t chunk_blocks;          // blocks [0, chunk_blocks): long-row chunks of every segment
  int32_t blocks_u, blocks_i;    // row blocks per user / item segment
  int32_t rows_u, rows_i;
};

__device__ __forceinline__ Epilogue seg_epilogue(const DirArgs& a, int k) {
  Epilogue e = a.ep;
  if (e.residual) e.residual += (int64_t)k * a.s_res;
  if (e.acc_in) e.acc_in += (int64_t)k * a.s_acc_in;
  if (e.out) e.out += (int64_t)k * a.s_out;
  if (e.acc_out) e.acc_out += (int64_t)k * a.s_acc_out;
  if (e.mask_out) e.mask_out += (int64_t)k * a.s_mask_out;
  if (e.mask_in) e.mask_in += (int64_t)k * a.s_mask_in;
  if (e.out2) e.out2 += (int64_t)k * a.s_out2;
  if (e.acc_in2) e.acc_in2 += (int64_t)k * a.s_acc_in2;
  return e;
}

template <int LPR, int RPW>
__global__ __launch_bounds__(kBlock) void spmm_rows_batch_kernel(const SegMeta* __restrict__ meta, BatchGeom g,
                                                                const int32_t* __restrict__ chunk_e0,
                                                                const int32_t* __restrict__ chunk_e1,
                                                                const int32_t* __restrict__ chunk_seg, int64_t n_chunks,
                                                                float* __restrict__ partial, int d, DirArgs au, DirArgs ai) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  if ((int)blockIdx.x < g.chunk_blocks) {
    const int grp = lane / LPR;
    const int col = 4 * (lane % LPR);
    const bool lane_on = col < d;
    const int64_t ci = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (ci >= n_chunks) return;
    const int seg = __builtin_amdgcn_readfirstlane(chunk_seg[ci]);
    const int dir = seg >= g.T, k = seg - dir * g.T;
    const DirArgs& a = dir ? ai : au;
    const float4 s = wave_row_sum<LPR, false>(meta[seg].colidx, chunk_e0[ci], chunk_e1[ci], a.X + (int64_t)k * a.s_X, a.ldx,
                                              lane, grp, col, lane_on);
    if (grp == 0 && lane_on) st4(partial + ci * (int64_t)d + col, s);
    return;
  }
  int rb = (int)blockIdx.x - g.chunk_blocks;
  const int dir = rb >= g.T * g.blocks_u;
  if (dir) rb -= g.T * g.blocks_u;
  const int per = dir ? g.blocks_i : g.blocks_u;
  const int k = rb / per;
  const int64_t n_rows = dir ? g.rows_i : g.rows_u;
  const int64_t row0 = ((int64_t)(rb - k * per) * kWavesPerBlock + wave) * RPW;
  if (row0 >= n_rows) return;
  const SegMeta m = meta[dir * g.T + k];
  const DirArgs& a = dir ? ai : au;
  const Epilogue ep = seg_epilogue(a, k);
  rows_wave<LPR, RPW>(m.rowptr, m.colidx, a.X + (int64_t)k * a.s_X, a.ldx, d, n_rows, row0, m.short_t, m.long_t, ep, lane);
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void spmm_fixup_batch_kernel(const int32_t* __restrict__ long_row,
                                                                 const int32_t* __restrict__ long_slot,
                                                                 const int32_t* __restrict__ long_seg, int64_t n_long,
                                                                 const float* __restrict__ partial, int d, int T,
                                                                 DirArgs au, DirArgs ai) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int grp = lane / LPR;
  const int col = 4 * (lane % LPR);
  const bool lane_on = col < d;
  const int64_t li = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (li >= n_long) return;
  const int seg = __builtin_amdgcn_readfirstlane(long_seg[li]);
  const int dir = seg >= T, k = seg - dir * T;
  const Epilogue ep = seg_epilogue(dir ? ai : au, k);
  const float4 s = wave_row_sum<LPR, true>(nullptr, long_slot[li], long_slot[li + 1], partial, d, lane, grp, col, lane_on);
  if (grp == 0 && lane_on) finish_row(ep, long_row[li], col, s);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// Plan
// ------------------------------------------------------------------------------------------
struct sagnn_spmm_plan {
  sagnn_spmm_plan_info info{};
  const int32_t* d_rowptr = nullptr;
  const int32_t* d_colidx = nullptr;
  // host copies of the chunk metadata (kept for tests / introspection)
  std::vector<int32_t> chunk_row, chunk_e0, chunk_e1;
  std::vector<int32_t> long_row, long_slot;  // long_slot has n_long+1 entries
  // device copies: one allocation [chunk_e0 | chunk_e1 | long_row | long_slot]
  int32_t* d_meta = nullptr;
  const int32_t *d_chunk_e0 = nullptr, *d_chunk_e1 = nullptr, *d_long_row = nullptr,
                *d_long_slot = nullptr;
};

namespace {
constexpr int kDefaultShort = 16;
// measured on MI355X: 256/256 beats 2048/1024 on small graphs (a 2k-edge row is
// a 60-100 us serial chain for one wave) and is on par at 100M edges
constexpr int kDefaultLong = 256;
constexpr int kDefaultChunk = 256;

int check_rowptr(const int32_t* rp, int64_t n_rows, int64_t nnz) {
  if (rp[0] != 0) return sagnn::fail(SAGNN_ERR_CSR, "rowptr[0] = %d, expected 0", rp[0]);
  for (int64_t r = 0; r < n_rows; ++r)
    if (rp[r + 1] < rp[r])
      return sagnn::fail(SAGNN_ERR_CSR, "rowptr decreases at row %lld", (long long)r);
  if (rp[n_rows] != nnz)
    return sagnn::fail(SAGNN_ERR_CSR, "rowptr[n_rows] = %d but nnz = %lld", rp[n_rows],
                       (long long)nnz);
  return SAGNN_OK;
}
}  // namespace

extern "C" int sagnn_csr_check_host(const int32_t* h_rowptr, const int32_t* h_colidx,
                                    int64_t n_rows, int64_t n_src, int64_t nnz) {
  if (!h_rowptr || (nnz > 0 && !h_colidx)) return sagnn::fail(SAGNN_ERR_NULL, "null CSR array");
  if (n_rows < 0 || n_src < 0 || nnz < 0 || nnz > INT32_MAX || n_rows >= INT32_MAX ||
      n_src > INT32_MAX)
    return sagnn::fail(SAGNN_ERR_ARG, "CSR sizes out of int32 range");
  if (int rc = check_rowptr(h_rowptr, n_rows, nnz)) return rc;
  for (int64_t e = 0; e < nnz; ++e)
    if (h_colidx[e] < 0 || h_colidx[e] >= n_src)
      return sagnn::fail(SAGNN_ERR_CSR, "colidx[%lld] = %d outside [0, %lld)", (long long)e,
                         h_colidx[e], (long long)n_src);
  return SAGNN_OK;
}

extern "C" int sagnn_spmm_plan_create(const int32_t* h_rowptr, const int32_t* d_rowptr,
                                      const int32_t* d_colidx, int64_t n_rows, int64_t n_src,
                                      int64_t nnz, const sagnn_spmm_tuning* tuning,
                                      sagnn_spmm_plan** plan_out) {
  if (!plan_out) return sagnn::fail(SAGNN_ERR_NULL, "plan_out is NULL");
  *plan_out = nullptr;
  if (!h_rowptr) return sagnn::fail(SAGNN_ERR_NULL, "h_rowptr is NULL");
  if ((d_rowptr == nullptr) != (d_colidx == nullptr) && nnz > 0)
    return sagnn::fail(SAGNN_ERR_NULL, "d_rowptr and d_colidx must both be given or both NULL");
  // edge cursors advance in steps of 64 past the last edge, so keep that much int32 headroom
  if (n_rows < 0 || n_src < 0 || nnz < 0 || nnz > INT32_MAX - 256 || n_rows >= INT32_MAX - 256 ||
      n_src > INT32_MAX)
    return sagnn::fail(SAGNN_ERR_ARG, "CSR sizes out of int32 range");
  if (int rc = check_rowptr(h_rowptr, n_rows, nnz)) return rc;

  sagnn_spmm_plan* p = new (std::nothrow) sagnn_spmm_plan();
  if (!p) return sagnn::fail(SAGNN_ERR_NOMEM, "out of host memory");
  int short_t = tuning && tuning->short_thresh > 0 ? tuning->short_thresh : kDefaultShort;
  int long_t = tuning && tuning->long_thresh > 0 ? tuning->long_thresh : kDefaultLong;
  int chunk = tuning && tuning->chunk_edges > 0 ? tuning->chunk_edges : kDefaultChunk;
  chunk = (chunk + kWave - 1) / kWave * kWave;
  if (long_t < short_t) long_t = short_t;

  int32_t max_deg = 0;
  try {
    p->long_slot.push_back(0);
    for (int64_t r = 0; r < n_rows; ++r) {
      const int32_t b = h_rowptr[r], e = h_rowptr[r + 1], deg = e - b;
      max_deg = std::max(max_deg, deg);
      if (deg <= long_t) continue;
      // balanced cut: nck chunks of equal length (multiple of 64, <= chunk)
      const int32_t nck = (deg + chunk - 1) / chunk;
      int32_t len = (deg + nck - 1) / nck;
      len = (len + kWave - 1) / kWave * kWave;
      for (int32_t s = b; s < e; s += len) {
        p->chunk_row.push_back((int32_t)r);
        p->chunk_e0.push_back(s);
        p->chunk_e1.push_back(std::min(e, s + len));
      }
      p->long_row.push_back((int32_t)r);
      p->long_slot.push_back((int32_t)p->chunk_row.size());
    }
  } catch (const std::bad_alloc&) {
    delete p;
    return sagnn::fail(SAGNN_ERR_NOMEM, "out of host memory building chunk list");
  }

  p->info.n_rows = n_rows;
  p->info.n_src = n_src;
  p->info.nnz = nnz;
  p->info.n_long_rows = (int64_t)p->long_row.size();
  p->info.n_chunks = (int64_t)p->chunk_row.size();
  p->info.short_thresh = short_t;
  p->info.long_thresh = long_t;
  p->info.chunk_edges = chunk;
  p->info.max_degree = max_deg;
  p->info.on_device = 0;
  p->d_rowptr = d_rowptr;
  p->d_colidx = d_colidx;

  if (d_rowptr) {
    const size_t nck = p->chunk_row.size(), nl = p->long_row.size();
    if (nck > 0) {
      const size_t words = 2 * nck + nl + (nl + 1);
      hipError_t e = hipMalloc((void**)&p->d_meta, words * sizeof(int32_t));
      if (e != hipSuccess) {
        delete p;
        return sagnn::hip_fail(e, "hipMalloc(plan metadata)");
      }
      std::vector<int32_t> host(words);
      std::copy(p->chunk_e0.begin(), p->chunk_e0.end(), host.begin());
      std::copy(p->chunk_e1.begin(), p->chunk_e1.end(), host.begin() + nck);
      std::copy(p->long_row.begin(), p->long_row.end(), host.begin() + 2 * nck);
      std::copy(p->long_slot.begin(), p->long_slot.end(), host.begin() + 2 * nck + nl);
      e = hipMemcpy(p->d_meta, host.data(), words * sizeof(int32_t), hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        (void)hipFree(p->d_meta);
        delete p;
        return sagnn::hip_fail(e, "hipMemcpy(plan metadata)");
      }
      p->d_chunk_e0 = p->d_meta;
      p->d_chunk_e1 = p->d_meta + nck;
      p->d_long_row = p->d_meta + 2 * nck;
      p->d_long_slot = p->d_meta + 2 * nck + nl;
    }
    p->info.on_device = 1;
  }
  *plan_out = p;
  return SAGNN_OK;
}

extern "C" int sagnn_spmm_plan_destroy(sagnn_spmm_plan* plan) {
  if (!plan) return SAGNN_OK;
  if (plan->d_meta) (void)hipFree(plan->d_meta);
  delete plan;
  return SAGNN_OK;
}

extern "C" int sagnn_spmm_plan_get_info(const sagnn_spmm_plan* plan, sagnn_spmm_plan_info* info) {
  if (!plan || !info) return sagnn::fail(SAGNN_ERR_NULL, "plan/info is NULL");
  *info = plan->info;
  return SAGNN_OK;
}

extern "C" int sagnn_spmm_plan_copy_chunks(const sagnn_spmm_plan* plan, int32_t* rows,
                                           int32_t* e_begin, int32_t* e_end, int64_t cap) {
  if (!plan) return sagnn::fail(SAGNN_ERR_NULL, "plan is NULL");
  const int64_t n = plan->info.n_chunks;
  if (cap < n) return sagnn::fail(SAGNN_ERR_ARG, "cap %lld < n_chunks %lld", (long long)cap, (long long)n);
  if (n > 0 && (!rows || !e_begin || !e_end)) return sagnn::fail(SAGNN_ERR_NULL, "null output array");
  std::copy(plan->chunk_row.begin(), plan->chunk_row.end(), rows);
  std::copy(plan->chunk_e0.begin(), plan->chunk_e0.end(), e_begin);
  std::copy(plan->chunk_e1.begin(), plan->chunk_e1.end(), e_end);
  return SAGNN_OK;
}

extern "C" size_t sagnn_spmm_workspace_bytes(const sagnn_spmm_plan* plan, int d) {
  if (!plan || d <= 0) return 0;
  return (size_t)plan->info.n_chunks * (size_t)d * sizeof(float);
}

// ------------------------------------------------------------------------------------------
// Launch
// ------------------------------------------------------------------------------------------
namespace {

template <int LPR>
int launch_spmm(const sagnn_spmm_plan* p, const float* X, int64_t ldx, int d, const Epilogue& ep,
                float* partial, hipStream_t stream) {
  const int64_t n_rows = p->info.n_rows;
  const int64_t n_chunks = p->info.n_chunks;
  const int64_t chunk_blocks = (n_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
  constexpr int G = kWave / LPR;
  constexpr int RPW_SMALL = G > 4 ? G : 4;        // one row per lane group (at least 4 rows)
  const bool small = n_rows < kSmallRows;
  const int64_t rows_per_block = (int64_t)kWavesPerBlock * (small ? RPW_SMALL : kRowsPerWave);
  const int64_t row_blocks = (n_rows + rows_per_block - 1) / rows_per_block;
  const int64_t blocks = chunk_blocks + row_blocks;
  if (blocks > INT32_MAX) return sagnn::fail(SAGNN_ERR_ARG, "grid too large");
  if (blocks > 0) {
    sagnn::ProfileScope prof(sagnn::kProfSpmmRows, stream, p->info.nnz, n_rows);
    if (small)
      hipLaunchKernelGGL((spmm_rows_kernel<LPR, RPW_SMALL>), dim3((unsigned)blocks), dim3(kBlock), 0, stream,
                         p->d_rowptr, p->d_colidx, X, ldx, d, n_rows, p->info.short_thresh,
                         p->info.long_thresh, p->d_chunk_e0, p->d_chunk_e1, n_chunks,
                         (int)chunk_blocks, partial, ep);
    else
      hipLaunchKernelGGL((spmm_rows_kernel<LPR, kRowsPerWave>), dim3((unsigned)blocks), dim3(kBlock), 0, stream,
                         p->d_rowptr, p->d_colidx, X, ldx, d, n_rows, p->info.short_thresh,
                         p->info.long_thresh, p->d_chunk_e0, p->d_chunk_e1, n_chunks,
                         (int)chunk_blocks, partial, ep);
    SAGNN_HIP_TRY(hipGetLastError());
  }
  const int64_t n_long = p->info.n_long_rows;
  if (n_long > 0) {
    sagnn::ProfileScope prof(sagnn::kProfSpmmFixup, stream, n_chunks, n_long);
    const int64_t fb = (n_long + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(spmm_fixup_kernel<LPR>, dim3((unsigned)fb), dim3(kBlock), 0, stream,
                       p->d_long_row, p->d_long_slot, n_long, partial, d, ep);
    SAGNN_HIP_TRY(hipGetLastError());
  }
  return SAGNN_OK;
}

int check_mat(const char* name, const void* ptr, int64_t ld, int d, bool required) {
  if (!ptr) return required ? sagnn::fail(SAGNN_ERR_NULL, "%s is NULL", name) : SAGNN_OK;
  if (!sagnn::aligned16(ptr) || (ld & 3) != 0)
    return sagnn::fail(SAGNN_ERR_ALIGN, "%s: pointer must be 16-byte aligned and ld a multiple of 4", name);
  if (ld < d) return sagnn::fail(SAGNN_ERR_ARG, "%s: ld %lld < d %d", name, (long long)ld, d);
  return SAGNN_OK;
}

}  // namespace

extern "C" int sagnn_spmm_ex_f32(const sagnn_spmm_plan* plan, const float* X, int64_t ldx, int d,
                                 const sagnn_spmm_epilogue* e, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  if (!plan || !e) return sagnn::fail(SAGNN_ERR_NULL, "plan/epilogue is NULL");
  if (!plan->info.on_device) return sagnn::fail(SAGNN_ERR_ARG, "plan was built host-only (no device CSR)");
  if (d < 4 || d > 256 || (d & 3)) return sagnn::fail(SAGNN_ERR_DIM, "d = %d: need a multiple of 4 in [4, 256]", d);
  if (!e->out && !e->acc_out && !e->out2) return sagnn::fail(SAGNN_ERR_NULL, "no output given");
  if (int rc = check_mat("X", X, ldx, d, plan->info.nnz > 0)) return rc;
  if (int rc = check_mat("residual", e->residual, e->ldr, d, false)) return rc;
  if (int rc = check_mat("out", e->out, e->ldo, d, false)) return rc;
  if (int rc = check_mat("acc_in", e->acc_in, e->ld_acc_in, d, false)) return rc;
  if (int rc = check_mat("acc_out", e->acc_out, e->ld_acc_out, d, false)) return rc;
  if (int rc = check_mat("acc_in2", e->acc_in2, e->ld_acc_in2, d, false)) return rc;
  if (e->acc_in2 && !e->acc_out) return sagnn::fail(SAGNN_ERR_ARG, "acc_in2 given without acc_out");
  if (int rc = check_mat("out2", e->out2, e->ldo2, d, false)) return rc;
  if ((e->out && e->out == X) || (e->acc_out && e->acc_out == X) || (e->out2 && e->out2 == X))
    return sagnn::fail(SAGNN_ERR_ARG, "outputs must not alias X");
  if (e->mask_in && !e->out2) return sagnn::fail(SAGNN_ERR_ARG, "mask_in given without out2");
  const size_t need = sagnn_spmm_workspace_bytes(plan, d);
  if (need > 0) {
    if (!workspace || workspace_bytes < need)
      return sagnn::fail(SAGNN_ERR_WORKSPACE, "workspace needs %zu bytes, got %zu", need,
                         workspace ? workspace_bytes : (size_t)0);
    if (!sagnn::aligned16(workspace)) return sagnn::fail(SAGNN_ERR_ALIGN, "workspace not 16-byte aligned");
  }
  if (plan->info.n_rows == 0) return SAGNN_OK;

  Epilogue ep{e->residual, e->ldr,       e->acc_in,  e->ld_acc_in, e->out,   e->ldo,    e->acc_out, e->ld_acc_out,
              e->leaky,    e->mask_out,  e->mask_in, e->out2,      e->ldo2,  e->slope2, d / 4,
              e->acc_in2,  e->ld_acc_in2};
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  switch (sagnn::lanes_per_row(d)) {
    case 8: return launch_spmm<8>(plan, X, ldx, d, ep, partial, s);
    case 16: return launch_spmm<16>(plan, X, ldx, d, ep, partial, s);
    case 32: return launch_spmm<32>(plan, X, ldx, d, ep, partial, s);
    default: return launch_spmm<64>(plan, X, ldx, d, ep, partial, s);
  }
}

extern "C" int sagnn_spmm_f32(const sagnn_spmm_plan* plan, const float* X, int64_t ldx, int d,
                              const float* residual, int64_t ldr, float leaky, float* out,
                              int64_t ldo, const float* acc_in, int64_t ld_acc_in, float* acc_out,
                              int64_t ld_acc_out, void* workspace, size_t workspace_bytes,
                              void* stream) {
  sagnn_spmm_epilogue e{};
  e.leaky = leaky;
  e.residual = residual;
  e.ldr = ldr;
  e.out = out;
  e.ldo = ldo;
  e.acc_in = acc_in;
  e.ld_acc_in = ld_acc_in;
  e.acc_out = acc_out;
  e.ld_acc_out = ld_acc_out;
  return sagnn_spmm_ex_f32(plan, X, ldx, d, &e, workspace, workspace_bytes, stream);
}

// out[r, :] = g[r, :] * (mask bit ? 1 : slope): the seed of the backward chain (what sagnn_gnn_interval_bwd_f32 does first),
// exposed for hosts that run the chain themselves on row slices (parallel.FractionalRunner.run_backward).
extern "C" int sagnn_mask_scale_f32(const float* g, int64_t ldg, const uint8_t* mask, float slope, float* out, int64_t ldo,
                                    int64_t n_rows, int d, void* stream) {
  if (!g || !mask || !out) return sagnn::fail(SAGNN_ERR_NULL, "null pointer");
  if (d < 4 || d > 256 || (d & 3)) return sagnn::fail(SAGNN_ERR_DIM, "d = %d: need a multiple of 4 in [4, 256]", d);
  if (n_rows < 0) return sagnn::fail(SAGNN_ERR_ARG, "n_rows = %lld", (long long)n_rows);
  if (int rc = check_mat("g", g, ldg, d, true)) return rc;
  if (int rc = check_mat("out", out, ldo, d, true)) return rc;
  const int64_t n = n_rows * (d / 4);
  if (n == 0) return SAGNN_OK;
  hipLaunchKernelGGL(mask_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), g, ldg,
                     mask, d / 4, slope, out, ldo, n_rows, d);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

namespace {
int check_interval_plans(const sagnn_spmm_plan* pu, const sagnn_spmm_plan* pi) {
  if (!pu || !pi) return sagnn::fail(SAGNN_ERR_NULL, "plan is NULL");
  if (pu->info.n_src != pi->info.n_rows || pi->info.n_src != pu->info.n_rows)
    return sagnn::fail(SAGNN_ERR_ARG, "plans are not a transposed pair: user %lldx%lld, item %lldx%lld",
                       (long long)pu->info.n_rows, (long long)pu->info.n_src, (long long)pi->info.n_rows,
                       (long long)pi->info.n_src);
  return SAGNN_OK;
}
}  // namespace

extern "C" int sagnn_gnn_interval_ex_f32(const sagnn_spmm_plan* plan_user, const sagnn_spmm_plan* plan_item,
                                         const float* u0, int64_t ld_u0, const float* i0, int64_t ld_i0,
                                         int d, int n_layers, float leaky, float* scratch_u,
                                         float* scratch_i, float* user_out, int64_t ld_uo, float* item_out,
                                         int64_t ld_io, uint8_t* mask_u, uint8_t* mask_i, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (int rc = check_interval_plans(plan_user, plan_item)) return rc;
  if (!u0 || !i0 || !user_out || !item_out) return sagnn::fail(SAGNN_ERR_NULL, "null embedding pointer");
  if (n_layers < 1) return sagnn::fail(SAGNN_ERR_ARG, "n_layers = %d: need >= 1", n_layers);
  if ((mask_u == nullptr) != (mask_i == nullptr)) return sagnn::fail(SAGNN_ERR_NULL, "give both masks or neither");
  const int64_t U = plan_user->info.n_rows, I = plan_item->info.n_rows;
  if (n_layers > 1 && (!scratch_u || !scratch_i))
    return sagnn::fail(SAGNN_ERR_NULL, "scratch buffers required for n_layers > 1");
  // e^l lives in cur; layer l writes e^{l+1} into the other half of the ping-pong scratch
  // (skipped for the last layer, whose only consumer is the running sum).
  const float* cu = u0;
  int64_t lcu = ld_u0;
  const float* ci = i0;
  int64_t lci = ld_i0;
  const int64_t mrow = d / 4;
  for (int l = 0; l < n_layers; ++l) {
    const bool last = (l + 1 == n_layers);
    sagnn_spmm_epilogue eu{}, ei{};
    eu.leaky = ei.leaky = leaky;
    eu.residual = cu;
    eu.ldr = lcu;
    ei.residual = ci;
    ei.ldr = lci;
    eu.out = last ? nullptr : scratch_u + (int64_t)(l & 1) * U * d;
    ei.out = last ? nullptr : scratch_i + (int64_t)(l & 1) * I * d;
    eu.ldo = ei.ldo = d;
    // Running sum sum_l e^l without a write that is only read back: layer 0 of a deeper stack
    // writes e^1 alone; layer 1 starts the sum from e^1 (its residual, already in registers);
    // the LAST layer adds e^0 on the way out (acc_in2). One layer: acc_out = e^0 + e^1 directly.
    const bool sum_here = last || l >= 1;
    if (sum_here) {
      eu.acc_in = (l <= 1) ? cu : user_out;
      eu.ld_acc_in = (l <= 1) ? lcu : ld_uo;
      ei.acc_in = (l <= 1) ? ci : item_out;
      ei.ld_acc_in = (l <= 1) ? lci : ld_io;
      eu.acc_out = user_out;
      eu.ld_acc_out = ld_uo;
      ei.acc_out = item_out;
      ei.ld_acc_out = ld_io;
      if (last && l >= 1) {
        eu.acc_in2 = u0;
        eu.ld_acc_in2 = ld_u0;
        ei.acc_in2 = i0;
        ei.ld_acc_in2 = ld_i0;
      }
    }
    if (mask_u) {
      eu.mask_out = mask_u + (int64_t)l * U * mrow;
      ei.mask_out = mask_i + (int64_t)l * I * mrow;
    }
    if (int rc = sagnn_spmm_ex_f32(plan_user, ci, lci, d, &eu, workspace, workspace_bytes, stream)) return rc;
    if (int rc = sagnn_spmm_ex_f32(plan_item, cu, lcu, d, &ei, workspace, workspace_bytes, stream)) return rc;
    cu = eu.out;
    lcu = d;
    ci = ei.out;
    lci = d;
  }
  return SAGNN_OK;
}

extern "C" int sagnn_gnn_interval_f32(const sagnn_spmm_plan* plan_user,
                                      const sagnn_spmm_plan* plan_item, const float* u0,
                                      int64_t ld_u0, const float* i0, int64_t ld_i0, int d,
                                      int n_layers, float leaky, float* scratch_u, float* scratch_i,
                                      float* user_out, int64_t ld_uo, float* item_out, int64_t ld_io,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  return sagnn_gnn_interval_ex_f32(plan_user, plan_item, u0, ld_u0, i0, ld_i0, d, n_layers, leaky, scratch_u,
                                   scratch_i, user_out, ld_uo, item_out, ld_io, nullptr, nullptr, workspace,
                                   workspace_bytes, stream);
}

// Backward of sagnn_gnn_interval_ex_f32. With g^l = dL/de^l (l = 0..L), G = dL/d(sum_l e^l):
//   g_u^L = G_u,  g_i^L = G_i
//   g_u^l = G_u + g_u^{l+1} + A   (g_i^{l+1} * m_i^{l+1})     (plan_user: rows = users)
//   g_i^l = G_i + g_i^{l+1} + A^T (g_u^{l+1} * m_u^{l+1})     (plan_item: rows = items)
// m^{l+1} = slope mask of the forward layer that produced e^{l+1} (1 where the activation passed
// the sum through, leaky elsewhere). Each step is the forward kernel with slope 1, residual =
// g^{l+1}, acc_in = G, and the masked copy for the next step written by the same epilogue.
extern "C" int sagnn_gnn_interval_bwd_f32(const sagnn_spmm_plan* plan_user, const sagnn_spmm_plan* plan_item,
                                          const float* G_u, int64_t ld_gu, const float* G_i, int64_t ld_gi,
                                          int d, int n_layers, float leaky, const uint8_t* mask_u,
                                          const uint8_t* mask_i, float* scratch_u, float* scratch_i,
                                          float* grad_u0, int64_t ld_du, float* grad_i0, int64_t ld_di,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_interval_plans(plan_user, plan_item)) return rc;
  if (!G_u || !G_i || !grad_u0 || !grad_i0 || !mask_u || !mask_i || !scratch_u || !scratch_i)
    return sagnn::fail(SAGNN_ERR_NULL, "null pointer");
  if (n_layers < 1) return sagnn::fail(SAGNN_ERR_ARG, "n_layers = %d: need >= 1", n_layers);
  if (d < 4 || d > 256 || (d & 3)) return sagnn::fail(SAGNN_ERR_DIM, "d = %d: need a multiple of 4 in [4, 256]", d);
  if (int rc = check_mat("G_u", G_u, ld_gu, d, true)) return rc;
  if (int rc = check_mat("G_i", G_i, ld_gi, d, true)) return rc;
  const int64_t U = plan_user->info.n_rows, I = plan_item->info.n_rows;
  const int64_t mrow = d / 4;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // scratch_x: [4][N][d]: slots 0/1 ping-pong the full gradients g^l, slots 2/3 the masked copies
  float* gfull_u[2] = {scratch_u, scratch_u + U * d};
  float* gmask_u[2] = {scratch_u + 2 * U * d, scratch_u + 3 * U * d};
  float* gfull_i[2] = {scratch_i, scratch_i + I * d};
  float* gmask_i[2] = {scratch_i + 2 * I * d, scratch_i + 3 * I * d};
  // seed: g^L * m^L
  {
    const int threads = 256;
    const int64_t nu = U * mrow, ni = I * mrow;
    if (nu > 0) {
      hipLaunchKernelGGL(mask_scale_kernel, dim3((unsigned)((nu + threads - 1) / threads)), dim3(threads), 0, s, G_u,
                         ld_gu, mask_u + (int64_t)(n_layers - 1) * U * mrow, (int)mrow, leaky, gmask_u[0], (int64_t)d, U, d);
      SAGNN_HIP_TRY(hipGetLastError());
    }
    if (ni > 0) {
      hipLaunchKernelGGL(mask_scale_kernel, dim3((unsigned)((ni + threads - 1) / threads)), dim3(threads), 0, s, G_i,
                         ld_gi, mask_i + (int64_t)(n_layers - 1) * I * mrow, (int)mrow, leaky, gmask_i[0], (int64_t)d, I, d);
      SAGNN_HIP_TRY(hipGetLastError());
    }
  }
  const float* gu_next = G_u;  // g^{l+1}
  int64_t ld_gun = ld_gu;
  const float* gi_next = G_i;
  int64_t ld_gin = ld_gi;
  int cur = 0;  // index of the masked buffers holding g^{l+1} * m^{l+1}
  for (int l = n_layers - 1; l >= 0; --l) {
    const bool final_step = (l == 0);
    sagnn_spmm_epilogue eu{}, ei{};
    eu.leaky = ei.leaky = 1.f;
    eu.residual = gu_next;
    eu.ldr = ld_gun;
    ei.residual = gi_next;
    ei.ldr = ld_gin;
    eu.acc_in = G_u;
    eu.ld_acc_in = ld_gu;
    ei.acc_in = G_i;
    ei.ld_acc_in = ld_gi;
    eu.acc_out = final_step ? grad_u0 : gfull_u[l & 1];
    eu.ld_acc_out = final_step ? ld_du : d;
    ei.acc_out = final_step ? grad_i0 : gfull_i[l & 1];
    ei.ld_acc_out = final_step ? ld_di : d;
    if (!final_step) {
      eu.mask_in = mask_u + (int64_t)(l - 1) * U * mrow;
      ei.mask_in = mask_i + (int64_t)(l - 1) * I * mrow;
      eu.out2 = gmask_u[cur ^ 1];
      ei.out2 = gmask_i[cur ^ 1];
      eu.ldo2 = ei.ldo2 = d;
      eu.slope2 = ei.slope2 = leaky;
    }
    if (int rc = sagnn_spmm_ex_f32(plan_user, gmask_i[cur], d, d, &eu, workspace, workspace_bytes, stream)) return rc;
    if (int rc = sagnn_spmm_ex_f32(plan_item, gmask_u[cur], d, d, &ei, workspace, workspace_bytes, stream)) return rc;
    gu_next = eu.acc_out;
    ld_gun = eu.ld_acc_out;
    gi_next = ei.acc_out;
    ld_gin = ei.ld_acc_out;
    cur ^= 1;
  }
  return SAGNN_OK;
}

// ------------------------------------------------------------------------------------------
// Batch over the T intervals: one launch per LAYER of the stack (+ one fix-up launch)
// ------------------------------------------------------------------------------------------
struct sagnn_spmm_batch {
  int T = 0;
  int64_t U = 0, I = 0;
  int64_t n_chunks = 0, n_long = 0, nnz = 0;
  // device: [SegMeta x 2T] and [chunk_e0 | chunk_e1 | chunk_seg | long_row | long_seg | long_slot (+1)]
  SegMeta* d_meta = nullptr;
  int32_t* d_ints = nullptr;
  const int32_t *d_chunk_e0 = nullptr, *d_chunk_e1 = nullptr, *d_chunk_seg = nullptr, *d_long_row = nullptr,
                *d_long_seg = nullptr, *d_long_slot = nullptr;
};

extern "C" int sagnn_spmm_batch_create(const sagnn_spmm_plan* const* plans_user, const sagnn_spmm_plan* const* plans_item,
                                       int n_intervals, sagnn_spmm_batch** batch_out) {
  if (!batch_out) return sagnn::fail(SAGNN_ERR_NULL, "batch_out is NULL");
  *batch_out = nullptr;
  if (!plans_user || !plans_item) return sagnn::fail(SAGNN_ERR_NULL, "plan table is NULL");
  if (n_intervals < 1 || n_intervals > 4096) return sagnn::fail(SAGNN_ERR_ARG, "n_intervals = %d", n_intervals);
  const int T = n_intervals;
  for (int k = 0; k < T; ++k) {
    if (int rc = check_interval_plans(plans_user[k], plans_item[k])) return rc;
    if (!plans_user[k]->info.on_device || !plans_item[k]->info.on_device)
      return sagnn::fail(SAGNN_ERR_ARG, "interval %d: plan was built host-only", k);
    if (plans_user[k]->info.n_rows != plans_user[0]->info.n_rows || plans_item[k]->info.n_rows != plans_item[0]->info.n_rows)
      return sagnn::fail(SAGNN_ERR_ARG, "interval %d: every interval must have the same user / item counts", k);
  }
  sagnn_spmm_batch* b = new (std::nothrow) sagnn_spmm_batch();
  if (!b) return sagnn::fail(SAGNN_ERR_NOMEM, "out of host memory");
  b->T = T;
  b->U = plans_user[0]->info.n_rows;
  b->I = plans_item[0]->info.n_rows;
  std::vector<SegMeta> meta(2 * (size_t)T);
  std::vector<int32_t> ce0, ce1, cseg, lrow, lseg, lslot;
  try {
    for (int s = 0; s < 2 * T; ++s) {
      const sagnn_spmm_plan* p = s < T ? plans_user[s] : plans_item[s - T];
      meta[s] = SegMeta{p->d_rowptr, p->d_colidx, p->info.short_thresh, p->info.long_thresh};
      const int32_t coff = (int32_t)ce0.size();
      ce0.insert(ce0.end(), p->chunk_e0.begin(), p->chunk_e0.end());
      ce1.insert(ce1.end(), p->chunk_e1.begin(), p->chunk_e1.end());
      cseg.insert(cseg.end(), p->chunk_e0.size(), (int32_t)s);
      for (size_t j = 0; j < p->long_row.size(); ++j) {
        lrow.push_back(p->long_row[j]);
        lseg.push_back((int32_t)s);
        lslot.push_back(coff + p->long_slot[j]);
      }
      b->nnz += p->info.nnz;
    }
    lslot.push_back((int32_t)ce0.size());
  } catch (const std::bad_alloc&) {
    delete b;
    return sagnn::fail(SAGNN_ERR_NOMEM, "out of host memory building the batch tables");
  }
  if (ce0.size() > (size_t)INT32_MAX - 256) {
    delete b;
    return sagnn::fail(SAGNN_ERR_ARG, "too many long-row chunks for one batch");
  }
  b->n_chunks = (int64_t)ce0.size();
  b->n_long = (int64_t)lrow.size();
  hipError_t e = hipMalloc((void**)&b->d_meta, meta.size() * sizeof(SegMeta));
  if (e == hipSuccess) e = hipMemcpy(b->d_meta, meta.data(), meta.size() * sizeof(SegMeta), hipMemcpyHostToDevice);
  if (e == hipSuccess && b->n_chunks > 0) {
    const size_t nck = ce0.size(), nl = lrow.size();
    std::vector<int32_t> host;
    host.reserve(3 * nck + 3 * nl + 1);
    host.insert(host.end(), ce0.begin(), ce0.end());
    host.insert(host.end(), ce1.begin(), ce1.end());
    host.insert(host.end(), cseg.begin(), cseg.end());
    host.insert(host.end(), lrow.begin(), lrow.end());
    host.insert(host.end(), lseg.begin(), lseg.end());
    host.insert(host.end(), lslot.begin(), lslot.end());
    e = hipMalloc((void**)&b->d_ints, host.size() * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(b->d_ints, host.data(), host.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    b->d_chunk_e0 = b->d_ints;
    b->d_chunk_e1 = b->d_ints + nck;
    b->d_chunk_seg = b->d_ints + 2 * nck;
    b->d_long_row = b->d_ints + 3 * nck;
    b->d_long_seg = b->d_ints + 3 * nck + nl;
    b->d_long_slot = b->d_ints + 3 * nck + 2 * nl;
  }
  if (e != hipSuccess) {
    if (b->d_meta) (void)hipFree(b->d_meta);
    if (b->d_ints) (void)hipFree(b->d_ints);
    delete b;
    return sagnn::hip_fail(e, "batch tables (hipMalloc / hipMemcpy)");
  }
  *batch_out = b;
  return SAGNN_OK;
}

extern "C" int sagnn_spmm_batch_destroy(sagnn_spmm_batch* b) {
  if (!b) return SAGNN_OK;
  if (b->d_meta) (void)hipFree(b->d_meta);
  if (b->d_ints) (void)hipFree(b->d_ints);
  delete b;
  return SAGNN_OK;
}

extern "C" size_t sagnn_spmm_batch_workspace_bytes(const sagnn_spmm_batch* b, int d) {
  if (!b || d <= 0) return 0;
  return (size_t)b->n_chunks * (size_t)d * sizeof(float);
}

namespace {

template <int LPR>
int launch_batch(const sagnn_spmm_batch* b, int d, const DirArgs& au, const DirArgs& ai, float* partial, hipStream_t stream) {
  constexpr int G = kWave / LPR;
  constexpr int RPW_SMALL = G > 4 ? G : 4;
  const bool small = (b->U > b->I ? b->U : b->I) < kSmallRows;
  const int64_t rpb = (int64_t)kWavesPerBlock * (small ? RPW_SMALL : kRowsPerWave);
  const int64_t chunk_blocks = (b->n_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
  const int64_t bu = (b->U + rpb - 1) / rpb, bi = (b->I + rpb - 1) / rpb;
  const int64_t blocks = chunk_blocks + (int64_t)b->T * (bu + bi);
  if (blocks > INT32_MAX || b->U > INT32_MAX || b->I > INT32_MAX) return sagnn::fail(SAGNN_ERR_ARG, "grid too large");
  const BatchGeom g{b->T, (int32_t)chunk_blocks, (int32_t)bu, (int32_t)bi, (int32_t)b->U, (int32_t)b->I};
  if (blocks > 0) {
    sagnn::ProfileScope prof(sagnn::kProfSpmmRows, stream, b->nnz, (int64_t)b->T * (b->U + b->I));
    if (small)
      hipLaunchKernelGGL((spmm_rows_batch_kernel<LPR, RPW_SMALL>), dim3((unsigned)blocks), dim3(kBlock), 0, stream, b->d_meta,
                         g, b->d_chunk_e0, b->d_chunk_e1, b->d_chunk_seg, b->n_chunks, partial, d, au, ai);
    else
      hipLaunchKernelGGL((spmm_rows_batch_kernel<LPR, kRowsPerWave>), dim3((unsigned)blocks), dim3(kBlock), 0, stream,
                         b->d_meta, g, b->d_chunk_e0, b->d_chunk_e1, b->d_chunk_seg, b->n_chunks, partial, d, au, ai);
    SAGNN_HIP_TRY(hipGetLastError());
  }
  if (b->n_long > 0) {
    sagnn::ProfileScope prof(sagnn::kProfSpmmFixup, stream, b->n_chunks, b->n_long);
    const int64_t fb = (b->n_long + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(spmm_fixup_batch_kernel<LPR>, dim3((unsigned)fb), dim3(kBlock), 0, stream, b->d_long_row,
                       b->d_long_slot, b->d_long_seg, b->n_long, partial, d, b->T, au, ai);
    SAGNN_HIP_TRY(hipGetLastError());
  }
  return SAGNN_OK;
}

int launch_batch_d(const sagnn_spmm_batch* b, int d, const DirArgs& au, const DirArgs& ai, float* partial, hipStream_t s) {
  switch (sagnn::lanes_per_row(d)) {
    case 8: return launch_batch<8>(b, d, au, ai, partial, s);
    case 16: return launch_batch<16>(b, d, au, ai, partial, s);
    case 32: return launch_batch<32>(b, d, au, ai, partial, s);
    default: return launch_batch<64>(b, d, au, ai, partial, s);
  }
}

struct Slab {           // a [T, N, d]-like operand: interval k's matrix starts at p + k * slab, rows ld apart
  float* p;
  int64_t ld, slab;
};

int check_slab(const char* name, const float* p, int64_t ld, int64_t slab, int d) {
  if (int rc = check_mat(name, p, ld, d, true)) return rc;
  if (slab & 3) return sagnn::fail(SAGNN_ERR_ALIGN, "%s: slab stride must be a multiple of 4", name);
  return SAGNN_OK;
}

int check_batch_ws(const sagnn_spmm_batch* b, int d, const void* workspace, size_t workspace_bytes) {
  const size_t need = sagnn_spmm_batch_workspace_bytes(b, d);
  if (need == 0) return SAGNN_OK;
  if (!workspace || workspace_bytes < need)
    return sagnn::fail(SAGNN_ERR_WORKSPACE, "workspace needs %zu bytes, got %zu", need, workspace ? workspace_bytes : (size_t)0);
  if (!sagnn::aligned16(workspace)) return sagnn::fail(SAGNN_ERR_ALIGN, "workspace not 16-byte aligned");
  return SAGNN_OK;
}

}  // namespace

// The whole GNN loop of model.py:118-129 — every interval, every layer — in L row launches (+ L fix-up launches when
// the graphs have long rows): sagnn_gnn_interval_ex_f32 for all T intervals at once. Operands are slabs.
extern "C" int sagnn_gnn_stack_f32(const sagnn_spmm_batch* b, const float* u0, int64_t ld_u0, int64_t slab_u0,
                                   const float* i0, int64_t ld_i0, int64_t slab_i0, int d, int n_layers, float leaky,
                                   float* scratch_u, float* scratch_i, float* user_out, int64_t ld_uo, int64_t slab_uo,
                                   float* item_out, int64_t ld_io, int64_t slab_io, uint8_t* mask_u, uint8_t* mask_i,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!b) return sagnn::fail(SAGNN_ERR_NULL, "batch is NULL");
  if (!u0 || !i0 || !user_out || !item_out) return sagnn::fail(SAGNN_ERR_NULL, "null embedding pointer");
  if (d < 4 || d > 256 || (d & 3)) return sagnn::fail(SAGNN_ERR_DIM, "d = %d: need a multiple of 4 in [4, 256]", d);
  if (n_layers < 1) return sagnn::fail(SAGNN_ERR_ARG, "n_layers = %d: need >= 1", n_layers);
  if ((mask_u == nullptr) != (mask_i == nullptr)) return sagnn::fail(SAGNN_ERR_NULL, "give both masks or neither");
  if (n_layers > 1 && (!scratch_u || !scratch_i)) return sagnn::fail(SAGNN_ERR_NULL, "scratch buffers required for n_layers > 1");
  if (int rc = check_slab("u0", u0, ld_u0, slab_u0, d)) return rc;
  if (int rc = check_slab("i0", i0, ld_i0, slab_i0, d)) return rc;
  if (int rc = check_slab("user_out", user_out, ld_uo, slab_uo, d)) return rc;
  if (int rc = check_slab("item_out", item_out, ld_io, slab_io, d)) return rc;
  if (n_layers > 1 && (!sagnn::aligned16(scratch_u) || !sagnn::aligned16(scratch_i)))
    return sagnn::fail(SAGNN_ERR_ALIGN, "scratch buffers must be 16-byte aligned");
  if (int rc = check_batch_ws(b, d, workspace, workspace_bytes)) return rc;
  const int64_t U = b->U, I = b->I, T = b->T;
  const int64_t mrow = d / 4;
  // e^l of interval k: layer 0 reads the callers' tables, layer l >= 1 the ping-pong scratch [2][T][N][d]
  Slab cu{const_cast<float*>(u0), ld_u0, slab_u0}, ci{const_cast<float*>(i0), ld_i0, slab_i0};
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int l = 0; l < n_layers; ++l) {
    const bool last = (l + 1 == n_layers);
    DirArgs au{}, ai{};
    au.ep.leaky = ai.ep.leaky = leaky;
    au.ep.mask_stride = ai.ep.mask_stride = (int)mrow;
    au.X = ci.p, au.ldx = ci.ld, au.s_X = ci.slab;       // rows = users gather item rows
    ai.X = cu.p, ai.ldx = cu.ld, ai.s_X = cu.slab;
    au.ep.residual = cu.p, au.ep.ldr = cu.ld, au.s_res = cu.slab;
    ai.ep.residual = ci.p, ai.ep.ldr = ci.ld, ai.s_res = ci.slab;
    Slab nu{nullptr, d, U * d}, ni{nullptr, d, I * d};
    if (!last) {
      nu.p = scratch_u + (int64_t)(l & 1) * T * U * d;
      ni.p = scratch_i + (int64_t)(l & 1) * T * I * d;
      au.ep.out = nu.p, au.ep.ldo = nu.ld, au.s_out = nu.slab;
      ai.ep.out = ni.p, ai.ep.ldo = ni.ld, ai.s_out = ni.slab;
    }
    // running sum as in sagnn_gnn_interval_ex_f32: layer 0 of a deeper stack writes e^1 alone, layer 1 starts the sum
    // from its residual, the last layer adds e^0 on the way out
    if (last || l >= 1) {
      if (l <= 1) {
        au.ep.acc_in = cu.p, au.ep.ld_acc_in = cu.ld, au.s_acc_in = cu.slab;
        ai.ep.acc_in = ci.p, ai.ep.ld_acc_in = ci.ld, ai.s_acc_in = ci.slab;
      } else {
        au.ep.acc_in = user_out, au.ep.ld_acc_in = ld_uo, au.s_acc_in = slab_uo;
        ai.ep.acc_in = item_out, ai.ep.ld_acc_in = ld_io, ai.s_acc_in = slab_io;
      }
      au.ep.acc_out = user_out, au.ep.ld_acc_out = ld_uo, au.s_acc_out = slab_uo;
      ai.ep.acc_out = item_out, ai.ep.ld_acc_out = ld_io, ai.s_acc_out = slab_io;
      if (last && l >= 1) {
        au.ep.acc_in2 = u0, au.ep.ld_acc_in2 = ld_u0, au.s_acc_in2 = slab_u0;
        ai.ep.acc_in2 = i0, ai.ep.ld_acc_in2 = ld_i0, ai.s_acc_in2 = slab_i0;
      }
    }
    if (mask_u) {   // [T][L][N][d/4]
      au.ep.mask_out = mask_u + (int64_t)l * U * mrow, au.s_mask_out = (int64_t)n_layers * U * mrow;
      ai.ep.mask_out = mask_i + (int64_t)l * I * mrow, ai.s_mask_out = (int64_t)n_layers * I * mrow;
    }
    if (int rc = launch_batch_d(b, d, au, ai, static_cast<float*>(workspace), s)) return rc;
    cu = nu;
    ci = ni;
  }
  return SAGNN_OK;
}

// Backward of sagnn_gnn_stack_f32 (see sagnn_gnn_interval_bwd_f32 for the recurrence): G_u / G_i are the gradients at the
// interval outputs as slabs, masks [T][L][N][d/4] as the forward recorded them; scratch_x: [4][T][N][d]. `b` must be the
// batch of the ADJOINT patterns (for canonical matrices: the same batch).
extern "C" int sagnn_gnn_stack_bwd_f32(const sagnn_spmm_batch* b, const float* G_u, int64_t ld_gu, int64_t slab_gu,
                                       const float* G_i, int64_t ld_gi, int64_t slab_gi, int d, int n_layers, float leaky,
                                       const uint8_t* mask_u, const uint8_t* mask_i, float* scratch_u, float* scratch_i,
                                       float* grad_u0, int64_t ld_du, int64_t slab_du, float* grad_i0, int64_t ld_di,
                                       int64_t slab_di, void* workspace, size_t workspace_bytes, void* stream) {
  if (!b) return sagnn::fail(SAGNN_ERR_NULL, "batch is NULL");
  if (!G_u || !G_i || !grad_u0 || !grad_i0 || !mask_u || !mask_i || !scratch_u || !scratch_i)
    return sagnn::fail(SAGNN_ERR_NULL, "null pointer");
  if (n_layers < 1) return sagnn::fail(SAGNN_ERR_ARG, "n_layers = %d: need >= 1", n_layers);
  if (d < 4 || d > 256 || (d & 3)) return sagnn::fail(SAGNN_ERR_DIM, "d = %d: need a multiple of 4 in [4, 256]", d);
  if (int rc = check_slab("G_u", G_u, ld_gu, slab_gu, d)) return rc;
  if (int rc = check_slab("G_i", G_i, ld_gi, slab_gi, d)) return rc;
  if (int rc = check_slab("grad_u0", grad_u0, ld_du, slab_du, d)) return rc;
  if (int rc = check_slab("grad_i0", grad_i0, ld_di, slab_di, d)) return rc;
  if (int rc = check_batch_ws(b, d, workspace, workspace_bytes)) return rc;
  const int64_t U = b->U, I = b->I, T = b->T;
  const int64_t mrow = d / 4;
  const int64_t s_mask_u = (int64_t)n_layers * U * mrow, s_mask_i = (int64_t)n_layers * I * mrow;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* gfull_u[2] = {scratch_u, scratch_u + T * U * d};
  float* gmask_u[2] = {scratch_u + 2 * T * U * d, scratch_u + 3 * T * U * d};
  float* gfull_i[2] = {scratch_i, scratch_i + T * I * d};
  float* gmask_i[2] = {scratch_i + 2 * T * I * d, scratch_i + 3 * T * I * d};
  // seed: g^L * m^L of every interval (one launch per node type)
  {
    const int threads = 256;
    const int64_t nu = U * mrow, ni = I * mrow;
    if (nu > 0) {
      hipLaunchKernelGGL(mask_scale_batch_kernel, dim3((unsigned)((nu + threads - 1) / threads), (unsigned)T), dim3(threads), 0, s,
                         G_u, ld_gu, slab_gu, mask_u + (int64_t)(n_layers - 1) * U * mrow, s_mask_u, (int)mrow, leaky,
                         gmask_u[0], (int64_t)d, U * d, U, d);
      SAGNN_HIP_TRY(hipGetLastError());
    }
    if (ni > 0) {
      hipLaunchKernelGGL(mask_scale_batch_kernel, dim3((unsigned)((ni + threads - 1) / threads), (unsigned)T), dim3(threads), 0, s,
                         G_i, ld_gi, slab_gi, mask_i + (int64_t)(n_layers - 1) * I * mrow, s_mask_i, (int)mrow, leaky,
                         gmask_i[0], (int64_t)d, I * d, I, d);
      SAGNN_HIP_TRY(hipGetLastError());
    }
  }
  Slab gun{const_cast<float*>(G_u), ld_gu, slab_gu}, gin{const_cast<float*>(G_i), ld_gi, slab_gi};   // g^{l+1}
  int cur = 0;
  for (int l = n_layers - 1; l >= 0; --l) {
    const bool final_step = (l == 0);
    DirArgs au{}, ai{};
    au.ep.leaky = ai.ep.leaky = 1.f;
    au.ep.mask_stride = ai.ep.mask_stride = (int)mrow;
    au.X = gmask_i[cur], au.ldx = d, au.s_X = I * d;
    ai.X = gmask_u[cur], ai.ldx = d, ai.s_X = U * d;
    au.ep.residual = gun.p, au.ep.ldr = gun.ld, au.s_res = gun.slab;
    ai.ep.residual = gin.p, ai.ep.ldr = gin.ld, ai.s_res = gin.slab;
    au.ep.acc_in = G_u, au.ep.ld_acc_in = ld_gu, au.s_acc_in = slab_gu;
    ai.ep.acc_in = G_i, ai.ep.ld_acc_in = ld_gi, ai.s_acc_in = slab_gi;
    Slab ou = final_step ? Slab{grad_u0, ld_du, slab_du} : Slab{gfull_u[l & 1], d, U * d};
    Slab oi = final_step ? Slab{grad_i0, ld_di, slab_di} : Slab{gfull_i[l & 1], d, I * d};
    au.ep.acc_out = ou.p, au.ep.ld_acc_out = ou.ld, au.s_acc_out = ou.slab;
    ai.ep.acc_out = oi.p, ai.ep.ld_acc_out = oi.ld, ai.s_acc_out = oi.slab;
    if (!final_step) {
      au.ep.mask_in = mask_u + (int64_t)(l - 1) * U * mrow, au.s_mask_in = s_mask_u;
      ai.ep.mask_in = mask_i + (int64_t)(l - 1) * I * mrow, ai.s_mask_in = s_mask_i;
      au.ep.out2 = gmask_u[cur ^ 1], au.ep.ldo2 = d, au.s_out2 = U * d;
      ai.ep.out2 = gmask_i[cur ^ 1], ai.ep.ldo2 = d, ai.s_out2 = I * d;
      au.ep.slope2 = ai.ep.slope2 = leaky;
    }
    if (int rc = launch_batch_d(b, d, au, ai, static_cast<float*>(workspace), s)) return rc;
    gun = ou;
    gin = oi;
    cur ^= 1;
  }
  return SAGNN_OK;
}
