// Interval LSTM at d = 128 (BASELINE config 3: MovieLens, --latdim 128) on the f16 matrix cores with
// split fp32 operands — the arithmetic of lstm_f16_kernel.h (f16_split.h: two round-to-nearest f16 pieces,
// three piece products into two fp32 accumulators), a different decomposition:
//
// at d = 128 a wave's slice of W[256, 512] for 16 hidden units is 256 registers of A fragments, so the
// register-resident-W, LDS-resident-h design of lstm_f16_kernel.h does not fit. Here ONE LAUNCH PER STEP
// computes h_t, c_t from x_t, h_{t-1}, c_{t-1} (the recurrence travels through HBM: 1.5 KB per row and
// step against 0.8 MFLOP), and the work is cut over (row tiles) x (4 hidden slices of 32 units):
//   * a workgroup of 4 waves owns one hidden slice for its row tiles (persistent over them); wave w owns
//     8 hidden units. Its two M tiles interleave the gates: C row 4q + r of tile A is gate (r & 1) of
//     hidden unit 2q + (r >> 1) with gates (i, j), tile B the same with (f, o) — so a lane holds i, j, f, o
//     of TWO hidden units of one row, and the wave's W slice is 2 tiles x 8 k-steps x 2 pieces = 128
//     registers, resident for the whole launch; two workgroups share a CU (64 KB of LDS each), so one's
//     gate math runs beside the other's products;
//   * x_t and h_{t-1} of a 64-row tile are split into pieces and shared through LDS as two f16 images
//     [64][256] (16-byte slots XOR-swizzled with the row: 512-byte rows start on the same bank);
//   * a tile that met a value outside the window of the split (f16_split.h, RANGE; or W holding one) is evaluated again by the same lanes
//     with fp32 fmaf chains (they still hold their c_{t-1}) and stored over the fast pass's results;
//   * c_t is kept in the caller's h buffer one interval AHEAD (slot ts + 1 is free until step ts + 1
//     writes h there; the same lane reads c and then writes h at those addresses), so the entry needs no
//     workspace; with saved cell states (training) it is read from / written to them instead.
// Output dropout (training form only): the emitted h is h * mask and the next launch re-makes the un-dropped h from the saved
// cell state and o gate instead of keeping a second copy. Reference: model.py:135-146, TF 1.14 BasicLSTMCell (see lstm_f16_kernel.h).
#include <type_traits>

#include "common.h"
#include "f16_split.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr int D = 128, NC = 4 * D, K2 = 2 * D;
constexpr int kRows = 64, kBT = 4;
constexpr int KS = K2 / 32;                  // 8 k-steps; the first 4 read x, the last 4 h
constexpr int PLANE = kRows * K2 * 2;        // bytes of one f16 image [64][256]

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// FIRST: zero state — h_{t-1} = 0 and c_{t-1} = 0 (h_prev / c_prev are not read, the h half is skipped).
template <bool SAVE, bool FIRST>
__global__ __launch_bounds__(256, 2) void lstm_step128_kernel(
    const float* __restrict__ x_t, int64_t ld_x, const float* __restrict__ h_prev, int64_t ld_hp,
    const float* c_prev, int64_t ld_cp, const float* __restrict__ W, const float* __restrict__ bias,
    float forget_bias, float* h_out, int64_t ld_h, float* c_out, int64_t ld_c,
    float* __restrict__ gates_out, int64_t ld_g, int64_t n, int64_t n_tiles, unsigned int* __restrict__ redo_ctr,
    const float* __restrict__ drop_t, int64_t ld_d, const float* __restrict__ go_prev) {
  // drop_t (training only, SAVE): the mask of DropoutWrapper(output_keep_prob) for this step, row stride ld_d. The emitted h
  // is h * mask, but the recurrence needs the UN-dropped h_{t-1}, and h's slot in HBM now holds the dropped one: the fill
  // re-makes it from the saved cell state (c_prev) and the saved o gate of step t-1 (go_prev, row stride ld_g) with the very
  // expression the previous launch evaluated — bit-identical, no second copy of h.
  // c_prev and h_out may be the SAME addresses (the cell state parked in h's next slot): no __restrict__ on them;
  // a lane reads its c before it stores its h.
  constexpr float kL2E = 1.44269504088896340736f;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, q = lane >> 4;
  const int hb = 32 * (int)blockIdx.y + 8 * wave;       // this wave's 8 hidden units
  const int hid = hb + 2 * q;                           // this lane's two (C rows 4q + r: unit r >> 1, gate r & 1)

  // ---- W slice as A fragments: A row mm = lane & 15 -> hidden hb + 2 (mm >> 2) + ((mm & 3) >> 1), gate 2 tile + (mm & 1);
  //      the gate's exp2 scale is folded in (lstm_split.hip)
  int* const flags = reinterpret_cast<int*>(lds + 2 * PLANE);   // [0], [1]: value outside the split's window in the tile of that parity; [2]: in W
  float k4096 = 4096.f;
  asm volatile("" : "+v"(k4096));
  if (tid < 3) flags[tid] = 0;
  __syncthreads();
  i32x4 wf[2][KS][2];
  {
    const int a_hid = hb + 2 * (m >> 2) + ((m & 3) >> 1);
    RangeTrack wr = range_init();
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
      const int gate = 2 * tile + (m & 1);
      const float sc = gate == 1 ? 2.f * kL2E : -kL2E;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = W[(size_t)(32 * ks + 8 * q + j) * NC + gate * D + a_hid] * sc;
        range_seg4(wr, wv[0], wv[1], wv[2], wv[3]);
        range_seg4(wr, wv[4], wv[5], wv[6], wv[7]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int hd = head2(wv[2 * e], wv[2 * e + 1]);
          wf[tile][ks][0][e] = hd;
          wf[tile][ks][1][e] = tail2(hd, wv[2 * e], wv[2 * e + 1], k4096);
        }
      }
    }
    if (range_bad(wr)) flags[2] = 1;    // ordered before its first reader by the tile loop's barriers
  }
  f32x4 bc[2];   // k * (bias [+ forget bias]) per C row of the two tiles
#pragma unroll
  for (int tile = 0; tile < 2; ++tile)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gate = 2 * tile + (r & 1);
      const float b = bias[gate * D + hid + (r >> 1)] + (gate == 2 ? forget_bias : 0.f);
      bc[tile][r] = (gate == 1 ? 2.f * kL2E : -kL2E) * b;
    }

  // fill mapping: 32 threads per row (float4 each), 8 rows per pass
  const int fr = tid >> 5, fc4 = (tid & 31) * 4;
  RangeTrack xr = range_init();   // range of what this thread moved into the images of the current tile (f16_split.h, RANGE)
  auto h_from_state = [&](float cv, float gv) {   // tanh(c) * o, as the gate math below forms it
    return fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(cv * (2.f * kL2E))), 1.f) * gv;
  };
  auto fill_half = [&](const float* src, int64_t ld, int64_t row0, int rows_valid, int kbase, auto is_h) {
    float4 v[kRows / 8];
#pragma unroll
    for (int p = 0; p < kRows / 8; ++p) {
      const int r = p * 8 + fr;
      v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows_valid) {
        if (decltype(is_h)::value && drop_t) {
          const float4 cv = *reinterpret_cast<const float4*>(c_prev + (row0 + r) * ld_cp + fc4);
          const float4 gv = *reinterpret_cast<const float4*>(go_prev + (row0 + r) * ld_g + fc4);
          v[p] = make_float4(h_from_state(cv.x, gv.x), h_from_state(cv.y, gv.y), h_from_state(cv.z, gv.z), h_from_state(cv.w, gv.w));
        } else {
          v[p] = *reinterpret_cast<const float4*>(src + (row0 + r) * ld + fc4);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < kRows / 8; ++p) {
      const int r = p * 8 + fr;
      if constexpr (decltype(is_h)::value) range_seg4_hi(xr, v[p].x, v[p].y, v[p].z, v[p].w);   // h: top of the window only (f16_split.h)
      else range_seg4(xr, v[p].x, v[p].y, v[p].z, v[p].w);
      const int p0 = head2(v[p].x, v[p].y), p1 = head2(v[p].z, v[p].w);
      const int col = kbase + fc4;
      const int off = r * (K2 * 2) + ((((col >> 3)) ^ (r & 31)) << 4) + ((col >> 2) & 1) * 8;
      *reinterpret_cast<i32x2*>(lds + off) = i32x2{p0, p1};
      *reinterpret_cast<i32x2*>(lds + PLANE + off) = i32x2{tail2(p0, v[p].x, v[p].y, k4096), tail2(p1, v[p].z, v[p].w, k4096)};
    }
  };

  int par = 0;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, par ^= 1) {
    const int64_t row0 = tile * kRows;
    const int rows_valid = (int)(n - row0 < kRows ? n - row0 : kRows);
    if (tid == 0) flags[par] = 0;   // two tiles (and their barriers) after its last reader
    xr = range_init();
    fill_half(x_t, ld_x, row0, rows_valid, 0, std::false_type{});
    if (!FIRST) fill_half(h_prev, ld_hp, row0, rows_valid, D, std::true_type{});
    lds_barrier();
    if (range_bad(xr)) flags[par] = 1;    // after the barrier that follows the reset; read after the next one

    int m_ = m, q_ = q;
    asm volatile("" : "+v"(m_), "+v"(q_));
    f32x2 cps[kBT];   // c_{t-1} of this lane's units: the slow pass needs it after h may have overwritten it
#pragma unroll
    for (int bt = 0; bt < kBT; ++bt) {
      const int row = bt * 16 + m_;
      const bool live = row < rows_valid;
      f32x2 cp = {0.f, 0.f};
      if (!FIRST && live) cp = *reinterpret_cast<const f32x2*>(c_prev + (row0 + row) * ld_cp + hid);
      cps[bt] = cp;
      f32x4 hi[2] = {bc[0], bc[1]}, lo[2];
#pragma unroll
      for (int ks = 0; ks < (FIRST ? KS / 2 : KS); ++ks) {
        const int off = row * (K2 * 2) + (((4 * ks + q_) ^ (row & 31)) << 4);
        const f16x8 b1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(lds + off));
        const f16x8 b2 = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(lds + PLANE + off));
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
          const f16x8 a1 = __builtin_bit_cast(f16x8, wf[tl][ks][0]);
          const f16x8 a2 = __builtin_bit_cast(f16x8, wf[tl][ks][1]);
          lo[tl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : lo[tl], 0, 0, 0);
          lo[tl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, lo[tl], 0, 0, 0);
          hi[tl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, hi[tl], 0, 0, 0);
        }
      }
      f32x4 acc[2];
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tl][r] = fmaf(lo[tl][r], kLoInv, hi[tl][r]);
      // ---- gate math: acc[0] = (i, j) and acc[1] = (f, o) of hidden units hid, hid + 1; the MFMAs delivered k (pre + bias)
      f32x2 gi, gj, gf, go, cn, hn;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        gi[e] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(acc[0][2 * e]));
        gj[e] = fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(acc[0][2 * e + 1])), 1.f);
        gf[e] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(acc[1][2 * e]));
        go[e] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(acc[1][2 * e + 1]));
        cn[e] = fmaf(cp[e], gf[e], gi[e] * gj[e]);
        hn[e] = fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(cn[e] * (2.f * kL2E))), 1.f) * go[e];
      }
      if (live) {
        const int64_t grow = row0 + row;
        f32x2 hv = hn;
        if (drop_t) hv *= *reinterpret_cast<const f32x2*>(drop_t + grow * ld_d + hid);
        *reinterpret_cast<f32x2*>(h_out + grow * ld_h + hid) = hv;
        if (c_out) *reinterpret_cast<f32x2*>(c_out + grow * ld_c + hid) = cn;
        if (SAVE) {
          float* g = gates_out + grow * ld_g + hid;
          *reinterpret_cast<f32x2*>(g) = gi;
          *reinterpret_cast<f32x2*>(g + D) = gj;
          *reinterpret_cast<f32x2*>(g + 2 * D) = gf;
          *reinterpret_cast<f32x2*>(g + 3 * D) = go;
        }
      }
    }
    lds_barrier();   // every wave has read the images before the next tile's fill
    if (flags[par] | flags[2]) {
      // ---- a value outside the split's window: this lane's rows and units again as fp32 fmaf chains, stored over the fast pass
      if (tid == 0 && redo_ctr) atomicAdd(redo_ctr, 1u);
      __syncthreads();   // the fast pass's stores have left
#pragma unroll 1
      for (int bt = 0; bt < kBT; ++bt) {
        const int row = bt * 16 + m_;
        if (row >= rows_valid) continue;
        const int64_t grow = row0 + row;
        f32x2 act[4], cn, hn;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float a[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) a[g] = bias[g * D + hid + e] + (g == 2 ? forget_bias : 0.f);
          for (int k = 0; k < D; ++k) {
            const float xv = x_t[grow * ld_x + k];
#pragma unroll
            for (int g = 0; g < 4; ++g) a[g] = fmaf(xv, W[(size_t)k * NC + g * D + hid + e], a[g]);
          }
          if (!FIRST)
            for (int k = 0; k < D; ++k) {
              const float hv = drop_t ? h_from_state(c_prev[grow * ld_cp + k], go_prev[grow * ld_g + k]) : h_prev[grow * ld_hp + k];
#pragma unroll
              for (int g = 0; g < 4; ++g) a[g] = fmaf(hv, W[(size_t)(D + k) * NC + g * D + hid + e], a[g]);
            }
          act[0][e] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-kL2E * a[0]));
          act[1][e] = tanhf(a[1]);   // relatively accurate near zero: a row that is tiny as a whole keeps its h (lstm_f16_kernel.h)
          act[2][e] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-kL2E * a[2]));
          act[3][e] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-kL2E * a[3]));
          cn[e] = fmaf(cps[bt][e], act[2][e], act[0][e] * act[1][e]);
          hn[e] = tanhf(cn[e]) * act[3][e];
        }
        f32x2 hv2 = hn;
        if (drop_t) hv2 *= *reinterpret_cast<const f32x2*>(drop_t + grow * ld_d + hid);
        *reinterpret_cast<f32x2*>(h_out + grow * ld_h + hid) = hv2;
        if (c_out) *reinterpret_cast<f32x2*>(c_out + grow * ld_c + hid) = cn;
        if (SAVE) {
          float* g = gates_out + grow * ld_g + hid;
          *reinterpret_cast<f32x2*>(g) = act[0];
          *reinterpret_cast<f32x2*>(g + D) = act[1];
          *reinterpret_cast<f32x2*>(g + 2 * D) = act[2];
          *reinterpret_cast<f32x2*>(g + 3 * D) = act[3];
        }
      }
    }
  }
}

}  // namespace

namespace sagnn {

bool lstm_split128_supported(int d) { return d == 128; }

template <bool SAVE, bool FIRST>
static int launch_step(const float* x_t, int64_t ld_x, const float* h_prev, int64_t ld_hp, const float* c_prev,
                       int64_t ld_cp, const float* W, const float* b, float forget_bias, float* h_out, int64_t ld_h,
                       float* c_out, int64_t ld_c, float* gates_out, int64_t ld_g, int64_t n, const float* drop_t, int64_t ld_d,
                       const float* go_prev, hipStream_t s) {
  const size_t lds = (size_t)2 * PLANE + 16;   // 64 KB + flags: two workgroups per CU
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&lstm_step128_kernel<SAVE, FIRST>), lds)) return rc;
  const int64_t n_tiles = (n + kRows - 1) / kRows;
  const int64_t per_slice = cu_count_current() / 2 > 0 ? cu_count_current() / 2 : 1;   // 4 hidden slices share 2 workgroup slots per CU
  const int64_t bx = n_tiles < per_slice ? n_tiles : per_slice;
  hipLaunchKernelGGL((lstm_step128_kernel<SAVE, FIRST>), dim3((unsigned)bx, 4), dim3(256), lds, s, x_t, ld_x, h_prev,
                     ld_hp, c_prev, ld_cp, W, b, forget_bias, h_out, ld_h, c_out, ld_c, gates_out, ld_g, n, n_tiles, redo_counter(), drop_t, ld_d,
                     go_prev);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

// Same contract as lstm_fwd_mfma at d = 128. h [n, t, d] with row stride ld_h >= t*d. drop [n, t, d] (dense) is taken only
// by the training form (gates_out / c_out given: the un-dropped h_{t-1} is re-made from them).
int lstm_fwd_split128(const float* x, int64_t ld_n, int64_t ld_t, int64_t n, int t, const float* W, const float* b,
                      float forget_bias, const float* drop, float* h, int64_t ld_h, float* gates_out, float* c_out,
                      const float* h_init, int64_t ld_hi, const float* c_init, float* c_final, hipStream_t s) {
  const bool save = gates_out != nullptr;
  if (drop && !save) return fail(SAGNN_ERR_ARG, "d = 128 LSTM: an output-dropout mask needs the training form (saved gates / cell)");
  if (n <= 0) return SAGNN_OK;
  ProfileScope prof(kProfLstm, s, n, t);
  for (int ts = 0; ts < t; ++ts) {
    const bool first = ts == 0 && h_init == nullptr;
    const float* h_prev = ts == 0 ? h_init : h + (int64_t)(ts - 1) * D;
    const int64_t ld_hp = ts == 0 ? ld_hi : ld_h;
    // cell state: saved states [n, t, d] when training; otherwise parked in h's NEXT slot (free until the next
    // step writes h there), and handed to c_final after the last step
    const float* c_prev;
    int64_t ld_cp;
    float* c_dst;
    int64_t ld_c;
    if (save) {
      c_prev = ts == 0 ? c_init : c_out + (int64_t)(ts - 1) * D;
      ld_cp = ts == 0 ? D : (int64_t)t * D;
      c_dst = c_out + (int64_t)ts * D;
      ld_c = (int64_t)t * D;
    } else {
      c_prev = ts == 0 ? c_init : h + (int64_t)ts * D;
      ld_cp = ts == 0 ? D : ld_h;
      const bool last = ts + 1 == t;
      c_dst = last ? c_final : h + (int64_t)(ts + 1) * D;
      ld_c = last ? D : ld_h;
    }
    const float* x_t = x + (int64_t)ts * ld_t;
    float* h_t = h + (int64_t)ts * D;
    float* g_t = save ? gates_out + (int64_t)ts * NC : nullptr;
    int rc;
#define SAGNN_GO(SV, FI) \
  rc = launch_step<SV, FI>(x_t, ld_n, h_prev, ld_hp, c_prev, ld_cp, W, b, forget_bias, h_t, ld_h, c_dst, ld_c, g_t, (int64_t)t * NC, n, \
                           drop ? drop + (int64_t)ts * D : nullptr, (int64_t)t * D,                                                 \
                           (drop && ts > 0) ? gates_out + (int64_t)(ts - 1) * NC + 3 * D : nullptr, s)
    if (save && first) SAGNN_GO(true, true);
    else if (save) SAGNN_GO(true, false);
    else if (first) SAGNN_GO(false, true);
    else SAGNN_GO(false, false);
#undef SAGNN_GO
    if (rc) return rc;
    if (save && ts + 1 == t && c_final)   // the training entry does not ask for it; keep the contract anyway
      SAGNN_HIP_TRY(hipMemcpy2DAsync(c_final, D * sizeof(float), c_out + (int64_t)ts * D, (size_t)t * D * sizeof(float),
                                     D * sizeof(float), (size_t)n, hipMemcpyDeviceToDevice, s));
  }
  return SAGNN_OK;
}

}  // namespace sagnn
