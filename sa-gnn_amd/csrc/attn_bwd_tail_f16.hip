// Tail of the attention backward pass on the f16 matrix cores (gradient of reference Utils/attention.py:31-45's
// three dense layers), one pass over dQ|dK|dV:
//
//   dy [R, D]    = dQKV [R, 3D] @ Wqkv^T [3D, D]      (written over y: y is dead after this kernel)
//   dWqkv [D,3D] += y^T [D, R] @ dQKV [R, 3D]
//   dbqkv [3D]   += column sums of dQKV
//
// attn_bwd_tail.hip runs the two products on v_mfma_f32_32x32x2_f32 and is bound by it (15 ms per step of the
// roofline workload). Here the operands are split as in f16_split.h (two round-to-nearest f16 pieces, three piece
// products, fp32 accumulation) and the kernel is bound by its 1.3 KB of HBM traffic per row instead.
//
// A block (8 waves at D = 64: two per SIMD; 4 waves and two blocks per CU at D = 32) walks 32-row chunks. A chunk's y | dQ | dK | dV rows are split once and
// kept as four [32][D] f16 sub-images (x 2 pieces, x 2 buffers = 64 KB at D = 64), 16-byte chunks XOR-swizzled
// with the row. dy reads dQKV ROW-wise (ds_read_b128: 8 consecutive k of a row) against W^T fragments resident in
// registers; dW sums over the chunk's ROWS, so both of its operands are read TRANSPOSED from the same images
// (ds_read_b64_tr_b16: a 16-lane group fetches a 4-row x 16-column block and each lane receives one column —
// no second copy, no cross-lane traffic). dW's 6 (3 at D = 32) head and residual accumulator tiles per wave live
// across all chunks of the block and are flushed once with float atomics; db rides on the same product as a tile row of
// row factors. dQKV is a gradient of arbitrary scale: its rows enter the images at exact power-of-two scales (GRADIENT
// RANGE below), so dy is accurate per row and dW / db relative to the sum of their terms' magnitudes at any scale.
// A chunk the window cannot hold takes no part in the MFMAs: the block evaluates it with fp32 fmaf chains (a W out of range: every chunk).
//
// The same kernel in its LSTM form (NX = 2, NQ = 4, no dy; lstm_dw_f16 below) takes the weight gradient of the interval LSTM
// (reference model.py:135-146 differentiated) from the gate gradients that the BPTT launch stored time-major:
//   dW [2D, 4D] += sum over steps s and nodes of [x_s | h_{s-1}]^T dG_s
// rows are (step, node) pairs, the left operand's two blocks come from x (its node / interval strides) and from h one
// step back (zero at s = 0), and each of a wave's 16 dW tiles keeps ONE accumulator: the two cross products of a chunk go
// through a transient tile that is folded in while the next tile's MFMAs run.
#include "common.h"
#include "f16_split.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int kRows = 32;

// byte offset of 16-byte chunk `chunk` (8 columns) of row `row` in a [32][D] f16 sub-image
template <int D>
__device__ __forceinline__ int img_off(int row, int chunk) {
  const int sw = D == 64 ? (row & 7) : ((row >> 1) & 3);
  return row * (D * 2) + ((chunk ^ sw) << 4);
}

// Transposed operand: rows 8 kq .. 8 kq + 7 of column 16 ct + (lane & 15) of a sub-image, as 8 consecutive k of an
// MFMA A / B operand. Lane 4 qq + pp of a 16-lane group supplies the address of row qq, columns 4 pp .. 4 pp + 3 of
// each 4-row block and receives column (lane & 15) of its four rows.
template <int D>
__device__ __forceinline__ i32x4 read_tr(const char* sub, int ct, int lane) {
  const int kq = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const char* p0 = sub + img_off<D>(8 * kq + qq, 2 * ct + (pp >> 1)) + 8 * (pp & 1);
  const char* p1 = sub + img_off<D>(8 * kq + 4 + qq, 2 * ct + (pp >> 1)) + 8 * (pp & 1);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
  const i32x2 lo = __builtin_bit_cast(i32x2, a), hi = __builtin_bit_cast(i32x2, b);
  return i32x4{lo[0], lo[1], hi[0], hi[1]};
}

// db rides on the dW product as a tile row of ones (db[o] = sum_r 1 * dQKV[r][o]: the 32 rows of a chunk are summed inside the
// MFMA, so a node's rows, whose dK sum to ~0, cancel before they meet the running total). Tile column b of a wave's NBW belongs
// to the wave with wm = bias_owner(b); bias_slot(b) numbers an owner's columns.
__host__ __device__ constexpr int bias_owner(int b, int nmt, int nbw) { return b * nmt / nbw; }
__host__ __device__ constexpr int bias_slot(int b, int nmt, int nbw) {
  int first = b;
  while (first > 0 && bias_owner(first - 1, nmt, nbw) == bias_owner(b, nmt, nbw)) --first;
  return b - first;
}

// max over the lanes of a row group (8 or 16 consecutive lanes) of a non-negative int, by DPP butterflies
template <int CTRL>
__device__ __forceinline__ int dpp_max(int v) {
  const int o = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
  return o > v ? o : v;
}
template <int LANES>
__device__ __forceinline__ int group_max(int v) {
  v = dpp_max<0xB1>(v);                            // quad_perm [1,0,3,2]
  v = dpp_max<0x4E>(v);                            // quad_perm [2,3,0,1]
  v = dpp_max<0x141>(v);                           // row_half_mirror: the other quad of the 8
  if constexpr (LANES == 16) v = dpp_max<0x140>(v);   // row_mirror: the other half of the 16
  return v;
}
__device__ __forceinline__ float pow2_field(int field) {   // 2^(field - 127), 0 for field <= 0
  return __builtin_bit_cast(float, (field < 0 ? 0 : field > 254 ? 254 : field) << 23);
}

// GRADIENT RANGE. dQKV is a gradient: its scale is arbitrary (an SSL-only row carries 1e-9, a hinge row 1e-3) and
// every row matters on its own, because dy feeds per-row chains (layer-norm backward, BPTT, embedding rows) and
// Adam is scale-free per element. The two-piece f16 split has an absolute floor of 2^-37 (f16_split.h), so the rows
// are brought into its window by EXACT powers of two:
//   * row r of the dQKV images holds dQKV[r] 2^(127 - k_r + kGUp), k_r = the biased fp32 exponent of the row's max |.|
//     (clamped to kGUp + 1 .. 253): the row's max is in [2^kGUp, 2^(kGUp + 1)) (below 2^(kGUp + 2) at the clamp), the top
//     of the window, so that the split's floor lies 2^-49 below it. dy[r] is descaled by 2^(k_r - 127 - kGUp)
//     at its store — per-row accuracy is that of rows of magnitude one, at any scale.
//   * dW = sum_r y[r]^T dQKV[r] needs one scale for all rows, so row r of the y image holds y[r] 2^(k_r - E + kUp):
//     every term arrives as y dQKV 2^(127 - E + kUp); E follows the largest k the block has met (its accumulators
//     are rescaled when E grows, a block-uniform event) and the flush undoes the factor. A row 2^delta below E keeps
//     its y to max(2^-23, 2^(delta - 43) / |y|): fp32-grade down to a million times below the largest row, degrading
//     from there — where such a row's terms are 2^-delta of the largest row's. db rides on the same product
//     with a "ones" operand that carries the row factor (a power of two: exact in two f16 pieces down to delta = 42).
//   * E for the chunk being committed is known one chunk LATE (row exponents meet in an LDS max that the next
//     barrier publishes), so a row may exceed it: the scaled y row then reaches 32768 (roughly: the row is > 100
//     times larger than anything before it), its chunk is flagged and evaluated with fp32 fmaf chains — as are
//     chunks holding a non-finite gradient, a y segment below 2^-14, or a W out of range. One chunk raises E by at
//     most 2^kJump (the first chunk: at most that much above its 4th largest row), so an isolated absurd entry
//     (2e30 in a field of ones) goes through the fp32 path without wiping out the rows after it.
constexpr int kUp = 6, kJump = 16;
constexpr float kUpInv = 1.f / 64.f;   // 2^-kUp
// A gradient row enters the images with its maximum in [2^kGUp, 2^(kGUp + 1)), not in [1, 2): the split's absolute floor
// (2^-37) then sits 2^-(37 + kGUp) below the row's maximum, so a COLUMN that is small in every row (a saturated gate's
// gradient: 2^-18 of its row's maximum and less) keeps its own relative accuracy in dW — found by tools/fuzz_gpu.py as
// 1.3e-6 of such an element's terms with the rows at [1, 2). 2^13 stays two binades under the window's top (32768).
constexpr int kGUp = 12;
constexpr float kGUpInv = 1.f / 4096.f;   // 2^-kGUp

// Where the rows come from. The attention tail (DY): one segment, x0 = y [rows, D] (overwritten with dy), g = dQKV [rows, 3 D].
// The LSTM weight gradient (NX = 2, NQ = 4, no dy): row r = (s, i) = (step, node), s = r / seg_rows; block 0 of the left
// operand is x_s of node i at x0 + s seg0 + i ld0, block 1 is h_{s-1} at x1 + (s - 1) seg1 + i ld1 (zero at s = 0: the
// initial state), g = the gate gradients [t n, 4 D], rows in r order.
struct TnArgs {
  float* x0;
  int64_t ld0, seg0;
  const float* x1;
  int64_t ld1, seg1;
  const float* g;
  int64_t ldg;
  uint32_t seg_rows;
};

// NX left-operand blocks of D columns, NQ gradient blocks of D columns: dW [NX D, NQ D] += X^T G (+ db, + dy when DY).
template <int D, int NX, int NQ, bool DY>
__global__ __launch_bounds__(D == 64 ? 512 : 256, D == 64 ? 1 : 2) void attn_bwd_tail_f16_kernel(TnArgs src, int64_t rows,
                                                                      const float* __restrict__ W,
                                                                      float* __restrict__ dW, float* __restrict__ db,
                                                                      int64_t n_chunks, unsigned int* __restrict__ redo_ctr) {
  static_assert(!DY || NX == 1, "dy goes over the single left operand");
  constexpr bool DB = DY;                // column sums of G ride on the product (the LSTM's db comes from its BPTT kernel)
  // NX = 2 has 16 dW tiles per wave: head AND residual accumulators for all of them (128 registers) spill. There the two
  // cross products of a tile meet in a transient accumulator that is folded into the tile's head accumulator (one fma per
  // element and chunk) while the next tile's MFMAs run.
  constexpr bool FOLD = NX == 2;
  constexpr int Q3 = NQ * D;             // gradient columns
  constexpr int NS = NX + NQ;            // staged float4 per lane, sub-images per piece
  constexpr int SUB = kRows * D * 2;     // bytes of one [32][D] f16 sub-image
  constexpr int PIECE = NS * SUB;        // y | dQ | dK | dV  (x | h | di | dj | df | do)
  constexpr int BUF = 2 * PIECE;         // heads, scaled residuals
  float* const y = src.x0;
  const float* const dqkv = src.g;
  constexpr int NWV = D / 8;             // waves per block: 8 at D = 64 (two per SIMD), 4 at D = 32 (two blocks per CU)
  constexpr int kBlock = 64 * NWV;
  constexpr int LPRW = D / 4;            // lanes per staged row: lane j holds float4 j of the row's y, dQ, dK and dV
  constexpr int NMT = NX * D / 16, NNT = Q3 / 16;   // dW: NMT x NNT tiles of 16 x 16
  constexpr int MB = 1;                  // dW tile rows per wave ...
  constexpr int NBW = NMT * NNT / NWV;   // ... and tile columns per wave (6 / 3)
  constexpr int KS1 = Q3 / 32;           // k-steps of dy (6 / 3)
  constexpr int NT1 = D / 16;            // column tiles of dy (4 / 2)
  constexpr int M1 = 1;                  // dy tiles per wave: (row tile wave / NT1, column tile wave % NT1)
  static_assert(kBlock == kRows * LPRW && (!DY || 2 * NT1 == NWV) && NWV % NMT == 0 && (NMT * NNT) % NWV == 0 && NNT % (NWV / NMT) == 0 &&
                    (NBW + NMT - 1) / NMT <= 2, "unsupported D / NX / NQ");

  extern __shared__ __attribute__((aligned(16))) char lds[];
  int* const flags = reinterpret_cast<int*>(lds + 2 * BUF);   // [0], [1]: chunk number (+1) in that buffer if it takes the fp32 path; [2]: W does
  int* const emax = flags + 4;                                 // [3]: largest row exponent of a chunk, rotating (see the loop)
  float* const inv = reinterpret_cast<float*>(flags + 8);      // [2][32]: 2^(k_r - 127), the descale of dy's rows
  char* const ones = reinterpret_cast<char*>(inv + 2 * kRows); // [2][2][32] f16: head and scaled residual of 2^(k_r - E + kUp)
  int* const ktab = reinterpret_cast<int*>(ones + 4 * kRows * 2);  // [32]: the first chunk's row exponents
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int sr = tid / LPRW, sc = tid % LPRW;   // staging: my row of the chunk, my float4 of each of its four parts
  float k4096 = 4096.f;
  asm volatile("" : "+v"(k4096));
  if (tid < 8) flags[tid] = 0;
  __syncthreads();

  // ---- dy: this wave's column tile of W^T as B fragments, resident: B[k = o][j] = Wqkv[16 nt1 + j][o]
  const int nt1 = wave % NT1;
  const int mt1_0 = wave / NT1;
  i32x4 wf[KS1][2];
  if constexpr (DY) {
    RangeTrack wr = range_init();
    const float* wrow = W + (size_t)(16 * nt1 + m) * Q3 + 8 * kq;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      const float4 a = *reinterpret_cast<const float4*>(wrow + 32 * ks), b = *reinterpret_cast<const float4*>(wrow + 32 * ks + 4);
      const float wv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int hd = head2(wv[2 * e], wv[2 * e + 1]);
        wf[ks][0][e] = hd;
        wf[ks][1][e] = tail2(hd, wv[2 * e], wv[2 * e + 1], k4096);
      }
      range_seg4(wr, a.x, a.y, a.z, a.w);
      range_seg4(wr, b.x, b.y, b.z, b.w);
    }
    if (range_bad(wr)) flags[2] = 1;   // read after the barriers below
  }
  // ---- dW: my tiles (wm * MB + a, wn * NBW + b), head and residual accumulators
  const int wm = wave % NMT, wn = wave / NMT;
  f32x4 hiw[MB][NBW], low[MB][NBW];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NBW; ++b) hiw[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}, low[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 hib[2], lob[2];                                  // db tiles of the columns I own (at most two)
  hib[0] = hib[1] = lob[0] = lob[1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // element (row r, column c) of left block b, or NULL where the block is zero (h_{-1}); r < 2^31 (host)
  auto xptr = [&](int b, uint32_t r) -> const float* {
    const uint32_t sg = r / src.seg_rows, i = r - sg * src.seg_rows;
    if (b == 0) return src.x0 + sg * src.seg0 + i * src.ld0;
    return sg == 0 ? nullptr : src.x1 + (sg - 1) * src.seg1 + i * src.ld1;
  };
  float4 stage[NS];                                      // y | dQ | dK | dV  (x | h | di | dj | df | do): my float4 of my row
  auto fetch = [&](int64_t ch) {
    const int64_t row0 = ch * kRows;
    const int last = (int)(rows - 1 - row0 < kRows - 1 ? rows - 1 - row0 : kRows - 1);  // uniform
    const uint32_t rc = sr < last ? sr : last;           // rows past the end read the last row, zeroed below
    const bool ok = sr <= last;
    const float* const gp = dqkv + (row0 + rc) * src.ldg + 4 * sc;
#pragma unroll
    for (int v = 0; v < NS; ++v) {
      const float* const xp = v < NX ? xptr(v, (uint32_t)(row0 + rc)) : nullptr;
      const bool have = ok && (v >= NX || xp != nullptr);
      const float4 val = *reinterpret_cast<const float4*>(v < NX ? (xp ? xp + 4 * sc : src.x0) : gp + (v - NX) * D);
      stage[v] = make_float4(have ? val.x : 0.f, have ? val.y : 0.f, have ? val.z : 0.f, have ? val.w : 0.f);
    }
  };
  // biased exponent of the largest |gradient| of my staged row (all lanes of the row agree)
  auto row_exponent = [&]() -> int {
    float mx = 0.f;
#pragma unroll
    for (int v = NX; v < NS; ++v) mx = max3abs(max3abs(mx, stage[v].x, stage[v].y), stage[v].z, stage[v].w);
    return group_max<LPRW>(__builtin_bit_cast(int, mx)) >> 23;
  };
  // What the running scale E follows. The attention tail's left operand is a layer-normed row (O(1)): the gradient row's
  // exponent alone. The LSTM's x | h rows have no such scale (hub rows of a propagated embedding reach 1e4): E follows the
  // exponent of the PRODUCT of the two rows' maxima, so that the left row, scaled by 2^(k - E + kUp), stays below 2^(kUp + 1)
  // whenever its row is no heavier than E — the factor every term carries, 2^(127 - E + kUp), does not depend on which
  // exponent E follows.
  auto row_weight = [&](int e) -> int {
    if constexpr (DY) return e;
    float mx = 0.f;
#pragma unroll
    for (int v = 0; v < NX; ++v) mx = max3abs(max3abs(mx, stage[v].x, stage[v].y), stage[v].z, stage[v].w);
    const int kx = group_max<LPRW>(__builtin_bit_cast(int, mx)) >> 23;
    return kx == 0 ? 0 : e + kx - 127;                  // an all-zero left row weighs nothing
  };
  // split the staged row into buffer `b` at the scales described above; `id` = chunk number + 1 marks the buffer for the fp32 path
  auto commit = [&](int b, int id, int e, int E) {
    char* const buf = lds + b * BUF;
    const int k = e > 253 ? 253 : e < kGUp + 1 ? kGUp + 1 : e;
    const float s = pow2_field(254 - k + kGUp);        // dQKV row scale: 2^(127 - k + kGUp)
    const float f = pow2_field(127 + k - E + kUp);     // y row scale
    char* const dst = buf + img_off<D>(sr, sc >> 1) + (sc & 1) * 8;
    bool bad = e == 255;
#pragma unroll
    for (int v = 0; v < NX; ++v) {
      const float4 yr = stage[v];
      const float4 yv = make_float4(yr.x * f, yr.y * f, yr.z * f, yr.w * f);
      const int p0 = head2(yv.x, yv.y), p1 = head2(yv.z, yv.w);
      *reinterpret_cast<i32x2*>(dst + v * SUB) = i32x2{p0, p1};
      *reinterpret_cast<i32x2*>(dst + v * SUB + PIECE) = i32x2{tail2(p0, yv.x, yv.y, k4096), tail2(p1, yv.z, yv.w, k4096)};
      const float ys = maxabs_acc(maxabs3(yr.x, yr.y, yr.z), yr.w);       // the raw segment: small as a whole?
      const float yt = maxabs_acc(maxabs3(yv.x, yv.y, yv.z), yv.w);       // the scaled one: does it fit?
      // DY: y's scale is not part of what E follows, so a y segment that is small as a whole would lose its terms to the split's
      // floor. LSTM form: E follows the weight of the PRODUCT, a small x | h segment (a saturated gate leaves h = 1e-14 next
      // to 0.9) is small against the heaviest row's terms, which is all the floor is measured against.
      bad = bad || yt >= kF16Lim || (DY && ys > 0.f && ys < 6.103515625e-05f);
    }
#pragma unroll
    for (int v = NX; v < NS; ++v) {
      int h0, t0, h1, t1;
      split2_scaled(stage[v].x, stage[v].y, s, k4096, h0, t0);
      split2_scaled(stage[v].z, stage[v].w, s, k4096, h1, t1);
      *reinterpret_cast<i32x2*>(dst + v * SUB) = i32x2{h0, h1};
      *reinterpret_cast<i32x2*>(dst + v * SUB + PIECE) = i32x2{t0, t1};
    }
    if (sc == 0) {
      inv[b * kRows + sr] = pow2_field(k - kGUp);      // 2^(k - 127 - kGUp): dy's descale
      const int oh = head2(f, 0.f);
      const int ot = tail2(oh, f, 0.f, k4096);
      *reinterpret_cast<short*>(ones + (b * 2 + 0) * (kRows * 2) + sr * 2) = (short)oh;
      *reinterpret_cast<short*>(ones + (b * 2 + 1) * (kRows * 2) + sr * 2) = (short)ot;
    }
    if (bad) flags[b] = id;
  };

  int64_t ch = blockIdx.x;
  int cur = 0, slot = 0;   // buffer of chunk ch; emax slot its rows posted to
  int e_run = 1;           // E the y rows of chunk ch were scaled against
  int e_acc = 0;           // E the accumulators are at (0: nothing accumulated yet)
  if (ch < n_chunks) {
    fetch(ch);
    const int e = row_exponent();
    const int ew = row_weight(e);
    if (sc == 0) ktab[sr] = ew > 253 ? 253 : ew < 1 ? 1 : ew;
    __syncthreads();
    // the first chunk is scaled against its own largest row, but no more than 2^kJump above its 4th largest
    int t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
      int v = ktab[j], a;
      a = t0 > v ? t0 : v, v = t0 > v ? v : t0, t0 = a;
      a = t1 > v ? t1 : v, v = t1 > v ? v : t1, t1 = a;
      a = t2 > v ? t2 : v, v = t2 > v ? v : t2, t2 = a;
      t3 = t3 > v ? t3 : v;
    }
    // (the LSTM form scales every chunk against a maximum that includes its OWN rows — see the loop — and needs no cap)
    // (a ragged last chunk of fewer than four rows that is a block's FIRST chunk has padding rows standing in for its 4th
    // largest: its rows then take the fp32 path — exact, and the only safe choice with nothing to measure an absurd row against)
    e_run = (!DY || t0 < t3 + kJump) ? t0 : t3 + kJump;
    commit(0, (int)(ch + 1), e, e_run);
  }
  __syncthreads();
  for (; ch < n_chunks; ch += gridDim.x) {
    const int64_t nxt = ch + gridDim.x;
    const bool slow = flags[cur] == (int)(ch + 1) || flags[2] != 0;   // block-uniform
    if (nxt < n_chunks) fetch(nxt);  // lands under the MFMAs below
    const char* const buf = lds + cur * BUF;
    const int64_t row0 = ch * kRows;
    const int rows_valid = (int)(rows - row0 < kRows ? rows - row0 : kRows);
    int lane_ = lane, m_ = m, kq_ = kq;
    asm volatile("" : "+v"(lane_), "+v"(m_), "+v"(kq_));

    if (!slow) {
      if (e_acc != e_run) {   // E grew (block-uniform, a handful of times per block): the accumulators follow
        if (e_acc != 0) {
          const float g = pow2_field(127 + e_acc - e_run);
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
              hiw[a][b] *= g;
              if constexpr (!FOLD) low[a][b] *= g;
            }
          hib[0] *= g, hib[1] *= g, lob[0] *= g, lob[1] *= g;
        }
        e_acc = e_run;
      }
      // ---- dW tiles += y^T dQKV (K = the chunk's 32 rows): both operands read transposed
      {
        i32x4 a1[MB], a2[MB];
#pragma unroll
        for (int a = 0; a < MB; ++a) {
          const int mt = wm * MB + a;                       // tile row of dW = 16 columns of left block 16 mt / D
          a1[a] = read_tr<D>(buf + (16 * mt / D) * SUB, (16 * mt % D) / 16, lane_);
          a2[a] = read_tr<D>(buf + (16 * mt / D) * SUB + PIECE, (16 * mt % D) / 16, lane_);
        }
        // the rows' factors as an A operand whose 16 M rows are all equal: A[.][k = 8 kq + j] = factor of row 8 kq + j
        f32x4 lt_prev = {0.f, 0.f, 0.f, 0.f};
        const i32x4 o1 = *reinterpret_cast<const i32x4*>(ones + (cur * 2 + 0) * (kRows * 2) + 16 * kq_);
        const i32x4 o2 = *reinterpret_cast<const i32x4*>(ones + (cur * 2 + 1) * (kRows * 2) + 16 * kq_);
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
          const int col = 16 * (wn * NBW + b);              // dQKV column of the tile
          const char* const sub = buf + (NX + col / D) * SUB;
          const i32x4 b1 = read_tr<D>(sub, (col % D) / 16, lane_);
          const i32x4 b2 = read_tr<D>(sub + PIECE, (col % D) / 16, lane_);
#pragma unroll
          for (int a = 0; a < MB; ++a) {
            if constexpr (FOLD) {
              static_assert(MB == 1, "one transient accumulator in flight");
              f32x4 lt = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2[a]), __builtin_bit_cast(f16x8, b1), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
              lt = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1[a]), __builtin_bit_cast(f16x8, b2), lt, 0, 0, 0);
              hiw[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1[a]), __builtin_bit_cast(f16x8, b1), hiw[a][b], 0, 0, 0);
              if (b > 0) hiw[a][b - 1] += lt_prev * kLoInv;       // the previous tile's cross terms: its MFMAs have drained by now
              lt_prev = lt;
            } else {
              low[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2[a]), __builtin_bit_cast(f16x8, b1), low[a][b], 0, 0, 0);
              low[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1[a]), __builtin_bit_cast(f16x8, b2), low[a][b], 0, 0, 0);
              hiw[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1[a]), __builtin_bit_cast(f16x8, b1), hiw[a][b], 0, 0, 0);
            }
          }
          if (DB && bias_owner(b, NMT, NBW) == wm) {   // wave-uniform
            const int sl = bias_slot(b, NMT, NBW);   // a constant once the loop is unrolled
            lob[sl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, o2), __builtin_bit_cast(f16x8, b1), lob[sl], 0, 0, 0);
            lob[sl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, o1), __builtin_bit_cast(f16x8, b2), lob[sl], 0, 0, 0);
            hib[sl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, o1), __builtin_bit_cast(f16x8, b1), hib[sl], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);   // one tile column's operands in flight at a time
        }
        if constexpr (FOLD) hiw[0][NBW - 1] += lt_prev * kLoInv;
      }
      // ---- dy[:, 16 nt1 .. + 15] = dQKV @ W^T for my row tiles: dQKV read row-wise
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(y + row0 * D, 0, DY ? rows_valid * D * 4 : 0, 0x00020000);
#pragma unroll
      for (int t1 = 0; t1 < (DY ? M1 : 0); ++t1) {
        const int mt = mt1_0 + t1;
        const int row = 16 * mt + m_;
        f32x4 hi = {0.f, 0.f, 0.f, 0.f}, lo = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          const int col = 32 * ks;                          // + 8 kq: never crosses a sub-image (D % 32 == 0)
          const char* const src = buf + (1 + col / D) * SUB + img_off<D>(row, (col % D) / 8 + kq_);
          const f16x8 g1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(src));
          const f16x8 g2 = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4*>(src + PIECE));
          lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(g2, __builtin_bit_cast(f16x8, wf[ks][0]), lo, 0, 0, 0);
          lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(g1, __builtin_bit_cast(f16x8, wf[ks][1]), lo, 0, 0, 0);
          hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(g1, __builtin_bit_cast(f16x8, wf[ks][0]), hi, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        // C layout: lane (n = lane & 15, mq = lane >> 4) holds rows 16 mt + 4 mq + r of column 16 nt1 + n; each row leaves at its own scale
        const f32x4 iv = *reinterpret_cast<const f32x4*>(inv + cur * kRows + 16 * mt + 4 * kq_);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, fmaf(lo[r], kLoInv, hi[r]) * iv[r]), rs,
                                                ((16 * mt + 4 * kq_ + r) * D + 16 * nt1 + m_) * 4, 0, 0);
      }
    } else {
      // ---- this chunk (or W) is outside the window: fp32 fmaf chains. dW first (it reads y), then dy.
      if (tid == 0 && redo_ctr) atomicAdd(redo_ctr, 1u);
      for (int idx = tid; idx < NX * D * Q3; idx += kBlock) {
        const int i = idx / Q3, o = idx - i * Q3;
        float acc = 0.f;
        for (int r = 0; r < rows_valid; ++r) {
          const float* const xp = xptr(i / D, (uint32_t)(row0 + r));
          if (xp) acc = fmaf(xp[i % D], dqkv[(row0 + r) * src.ldg + o], acc);
        }
        atomicAdd(dW + idx, acc);
      }
      if constexpr (DB) {
        for (int o = tid; o < Q3; o += kBlock) {
          float acc = 0.f;
          for (int r = 0; r < rows_valid; ++r) acc += dqkv[(row0 + r) * src.ldg + o];
          atomicAdd(db + o, acc);
        }
      }
      if constexpr (DY) {
        __syncthreads();
        for (int idx = tid; idx < rows_valid * D; idx += kBlock) {
          const int r = idx / D, i = idx - r * D;
          float acc = 0.f;
          for (int o = 0; o < Q3; ++o) acc = fmaf(dqkv[(row0 + r) * Q3 + o], W[(size_t)i * Q3 + o], acc);
          y[(row0 + r) * D + i] = acc;
        }
      }
    }
    int e_next = e_run;
    const int nslot = slot == 2 ? 0 : slot + 1;
    if (nxt < n_chunks) {
      const int ec = emax[slot];                        // chunk ch's own rows: complete since the barrier that closed its commit
      e_next = ec > e_run ? ec : e_run;
      const int e = row_exponent();
      const int ew = row_weight(e);
      if constexpr (DY) {
        if (sc == 0) {                                    // what chunk nxt's rows ask of the chunk after it
          const int k = ew > 253 ? 253 : ew < 1 ? 1 : ew;
          atomicMax(emax + nslot, k < e_next + kJump ? k : e_next + kJump);
        }
      } else {
        // LSTM form: row weights are heavy-tailed (a hub node's row is 2^25 heavier than its neighbours'), and a chunk scaled
        // against a maximum that lags by one chunk would overflow and take the fp32 path at every such row — once per block
        // and launch at least. One more barrier per chunk and the chunk's own rows are in the maximum it is scaled against:
        // nothing overflows, so nothing caps the step either (an absurd finite row makes the rows after it negligible in
        // the sum it dominates, which is what they are).
        if (sc == 0) atomicMax(emax + nslot, ew > 253 ? 253 : ew < 1 ? 1 : ew);
        __syncthreads();
        const int own = emax[nslot];
        e_next = own > e_next ? own : e_next;
      }
      commit(cur ^ 1, (int)(nxt + 1), e, e_next);
    }
    if (tid == 0) emax[nslot == 2 ? 0 : nslot + 1] = 0;   // last read one iteration ago, next posted to one iteration ahead
    __syncthreads();  // next buffer complete and marked; this one free next time round
    cur ^= 1;
    slot = nslot;
    e_run = e_next;
  }

  // ---- flush: the accumulators hold the sums times 2^(127 - e_acc + kUp) --------------------------------------
  if (e_acc == 0) return;   // nothing went through the matrix cores
  const float u1 = pow2_field(e_acc), u2 = kUpInv * kGUpInv;
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
      const int mt = wm * MB + a, nt = wn * NBW + b;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        atomicAdd(dW + (size_t)(16 * mt + 4 * kq + r) * Q3 + 16 * nt + m,
                  (FOLD ? hiw[a][b][r] : fmaf(low[a][b][r], kLoInv, hiw[a][b][r])) * u1 * u2);
    }
#pragma unroll
  for (int b = 0; b < NBW; ++b)
    if (DB && bias_owner(b, NMT, NBW) == wm && kq == 0) {   // every C row of a ones tile holds the column sums: row 0 speaks
      const int sl = bias_slot(b, NMT, NBW);
      const float v0 = fmaf(sl == 0 ? lob[0][0] : lob[1][0], kLoInv, sl == 0 ? hib[0][0] : hib[1][0]);
      atomicAdd(db + 16 * (wn * NBW + b) + m, v0 * u1 * u2);
    }
}

template <int D, int NX, int NQ, bool DY>
int launch(const TnArgs& src, int64_t rows, const float* W, float* dW, float* db, hipStream_t s) {
  if (rows > INT32_MAX) return sagnn::fail(SAGNN_ERR_ARG, "f16 x 2 weight-gradient kernel: %lld rows in one call (limit 2^31 - 1)", (long long)rows);
  // images: 64 KB at D = 64 for the attention tail (y | dQ | dK | dV), 96 KB for the LSTM (x | h | 4 gates); + flags, row scales
  const size_t lds = (size_t)2 * 2 * (NX + NQ) * kRows * D * 2 + 32 + 2 * kRows * 4 + 4 * kRows * 2 + kRows * 4;
  if (int rc = sagnn::ensure_dynamic_lds(reinterpret_cast<const void*>(&attn_bwd_tail_f16_kernel<D, NX, NQ, DY>), lds)) return rc;
  const int64_t n_chunks = (rows + kRows - 1) / kRows;
  const int64_t want = (D == 64 ? 1 : 2) * (int64_t)sagnn::cu_count_current();
  const int64_t blocks = n_chunks < want ? n_chunks : want;
  hipLaunchKernelGGL((attn_bwd_tail_f16_kernel<D, NX, NQ, DY>), dim3((unsigned)blocks), dim3(D == 64 ? 512 : 256), lds, s, src, rows, W,
                     dW, db, n_chunks, sagnn::redo_counter());
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

}  // namespace

namespace sagnn {

int attn_bwd_tail_f16(float* y, const float* dqkv, int64_t rows, int d, const float* Wqkv, float* dWqkv, float* dbqkv,
                      hipStream_t s) {
  const TnArgs src = {y, (int64_t)d, 0, nullptr, 0, 0, dqkv, (int64_t)3 * d, (uint32_t)(rows > 0 ? rows : 1)};   // one segment
  if (d == 64) return launch<64, 1, 3, true>(src, rows, Wqkv, dWqkv, dbqkv, s);
  if (d == 32) return launch<32, 1, 3, true>(src, rows, Wqkv, dWqkv, dbqkv, s);
  return fail(SAGNN_ERR_DIM, "f16 attn_bwd_tail: d must be 32 or 64 (got %d)", d);
}

// dW [2d, 4d] += sum over steps of [x_s | h_{s-1}]^T dG_s: the LSTM's weight gradient from the gate gradients the BPTT
// kernel stored time-major (dg [t, n, 4d]), x [n, t, d] in its node / interval strides, h [n, t, d] dense (un-dropped),
// h_{-1} = 0. Same engine, same GRADIENT RANGE scheme as the attention tail: gate-gradient rows enter at power-of-two
// scales, chunks outside the window take the fp32 path.
int lstm_dw_f16(const float* x, int64_t ld_n, int64_t ld_t, const float* h, const float* dg, int64_t n, int t, int d, float* dW,
                hipStream_t s) {
  if (n <= 0 || t <= 0) return SAGNN_OK;
  const TnArgs src = {const_cast<float*>(x), ld_n, ld_t, h, (int64_t)t * d, (int64_t)d, dg, (int64_t)4 * d, (uint32_t)n};
  if (d == 64) return launch<64, 2, 4, false>(src, n * t, nullptr, dW, nullptr, s);
  if (d == 32) return launch<32, 2, 4, false>(src, n * t, nullptr, dW, nullptr, s);
  return fail(SAGNN_ERR_DIM, "f16 lstm_dw: d must be 32 or 64 (got %d)", d);
}

}  // namespace sagnn
