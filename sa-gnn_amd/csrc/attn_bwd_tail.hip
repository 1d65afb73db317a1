// Tail of the attention backward pass (gradient of reference Utils/attention.py:31-45's three dense
// layers) in one pass over dQ|dK|dV:
//
//   dy [R, D]    = dQKV [R, 3D] @ Wqkv^T [3D, D]      (written over y: y is dead after this kernel)
//   dWqkv [D,3D] += y^T [D, R] @ dQKV [R, 3D]
//   dbqkv [3D]   += column sums of dQKV
//
// The un-fused form reads dQKV (768 B per row at D = 64) once per product. Here a block (4 waves)
// walks 32-row chunks: the chunk's y | dQKV rows sit in a double-buffered, slot-swizzled LDS tile
// (next chunk in flight in registers), two blocks per CU (80 KB of LDS each; one block's barrier and
// load waits are the other's compute: 18.4 -> 14.7 ms), every wave runs 48 MFMAs of each product
// per chunk:
//   product 2: wave w owns dW tiles (ta, tb) — accumulators live across all chunks of the block;
//   product 1: wave w owns output column tile w % (D/32) and a slice of the K = 3D reduction;
//              the partial tiles of the other slices reach the tile's owner through LDS.
// Operands are read a group of k-steps ahead (an MFMA does not cover an LDS round trip) and the
// MFMA loops carry no branches and no 64-bit address arithmetic (fp32 VALU work is additive to
// fp32 MFMA work on this chip).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBlock = 256;
constexpr int kRows = 32;

__device__ __forceinline__ int crow(int r, int rh) { return (r & 3) + 8 * (r >> 2) + 4 * rh; }
__device__ __forceinline__ int kcol(int m, int kh) { return 8 * (m >> 2) + 4 * kh + (m & 3); }

template <int D>
__global__ __launch_bounds__(kBlock, 2) void attn_bwd_tail_kernel(float* __restrict__ y, const float* __restrict__ dqkv,
                                                                  int64_t rows, const float* __restrict__ W,
                                                                  float* __restrict__ dW, float* __restrict__ db,
                                                                  int64_t n_chunks) {
  constexpr int Q3 = 3 * D;          // dQKV columns
  constexpr int WD = 4 * D;          // tile row: y | dQKV
  constexpr int S4 = WD / 4;         // float4 slots per tile row
  constexpr int NCT = D / 32;        // column tiles of dy (= tile rows of dW)
  constexpr int NTB = Q3 / 32;       // tile columns of dW
  constexpr int KSPLIT = 4 / NCT;    // waves sharing one dy tile (K slices)
  constexpr int KS1 = Q3 / 2 / KSPLIT;  // k-steps of product 1 per wave (48 at D = 64, 12 at D = 32)
  constexpr int NQ1 = KS1 / 4;       // b128 operand reads of product 1 per wave
  constexpr int TPW = (NCT * NTB + 3) / 4;  // dW tiles per wave (3 at D = 64, 1 at D = 32)
  constexpr int NV = kRows * S4 / kBlock;   // float4 per thread and chunk (8 at D = 64, 4 at D = 32)
  static_assert(KS1 % 4 == 0 && NV * kBlock == kRows * S4, "unsupported D");

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tiles = lds;                                            // [2][32][WD], slot-swizzled
  float* part = lds + 2 * kRows * WD;   // [2][4 - NCT][32*32] partial dy tiles
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int ct = wave % NCT;       // my dy column tile
  const int kp = wave / NCT;       // my K slice of product 1 (0 = owner of the tile)

  // ---- W^T fragments of my (column tile, K slice), in registers: fragment q of lane (li, kh) =
  // Wqkv[32ct + li][8gq + 4kh .. + 3] with gq = kp*NQ1 + q: the B operands of k-steps 4gq .. 4gq+3 for
  // output column 32ct + li (k order = kcol, matching the b128 row reads of the tile). 80 KB of LDS
  // per block then leave room for two blocks per CU: one block's barrier / load waits are the
  // other's compute.
  float4 wreg[NQ1];
#pragma unroll
  for (int q = 0; q < NQ1; ++q)
    wreg[q] = *reinterpret_cast<const float4*>(W + (size_t)(32 * ct + li) * Q3 + 8 * (kp * NQ1 + q) + 4 * kh);

  // my dW tiles: tt = wave*TPW + j -> (ta, tb); past the last tile recompute the last one, dropped at the flush
  int aoff[TPW], boff[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    int tt = wave * TPW + j;
    if (tt >= NCT * NTB) tt = NCT * NTB - 1;
    aoff[j] = (tt / NTB) * 32 + li;            // y column
    boff[j] = D + (tt % NTB) * 32 + li;        // dQKV column (tile coordinates)
  }
  // byte offset of my operand columns within a tile row, with the lane part of the slot swizzle
  // folded in: (((col >> 2) ^ kh) << 4) + (col & 3) * 4   (see product 2)
  uint32_t aswz[TPW], bswz[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    aswz[j] = ((((uint32_t)aoff[j] >> 2) ^ (uint32_t)kh) << 4) + ((uint32_t)aoff[j] & 3) * 4;
    bswz[j] = ((((uint32_t)boff[j] >> 2) ^ (uint32_t)kh) << 4) + ((uint32_t)boff[j] & 3) * 4;
  }
  f32x16 accw[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[j][r] = 0.f;
  float colsum = 0.f;

  auto elem = [](const float* tile, int row, int col) -> const float* {
    return tile + row * WD + ((((col >> 2) ^ (row & 15))) << 2) + (col & 3);
  };

  float4 stage[NV];
  auto fetch = [&](int64_t ch) {
    const int64_t row0 = ch * kRows;
    const int last = (int)(rows - 1 - row0 < kRows - 1 ? rows - 1 - row0 : kRows - 1);  // uniform
    const float* ybase = y + row0 * D;
    const float* gbase = dqkv + row0 * Q3;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int i = tid + v * kBlock;
      const int r = i / S4, s = i - r * S4;
      const uint32_t rc = r < last ? r : last;          // rows past the end read the last row, zeroed below
      const float4 val = s < D / 4 ? *reinterpret_cast<const float4*>(ybase + (rc * D + 4 * s))
                                   : *reinterpret_cast<const float4*>(gbase + (rc * Q3 + 4 * (s - D / 4)));
      const bool ok = r <= last;
      stage[v] = make_float4(ok ? val.x : 0.f, ok ? val.y : 0.f, ok ? val.z : 0.f, ok ? val.w : 0.f);
    }
  };
  auto commit = [&](float* tile) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int i = tid + v * kBlock;
      const int r = i / S4, s = i - r * S4;
      reinterpret_cast<float4*>(tile + r * WD)[s ^ (r & 15)] = stage[v];
    }
  };

  int64_t ch = blockIdx.x;
  int cur = 0;
  if (ch < n_chunks) {
    fetch(ch);
    commit(tiles);
  }
  __syncthreads();
  for (; ch < n_chunks; ch += gridDim.x) {
    const int64_t nxt = ch + gridDim.x;
    if (nxt < n_chunks) fetch(nxt);  // lands under the MFMAs below
    const float* tile = tiles + cur * kRows * WD;
    int li_ = li, kh_ = kh;
    asm volatile("" : "+v"(li_), "+v"(kh_));

    // ---- bias gradient: column sums of the dQKV part, one column per thread -----------------------
    if (tid < Q3) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < kRows; ++r) s += *elem(tile, r, D + tid);
      colsum += s;
    }

    // ---- product 2: dW tiles += y^T dQKV (K = the chunk's 32 rows) ---------------------------------
    // Element (row, col) of the tile sits at byte row*WD*4 + (((col>>2) ^ (row&15)) << 4) + (col&3)*4.
    // With row = 2kk + kh: row & 15 = (2kk & 15) ^ kh, so the lane part ((col>>2) ^ kh) is fixed per
    // tile and each read costs one XOR with a compile-time constant; 2kk*WD*4 is an immediate offset.
    {
      constexpr int GK = 2;
      constexpr int NG = 16 / GK;
      const char* tb = reinterpret_cast<const char*>(tile) + kh_ * (WD * 4);
      float opa[2][GK][TPW], opb[2][GK][TPW];
      auto read_group = [&](float (&a)[GK][TPW], float (&b)[GK][TPW], int g) {
#pragma unroll
        for (int u = 0; u < GK; ++u) {
          const int kk = g * GK + u;
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            a[u][j] = *reinterpret_cast<const float*>(tb + 2 * kk * (WD * 4) + (aswz[j] ^ (((2 * kk) & 15) << 4)));
            b[u][j] = *reinterpret_cast<const float*>(tb + 2 * kk * (WD * 4) + (bswz[j] ^ (((2 * kk) & 15) << 4)));
          }
        }
      };
      read_group(opa[0], opb[0], 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {   // fully unrolled: the two operand sets alternate, no copies
        if (g + 1 < NG) read_group(opa[(g + 1) & 1], opb[(g + 1) & 1], g + 1);
#pragma unroll
        for (int u = 0; u < GK; ++u)
#pragma unroll
          for (int j = 0; j < TPW; ++j)
            accw[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(opa[g & 1][u][j], opb[g & 1][u][j], accw[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- product 1: my K slice of dy[:, 32ct .. 32ct+31] = dQKV @ W^T ---------------------------------
    f32x16 c0, c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) c0[r] = 0.f, c1[r] = 0.f;
    {
      // slot (D/4 + 2gq + kh) ^ (row & 15) = (D/4 + 2gq) ^ (kh ^ (row & 15)): the first term is even
      const float4* arow = reinterpret_cast<const float4*>(tile + li_ * WD);
      const int lsw = kh_ ^ (li_ & 15);
      const int q0 = D / 4 + 2 * kp * NQ1;   // wave-uniform
      float4 a = arow[q0 ^ lsw], an;
#pragma unroll
      for (int q = 0; q < NQ1; ++q) {
        if (q + 1 < NQ1) an = arow[(q0 + 2 * (q + 1)) ^ lsw];   // requested before these four MFMAs
        const float4 w = wreg[q];
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, c1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a = an;
      }
    }
    float* pbuf = part + cur * (4 - NCT) * 1024;
    if (kp > 0) {  // hand my partial tile to the owner (lane-major: conflict-free, no layout needed)
      float* dst = pbuf + (wave - NCT) * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64] = c0[r] + c1[r];
    }
    if (nxt < n_chunks) commit(tiles + (cur ^ 1) * kRows * WD);
    __syncthreads();  // partial tiles visible; next buffer complete; this one free next time round
    if (kp == 0) {
      const int64_t row0 = ch * kRows;
      const int rows_valid = (int)(rows - row0 < kRows ? rows - row0 : kRows);
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(y + row0 * D, 0, rows_valid * D * 4, 0x00020000);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = c0[r] + c1[r];
#pragma unroll
        for (int k = 1; k < KSPLIT; ++k) v += pbuf[((k * NCT + ct) - NCT) * 1024 + r * 64 + lane];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs, (crow(r, kh_) * D + 32 * ct + li_) * 4, 0, 0);
      }
    }
    cur ^= 1;
  }

  // ---- flush ------------------------------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int tt = wave * TPW + j;
    if (tt < NCT * NTB) {
      const int ta = tt / NTB, tb = tt % NTB;
#pragma unroll
      for (int r = 0; r < 16; ++r) atomicAdd(dW + (size_t)(32 * ta + crow(r, kh)) * Q3 + 32 * tb + li, accw[j][r]);
    }
  }
  if (tid < Q3) atomicAdd(db + tid, colsum);
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

template <int D>
int launch(float* y, const float* dqkv, int64_t rows, const float* W, float* dW, float* db, hipStream_t s) {
  constexpr int WD = 4 * D, NCT = D / 32;
  const size_t lds = ((size_t)2 * kRows * WD + (size_t)2 * (4 - NCT) * 1024) * sizeof(float);   // 80 KB at D = 64
  if (int rc = sagnn::ensure_dynamic_lds(reinterpret_cast<const void*>(&attn_bwd_tail_kernel<D>), lds)) return rc;
  const int64_t n_chunks = (rows + kRows - 1) / kRows;
  const int64_t want = 2 * (int64_t)cu_count();
  const int64_t blocks = n_chunks < want ? n_chunks : want;
  hipLaunchKernelGGL(attn_bwd_tail_kernel<D>, dim3((unsigned)blocks), dim3(kBlock), lds, s, y, dqkv, rows, W, dW, db,
                     n_chunks);
  SAGNN_HIP_TRY(hipGetLastError());
  return SAGNN_OK;
}

}  // namespace

extern "C" int sagnn_attn_bwd_tail_supported(int d) { return (d == 32 || d == 64) && !sagnn::force_valu(); }

extern "C" int sagnn_attn_bwd_tail_f32(float* y, const float* dqkv, int64_t rows, int d, const float* Wqkv, float* dWqkv,
                                       float* dbqkv, void* stream) {
  if (rows < 0) return sagnn::fail(SAGNN_ERR_DIM, "bad row count");
  if (d != 32 && d != 64) return sagnn::fail(SAGNN_ERR_DIM, "attn_bwd_tail: d must be 32 or 64 (got %d)", d);
  if (!y || !dqkv || !Wqkv || !dWqkv || !dbqkv) return sagnn::fail(SAGNN_ERR_NULL, "null tensor pointer");
  if (!sagnn::aligned16(y) || !sagnn::aligned16(dqkv) || !sagnn::aligned16(Wqkv))
    return sagnn::fail(SAGNN_ERR_ALIGN, "attn_bwd_tail: need 16-byte aligned buffers");
  if (rows == 0) return SAGNN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!sagnn::force_f32_mfma()) return sagnn::attn_bwd_tail_f16(y, dqkv, rows, d, Wqkv, dWqkv, dbqkv, s);
  if (d == 64) return launch<64>(y, dqkv, rows, Wqkv, dWqkv, dbqkv, s);
  return launch<32>(y, dqkv, rows, Wqkv, dWqkv, dbqkv, s);
}
