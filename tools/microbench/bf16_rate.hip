// Diagnostic: SUSTAINED rate (several hundred ms, so the power limit has settled) of the two bf16 MFMA
// shapes with register operands holding random data: v_mfma_f32_16x16x32_bf16 (what the split
// kernels use) against v_mfma_f32_32x32x16_bf16 (twice the flops per operand register read).
// Question: is the power-bound split LSTM leaving clock on the table by its choice of tile shape?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned rnd(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return s;
}
// random bf16 pairs with exponents near 1.0 (sign and mantissa bits toggling)
__device__ __forceinline__ int rnd_pair(unsigned& s) {
  const unsigned r = rnd(s);
  return (int)((r & 0x80FF80FFu) | 0x3F003F00u);
}

template <int SHAPE, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD, 1) void k(float* out, unsigned long long* cyc, int iters) {
  unsigned s = threadIdx.x * 7919u + blockIdx.x * 104729u + 1u;
  i32x4 a[6], b[6];
  for (int i = 0; i < 6; ++i)
    for (int e = 0; e < 4; ++e) a[i][e] = rnd_pair(s), b[i][e] = rnd_pair(s);
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  float sum = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[u]),
                                                           __builtin_bit_cast(bf16x8, b[(u + j) % 6]), acc[j], 0, 0, 0);
    }
    for (int j = 0; j < 8; ++j) sum += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  } else {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[u]),
                                                           __builtin_bit_cast(bf16x8, b[(u + j) % 6]), acc[j], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) sum += acc[j][r];
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int WPS>
void run(const char* name, float* out, unsigned long long* cyc, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, WPS>), dim3(256), dim3(256 * WPS), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long c;
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  // per wave and iteration: SHAPE 16: 48 MFMAs x 16*16*32*2 flop; SHAPE 32: 24 MFMAs x 32*32*16*2 flop
  const double flop_wave_iter = SHAPE == 16 ? 48.0 * 16384 : 24.0 * 32768;
  const double flops = flop_wave_iter * iters * 256.0 * 4 * WPS;
  printf("%-52s %8.2f ms  %7.1f TFLOP/s  (%.2f s_memtime ticks per MFMA per wave)\n", name, ms, flops / (ms * 1e-3) / 1e12,
         (double)c / ((SHAPE == 16 ? 48.0 : 24.0) * iters));
}

int main() {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&cyc, 256 * 8);
  const int iters = 400000;   // ~0.3 s per launch at 2 PFLOP/s
  run<16, 1>("16x16x32 bf16, 8 chains, 1 wave/SIMD", out, cyc, iters);
  run<32, 1>("32x32x16 bf16, 4 chains, 1 wave/SIMD", out, cyc, iters);
  run<16, 2>("16x16x32 bf16, 8 chains, 2 waves/SIMD", out, cyc, iters / 2);
  run<32, 2>("32x32x16 bf16, 4 chains, 2 waves/SIMD", out, cyc, iters / 2);
  run<16, 1>("16x16x32 bf16 again (thermal state)", out, cyc, iters);
  return 0;
}
