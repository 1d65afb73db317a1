// Diagnostic: do MFMA phases of one wave overlap VALU phases of another wave on the same SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// per iteration: NM MFMAs (operands from LDS, b128 per 4) then NV rounds of sigmoid-like VALU on 32 values
template <bool SMALL>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int iters, int nv) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 1e-4f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float a = threadIdx.x * 1e-3f;
  float v[32];
  for (int j = 0; j < 32; ++j) v[j] = j * 0.01f + a;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (threadIdx.x >= 256) {  // skew the second wave of every SIMD so the phases interleave
    for (int r = 0; r < 12; ++r)
#pragma unroll
      for (int j = 0; j < 32; ++j) v[j] = __builtin_amdgcn_rcpf(1.f + __expf(-v[j]));
  }
  if (SMALL) {
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 32; ++q) {   // 32 k-steps x 16 tiles = 512 MFMA 16x16x4 (= 256 of the 32x32x2)
#pragma unroll
        for (int hf = 0; hf < 4; ++hf) {
          const float4 w = reinterpret_cast<const float4*>(lds)[((q * 4 + hf) & 31) * 64 + lane];
          acc[4 * hf + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.x, acc[4 * hf + 0], 0, 0, 0);
          acc[4 * hf + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.y, acc[4 * hf + 1], 0, 0, 0);
          acc[4 * hf + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.z, acc[4 * hf + 2], 0, 0, 0);
          acc[4 * hf + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w.w, acc[4 * hf + 3], 0, 0, 0);
        }
      }
      for (int r = 0; r < nv; ++r)
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = __builtin_amdgcn_rcpf(1.f + __expf(-v[j])) + acc[j & 15][j & 3] * 1e-30f;
    }
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) a += acc[j][r];
  } else {
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 32; ++q) {   // 32 k-steps x 8 tiles = 256 MFMA 32x32x2
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const float4 w = reinterpret_cast<const float4*>(lds)[((q * 2 + hf) & 31) * 64 + lane];
          acc[4 * hf + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w.x, acc[4 * hf + 0], 0, 0, 0);
          acc[4 * hf + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w.y, acc[4 * hf + 1], 0, 0, 0);
          acc[4 * hf + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w.z, acc[4 * hf + 2], 0, 0, 0);
          acc[4 * hf + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w.w, acc[4 * hf + 3], 0, 0, 0);
        }
      }
      for (int r = 0; r < nv; ++r)
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = __builtin_amdgcn_rcpf(1.f + __expf(-v[j])) + acc[j & 7][j & 15] * 1e-30f;
    }
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) a += acc[j][r];
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  for (int j = 0; j < 32; ++j) a += v[j];
  out[blockIdx.x * 512 + threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <bool SMALL>
void run(const char* name, int threads, int nv, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SMALL>), dim3(256), dim3(threads), 32768, 0, out, cyc, iters, nv);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double flop = (double)iters * 256 * 4096 * (threads / 64) * 256;   // both forms: 256 "32x32x2 equivalents" per iteration
  printf("%-34s threads=%d nv=%d: %.2f ms, %.0f ticks/iter, MFMA %.1f TFLOP/s\n", name, threads, nv, ms, (double)c / iters, flop / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
  for (int nv : {0, 2, 4}) {
    run<false>("32x32x2, 1 wave/SIMD", 256, nv, out, cyc);
    run<false>("32x32x2, 2 waves/SIMD", 512, nv, out, cyc);
    run<true>("16x16x4, 1 wave/SIMD", 256, nv, out, cyc);
    run<true>("16x16x4, 2 waves/SIMD", 512, nv, out, cyc);
  }
  return 0;
}
