// Diagnostic: sustained rate of v_mfma_f32_32x32x2_f32 under different operand sources.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NT, int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  f32x16 acc[NT];
  for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  const int lane = threadIdx.x & 63;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  float bc[2][8], bn[2][8];
  if (MODE == 1 || MODE == 3) for (int u = 0; u < 2; ++u) for (int j = 0; j < 8; ++j) bc[u][j] = lds[(u * 8 + j) * 64 + lane];
  float4 wc, wn;
  if (MODE == 2) wc = reinterpret_cast<float4*>(lds)[lane];
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 32 / NT; ++u)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    } else if (MODE == 1 || MODE == 3) {  // one ds_read_b32 per MFMA, a group of 16 ahead (NT == 8)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        int off = ((i * 2 + g) & 3) * 1024 + lane;
        if (MODE == 3) { int x = off; x ^= (x >> 3) & 7; x = (x << 1) ^ (x >> 5); off = (x & 1023) + lane; }  // some address VALU
        for (int u = 0; u < 2; ++u) for (int j = 0; j < 8; ++j) bn[u][j] = lds[off + (u * 8 + j) * 64];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bc[u][j], acc[j % NT], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        for (int u = 0; u < 2; ++u) for (int j = 0; j < 8; ++j) bc[u][j] = bn[u][j];
      }
    } else if (MODE == 4) {
      const int li = lane & 31, kh = lane >> 5;
      const float4* arow = reinterpret_cast<const float4*>(lds) + li * 64;
      const int sw = (li & 15) ^ ((li & 1) << 3);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 av = arow[(2 * q + kh) ^ sw];
        const float4 w = reinterpret_cast<const float4*>(lds)[((i * 8 + q) & 15) * 64 + lane];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, w.x, acc[0], 0, 0, 0);
        acc[1 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, w.y, acc[1 % NT], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, w.z, acc[0], 0, 0, 0);
        acc[1 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, w.w, acc[1 % NT], 0, 0, 0);
      }
    } else {  // MODE 2: one ds_read_b128 per 4 MFMAs, one ahead; 2 accumulator chains (NT = 2) or more
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        wn = reinterpret_cast<float4*>(lds)[((i * 8 + q) & 15) * 64 + lane];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wc.x, acc[0], 0, 0, 0);
        acc[1 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wc.y, acc[1 % NT], 0, 0, 0);
        acc[2 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wc.z, acc[2 % NT], 0, 0, 0);
        acc[3 % NT] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wc.w, acc[3 % NT], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wc = wn;
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NT, int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 10000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NT, MODE>), dim3(256), dim3(256), 32768, 0, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double nm = (double)iters * 32;
  printf("%-44s %.2f ms, %.1f ticks/MFMA, %.1f TFLOP/s\n", name, ms, c / nm, 1024 * nm * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 4 * 8);
  run<8, 0>("regs, 8 accumulators", out, cyc);
  run<4, 0>("regs, 4 accumulators", out, cyc);
  run<2, 0>("regs, 2 accumulators", out, cyc);
  run<1, 0>("regs, 1 accumulator", out, cyc);
  run<8, 1>("B: ds_read_b32 per MFMA, group ahead", out, cyc);
  run<8, 3>("same + address VALU", out, cyc);
  run<2, 2>("B: ds_read_b128 per 4 MFMA, 2 chains", out, cyc);
  run<4, 2>("B: ds_read_b128 per 4 MFMA, 4 chains", out, cyc);
  run<2, 4>("A+B: 2 ds_read_b128 per 4 MFMA, 2 chains", out, cyc);
  return 0;
}
