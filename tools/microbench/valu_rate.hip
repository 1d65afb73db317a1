// Diagnostic: issue/execute cost of the VALU instructions the f16 LSTM's gate math is made of, one wave per SIMD
// and two, as cycles per instruction of a long independent stream (s_memtime around 64 x 200 instructions).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int iters, float seed) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
  int p = __builtin_bit_cast(int, seed);
  float k4096 = 4096.f;
  asm volatile("" : "+v"(k4096));
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        if (KIND == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
        if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v[i]));
        if (KIND == 3) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(v[i]));
        if (KIND == 4) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(p));
        if (KIND == 5) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(k4096));
        if (KIND == 6) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
        if (KIND == 7) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
        if (KIND == 8) {   // exp, add, rcp, fma alternating (the gate math's own mix: 1 transcendental in 2)
          if (i & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
          else asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v[i]));
        }
        if (KIND == 9) {   // 1 transcendental in 4
          if ((i & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
          else asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(v[i]));
        }
      }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, float* out, unsigned long long* cyc) {
  for (int waves = 1; waves <= 2; ++waves) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256 * waves), 0, 0, out, cyc, iters, 1.5f);
    (void)hipDeviceSynchronize();
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s %d wave(s)/SIMD: %.2f cycles per instruction per wave, %.2f per SIMD\n", name, waves, (double)c / (64.0 * iters),
           (double)c / (64.0 * iters) / waves);
  }
}
int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
  run<0>("v_exp_f32", out, cyc);
  run<1>("v_rcp_f32", out, cyc);
  run<2>("v_fma_f32", out, cyc);
  run<3>("v_add_f32", out, cyc);
  run<4>("v_fma_mix_f32 (f16 source)", out, cyc);
  run<5>("v_fma_mixlo_f16", out, cyc);
  run<6>("v_cvt_pk_f16_f32", out, cyc);
  run<7>("v_max3_f32 |.|", out, cyc);
  run<8>("v_exp_f32 : v_fma_f32 = 1 : 1", out, cyc);
  run<9>("v_exp_f32 : v_fma_f32 = 1 : 3", out, cyc);
  return 0;
}
