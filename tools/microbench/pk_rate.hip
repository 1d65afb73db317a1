// Diagnostic: do the packed fp32 forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 operations per lane and
// instruction) issue at the rate of their scalar forms on gfx950? Cycles per instruction of a long independent
// stream, one and two waves per SIMD (s_memtime around 64 x iters instructions), as tools/microbench/valu_rate.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int iters, float seed) {
  f2 v[16];
  for (int i = 0; i < 16; ++i) v[i] = f2{seed + threadIdx.x * 1e-3f + i, seed - i};
  f2 b = {1.0001f, 0.9999f};
  asm volatile("" : "+v"(b));
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(b));
        if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
        if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
        if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(v[i]) : "v"(b));   // scalar broadcast of src1
        if (KIND == 4) {  // the scalar pair it would replace
          asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i].x) : "v"(b.x));
          asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i].y) : "v"(b.y));
        }
        if (KIND == 5) {  // 1 exp : 2 pk_fma (attention mix)
          if ((i % 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i].x));
          else asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(b));
        }
      }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, float* out, unsigned long long* cyc, int per_slot) {
  for (int waves = 1; waves <= 2; ++waves) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256 * waves), 0, 0, out, cyc, iters, 1.5f);
    (void)hipDeviceSynchronize();
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %d wave(s)/SIMD: %.2f cycles per instruction per wave, %.2f per SIMD\n", name, waves,
           (double)c / (64.0 * iters * per_slot), (double)c / (64.0 * iters * per_slot) / waves);
  }
}
int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
  run<0>("v_pk_fma_f32", out, cyc, 1);
  run<1>("v_pk_mul_f32", out, cyc, 1);
  run<2>("v_pk_add_f32", out, cyc, 1);
  run<3>("v_pk_fma_f32 op_sel_hi:[1,0,1]", out, cyc, 1);
  run<4>("v_fma_f32 x 2 (the scalar pair)", out, cyc, 2);
  run<5>("v_exp_f32 : v_pk_fma_f32 = 1 : 2", out, cyc, 1);
  return 0;
}
