// Semantics check of the op_sel / op_sel_hi broadcast forms of v_pk_fma_f32 / v_pk_mul_f32 used by attn_split.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(float* o) {
  f2 a = {2.f, 3.f}, b = {10.f, 100.f}, c = {1.f, 5.f}, d;
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));              // a * b.lo + c
  o[0] = d.x, o[1] = d.y;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); // a * b.hi + c
  o[2] = d.x, o[3] = d.y;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));                               // a * b.lo
  o[4] = d.x, o[5] = d.y;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));                  // a * b.hi
  o[6] = d.x, o[7] = d.y;
}
int main() {
  float* o; (void)hipMalloc(&o, 64); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o); float h[8]; (void)hipMemcpy(h, o, 32, hipMemcpyDeviceToHost);
  printf("fma lo: %g %g (want 21 35)\nfma hi: %g %g (want 201 305)\nmul lo: %g %g (want 20 30)\nmul hi: %g %g (want 200 300)\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  return 0;
}
