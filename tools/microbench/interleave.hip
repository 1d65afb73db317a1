// Diagnostic: cost of VALU fillers placed between MFMAs of the SAME wave (program-order interleave).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NF, bool TRANS>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float a = threadIdx.x * 1e-3f;
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = j * 0.01f + a;
  float4 wc[2], wn[2];
  for (int hf = 0; hf < 2; ++hf) wc[hf] = reinterpret_cast<const float4*>(lds)[hf * 64 + lane];
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      for (int hf = 0; hf < 2; ++hf) wn[hf] = reinterpret_cast<const float4*>(lds)[(((i * 8 + q + 1) * 2 + hf) & 31) * 64 + lane];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float w = e == 0 ? wc[hf].x : e == 1 ? wc[hf].y : e == 2 ? wc[hf].z : wc[hf].w;
          acc[4 * hf + e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w, acc[4 * hf + e], 0, 0, 0);
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            float& z = v[(e * NF + f) & 15];
            z = TRANS ? __builtin_amdgcn_rcpf(1.f + __expf(-z)) : fmaf(z, 1.0001f, 0.5f);
          }
          __builtin_amdgcn_sched_barrier(0);   // keep the fillers where they are written
        }
      }
      for (int hf = 0; hf < 2; ++hf) wc[hf] = wn[hf];
    }
  }
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) a += acc[j][r];
  for (int j = 0; j < 16; ++j) a += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = a;
}
template <int NF, bool TRANS>
void run(float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NF, TRANS>), dim3(256), dim3(256), 32768, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double nm = (double)iters * 64;
  printf("%d %s fillers per MFMA: %.2f ms, %.1f ns*2.4GHz cycles per MFMA, MFMA %.1f TFLOP/s\n", NF, TRANS ? "sigmoid (exp+rcp+2)" : "fma", ms,
         ms * 1e-3 / nm * 2.4e9, 1024 * nm * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  run<0, false>(out); run<2, false>(out); run<4, false>(out); run<8, false>(out); run<12, false>(out); run<16, false>(out);
  run<1, true>(out); run<2, true>(out); run<3, true>(out); run<4, true>(out);
  return 0;
}
