// Do a matrix-only wave and a vector-only wave that share a SIMD run concurrently?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) issue independent
// v_mfma_f32_16x16x4_f32, waves 4-7 issue v_fma_f32 - dependent chains (DEP=1, like a
// factorisation) or independent ones.  mode 0: matrix waves only, 1: vector waves only, 2: both.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEP>
__global__ __launch_bounds__(512) void k(int mode, int iters, float *out) {
  __shared__ float pad[24 * 1024];  // 96 KB: one workgroup per CU
  const int wv = threadIdx.x >> 6;
  float r = threadIdx.x * 1e-3f;
  if (wv < 4) {
    if (mode == 1) return;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    for (int i = 0; i < iters; i++) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, r, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, r, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, r, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, r, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    if (mode == 0) return;
    float x0 = r, x1 = r + 1, x2 = r + 2, x3 = r + 3;
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (DEP) {
          x0 = fmaf(x0, 1.0001f, 0.5f);
          x0 = fmaf(x0, 0.9999f, 0.25f);
          x0 = fmaf(x0, 1.0001f, 0.5f);
          x0 = fmaf(x0, 0.9999f, 0.25f);
        } else {
          x0 = fmaf(x0, 1.0001f, 0.5f);
          x1 = fmaf(x1, 0.9999f, 0.25f);
          x2 = fmaf(x2, 1.0001f, 0.5f);
          x3 = fmaf(x3, 0.9999f, 0.25f);
        }
      }
    }
    r = x0 + x1 + x2 + x3;
  }
  if (r == 123.456f) out[threadIdx.x] = r + pad[threadIdx.x];
}

template <int DEP> void run(const char *name) {
  float *out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 3; mode++) {
    k<DEP><<<256, 512>>>(mode, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<DEP><<<256, 512>>>(mode, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s mode %d (%s): %.3f ms  [4 MFMA + 32 FMA per iteration, %d iterations]\n", name, mode,
           mode == 0 ? "matrix waves" : mode == 1 ? "vector waves" : "both", ms, iters);
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed fp32 (v_pk_fma_f32): same instruction count as the scalar loop, twice the flops
__global__ __launch_bounds__(512) void kpk(int iters, float *out) {
  __shared__ float pad[24 * 1024];
  const int wv = threadIdx.x >> 6;
  if (wv < 4) return;
  float r = threadIdx.x * 1e-3f;
  f32x2 x0 = {r, r + 1}, x1 = {r + 2, r + 3}, x2 = {r + 4, r + 5}, x3 = {r + 6, r + 7};
  const f32x2 m0 = {1.0001f, 0.9999f}, c0 = {0.5f, 0.25f};
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      x0 = __builtin_elementwise_fma(x0, m0, c0);
      x1 = __builtin_elementwise_fma(x1, m0, c0);
      x2 = __builtin_elementwise_fma(x2, m0, c0);
      x3 = __builtin_elementwise_fma(x3, m0, c0);
    }
  }
  r = x0.x + x0.y + x1.x + x1.y + x2.x + x2.y + x3.x + x3.y;
  if (r == 123.456f) out[threadIdx.x] = r + pad[threadIdx.x];
}

int main() {
  {
    float *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kpk<<<256, 512>>>(20000, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kpk<<<256, 512>>>(20000, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("packed fma (32 v_pk_fma_f32 per iteration, vector waves only): %.3f ms\n", ms);
  }
  run<0>("independent fma");
  run<1>("dependent fma  ");
  return 0;
}
