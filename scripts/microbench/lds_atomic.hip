// Microbenchmark: LDS atomic throughput per CU on gfx950 (development aid for knn.hip).
//   hipcc -O3 --offload-arch=gfx950 lds_atomic.hip -o lds_atomic && ./lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(1024) void bench(const int32_t *cols, int n_per_thread, int width, double *out) {
  extern __shared__ double acc[];
  for (int i = threadIdx.x; i < width; i += blockDim.x) acc[i] = 0.0;
  __syncthreads();
  const int dist = cols[0];
  unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll 8
  for (int it = 0; it < n_per_thread; it++) {
    st = st * 1664525u + 1013904223u;
    int j;
    if (dist == 0) j = (st >> 8) & (width - 1);                          // uniform random
    else if (dist == 1) j = (threadIdx.x & 63) + 64 * ((it + (threadIdx.x >> 6) * 13) & 255);  // conflict-free in a wave
    else if (dist == 2) j = ((st >> 8) & (st >> 16) & (st >> 3)) & (width - 1);  // skewed towards low columns
    else j = ((threadIdx.x & 31) * 2) + 64 * (it & 15);                // 2 lanes per address
    if (MODE == 0) atomicAdd(&acc[j], 1.0);
    if (MODE == 1) atomicAdd(reinterpret_cast<unsigned *>(acc) + j, 1u);
    if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long *>(acc) + j, 1ull);
    if (MODE == 3) acc[j] = 1.0;                      // plain store
    if (MODE == 4) atomicAdd(reinterpret_cast<float *>(acc) + j, 1.0f);
    if (MODE == 5) __hip_atomic_fetch_add(&acc[j], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  double s = 0;
  for (int i = threadIdx.x; i < width; i += blockDim.x) s += acc[i];
  if (s == 12345.678) out[blockIdx.x] = s;
}

int main() {
  const int width = 16384, threads = 1024, n_per = 2048, blocks = 256;
  const char *names[] = {"f64 add", "u32 add", "u64 add", "f64 store", "f32 add", "f64 add wg-scope"};
  for (int dist = 0; dist < 4; dist++) {
    std::vector<int32_t> h((size_t)blocks * threads * n_per);
    uint64_t st = 88172645463325252ull;
    h.resize(16); h[0] = dist; (void)st;
    int32_t *d; double *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, blocks * 8);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto k, const char *name) {
      hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, width * 8);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), width * 8, 0, d, n_per, width, o);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), width * 8, 0, d, n_per, width, o);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double ops = (double)blocks * threads * n_per;
      printf("dist %d %-18s %8.3f ms  %7.1f Gop/s  %5.2f ops/clk/CU (2.4 GHz, %d WGs)\n", dist, name, ms, ops / ms / 1e6,
             ops / (ms * 1e-3) / 2.4e9 / blocks, blocks);
    };
    run(bench<0>, names[0]); run(bench<1>, names[1]); run(bench<2>, names[2]); run(bench<3>, names[3]);
    run(bench<4>, names[4]); run(bench<5>, names[5]);
    hipFree(d); hipFree(o);
  }
  return 0;
}
