// Measured ceilings of the device this process runs on ("a device copy / triad
// GB/s and a pure-MFMA fp32 GEMM TF measured in the same run next to the spec peaks").
// A few tiny kernels, HIP-event timed: bench.py prints fraction-of-spec and
// fraction-of-measured for its roofline objects.  No reference counterpart.
#include <algorithm>

#include "common.hpp"

namespace irs {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// dst = src, 16 B per lane, grid-stride (HBM read + write)
__global__ __launch_bounds__(256) void copy_kernel(const f32x4 *__restrict__ src,
                                                   f32x4 *__restrict__ dst, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = __builtin_nontemporal_load(src + i);
}

// the same with four 16-byte loads in flight per lane before the first store (plain loads)
__global__ __launch_bounds__(256) void copy4_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst,
                                                    size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const f32x4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
    dst[i] = v0;
    dst[i + stride] = v1;
    dst[i + 2 * stride] = v2;
    dst[i + 3 * stride] = v3;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

// a = b + s * c (STREAM triad: two reads, one write)
__global__ __launch_bounds__(256) void triad_kernel(const f32x4 *__restrict__ b,
                                                    const f32x4 *__restrict__ c,
                                                    f32x4 *__restrict__ a, float s, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    a[i] = b[i] + s * c[i];
}

// Nothing but v_mfma_f32_16x16x4_f32 on 8 independent accumulators per wave (the GEMM inner
// loop with the operands already in registers): 2048 flop per instruction.
__global__ __launch_bounds__(256) void mfma_f32_kernel(int iters, float *__restrict__ out) {
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  // 32 instructions per trip: the loop's scalar bookkeeping is ~1 % of the issue slots, not ~4 %
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 8; i++) s += acc[i];
  if (s.x == 12345.678f) out[threadIdx.x] = s.x + s.y + s.z + s.w;  // never true: keeps the loop
}

// ds_add_u32 with one random bank per lane (the kNN count accumulation's instruction)
__global__ __launch_bounds__(1024) void lds_atomic_kernel(int n_per_thread, unsigned *__restrict__ out) {
  __shared__ unsigned acc[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) acc[i] = 0;
  __syncthreads();
  unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll 8
  for (int it = 0; it < n_per_thread; it++) {
    st = st * 1664525u + 1013904223u;
    atomicAdd(&acc[(st >> 8) & 16383], 1u);
  }
  __syncthreads();
  unsigned s = 0;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) s += acc[i];
  if (s == 0xdeadbeefu) out[blockIdx.x] = s;
}

// Random gather of whole rows (ROW_F4 x 16 bytes each, 16 lanes per row) out of a 1 GiB table:
// the access pattern of the iALS rank update and of its short-row kernels (one gathered factor
// row per stored entry).  Eight row loads are in flight per lane group; the rows are drawn by a
// per-group LCG, uniformly - no reuse beyond what 2 M random rows leave in the caches.
template <int ROW_F4>
__global__ __launch_bounds__(256) void gather_rows_kernel(const f32x4 *__restrict__ table, unsigned row_mask,
                                                          int rows_per_group, f32x4 *__restrict__ out) {
  constexpr int PER_LANE = ROW_F4 / 16;  // float4 loads per lane and row (1: 256 B rows, 2: 512 B rows)
  const unsigned group = (blockIdx.x * 256u + threadIdx.x) >> 4, m = threadIdx.x & 15;
  unsigned st = group * 2654435761u + 12345u;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < rows_per_group; it += 8) {
    f32x4 v[8][PER_LANE];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      st = st * 1664525u + 1013904223u;
      const f32x4 *row = table + static_cast<size_t>((st >> 8) & row_mask) * ROW_F4;
#pragma unroll
      for (int q = 0; q < PER_LANE; q++) v[u][q] = row[16 * q + m];
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int q = 0; q < PER_LANE; q++) acc += v[u][q];
  }
  if (acc.x == 12345.678f) out[threadIdx.x] = acc;  // never true (the table is zero): keeps the loads
}

template <class F> double best_ms(F &&launch, hipStream_t s, int reps) {
  hipEvent_t e0, e1;
  IRS_HIP(hipEventCreate(&e0));
  IRS_HIP(hipEventCreate(&e1));
  launch();  // warm-up
  double best = 1e30;
  for (int r = 0; r < reps; r++) {
    IRS_HIP(hipEventRecord(e0, s));
    launch();
    IRS_HIP(hipEventRecord(e1, s));
    IRS_HIP(hipEventSynchronize(e1));
    float ms = 0;
    IRS_HIP(hipEventElapsedTime(&ms, e0, e1));
    best = std::min<double>(best, ms);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  IRS_HIP(hipGetLastError());
  return best;
}

}  // namespace
}  // namespace irs

using namespace irs;

extern "C" irs_status irs_measure_ceilings(int32_t device, irs_ceilings *out) {
  return guard([&] {
    check_arg(out != nullptr, "null argument.");
    require_device(device);
    hipStream_t s = nullptr;
    int n_cu = 0;
    IRS_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device));
    n_cu = std::max(n_cu, 1);
    // 1 GiB per array: far beyond the 256 MB Infinity Cache
    const size_t n = (size_t(1) << 30) / sizeof(f32x4);
    DeviceBuffer<f32x4> a, b, c;
    a.alloc(n);
    b.alloc(n);
    c.alloc(n);
    IRS_HIP(hipMemsetAsync(a.ptr, 0, n * sizeof(f32x4), s));
    IRS_HIP(hipMemsetAsync(b.ptr, 0, n * sizeof(f32x4), s));
    IRS_HIP(hipMemsetAsync(c.ptr, 0, n * sizeof(f32x4), s));
    const int grid = n_cu * 16;
    const double bytes = static_cast<double>(n) * sizeof(f32x4);
    // the best of: non-temporal grid-stride copy, four loads in flight per lane at 8 / 16 / 32
    // workgroups per compute unit, and the runtime's own device-to-device copy
    double ms = best_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, s, b.ptr, a.ptr, n); }, s, 5);
    for (int per_cu : {8, 16, 32})
      ms = std::min(ms, best_ms([&] { hipLaunchKernelGGL(copy4_kernel, dim3(n_cu * per_cu), dim3(256), 0, s, b.ptr, a.ptr, n); }, s, 3));
    ms = std::min(ms, best_ms([&] { (void)hipMemcpyAsync(a.ptr, b.ptr, n * sizeof(f32x4), hipMemcpyDeviceToDevice, s); }, s, 3));
    out->copy_gbs = 2.0 * bytes / (ms * 1e-3) / 1e9;
    ms = best_ms([&] { hipLaunchKernelGGL(triad_kernel, dim3(grid), dim3(256), 0, s, b.ptr, c.ptr, a.ptr, 0.5f, n); }, s, 5);
    out->triad_gbs = 3.0 * bytes / (ms * 1e-3) / 1e9;
    // 1, 2 or 4 workgroups of 4 waves per CU (= waves per SIMD): the best rate
    out->mfma_f32_tflops = 0.0;
    for (int per_cu : {1, 2, 4}) {
      const int iters = 8192, mgrid = n_cu * per_cu;
      ms = best_ms([&] { hipLaunchKernelGGL(mfma_f32_kernel, dim3(mgrid), dim3(256), 0, s, iters, reinterpret_cast<float *>(a.ptr)); }, s, 5);
      out->mfma_f32_tflops = std::max(out->mfma_f32_tflops,
                                      static_cast<double>(mgrid) * 4 * iters * 8 * 2048.0 / (ms * 1e-3) / 1e12);
    }
    const int per = 4096;
    ms = best_ms([&] { hipLaunchKernelGGL(lds_atomic_kernel, dim3(n_cu), dim3(1024), 0, s, per, reinterpret_cast<unsigned *>(a.ptr)); }, s, 3);
    out->lds_atomic_u32_gops = static_cast<double>(n_cu) * 1024 * per / (ms * 1e-3) / 1e9;
    {
      // 1 GiB table: 4 M rows of 256 B / 2 M rows of 512 B; 8 workgroups per CU, 16 row groups each
      const int ggrid = n_cu * 8, per_group = 512;
      const double rows = static_cast<double>(ggrid) * 16 * per_group;
      ms = best_ms([&] { hipLaunchKernelGGL(gather_rows_kernel<16>, dim3(ggrid), dim3(256), 0, s, b.ptr, (1u << 22) - 1, per_group, a.ptr); }, s, 3);
      out->gather256_gbs = rows * 256.0 / (ms * 1e-3) / 1e9;
      ms = best_ms([&] { hipLaunchKernelGGL(gather_rows_kernel<32>, dim3(ggrid), dim3(256), 0, s, b.ptr, (1u << 21) - 1, per_group, a.ptr); }, s, 3);
      out->gather512_gbs = rows * 512.0 / (ms * 1e-3) / 1e9;
    }
    out->n_cu = n_cu;
    int clk_khz = 0;
    IRS_HIP(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, device));
    out->clock_mhz = clk_khz / 1000.0;
  });
}
