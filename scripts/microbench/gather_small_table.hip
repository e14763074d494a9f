// Microbenchmark (development): how fast can every CU gather random ROWS of a small table that lives in L2 /
// Infinity Cache, in the access shape of the iALS rank update - lane (g, m) of a wave reads a 24-byte (or
// 16-byte) piece of the row its 16-lane group was handed, eight rows per lane in flight?
//   rows x row_bytes: 26,744 x 384 (the bf16x3 pre-split item table, 10 MB), 26,744 x 256 (the fp32 one),
//   138,493 x 384 / 256 (the user table).  Row ids: uniform, or Zipf-like (popular rows).
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_small_table gather_small_table.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      std::exit(1);                                                                    \
    }                                                                                  \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

// every wave walks `per_wave` index blocks of 32 rows (its lanes' groups take 8 rows each)
template <int PIECE>  // bytes per lane per row: 16 (one dwordx4) or 24 (two dwordx3)
__global__ __launch_bounds__(256) void gather_kernel(const char *__restrict__ table, const int32_t *__restrict__ idx,
                                                     int row_bytes, int per_wave, float *__restrict__ out) {
  const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
  const int64_t wave = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int32_t *ip = idx + wave * per_wave * 32;
  float acc = 0.f;
  for (int b = 0; b < per_wave; b++) {
    unsigned r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) r[k] = static_cast<unsigned>(ip[b * 32 + 8 * g + k]);
    if (PIECE == 16) {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = *reinterpret_cast<const f32x4 *>(table + r[k] * static_cast<unsigned>(row_bytes) + 16 * m);
#pragma unroll
      for (int k = 0; k < 8; k++) acc += v[k].x + v[k].w;
    } else {
      u32x3 v[8][2];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const char *p = table + r[k] * static_cast<unsigned>(row_bytes) + 24 * m;
        v[k][0] = *reinterpret_cast<const u32x3 *>(p);
        v[k][1] = *reinterpret_cast<const u32x3 *>(p + 12);
      }
#pragma unroll
      for (int k = 0; k < 8; k++) acc += __uint_as_float(v[k][0].x) + __uint_as_float(v[k][1].z);
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

int main() {
  int n_cu = 0;
  CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
  const int waves_per_cu = 12, per_wave = 256;
  const int64_t n_waves = static_cast<int64_t>(n_cu) * waves_per_cu * 8;  // eight rounds of resident waves
  const int64_t n_idx = n_waves * per_wave * 32;
  int32_t *d_idx;
  float *d_out;
  CK(hipMalloc(&d_idx, n_idx * sizeof(int32_t)));
  CK(hipMalloc(&d_out, 64));
  std::vector<int32_t> h(n_idx);
  std::mt19937_64 rng(1);
  for (const int rows : {26744, 138493}) {
    for (const int row_bytes : {256, 384}) {
      char *d_table;
      CK(hipMalloc(&d_table, static_cast<size_t>(rows) * row_bytes));
      CK(hipMemset(d_table, 1, static_cast<size_t>(rows) * row_bytes));
      for (const int zipf : {0, 1}) {
        std::vector<double> cdf(rows);
        double s = 0;
        for (int i = 0; i < rows; i++) cdf[i] = (s += zipf ? 1.0 / (i + 1.0) : 1.0);
        std::uniform_real_distribution<double> U(0.0, s);
        std::vector<int32_t> perm(rows);
        for (int i = 0; i < rows; i++) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), rng);
        for (int64_t i = 0; i < n_idx; i++)
          h[i] = perm[std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin()];
        CK(hipMemcpy(d_idx, h.data(), n_idx * sizeof(int32_t), hipMemcpyHostToDevice));
        for (const int piece : {16, 24}) {
          if (piece * 16 > row_bytes) continue;
          hipEvent_t a, b;
          CK(hipEventCreate(&a));
          CK(hipEventCreate(&b));
          float best = 1e30f;
          for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(a));
            if (piece == 16)
              hipLaunchKernelGGL(gather_kernel<16>, dim3(n_waves / 4), dim3(256), 0, 0, d_table, d_idx, row_bytes, per_wave, d_out);
            else
              hipLaunchKernelGGL(gather_kernel<24>, dim3(n_waves / 4), dim3(256), 0, 0, d_table, d_idx, row_bytes, per_wave, d_out);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            best = std::min(best, ms);
          }
          const double bytes = static_cast<double>(n_idx) * piece * 16;
          std::printf("rows %6d x %3d B (%5.1f MB) %s piece %2d B/lane: %7.1f GB/s useful, %6.1f M rows/ms\n", rows, row_bytes,
                      rows * static_cast<double>(row_bytes) / 1e6, zipf ? "zipf   " : "uniform", piece,
                      bytes / (best * 1e-3) / 1e9, n_idx / best / 1e6);
        }
      }
      CK(hipFree(d_table));
    }
  }
  return 0;
}
