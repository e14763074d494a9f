// Shared host-side helpers of libirspack_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <utility>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/irspack_amd.h"
#include "host_util.hpp"

namespace irs {

#ifdef __HIPCC__
// The wave's index inside its workgroup as a scalar (`threadIdx.x >> 6` alone is a per-lane value to
// the compiler: the wave's row, its bounds and pointers then live in vector registers and are
// computed by vector instructions).  ials_kernels.hpp has the same helper for the iALS kernels.
__device__ __forceinline__ int wave_index_in_block() {
  return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
}
#endif

std::string &last_error();

// The reference signals errors with std::invalid_argument (-> ValueError) and
// std::runtime_error (-> RuntimeError), cpp_source/argcheck.hpp:7-35.
template <class F> irs_status guard(F &&f) {
  try {
    f();
    return IRS_OK;
  } catch (const std::invalid_argument &e) {
    last_error() = e.what();
    return IRS_INVALID_ARGUMENT;
  } catch (const std::exception &e) {
    last_error() = e.what();
    return IRS_RUNTIME_ERROR;
  }
}

#define IRS_HIP(expr)                                                          \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess)                                                      \
      throw std::runtime_error(std::string("HIP error: ") +                    \
                               hipGetErrorString(_e) + " at " + __FILE__ +     \
                               ":" + std::to_string(__LINE__) + " (" #expr ")"); \
  } while (0)

// The product path has no CPU fallback: without a visible gfx950 device every
// compute entry point fails loudly.
void require_device(int device);

template <class T> struct DeviceBuffer {
  T *ptr = nullptr;
  size_t count = 0;
  bool owned = true;  // false: `ptr` is another buffer's memory (borrow)
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
  ~DeviceBuffer() { release(); }
  void release() {
    if (ptr && owned) (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
    owned = true;
  }
  // read-only alias of another buffer (which must outlive this one)
  void borrow(const DeviceBuffer &other) {
    release();
    ptr = other.ptr;
    count = other.count;
    owned = false;
  }
  // n elements at `p` inside a larger allocation somebody else owns (an arena carved per call); alloc()
  // must not be called on a view that is too small - it would leave the arena for memory of its own
  void view(void *p, size_t n) {
    release();
    ptr = static_cast<T *>(p);
    count = n;
    owned = false;
  }
  void alloc(size_t n) {
    if (n <= count && ptr && owned) return;
    release();
    if (n == 0) n = 1;
    IRS_HIP(hipMalloc(reinterpret_cast<void **>(&ptr), n * sizeof(T)));
    count = n;
  }
  void upload(const T *host, size_t n, hipStream_t s) {
    alloc(n);
    if (n) IRS_HIP(hipMemcpyAsync(ptr, host, n * sizeof(T), hipMemcpyHostToDevice, s));
  }
  template <class A> void upload(const std::vector<T, A> &v, hipStream_t s) {
    upload(v.data(), v.size(), s);
  }
  void zero(hipStream_t s) {
    if (ptr) IRS_HIP(hipMemsetAsync(ptr, 0, count * sizeof(T), s));
  }
};

// device_sort.hip: key-value sort of n float keys (ascending or descending, stable) on stream s;
// `tmp` is scratch that grows as needed and may be kept between calls
void sort_pairs_f32(bool descending, const float *keys_in, float *keys_out, const int32_t *vals_in,
                    int32_t *vals_out, size_t n, DeviceBuffer<char> &tmp, hipStream_t s);

// device_sort.hip: X^T of a device-resident CSR (stable: entries of a column stay in row order); t_count
// (host) receives the stored entries per column; synchronises `s`
void transpose_csr_device(const int32_t *indptr, const int32_t *indices, const float *data, int64_t rows,
                          int64_t cols, int64_t nnz, int32_t *t_indices, float *t_data,
                          std::vector<int32_t> &t_count, DeviceBuffer<char> &tmp, hipStream_t s);
void transpose_csr_device(const int32_t *indptr, const int32_t *indices, const double *data, int64_t rows,
                          int64_t cols, int64_t nnz, int32_t *t_indices, double *t_data,
                          std::vector<int32_t> &t_count, DeviceBuffer<char> &tmp, hipStream_t s);

}  // namespace irs
