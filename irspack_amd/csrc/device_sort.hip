// Device-wide key-value sorts (rocPRIM's radix sort), set-up work only, kept in their own translation
// unit because the rocPRIM headers are heavy:
//   sort_pairs_f32        float keys: the item norms and the per-user pruning radii of the evaluator's
//                         bounded path (~10^4 .. 10^6 keys per call)
//   transpose_csr_device  X^T of a device-resident CSR (IALSTrainer's X.transpose(), hpp:713; the kNN
//                         computers' X_arg^T, knn.hpp:30-41, float64 values): a stable
//                         sort of the entry numbers by column + one gather - once per trainer
// No reference counterpart.
#include <cstring>

#include "common.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace irs {

namespace {

__global__ __launch_bounds__(256) void column_count_kernel(const int32_t *__restrict__ indices, int64_t nnz,
                                                           int32_t *__restrict__ count) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < nnz; p += stride)
    atomicAdd(count + indices[p], 1);
}

// entry q of X^T is entry perm[q] of X: its row = the row whose [indptr[r], indptr[r + 1]) holds perm[q]
template <class V>
__global__ __launch_bounds__(256) void transposed_entries_kernel(const int32_t *__restrict__ perm,
                                                                 const int32_t *__restrict__ indptr, int32_t rows,
                                                                 const V *__restrict__ data, int64_t nnz,
                                                                 int32_t *__restrict__ t_indices,
                                                                 V *__restrict__ t_data) {
  const int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= nnz) return;
  const int32_t p = perm[q];
  int32_t lo = 0, hi = rows;  // the last r in [0, rows) with indptr[r] <= p
  while (hi - lo > 1) {
    const int32_t mid = lo + ((hi - lo) >> 1);
    if (indptr[mid] <= p) lo = mid;
    else hi = mid;
  }
  t_indices[q] = lo;
  if (t_data) t_data[q] = data[p];
}

}  // namespace

// X^T of the CSR (indptr [rows + 1], indices, data or null - all on the device) into t_indices /
// t_data (device, nnz entries; t_data null when data is) and t_count (host: stored entries per column).
// Entries of a column keep their row order (the radix sort is stable and the entries arrive in row
// order): the result is the sequential counting-sort transpose, bit for bit.
template <class V>
static void transpose_csr_device_impl(const int32_t *indptr, const int32_t *indices, const V *data, int64_t rows,
                                      int64_t cols, int64_t nnz, int32_t *t_indices, V *t_data,
                                      std::vector<int32_t> &t_count, DeviceBuffer<char> &tmp, hipStream_t s) {
  t_count.assign(static_cast<size_t>(std::max<int64_t>(cols, 0)), 0);
  if (nnz == 0 || cols == 0) return;
  int end_bit = 1;
  while (end_bit < 32 && (int64_t(1) << end_bit) < cols) end_bit++;
  const size_t n = static_cast<size_t>(nnz);
  size_t sort_bytes = 0;
  rocprim::counting_iterator<int32_t> entry(0);
  int32_t *none = nullptr;
  IRS_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, indices, none, entry, none, n, 0, end_bit, s));
  // scratch: [column counts | sorted keys | permutation | rocPRIM's own]
  auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t o_cnt = 0, o_keys = o_cnt + up(static_cast<size_t>(cols) * 4), o_perm = o_keys + up(n * 4),
               o_sort = o_perm + up(n * 4);
  // (a caller's view that is large enough - scratch carved out of an arena - is used as it is)
  if (!tmp.ptr || tmp.count < o_sort + sort_bytes) tmp.alloc(o_sort + sort_bytes);
  int32_t *d_cnt = reinterpret_cast<int32_t *>(tmp.ptr + o_cnt);
  int32_t *d_keys = reinterpret_cast<int32_t *>(tmp.ptr + o_keys);
  int32_t *d_perm = reinterpret_cast<int32_t *>(tmp.ptr + o_perm);
  IRS_HIP(hipMemsetAsync(d_cnt, 0, static_cast<size_t>(cols) * 4, s));
  hipLaunchKernelGGL(column_count_kernel, dim3(static_cast<unsigned>(std::min<int64_t>((nnz + 255) / 256, 8192))),
                     dim3(256), 0, s, indices, nnz, d_cnt);
  IRS_HIP(hipMemcpyAsync(t_count.data(), d_cnt, static_cast<size_t>(cols) * 4, hipMemcpyDeviceToHost, s));
  IRS_HIP(rocprim::radix_sort_pairs(tmp.ptr + o_sort, sort_bytes, indices, d_keys, entry, d_perm, n, 0, end_bit, s));
  hipLaunchKernelGGL(transposed_entries_kernel<V>, dim3(static_cast<unsigned>((nnz + 255) / 256)), dim3(256), 0, s,
                     static_cast<const int32_t *>(d_perm), indptr, static_cast<int32_t>(rows), data, nnz, t_indices,
                     t_data);
  IRS_HIP(hipGetLastError());
  IRS_HIP(hipStreamSynchronize(s));
}

void transpose_csr_device(const int32_t *indptr, const int32_t *indices, const float *data, int64_t rows,
                          int64_t cols, int64_t nnz, int32_t *t_indices, float *t_data,
                          std::vector<int32_t> &t_count, DeviceBuffer<char> &tmp, hipStream_t s) {
  transpose_csr_device_impl<float>(indptr, indices, data, rows, cols, nnz, t_indices, t_data, t_count, tmp, s);
}
void transpose_csr_device(const int32_t *indptr, const int32_t *indices, const double *data, int64_t rows,
                          int64_t cols, int64_t nnz, int32_t *t_indices, double *t_data,
                          std::vector<int32_t> &t_count, DeviceBuffer<char> &tmp, hipStream_t s) {
  transpose_csr_device_impl<double>(indptr, indices, data, rows, cols, nnz, t_indices, t_data, t_count, tmp, s);
}

void sort_pairs_f32(bool descending, const float *keys_in, float *keys_out, const int32_t *vals_in,
                    int32_t *vals_out, size_t n, DeviceBuffer<char> &tmp, hipStream_t s) {
  if (n == 0) return;
  size_t bytes = 0;
  if (descending)
    IRS_HIP(rocprim::radix_sort_pairs_desc(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 32, s));
  else
    IRS_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 32, s));
  tmp.alloc(bytes);
  if (descending)
    IRS_HIP(rocprim::radix_sort_pairs_desc(tmp.ptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 32, s));
  else
    IRS_HIP(rocprim::radix_sort_pairs(tmp.ptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 32, s));
}

}  // namespace irs
