// Device-wide key-value sort of float keys (rocPRIM's radix sort): orders the item norms and the
// per-user pruning radii of the evaluator's bounded path.  Set-up work on ~10^4..10^6 keys per
// call, not a hot kernel; kept in its own translation unit because the rocPRIM headers are heavy.
// No reference counterpart.
#include <cstring>

#include "common.hpp"

#include <rocprim/device/device_radix_sort.hpp>

namespace irs {

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
