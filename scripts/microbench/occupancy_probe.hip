// Development probe: resident workgroups per CU the runtime reports for the evaluator's big kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../irspack_amd/csrc -o occupancy_probe occupancy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "common.hpp"
#include "eval_fused_kernels.hpp"
using namespace irs::eval;
int main() {
  int n = -1;
  const size_t lds = 4 * 64 * FZ_SROW * sizeof(float) + 4 * 64 * sizeof(int32_t);
  hipFuncSetAttribute(reinterpret_cast<const void *>(score_emit_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, score_emit_kernel<64, true>, 256, lds);
  std::printf("score_emit_kernel<64, true>: %d workgroups of 256 per CU with %zu B of LDS (%s)\n", n, lds, hipGetErrorString(e));
  hipFuncAttributes a;
  hipFuncGetAttributes(&a, reinterpret_cast<const void *>(score_emit_kernel<64, true>));
  std::printf("  numRegs %d sharedSizeBytes %zu maxDynamicSharedSizeBytes %d localSizeBytes %zu\n", a.numRegs, a.sharedSizeBytes, a.maxDynamicSharedSizeBytes, a.localSizeBytes);
  hipFuncSetAttribute(reinterpret_cast<const void *>(sample_tau_fused_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(SF_LDS_BYTES));
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sample_tau_fused_kernel<64>, 512, SF_LDS_BYTES);
  std::printf("sample_tau_fused_kernel<64>: %d workgroups of 512 per CU with %zu B of LDS (%s)\n", n, SF_LDS_BYTES, hipGetErrorString(e));
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  std::printf("device: %zu B LDS per CU (maxSharedMemoryPerMultiProcessor), %zu per block, regsPerMultiprocessor %d\n", pr.maxSharedMemoryPerMultiProcessor, pr.sharedMemPerBlock, pr.regsPerMultiprocessor);
  return 0;
}
