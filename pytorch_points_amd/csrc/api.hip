// api.hip -- library identification and host-side helpers of the C ABI (include/pp_hip.h).
#include <algorithm>
#include <cmath>

#include "pp_common.h"

extern "C" const char* pp_version(void) { return "pp_hip 0.1.0 gfx950"; }

// Same expression as the reference's host helper (_ext/cuda_utils.h:11-16), evaluated in double on
// the host exactly as there: 2^floor(log2(work_size)) clamped to [1, 512].
extern "C" int pp_opt_n_threads(int work_size) {
  const int pow_2 = (int)(std::log(static_cast<double>(work_size)) / std::log(2.0));
  return std::max(std::min(1 << pow_2, 512), 1);
}

// ---- the chip's store ceiling, measured (bench.py: roofline.peak_measured of the group_points line) ----
// A pure streaming store: every lane 16-byte non-temporal stores of a constant, consecutive lanes consecutive
// addresses, one 1024-thread workgroup per CU walking its share of the buffer -- no loads, no index arithmetic worth
// the name: what any kernel that writes `bytes` can at best approach on this device at this moment.
namespace {
__global__ __launch_bounds__(1024) void store_ceiling_kernel(pp::f4* __restrict__ out, size_t n16, int nt) {
  const pp::f4 v = {1.0f, 2.0f, 3.0f, 4.0f};
  const size_t stride = (size_t)gridDim.x * 1024;
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += stride) {
    if (nt)
      __builtin_nontemporal_store(v, out + i);
    else
      out[i] = v;
  }
}
}  // namespace

extern "C" int pp_debug_store_ceiling(void* buf, size_t bytes, int nontemporal, int workgroups, void* stream) {
  if (!buf || bytes < 16 || workgroups <= 0) return PP_EINVAL;
  store_ceiling_kernel<<<dim3((unsigned)workgroups), dim3(1024), 0, (hipStream_t)stream>>>((pp::f4*)buf, bytes / 16,
                                                                                          nontemporal);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}
