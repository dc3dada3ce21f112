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
