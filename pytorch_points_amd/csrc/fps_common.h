// fps_common.h -- helpers shared by the furthest-point-sampling kernels (fps.hip, fps_bucket.hip).
#pragma once
#include "pp_common.h"

namespace ppfps {

typedef unsigned long long u64;

// Wave-wide unsigned 64-bit max, result in every lane.  DPP row shifts / row broadcasts (VALU
// speed, no LDS crossbar): six steps leave the maximum in lane 63, one v_readlane pair broadcasts
// it.  max is idempotent, so full row/bank masks are fine (an element may be folded in twice).
// (2.10 -> 1.78 us per pick at config 3 against the __shfl_xor butterfly.)
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_max_step(unsigned long long v) {
  const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
  const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
  return o > v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = dpp_max_step<0x111>(v);  // row_shr:1
  v = dpp_max_step<0x112>(v);  // row_shr:2
  v = dpp_max_step<0x114>(v);  // row_shr:4
  v = dpp_max_step<0x118>(v);  // row_shr:8   -> lane 15 of each row holds the row maximum
  v = dpp_max_step<0x142>(v);  // row_bcast:15
  v = dpp_max_step<0x143>(v);  // row_bcast:31 -> lane 63 holds the wave maximum
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

// The same maximum as two 32-bit passes -- the high words, then the low words of the lanes that hold the
// maximal high word -- each a chain of six in-place `v_max_u32_dpp` (lanes without a source keep their
// value; a nop between dependent DPP operations, which need two wait states after the VALU write): 12
// VALU instructions instead of the 30 of the 64-bit compare-and-select steps above.  Every lane active.
template <int STEPS>
__device__ __forceinline__ unsigned wave_max_u32(unsigned x) {
  static_assert(STEPS == 3 || STEPS == 4 || STEPS == 6, "");
  if constexpr (STEPS == 3) {  // values in lanes 0..7 -> lane 7
    asm volatile(
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
        : "+v"(x));
    return (unsigned)__builtin_amdgcn_readlane((int)x, 7);
  } else if constexpr (STEPS == 4) {  // values in lanes 0..15 -> lane 15
    asm volatile(
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
        : "+v"(x));
    return (unsigned)__builtin_amdgcn_readlane((int)x, 15);
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
        : "+v"(x));
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
  }
}
// STEPS = 6: all 64 lanes; 4: the values sit in lanes 0..15; 3: in lanes 0..7 (what the other lanes hold is
// ignored: row shifts only move values towards higher lanes)
template <int STEPS>
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long v) {
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const unsigned mh = wave_max_u32<STEPS>(hi);
  const unsigned ml = wave_max_u32<STEPS>(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | ml;
}

struct TieOrder {
  int t_mask;   // T - 1
  int t_shift;  // log2(T)
  int rows;     // ceil(N / T)
  __device__ __forceinline__ unsigned rank(int k) const {
    return (unsigned)((k & t_mask) * rows + (k >> t_shift));
  }
  __device__ __forceinline__ int unrank(unsigned r) const {
    return (int)((r % (unsigned)rows) << t_shift) + (int)(r / (unsigned)rows);
  }
};

// fps_bucket.hip: the bucketed form (one workgroup per batch element over a spatially sorted cloud)
bool bucket_applies(int B, int N, int npoint);
size_t bucket_workspace_bytes(int B, int N);  // scratch behind the status word and the cluster kernel's ring
int bucket_launch(const float* xyz, float* temp, int* idx, int B, int N, int npoint, int seed, TieOrder order,
                  void* ws, float* sampled, int cf, hipStream_t s);

}  // namespace ppfps
