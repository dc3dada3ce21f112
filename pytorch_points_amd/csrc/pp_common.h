// pp_common.h -- shared device helpers for libpp_hip.so (gfx950 only).
//
// Canonical fp32 arithmetic (DESIGN.md "Arithmetic contract"): the library is compiled with
// -ffp-contract=off and every fused multiply-add is an explicit __builtin_fmaf, so the rounding
// sequence is fixed by the source, not by the optimiser:
//   distc (Chamfer, ref _ext/nmdistance_cuda.cu:31-35):  t_c = ref_c - query_c;
//         d = t_0*t_0; d = fma(t_1,t_1,d); d = fma(t_2,t_2,d) ...
//   dist3 (FPS / ball_query / three_nn, ref _ext/sampling_cuda.cu:202,364, interpolate_gpu.cu:36):
//         fma(dz,dz, fma(dx,dx, dy*dy))
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>

#include "pp_hip.h"
#include "pp_hip_debug.h"

#define PP_WAVE 64

#define PP_RETURN_IF_LAUNCH_FAILED()        \
  do {                                      \
    hipError_t e__ = hipGetLastError();     \
    if (e__ != hipSuccess) return (int)e__; \
  } while (0)

namespace pp {

// Process-wide state of the library is limited to two kinds of atomics: the debug knobs declared in
// include/pp_hip_debug.h (tests and benchmarks select kernel variants with them; product code never sets one)
// and the "this kernel's LDS limit has been raised on device d" flags below (idempotent).
struct Knob {
  std::atomic<int> v{0};
  operator int() const { return v.load(std::memory_order_relaxed); }
  void set(int x) { v.store(x, std::memory_order_relaxed); }
};
struct DeviceFlags {
  std::atomic<bool> f[64] = {};
};

// Raise a kernel's dynamic-LDS limit (needed above 64 KiB).  Once per device per kernel: `flags`
// is a zero-initialised static owned by the call site.
template <typename K>
inline hipError_t allow_big_lds(K kernel, int bytes, DeviceFlags& flags) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) dev = 63;
  if (dev != 63 && flags.f[dev].load(std::memory_order_acquire)) return hipSuccess;
  e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) flags.f[dev].store(true, std::memory_order_release);
  return e;
}

__device__ __forceinline__ float chamfer_d3(float rx, float ry, float rz, float qx, float qy,
                                            float qz) {
  const float t0 = rx - qx, t1 = ry - qy, t2 = rz - qz;
  return __builtin_fmaf(t2, t2, __builtin_fmaf(t1, t1, t0 * t0));
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

// two queries against one wave-uniform reference point, element-wise the same operations as
// chamfer_d3 (v_pk_add_f32 with the SGPR broadcast, v_pk_mul_f32, v_pk_fma_f32)
__device__ __forceinline__ f2 chamfer_d3_pk(float rx, float ry, float rz, f2 qx, f2 qy, f2 qz) {
  const f2 t0 = (f2)(rx) - qx, t1 = (f2)(ry) - qy, t2 = (f2)(rz) - qz;
  return __builtin_elementwise_fma(t2, t2, __builtin_elementwise_fma(t1, t1, t0 * t0));
}

__device__ __forceinline__ float dist3(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
}

// two points b against one point a, element-wise the operations of dist3 (v_pk_add_f32 with a negated operand,
// v_pk_mul_f32, v_pk_fma_f32: IEEE results, the same bits as the scalar form, at twice the rate)
__device__ __forceinline__ f2 dist3_pk(float ax, float ay, float az, f2 bx, f2 by, f2 bz) {
  const f2 dx = (f2)(ax) - bx, dy = (f2)(ay) - by, dz = (f2)(az) - bz;
  return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
}

// In-place ascending sort of a[0..n) by key(a[i]) by ONE lane: insertion sort for the short lists the callers
// normally see (a handful of entries: its cost is the number of inversions), heapsort beyond -- O(n log n) whatever
// the arrival order, so that a degenerate input (thousands of sources sharing one destination) costs milliseconds,
// not the 10^8 steps of an insertion sort.  swap(i, j) exchanges entries i and j (and whatever travels
// with them); keys are distinct in both callers (source positions), so the order is total.
template <typename KEY, typename SWAP>
__device__ __forceinline__ void lane_sort(unsigned n, KEY key, SWAP swap) {
  if (n < 2) return;
  if (n <= 24) {
    for (unsigned i = 1; i < n; ++i)
      for (unsigned j = i; j > 0 && key(j - 1) > key(j); --j) swap(j - 1, j);
    return;
  }
  auto sift = [&](unsigned root, unsigned end) {  // max-heap on [0, end)
    for (;;) {
      unsigned child = 2 * root + 1;
      if (child >= end) return;
      if (child + 1 < end && key(child + 1) > key(child)) ++child;
      if (!(key(child) > key(root))) return;
      swap(root, child);
      root = child;
    }
  };
  for (unsigned i = n / 2; i-- > 0;) sift(i, n);
  for (unsigned end = n - 1; end > 0; --end) {
    swap(0, end);
    sift(0, end);
  }
}

// v_min3_f32: one VALU op for two candidates.  Operands are results of fma/mul chains (never
// signalling NaNs), so no canonicalisation is needed; spelled in asm so that the compiler cannot
// split it back into two v_min_f32 or insert v_max canonicalisations.
__device__ __forceinline__ float min3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Wave-wide max / sum of a float through DPP (row shifts, then row broadcasts; lane 63 ends up with the
// result, which is read back as a wave-uniform value).  VALU-rate, unlike the ds_bpermute behind __shfl_xor.
// Every lane must be active.
template <bool SUM>
__device__ __forceinline__ float wave_reduce_dpp(float v) {
  auto step = [](float x, auto ctrl, auto row_mask) {
    // bound_ctrl = false, old = x: lanes without a source (and rows masked out) combine x with itself for
    // max, and must add nothing for a sum -- hence old = 0 there
    const float o = __int_as_float(__builtin_amdgcn_update_dpp(SUM ? 0 : __float_as_int(x), __float_as_int(x),
                                                               decltype(ctrl)::value, decltype(row_mask)::value,
                                                               0xf, false));
    return SUM ? x + o : fmaxf(x, o);
  };
  using I = std::integral_constant<int, 0>;
  (void)sizeof(I);
  v = step(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});  // row_shr:1
  v = step(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});  // row_shr:2
  v = step(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});  // row_shr:4
  v = step(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});  // row_shr:8
  v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15 -> rows 1, 3
  v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});  // row_bcast:31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Six wave reductions at once (max or sum), in place: `op v, v, v row_shr:k` leaves lanes without a source
// untouched (bound_ctrl off), so after row_shr 1, 2, 4, 8 lane 15 of every row of 16 holds the row's result
// (STEPS = 4), and after row_bcast 15 / 31 lane 63 holds the wave's (STEPS = 6).  The six chains are
// interleaved: dependent DPP operations sit six instructions apart (a VGPR written by a VALU instruction
// needs two wait states before a DPP read), and the block opens and closes with a nop for its neighbours.
// Half the instructions of update_dpp + op, which the compiler does not fuse.  Every lane must be active.
#define PP_DPP6_STEP(OP, CTRL)                                                                       \
  OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\t" OP " %2, %2, %2 " CTRL "\n\t" OP     \
     " %3, %3, %3 " CTRL "\n\t" OP " %4, %4, %4 " CTRL "\n\t" OP " %5, %5, %5 " CTRL "\n\t"
#define PP_DPP6_ROWS(OP)                                                                             \
  "s_nop 1\n\t" PP_DPP6_STEP(OP, "row_shr:1 row_mask:0xf bank_mask:0xf")                            \
      PP_DPP6_STEP(OP, "row_shr:2 row_mask:0xf bank_mask:0xf")                                       \
          PP_DPP6_STEP(OP, "row_shr:4 row_mask:0xf bank_mask:0xf")                                   \
              PP_DPP6_STEP(OP, "row_shr:8 row_mask:0xf bank_mask:0xf")
#define PP_DPP6_WAVE(OP)                                                                             \
  PP_DPP6_ROWS(OP) PP_DPP6_STEP(OP, "row_bcast:15 row_mask:0xa bank_mask:0xf")                       \
      PP_DPP6_STEP(OP, "row_bcast:31 row_mask:0xc bank_mask:0xf")
template <bool SUM, int STEPS>
__device__ __forceinline__ void wave_reduce6_dpp(float (&v)[6]) {
  static_assert(STEPS == 4 || STEPS == 6, "");
  if constexpr (SUM && STEPS == 6)
    asm volatile(PP_DPP6_WAVE("v_add_f32_dpp") "s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
  else if constexpr (SUM)
    asm volatile(PP_DPP6_ROWS("v_add_f32_dpp") "s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
  else if constexpr (STEPS == 6)
    asm volatile(PP_DPP6_WAVE("v_max_f32_dpp") "s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
  else
    asm volatile(PP_DPP6_ROWS("v_max_f32_dpp") "s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
}

// Wave-wide minimum of an unsigned word through DPP (VALU-rate; __shfl_xor is six dependent ds_bpermute round trips):
// the result as a wave-uniform value.  Every lane must be active.
__device__ __forceinline__ unsigned wave_min_u32_dpp(unsigned v) {
#define PP_MIN_STEP(CTRL, ROWS)                                                                  \
  {                                                                                               \
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWS, 0xf, false); \
    v = o < v ? o : v;                                                                            \
  }
  PP_MIN_STEP(0x111, 0xf)  // row_shr:1
  PP_MIN_STEP(0x112, 0xf)  // row_shr:2
  PP_MIN_STEP(0x114, 0xf)  // row_shr:4
  PP_MIN_STEP(0x118, 0xf)  // row_shr:8
  PP_MIN_STEP(0x142, 0xa)  // row_bcast:15 -> rows 1, 3
  PP_MIN_STEP(0x143, 0xc)  // row_bcast:31 -> rows 2, 3
#undef PP_MIN_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// ... of 64-bit keys (high word first; the low words only decide among the lanes that hold the smallest high word)
__device__ __forceinline__ unsigned long long wave_min_u64_dpp(unsigned long long key) {
  const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
  const unsigned mh = wave_min_u32_dpp(hi);
  const unsigned ml = wave_min_u32_dpp(hi == mh ? lo : 0xffffffffu);
  return ((unsigned long long)mh << 32) | ml;
}

__device__ __forceinline__ int wave_id_uniform() {
  return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

// Blocks b and b+8 share an XCD (its private 4 MiB L2) under the observed round-robin dispatch;
// remap so that each XCD walks a contiguous range of virtual block ids.  Speed only: any
// placement gives the same results.  per_xcd = ceil(total / 8); grid = 8 * per_xcd.
__device__ __forceinline__ int xcd_virtual_block(int linear, int per_xcd) {
  return (linear & 7) * per_xcd + (linear >> 3);
}

}  // namespace pp
