// valu_microbench.hip -- measures fp32 VALU issue rates on gfx950 (which roof binds the distance
// kernels): v_fma_f32, v_pk_fma_f32, v_sub_f32 with an SGPR operand, v_min3_f32.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o tools/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                        "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));)
    } else if (KIND == 1) {
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                        "v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
    } else if (KIND == 2) {
      REP8(asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n"
                        "v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));)
    } else if (KIND == 3) {
      REP8(asm volatile("v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_min3_f32 %2, %2, %3, %4\n v_min3_f32 %3, %3, %4, %5\n"
                        "v_min3_f32 %4, %4, %5, %6\n v_min3_f32 %5, %5, %6, %7\n v_min3_f32 %6, %6, %7, %0\n v_min3_f32 %7, %7, %0, %1\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 4) {  // the Chamfer mix: 3 sub(sgpr) 1 mul 2 fmac + min3 per 2
      REP8(asm volatile("v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %8, %5\n v_sub_f32 %2, %8, %6\n v_mul_f32 %0, %0, %0\n"
                        "v_fmac_f32 %0, %1, %1\n v_fmac_f32 %0, %2, %2\n v_min3_f32 %3, %0, %7, %3\n v_sub_f32 %7, %8, %4\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));)
    } else if (KIND == 5) {  // pk_add with sgpr pair? use pk_mul + pk_add mix
      REP8(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %3\n"
                        "v_pk_add_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %5\n v_pk_add_f32 %6, %6, %7\n v_pk_mul_f32 %7, %7, %7\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
    }
  }
  if (KIND == 1 || KIND == 5) { a0 = p0.x + p1.x + p2.y + p3.x + p4.x + p5.y + p6.x + p7.x; }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
void run(const char* name, int waves_per_simd, float lanes_per_inst) {
  const int blocks = 256 * waves_per_simd;  // 4 waves per block -> waves_per_simd per SIMD if spread evenly
  const int iters = 4000;
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<blocks, 256>>>(out, 100, 1.0001f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(out, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double inst = (double)blocks * 4 * iters * 64;  // wave-instructions
  const double laneops = inst * 64 * lanes_per_inst;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.2f Twave-inst/s  %.1f T lane-ops/s  (cycles/inst/SIMD at 2.4GHz: %.2f)\n", name,
         waves_per_simd, best, inst / best * 1e-9, laneops / best * 1e-9, best * 1e-3 * 2.4e9 / (inst / 1024.0));
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", w, 1);
    run<1>("v_pk_fma_f32", w, 2);
    run<2>("v_sub_f32 (sgpr src0)", w, 1);
    run<3>("v_min3_f32", w, 1);
    run<4>("chamfer mix", w, 1);
    run<5>("v_pk_add/mul_f32", w, 2);
  }
  return 0;
}
