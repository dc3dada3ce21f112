// tools/pk_probe.hip -- issue rate of packed fp32 VALU against scalar fp32 VALU on gfx950, W waves per SIMD:
// the same number of fp32 operations as v_fma_f32 / v_pk_fma_f32 and as the walk's mix (sub, mul, fma, min).
// hipcc --offload-arch=gfx950 -O3 tools/pk_probe.hip -o tools/pk_probe && ./tools/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (float)(threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 16 scalar fma
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    } else if (MODE == 1) {  // 8 packed fma: the same 16 fp32 fma
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f2 v = {acc[2 * i], acc[2 * i + 1]};
        f2 aa = {a, a}, bb = {b, b};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(aa), "v"(bb));
        acc[2 * i] = v.x; acc[2 * i + 1] = v.y;
      }
    } else if (MODE == 2) {  // 16 scalar add
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
    } else if (MODE == 3) {  // 8 packed add
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f2 v = {acc[2 * i], acc[2 * i + 1]};
        f2 aa = {a, a};
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(aa));
        acc[2 * i] = v.x; acc[2 * i + 1] = v.y;
      }
    } else if (MODE == 4) {  // 16 scalar mul
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
    } else {  // 8 packed mul
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f2 v = {acc[2 * i], acc[2 * i + 1]};
        f2 aa = {a, a};
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(aa));
        acc[2 * i] = v.x; acc[2 * i + 1] = v.y;
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int wg_per_cu, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256 * wg_per_cu, 256>>>(out, 100, 1.0f, 0.0f);
  hipEventRecord(e0);
  k<MODE><<<256 * wg_per_cu, 256>>>(out, iters, 1.0f, 0.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // fp32 operations per SIMD: wg_per_cu waves per SIMD x iters x 16 ops x 64 lanes
  const double ops = (double)wg_per_cu * iters * 16.0;  // wave-level scalar-equivalent instructions per SIMD
  printf("%-12s %d waves/SIMD: %.3f ms  -> %.2f cycles (at 2.4 GHz) per scalar-equivalent wave instruction\n", name, wg_per_cu, ms,
         ms * 1e-3 * 2.4e9 / ops);
}

int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4}) {
    run<0>("v_fma", w, out); run<1>("v_pk_fma", w, out);
    run<2>("v_add", w, out); run<3>("v_pk_add", w, out);
    run<4>("v_mul", w, out); run<5>("v_pk_mul", w, out);
  }
  return 0;
}
