// valu_microbench2.hip -- per-opcode VALU issue cost on gfx950, in shader cycles (s_memtime) and
// with the clock the chip holds (s_memtime / s_memrealtime).  Decides how the distance kernels
// are written (DESIGN.md "VALU roof").
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define OPS8(fmt)                                                                             \
  REP8(asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)                   \
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                      "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3)                                   \
                    : "s"(s), "v"(c0), "v"(c1));)
#define POPS8(fmt)                                                                            \
  REP8(asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)                   \
                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), \
                      "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)                                   \
                    : "s"(sp), "v"(c0), "v"(c1));)
// operands: %0..%7 chains, %8..%11 temps, %12 sgpr, %13,%14 constant vgprs
#define F_FMA(i) "v_fma_f32 %" #i ", %" #i ", %13, %14\n"
#define F_FMAC(i) "v_fmac_f32 %" #i ", %13, %14\n"
#define F_MUL(i) "v_mul_f32 %" #i ", %" #i ", %13\n"
#define F_ADD(i) "v_add_f32 %" #i ", %" #i ", %13\n"
#define F_SUBV(i) "v_sub_f32 %" #i ", %13, %" #i "\n"
#define F_SUBS(i) "v_sub_f32 %" #i ", %12, %" #i "\n"
#define F_SUBS2(i) "v_sub_f32 %" #i ", %12, %13\n"
#define F_MIN(i) "v_min_f32 %" #i ", %" #i ", %13\n"
#define F_MIN3(i) "v_min3_f32 %" #i ", %" #i ", %13, %14\n"
#define F_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %13, %14\n"
#define F_CMPSEL(i) "v_cmp_lt_f32 vcc, %13, %" #i "\n v_cndmask_b32 %" #i ", %" #i ", %13, vcc\n"
#define F_MOV(i) "v_mov_b32 %" #i ", %13\n"
#define F_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %13, %14\n"
#define F_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %13\n"
#define F_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %13\n"
#define F_PKADDS(i) "v_pk_add_f32 %" #i ", %12, %13 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float s) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float t0 = 1, t1 = 2, t2 = 3, t3 = 4;
  float c0 = 0.999f + threadIdx.x * 1e-6f, c1 = 0.5f;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  f2 u0 = {1, 2}, u1 = {3, 4}, u2 = {5, 6}, u3 = {7, 8};
  f2 sp = {s, s};
  f2 c0p = {c0, c0}, c1p = {c1, c1};
  long long tA = __builtin_amdgcn_s_memtime();
  long long rA = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { OPS8(F_FMA) }
    else if (KIND == 1) { OPS8(F_FMAC) }
    else if (KIND == 2) { OPS8(F_MUL) }
    else if (KIND == 3) { OPS8(F_ADD) }
    else if (KIND == 4) { OPS8(F_SUBV) }
    else if (KIND == 5) { OPS8(F_SUBS) }
    else if (KIND == 6) { OPS8(F_SUBS2) }
    else if (KIND == 7) { OPS8(F_MIN) }
    else if (KIND == 8) { OPS8(F_MIN3) }
    else if (KIND == 9) { OPS8(F_CMPSEL) }
    else if (KIND == 10) { OPS8(F_MOV) }
    else if (KIND == 11) { OPS8(F_MAX3) }
    else if (KIND == 12) { f2 c0 = c0p, c1 = c1p; POPS8(F_PKFMA) }
    else if (KIND == 13) { f2 c0 = c0p, c1 = c1p; POPS8(F_PKMUL) }
    else if (KIND == 14) { f2 c0 = c0p, c1 = c1p; POPS8(F_PKADD) }
    else if (KIND == 15) { f2 c0 = c0p, c1 = c1p; POPS8(F_PKADDS) }
  }
  long long tB = __builtin_amdgcn_s_memtime();
  long long rB = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    cyc[2 * w] = tB - tA;
    cyc[2 * w + 1] = rB - rA;
  }
  a0 += p0.x + p1.x + p2.y + p3.x + p4.x + p5.y + p6.x + p7.x + u0.x + u1.y + u2.x + u3.y + t0 + t1 + t2 + t3;
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
void run(const char* name, int ninst_per_slot) {
  printf("%-22s", name);
  for (int w : {1, 2, 4, 8}) {
    const int blocks = 256 * w, iters = 2000;
    float* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    (void)hipMalloc(&cyc, sizeof(long long) * blocks * 8);
    k<KIND><<<blocks, 256>>>(out, cyc, 50, 1.0001f);
    k<KIND><<<blocks, 256>>>(out, cyc, iters, 1.0001f);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 8, hipMemcpyDeviceToHost);
    std::vector<double> c, f;
    for (int i = 0; i < blocks * 4; ++i) { c.push_back((double)h[2 * i]); f.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0); }
    std::sort(c.begin(), c.end()); std::sort(f.begin(), f.end());
    const double inst = (double)iters * 64 * ninst_per_slot;
    // cycles per instruction per SIMD = wave cycles / inst / waves-per-SIMD
    printf("  w%d: %5.2f cyc/inst/SIMD @%4.0fMHz", w, c[c.size() / 2] / inst / w, f[f.size() / 2]);
    (void)hipFree(out); (void)hipFree(cyc);
  }
  printf("\n");
}

int main() {
  run<0>("v_fma_f32", 1);
  run<1>("v_fmac_f32", 1);
  run<2>("v_mul_f32", 1);
  run<3>("v_add_f32", 1);
  run<4>("v_sub_f32 v,v", 1);
  run<5>("v_sub_f32 s,v (dst=src)", 1);
  run<6>("v_sub_f32 s,v (dst!=src)", 1);
  run<7>("v_min_f32", 1);
  run<8>("v_min3_f32", 1);
  run<11>("v_max3_f32", 1);
  run<9>("v_cmp_lt+v_cndmask", 2);
  run<10>("v_mov_b32", 1);
  run<12>("v_pk_fma_f32", 1);
  run<13>("v_pk_mul_f32", 1);
  run<14>("v_pk_add_f32", 1);
  run<15>("v_pk_add_f32 s-bcast", 1);
  return 0;
}
