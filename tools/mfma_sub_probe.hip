// mfma_sub_probe.hip -- can the fp32 matrix pipe serve as an EXACT subtractor beside the VALU?
// D = A*B with A[i][0] = r_i, A[i][1] = -1, B[0][j] = 1, B[1][j] = q_j gives
// D[i][j] = fma(-1, q_j, r_i*1) = r_i - q_j (one rounding), 1024 differences per
// v_mfma_f32_32x32x2_f32.  Checks bit-exactness against v_sub_f32 and times three loop bodies:
//   VALU  : 3 sub + mul + 2 fma per pair + min3 per two pairs (the shipped kernel's mix)
//   MFMA3 : 3 MFMA (x,y,z differences) + mul + 2 fma + min3/2 on the accumulators
//   MIXED : waves 0-1 of each 4-wave workgroup run MFMA3, waves 2-3 run VALU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float min3(float a, float b, float c) {
  float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}

// exactness: out[i*32+j] = D[i][j] for 32 refs x 32 queries
__global__ void exact_kernel(const float* r, const float* q, float* out) {
  const int lane = threadIdx.x;
  const float a = lane < 32 ? r[lane] : -1.0f;
  const float b = lane < 32 ? 1.0f : q[lane & 31];
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    out[row * 32 + (lane & 31)] = c[reg];
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ ref, const float* __restrict__ qry, float* out, int ntiles) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool use_mfma = MODE == 1 || (MODE == 2 && wave < 2) || (MODE == 3 && wave < 3);
  float best = 1e30f;
  if (use_mfma) {
    // 32 queries per wave; B operands fixed for the whole scan
    const float qx = qry[(lane & 31) * 3], qy = qry[(lane & 31) * 3 + 1], qz = qry[(lane & 31) * 3 + 2];
    const float bx = lane < 32 ? 1.0f : qx, by = lane < 32 ? 1.0f : qy, bz = lane < 32 ? 1.0f : qz;
    float b0 = 1e30f, b1 = 1e30f;
    for (int t = 0; t < ntiles; ++t) {
      float ax = -1.0f, ay = -1.0f, az = -1.0f;
      if (lane < 32) { const float* p = ref + (size_t)(t * 32 + lane) * 3; ax = p[0]; ay = p[1]; az = p[2]; }
      f16v z = {0};
      f16v dx = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bx, z, 0, 0, 0);
      f16v dy = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, by, z, 0, 0, 0);
      f16v dz = __builtin_amdgcn_mfma_f32_32x32x2f32(az, bz, z, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float d0 = __builtin_fmaf(dz[r], dz[r], __builtin_fmaf(dy[r], dy[r], dx[r] * dx[r]));
        const float d1 = __builtin_fmaf(dz[r + 1], dz[r + 1], __builtin_fmaf(dy[r + 1], dy[r + 1], dx[r + 1] * dx[r + 1]));
        if (r & 2) b1 = min3(d0, d1, b1); else b0 = min3(d0, d1, b0);
      }
    }
    best = fminf(b0, b1);
  } else {
    // 64 queries per wave (one per lane), reference wave-uniform; 16 refs per trip = same pairs per lane as a tile
    const float qx = qry[lane * 3], qy = qry[lane * 3 + 1], qz = qry[lane * 3 + 2];
    float b0 = 1e30f, b1 = 1e30f;
    for (int t = 0; t < ntiles; ++t) {
      const float* rp = ref + (size_t)t * 48;
      float rr[48];
#pragma unroll
      for (int e = 0; e < 48; ++e) rr[e] = rp[e];
#pragma unroll
      for (int p = 0; p < 16; p += 2) {
        const float t0 = rr[3 * p] - qx, t1 = rr[3 * p + 1] - qy, t2 = rr[3 * p + 2] - qz;
        const float d0 = __builtin_fmaf(t2, t2, __builtin_fmaf(t1, t1, t0 * t0));
        const float u0 = rr[3 * p + 3] - qx, u1 = rr[3 * p + 4] - qy, u2 = rr[3 * p + 5] - qz;
        const float d1 = __builtin_fmaf(u2, u2, __builtin_fmaf(u1, u1, u0 * u0));
        if (p & 2) b1 = min3(d0, d1, b1); else b0 = min3(d0, d1, b0);
      }
    }
    best = fminf(b0, b1);
  }
  out[blockIdx.x * 256 + threadIdx.x] = best;
}

template <int MODE>
void run(const char* name, const float* ref, const float* qry, float* out, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, ntiles = 4096;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(ref, qry, out, 64);
  float best = 1e9;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(ref, qry, out, ntiles);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  // pairs: each wave-tile = 1024 pairs in both forms (MFMA: 32x32; VALU: 64 lanes x 16 refs)
  const double pairs = (double)blocks * 4 * ntiles * 1024;
  printf("%-8s waves/SIMD=%d  %.3f ms  %.2f Tpairs/s\n", name, waves_per_simd, best, pairs / best / 1e9);
}

int main() {
  // exactness on adversarial-ish random data (wide exponent range, denormal-producing differences)
  const int T = 2000;
  std::vector<float> r(32 * T), q(32 * T);
  srand(7);
  auto rnd = [] { float m = (float)rand() / RAND_MAX * 2 - 1; int e = rand() % 60 - 40; return ldexpf(m, e); };
  for (auto& v : r) v = rnd();
  for (auto& v : q) v = (rand() % 8 == 0) ? r[rand() % r.size()] * (1 + ((rand() % 3) - 1) * 1.1920929e-7f) : rnd();
  float *dr, *dq, *dout; (void)hipMalloc(&dr, r.size() * 4); (void)hipMalloc(&dq, q.size() * 4); (void)hipMalloc(&dout, 1024 * 4);
  (void)hipMemcpy(dr, r.data(), r.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice);
  long bad = 0; std::vector<float> o(1024);
  for (int t = 0; t < T; ++t) {
    exact_kernel<<<1, 64>>>(dr + t * 32, dq + t * 32, dout);
    (void)hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      const float e = r[t * 32 + i] - q[t * 32 + j];
      if (memcmp(&e, &o[i * 32 + j], 4) != 0) { if (bad < 5) printf("MISMATCH r=%a q=%a mfma=%a valu=%a\n", r[t*32+i], q[t*32+j], o[i*32+j], e); ++bad; }
    }
  }
  printf("exactness: %ld mismatches of %d differences\n", bad, T * 1024);
  float *ref, *qry, *out; (void)hipMalloc(&ref, 4096 * 48 * 4 + 1024); (void)hipMalloc(&qry, 64 * 3 * 4); (void)hipMalloc(&out, 2048 * 256 * 4);
  std::vector<float> h(4096 * 48 + 256); for (auto& v : h) v = (float)rand() / RAND_MAX; 
  (void)hipMemcpy(ref, h.data(), h.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(qry, h.data(), 64 * 3 * 4, hipMemcpyHostToDevice);
  for (int w : {2, 4, 8}) {
    run<0>("VALU", ref, qry, out, w);
    run<1>("MFMA3", ref, qry, out, w);
    run<2>("MIXED2", ref, qry, out, w);
    run<3>("MIXED3", ref, qry, out, w);
  }
  return 0;
}
