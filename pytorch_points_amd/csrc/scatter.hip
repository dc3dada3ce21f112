// scatter.hip -- segmented scatter-add for the three backward passes of the path that the
// reference implements with one global fp32 atomic per element:
//   gather_points_grad      (_ext/sampling_cuda.cu:47-64)      grad_points[b,c,idx[b,m]]      += grad_out[b,c,m]
//   group_points_grad       (_ext/sampling_cuda.cu:482-503)    grad_points[b,c,idx[b,j,k]]    += grad_out[b,c,j,k]
//   three_interpolate_grad  (_ext/interpolate_gpu.cu:120-142)  grad_points[b,c,idx[b,n,k]]    += grad_out[b,c,n]*w[b,n,k]
// All three have the same shape: per batch element a list of P (source, destination[, weight])
// triples that is THE SAME FOR EVERY CHANNEL.  On gfx950 neither atomic route is fast: scattered
// global fp32 atomics run at ~20 G/s and ds_add_f32 at 0.36 lanes/clk/CU (tools/lds_atomic_probe.hip).
// So the triples are sorted once per call by (source chunk, destination) with integer LDS atomics
// (ssa_build_kernel) and every (batch, channel) then needs no atomics at all (ssa_apply_kernel):
// it stages 64 KiB of its source row in LDS, and thread t -- which exclusively owns destinations
// [t*npt, (t+1)*npt) -- walks its contiguous slice of the sorted triples, reading the staged value
// and accumulating into its own LDS slots with plain read-modify-writes.  The summation order
// within a destination is unspecified, as it is with the reference's atomics -- unless the ORDERED form
// is asked for (pp_*_ordered_f32, selected by torch.use_deterministic_algorithms): the build kernel then
// also sorts the triples of every (chunk, destination) group by source position, so that a destination
// receives its terms in ascending source order.  That is the order in which a sequential loop over the
// reference's launch (the CPU oracle) adds them: results are reproducible bit for bit from run to run AND
// equal to the oracle's.
#include "pp_common.h"

namespace {

constexpr int kSsaThreads = 1024;
constexpr int kSsaMaxChunk = 16384;  // source elements staged per pass (chosen per problem so that
                                     // values + triples + accumulators fit the LDS)
constexpr int kSsaSrcBits = 14;      // log2(kSsaMaxChunk)
constexpr int kSsaMaxDst = 20480;    // destinations per batch element: accumulators live in LDS
// a sorted triple: destination << 16 | source in chunk << 2 | triple of that source (R <= 3): the low 16
// bits order the triples of one destination by source position
constexpr int kSsaDstShift = 16;
__device__ __forceinline__ unsigned ssa_src(unsigned e) { return (e >> 2) & ((1u << kSsaSrcBits) - 1u); }

struct SsaLayout {
  size_t entries, weights, offsets, total;
};
__host__ __device__ inline SsaLayout ssa_layout(int B, long long P, int nchunks, bool weighted) {
  SsaLayout L;
  L.entries = 0;
  L.weights = L.entries + 4 * (size_t)B * P;
  L.offsets = L.weights + (weighted ? 4 * (size_t)B * P : 0);
  L.total = L.offsets + 4 * (size_t)B * nchunks * (kSsaThreads + 1);
  return L;
}

// One workgroup per (batch, source chunk): counting sort of the chunk's triples by destination.
// R triples per source element (R = 3 for three_interpolate, else 1): triple p belongs to source p / R.
__global__ __launch_bounds__(kSsaThreads) void ssa_build_kernel(const int* __restrict__ dst,
                                                                const float* __restrict__ weight,
                                                                unsigned char* __restrict__ ws, int B,
                                                                long long P, int R, int Nd, int nchunks,
                                                                int S, int ordered) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_cnt[];  // [Nd + Nd/32] skewed
  __shared__ unsigned s_wave[16];
  const int b = blockIdx.x / nchunks, q = blockIdx.x - b * nchunks;
  const SsaLayout L = ssa_layout(B, P, nchunks, weight != nullptr);
  const long long p0 = (long long)q * S * R;
  const long long p1 = min(P, p0 + (long long)S * R);
  const int* __restrict__ d = dst + (size_t)b * P;
  unsigned* entries = reinterpret_cast<unsigned*>(ws + L.entries) + (size_t)b * P + p0;
  float* wout = weight ? reinterpret_cast<float*>(ws + L.weights) + (size_t)b * P + p0 : nullptr;
  const float* __restrict__ win = weight ? weight + (size_t)b * P : nullptr;
  unsigned* offs = reinterpret_cast<unsigned*>(ws + L.offsets) + ((size_t)b * nchunks + q) * (kSsaThreads + 1);
  const int t = threadIdx.x;
  auto sk = [](int c) { return c + (c >> 5); };
  for (int c = t; c < Nd; c += kSsaThreads) s_cnt[sk(c)] = 0;
  __syncthreads();
  for (long long p = p0 + t; p < p1; p += kSsaThreads) atomicAdd(&s_cnt[sk(d[p])], 1u);
  __syncthreads();
  const int npt = (Nd + kSsaThreads - 1) / kSsaThreads;  // destinations owned by one thread
  const int c0 = min(t * npt, Nd), c1 = min(c0 + npt, Nd);
  unsigned sum = 0;
  for (int c = c0; c < c1; ++c) sum += s_cnt[sk(c)];
  unsigned incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned o = __shfl_up(incl, off);
    if ((t & 63) >= off) incl += o;
  }
  if ((t & 63) == 63) s_wave[t >> 6] = incl;
  __syncthreads();
  unsigned run = incl - sum;
  for (int w = 0; w < (t >> 6); ++w) run += s_wave[w];
  const unsigned first = run;
  offs[t] = run;  // first triple of thread t's destinations in this chunk
  if (t == kSsaThreads - 1) offs[kSsaThreads] = (unsigned)(p1 - p0);
  for (int c = c0; c < c1; ++c) {
    const unsigned v = s_cnt[sk(c)];
    s_cnt[sk(c)] = run;  // scatter cursor
    run += v;
  }
  __syncthreads();
  for (long long p = p0 + t; p < p1; p += kSsaThreads) {
    const int dd = d[p];
    const unsigned pos = atomicAdd(&s_cnt[sk(dd)], 1u);
    const unsigned rel = (unsigned)(p - p0);
    const unsigned src_local = rel / (unsigned)R;
    entries[pos] = ((unsigned)dd << kSsaDstShift) | (src_local << 2) | (rel - src_local * (unsigned)R);
    if (win) wout[pos] = win[p];
  }
  if (!ordered) return;
  // ORDERED: the cursors handed the positions inside a (chunk, destination) group out in arrival order;
  // the owner of the destination sorts the group by source position (pp::lane_sort: insertion sort for the usual
  // short groups, heapsort for long ones)
  __threadfence_block();
  __syncthreads();
  unsigned g0 = first;
  for (int c = c0; c < c1; ++c) {
    const unsigned g1 = s_cnt[sk(c)];  // the cursor now stands at the group's end
    unsigned* grp = entries + g0;
    float* gw = wout ? wout + g0 : nullptr;
    pp::lane_sort(
        g1 - g0, [&](unsigned i) { return grp[i] & 0xffffu; },
        [&](unsigned i, unsigned j) {
          const unsigned e = grp[i];
          grp[i] = grp[j];
          grp[j] = e;
          if (gw) {
            const float w = gw[i];
            gw[i] = gw[j];
            gw[j] = w;
          }
        });
    g0 = g1;
  }
}

// One workgroup per (batch, channel).  Per source chunk the values AND the chunk's sorted triples
// (and weights) are staged in LDS with coalesced loads; a thread then walks its own slice of the
// triples out of LDS, four at a time (reads first, then the read-modify-writes in order).
template <bool WEIGHTED>
__global__ __launch_bounds__(kSsaThreads) void ssa_apply_kernel(const float* __restrict__ src,
                                                                const unsigned char* __restrict__ ws,
                                                                float* __restrict__ out, int B, int C,
                                                                long long P, int R, long long Ps, int Nd,
                                                                int nchunks, int S, long long src_bstride) {
  extern __shared__ __attribute__((aligned(16))) float s_f[];  // s_val[S] | s_ent[S*R] | s_w[S*R]? | s_acc[Nd]
  float* s_val = s_f;
  unsigned* s_ent = reinterpret_cast<unsigned*>(s_f + S);
  float* s_w = s_f + S + (size_t)S * R;
  float* s_acc = s_f + S + (size_t)S * R * (WEIGHTED ? 2 : 1);
  const int x = blockIdx.x & 7, y = blockIdx.x >> 3;  // channels of one batch element share an XCD
  const int b = x + 8 * (y / C);
  const int c = y % C;
  if (b >= B) return;
  const SsaLayout L = ssa_layout(B, P, nchunks, WEIGHTED);
  const unsigned* __restrict__ entries = reinterpret_cast<const unsigned*>(ws + L.entries) + (size_t)b * P;
  const float* __restrict__ wts = WEIGHTED ? reinterpret_cast<const float*>(ws + L.weights) + (size_t)b * P : nullptr;
  const unsigned* __restrict__ offs = reinterpret_cast<const unsigned*>(ws + L.offsets) + (size_t)b * nchunks * (kSsaThreads + 1);
  const float* __restrict__ row = src + (size_t)b * src_bstride + (size_t)c * Ps;
  const int t = threadIdx.x;
  for (int k = t; k < Nd; k += kSsaThreads) s_acc[k] = 0.0f;
  // The next chunk's values / triples / weights travel through registers while the current chunk is
  // consumed (loads unconditional, indices clamped, so that all of them are in flight together):
  // KV = S / 1024 values and KV * R triples per thread.
  constexpr int KVMAX = kSsaMaxChunk / kSsaThreads;  // 16
  const int kv = S / kSsaThreads;
  float rv[KVMAX];
  unsigned re[KVMAX];   // R == 1 fast path keeps triples in registers too; R > 1 stages them directly
  auto fetch = [&](int q) {
    const long long s0 = (long long)q * S;
    const long long last = Ps - 1;
#pragma unroll
    for (int u = 0; u < KVMAX; ++u)
      if (u < kv) {
        const long long sidx = s0 + t + (long long)kSsaThreads * u;
        rv[u] = row[sidx < Ps ? sidx : last];
        if (R == 1) re[u] = entries[sidx < Ps ? sidx : last];
      }
  };
  fetch(0);
  for (int q = 0; q < nchunks; ++q) {
    const long long s0 = (long long)q * S;
    const int ns = (int)min((long long)S, Ps - s0);
    const size_t base = (size_t)s0 * R;
    const int ne = ns * R;
    __syncthreads();  // previous chunk fully consumed (and s_acc zeroed, first time)
#pragma unroll
    for (int u = 0; u < KVMAX; ++u)
      if (u < kv) {
        s_val[t + kSsaThreads * u] = rv[u];
        if (R == 1) s_ent[t + kSsaThreads * u] = re[u];
      }
    if (R != 1)
      for (int e = t; e < ne; e += kSsaThreads) s_ent[e] = entries[base + e];
    if (WEIGHTED)
      for (int e = t; e < ne; e += kSsaThreads) s_w[e] = wts[base + e];
    const unsigned* __restrict__ of = offs + (size_t)q * (kSsaThreads + 1);
    unsigned e = of[t];
    const unsigned e1 = of[t + 1];
    __syncthreads();
    if (q + 1 < nchunks) fetch(q + 1);
    for (; e + 4 <= e1; e += 4) {
      unsigned ent[4];
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ent[u] = s_ent[e + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = s_val[ssa_src(ent[u])];
        if (WEIGHTED) v[u] *= s_w[e + u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) s_acc[ent[u] >> kSsaDstShift] += v[u];  // this thread is their only writer
    }
    for (; e < e1; ++e) {
      const unsigned en = s_ent[e];
      float v = s_val[ssa_src(en)];
      if (WEIGHTED) v *= s_w[e];
      s_acc[en >> kSsaDstShift] += v;
    }
  }
  __syncthreads();
  float* __restrict__ o = out + ((size_t)b * C + c) * Nd;
  for (int k = t; k < Nd; k += kSsaThreads) o[k] += s_acc[k];  // accumulate: the ABI's contract
}

}  // namespace

namespace pp {

// source elements per chunk: the largest power of two <= 16384 with values + triples (+ weights) +
// accumulators inside ~150 KiB of LDS; 0 if even 1024 does not fit
static int ssa_chunk(int R, int Nd, bool weighted) {
  for (int S = kSsaMaxChunk; S >= 1024; S >>= 1) {
    const size_t bytes = 4 * ((size_t)S + (size_t)S * R * (weighted ? 2 : 1) + (size_t)Nd);
    if (bytes <= 150 * 1024) return S;
  }
  return 0;
}

// bytes of scratch for a segmented scatter-add of P triples per batch element (0: not applicable)
size_t ssa_workspace_bytes(int B, long long P, int R, int Nd, bool weighted) {
  if (B <= 0 || P <= 0 || Nd <= 0 || Nd > kSsaMaxDst || P % R != 0 || R > 3) return 0;
  const int S = ssa_chunk(R, Nd, weighted);
  if (S == 0) return 0;
  const long long Ps = P / R;
  const long long nchunks = (Ps + S - 1) / S;
  if (nchunks > 8192 || (long long)B * nchunks > 0x7fffffffLL) return 0;
  return ssa_layout(B, P, (int)nchunks, weighted).total;
}

// out[b,c,dst[b,p]] += (weight ? weight[b,p] : 1) * src[b*src_bstride + c*(P/R) + p/R]
int ssa_run(const float* src, const int* dst, const float* weight, float* out, int B, int C,
            long long P, int R, int Nd, long long src_bstride, void* workspace, hipStream_t s, bool ordered) {
  const long long Ps = P / R;
  const int S = ssa_chunk(R, Nd, weight != nullptr);
  const int nchunks = (int)((Ps + S - 1) / S);
  static pp::DeviceFlags ok_build, ok_a, ok_b;
  hipError_t e = allow_big_lds(ssa_build_kernel, 152 * 1024, ok_build);
  if (e != hipSuccess) return (int)e;
  e = allow_big_lds(ssa_apply_kernel<false>, 160 * 1024, ok_a);
  if (e != hipSuccess) return (int)e;
  e = allow_big_lds(ssa_apply_kernel<true>, 160 * 1024, ok_b);
  if (e != hipSuccess) return (int)e;
  unsigned char* ws = (unsigned char*)workspace;
  ssa_build_kernel<<<dim3((unsigned)(B * nchunks)), dim3(kSsaThreads), (size_t)(Nd + Nd / 32 + 1) * sizeof(unsigned), s>>>(
      dst, weight, ws, B, P, R, Nd, nchunks, S, ordered ? 1 : 0);
  PP_RETURN_IF_LAUNCH_FAILED();
  const size_t lds = 4 * ((size_t)S + (size_t)S * R * (weight ? 2 : 1) + (size_t)Nd);
  const unsigned blocks = (unsigned)(8 * ((B + 7) / 8) * C);
  if (weight)
    ssa_apply_kernel<true><<<dim3(blocks), dim3(kSsaThreads), lds, s>>>(src, ws, out, B, C, P, R, Ps, Nd, nchunks, S, src_bstride);
  else
    ssa_apply_kernel<false><<<dim3(blocks), dim3(kSsaThreads), lds, s>>>(src, ws, out, B, C, P, R, Ps, Nd, nchunks, S, src_bstride);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

}  // namespace pp
