// shard.hip -- packing of a rank's Chamfer outputs for the one-collective-per-step exchange of the
// batch-sharded operator (pytorch_points_amd/sharded.py, DESIGN.md "Multi-GPU"), and unpacking of the
// gathered buffer.  Layout of one rank's packed bytes (n1 = B_local*N, n2 = B_local*M):
//     [ dist1: n1 f32 | dist2: n2 f32 | idx1: n1 (u16 | i32) | idx2: n2 (u16 | i32) ]
// Indices travel as 16-bit words when every index fits (compact != 0): 6 instead of 8 bytes per
// point pair over xGMI.  The word 0xFFFF stands for -1 (labeled Chamfer's "no partner with this label"),
// so the compact form needs every real index <= 65534, i.e. N, M <= 65535 -- the caller's condition.
// One launch each way instead of a dozen small tensor ops.
#include "pp_common.h"

namespace {

__global__ __launch_bounds__(256) void shard_pack_kernel(const float* __restrict__ d1, const float* __restrict__ d2,
                                                         const int* __restrict__ i1, const int* __restrict__ i2,
                                                         unsigned char* __restrict__ out, long long n1, long long n2,
                                                         int compact) {
  float* od1 = reinterpret_cast<float*>(out);
  float* od2 = od1 + n1;
  unsigned char* oi = reinterpret_cast<unsigned char*>(od2 + n2);
  const long long stride = (long long)gridDim.x * 256;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n1 + n2; k += stride) {
    const bool first = k < n1;
    const long long j = first ? k : k - n1;
    if (d1) (first ? od1 : od2)[j] = (first ? d1 : d2)[j];  // (nullptr: the search wrote the distances in place)
    const int v = (first ? i1 : i2)[j];
    if (compact)
      reinterpret_cast<unsigned short*>(oi)[k] = (unsigned short)v;  // k: idx1 then idx2, contiguous
    else
      reinterpret_cast<int*>(oi)[k] = v;
  }
}

// in: world rows of `stride` bytes, each packed as above; outputs are the global-batch tensors in
// rank order: D1/I1 (world*n1), D2/I2 (world*n2)
__global__ __launch_bounds__(256) void shard_unpack_kernel(const unsigned char* __restrict__ in, int world,
                                                           long long stride_bytes, long long n1, long long n2,
                                                           int compact, float* __restrict__ D1,
                                                           float* __restrict__ D2, int* __restrict__ I1,
                                                           int* __restrict__ I2) {
  const long long per = n1 + n2;
  const long long total = per * world;
  const long long step = (long long)gridDim.x * 256;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += step) {
    const int r = (int)(e / per);
    const long long k = e - (long long)r * per;
    const unsigned char* row = in + (long long)r * stride_bytes;
    const float* rd = reinterpret_cast<const float*>(row);
    const unsigned char* ri = row + 4 * per;
    const bool first = k < n1;
    const long long j = first ? k : k - n1;
    int v;
    if (compact) {
      const unsigned short w16 = reinterpret_cast<const unsigned short*>(ri)[k];
      v = w16 == 0xFFFFu ? -1 : (int)w16;
    } else {
      v = reinterpret_cast<const int*>(ri)[k];
    }
    if (first) {
      if (D1) D1[(long long)r * n1 + j] = rd[k];  // (nullptr: the caller reads the distances where they were gathered)
      I1[(long long)r * n1 + j] = v;
    } else {
      if (D1) D2[(long long)r * n2 + j] = rd[k];
      I2[(long long)r * n2 + j] = v;
    }
  }
}

}  // namespace

extern "C" size_t pp_shard_packed_bytes(long long n1, long long n2, int compact) {
  if (n1 < 0 || n2 < 0) return 0;
  const size_t raw = (size_t)(n1 + n2) * (compact ? 6 : 8);
  return (raw + 15) / 16 * 16;
}

extern "C" int pp_shard_pack_f32(const float* dist1, const float* dist2, const int* idx1, const int* idx2,
                                 void* packed, long long n1, long long n2, int compact, void* stream) {
  if (n1 < 0 || n2 < 0) return PP_EINVAL;
  if (n1 + n2 == 0) return PP_OK;
  if (!packed || (n1 > 0 && !idx1) || (n2 > 0 && !idx2)) return PP_EINVAL;
  // dist1 == dist2 == NULL: the distances are in place already (the caller handed the packed buffer's own first
  // n1 + n2 floats to the search as its dist1 / dist2 outputs); only the indices are packed behind them
  const bool in_place = dist1 == nullptr && dist2 == nullptr;
  if (!in_place && ((n1 > 0 && !dist1) || (n2 > 0 && !dist2))) return PP_EINVAL;
  const long long blocks = (n1 + n2 + 256 * 4 - 1) / (256 * 4);
  shard_pack_kernel<<<dim3((unsigned)(blocks > 65535 * 16 ? 65535 * 16 : blocks)), dim3(256), 0,
                      (hipStream_t)stream>>>(in_place ? nullptr : dist1, dist2, idx1, idx2, (unsigned char*)packed, n1, n2,
                                             compact);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}

extern "C" int pp_shard_unpack_f32(const void* gathered, int world, long long stride_bytes, long long n1,
                                   long long n2, int compact, float* dist1, float* dist2, int* idx1, int* idx2,
                                   void* stream) {
  if (world < 0 || n1 < 0 || n2 < 0 || stride_bytes < (long long)pp_shard_packed_bytes(n1, n2, compact))
    return PP_EINVAL;
  if (world == 0 || n1 + n2 == 0) return PP_OK;
  // dist1 == dist2 == NULL: indices only (the distances are read in the gathered buffer itself, as strided views)
  const bool idx_only = dist1 == nullptr && dist2 == nullptr;
  if (!gathered || (n1 > 0 && !idx1) || (n2 > 0 && !idx2)) return PP_EINVAL;
  if (!idx_only && ((n1 > 0 && !dist1) || (n2 > 0 && !dist2))) return PP_EINVAL;
  const long long total = (n1 + n2) * world;
  const long long blocks = (total + 256 * 4 - 1) / (256 * 4);
  shard_unpack_kernel<<<dim3((unsigned)(blocks > 65535 * 16 ? 65535 * 16 : blocks)), dim3(256), 0,
                        (hipStream_t)stream>>>((const unsigned char*)gathered, world, stride_bytes, n1, n2, compact,
                                               idx_only ? nullptr : dist1, dist2, idx1, idx2);
  PP_RETURN_IF_LAUNCH_FAILED();
  return PP_OK;
}
