// cuda_shim.h -- CPU emulation of the CUDA execution model, just enough to RUN THE REFERENCE'S OWN
// __global__ KERNEL BODIES in the build container (oracle/xcheck/ref_xcheck.py).  Test infrastructure.
//
// WHAT THIS IS NOT: a build of the reference.  The reference's extension cannot be built in this image (no
// nvcc / CUDA runtime / THC headers / cuSOLVER: DESIGN.md §3), and running its kernel source on stand-ins for
// the CUDA execution model pins nothing in the sense of the parity rules -- parity of this repository stays
// UNPINNED.  It is a cross-check ("O2" of SURVEY.md §8c): the text of the kernels, extracted by line range
// from /root/reference at run time and never stored in this repository, is executed and its outputs are
// compared with the CPU oracle's (O1) -- indices equal, distances within 2 ulp (fp contraction is the
// compiler's choice in the reference) -- so that a misreading of the reference in the line-by-line
// restatement would show.
//
// Model: a block's threads are std::threads meeting at a std::barrier (__syncthreads); blocks run one after
// the other; __shared__ is `static`; the dynamic shared array is one global buffer; atomicAdd is a CAS loop.
#pragma once
#include <algorithm>
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
static thread_local dim3 threadIdx, blockIdx;
static dim3 blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __align__(x)

alignas(16) static unsigned char my_smem[1 << 16];  // the kernels' `extern __shared__ ... my_smem[]`

static std::barrier<>* g_block_barrier = nullptr;
static inline void __syncthreads() { g_block_barrier->arrive_and_wait(); }

template <class T>
static inline T atomicAdd(T* addr, T v) {
  std::atomic_ref<T> r(*addr);
  T old = r.load(std::memory_order_relaxed);
  while (!r.compare_exchange_weak(old, old + v, std::memory_order_relaxed)) {
  }
  return old;
}

using std::max;
using std::min;
static inline unsigned min(unsigned a, int b) { return b < 0 ? 0u : std::min(a, (unsigned)b); }  // CUDA has this overload
static inline int min(int a, unsigned b) { return (int)std::min((unsigned)std::max(a, 0), b); }

// kernel<<<grid, block>>>(...): `body` is a closure that calls the kernel function
static inline void launch(dim3 grid, dim3 block, const std::function<void()>& body, bool needs_barrier) {
  gridDim = grid;
  blockDim = block;
  const unsigned nthreads = block.x * block.y * block.z;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        if (!needs_barrier) {  // no __syncthreads in the kernel: the threads may as well run in turn
          blockIdx = dim3(bx, by, bz);
          for (unsigned t = 0; t < nthreads; ++t) {
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            body();
          }
          continue;
        }
        std::barrier<> bar((std::ptrdiff_t)nthreads);
        g_block_barrier = &bar;
        std::vector<std::thread> pool;
        pool.reserve(nthreads);
        for (unsigned t = 0; t < nthreads; ++t)
          pool.emplace_back([=, &body]() {
            blockIdx = dim3(bx, by, bz);
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            body();
          });
        for (auto& th : pool) th.join();
      }
}
