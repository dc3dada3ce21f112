// torch_bridge.cpp -- the Chamfer operators as C++ torch::autograd::Function nodes over the C ABI.
//
// The kernels of one nndistance step take ~0.08 ms at BASELINE config 2; issued from Python
// (torch.autograd.Function: model_loss.NmDistanceFunction) the same step costs ~0.12 ms of host time -- the
// autograd engine hands the backward to its device thread, which has to take the GIL to run Python -- so the
// operator was host-bound.  Here the operator's host side is native, as the reference's is
// (a pybind11 C++ extension, _ext/nmdistance.cpp): input checks, four output allocations, the C-ABI call on
// torch's current HIP stream, and a backward node that runs on the engine's thread without Python.
// Same checks, same outputs and gradients as the Python classes (tests/test_gpu_chamfer.py runs both).
//
// PyTorch is plumbing here: tensors, the caching allocator, the current stream, autograd bookkeeping.  The
// compute is libpp_hip.so (include/pp_hip.h), which this module links and which has no torch types in it.
// Built in-tree by pytorch_points_amd/_build.py with g++ against the installed torch (ROCm build).
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>  // a ROCm build of torch names its HIP devices "cuda"
#include <c10/core/DeviceGuard.h>
#include <chrono>
#include <cstring>
#include <rccl/rccl.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>
#include <torch/extension.h>

#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <string>
#include <thread>
#include <mutex>
#include <tuple>
#include <vector>

#include "pp_hip.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// Scratch of the grid search: one growing buffer per (device, stream), as in _lib.workspace (Python); while the
// stream is being captured into a graph nothing is cached (the buffer then belongs to the capturing graph's pool).
std::mutex g_ws_mutex;
std::map<std::tuple<int, void*, bool>, Tensor> g_ws;

Tensor workspace(const c10::Device& dev, hipStream_t stream, size_t nbytes, bool labeled) {
  if (nbytes == 0) return Tensor();
  const auto opts = torch::TensorOptions().dtype(torch::kUInt8).device(dev);
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
    return torch::empty({(int64_t)nbytes}, opts);
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  if (g_ws.size() > 16) g_ws.clear();
  Tensor& slot = g_ws[std::make_tuple((int)dev.index(), (void*)stream, labeled)];
  if (!slot.defined() || (size_t)slot.numel() < nbytes) slot = torch::empty({(int64_t)nbytes}, opts);
  return slot;
}

struct Shapes {
  int b, n, m, c;
};

// the checks the reference leaves out (its launcher validates nothing, _ext/nmdistance.cpp:13-15), with the
// messages of the Python operator
Shapes check_inputs(const Tensor& xyz1, const Tensor& xyz2) {
  TORCH_CHECK(xyz1.scalar_type() == xyz2.scalar_type(), "xyz1 and xyz2 must have the same dtype");
  TORCH_CHECK(xyz1.scalar_type() == torch::kFloat32, "xyz1 must be a float tensor");
  TORCH_CHECK(xyz1.is_cuda(), "xyz1 must be a CUDA tensor");
  TORCH_CHECK(xyz2.is_cuda(), "xyz2 must be a CUDA tensor");
  TORCH_CHECK(xyz2.device() == xyz1.device(), "xyz2 is on ", xyz2.device(), ", expected ", xyz1.device());
  TORCH_CHECK(xyz1.dim() == 3 && xyz2.dim() == 3, "xyz1 and xyz2 must be (B, N, C) and (B, M, C)");
  TORCH_CHECK(xyz1.size(0) == xyz2.size(0) && xyz1.size(2) == xyz2.size(2), "xyz1 ", xyz1.sizes(), " and xyz2 ",
              xyz2.sizes(), " disagree in batch or point dimension");
  TORCH_CHECK(xyz1.size(0) < (1LL << 31) && xyz1.size(1) < (1LL << 31) && xyz2.size(1) < (1LL << 31) &&
                  xyz1.size(2) < (1LL << 31),
              "sizes beyond int32");
  return {(int)xyz1.size(0), (int)xyz1.size(1), (int)xyz2.size(1), (int)xyz1.size(2)};
}

void check_code(int code, const char* what) {
  TORCH_CHECK(code == 0, "pytorch_points_amd: ", what, " failed with HIP error ", code);
}

variable_list chamfer_backward(AutogradContext* ctx, const variable_list& grads) {
  const auto saved = ctx->get_saved_variables();
  const Tensor &xyz1 = saved[0], &xyz2 = saved[1], &idx1 = saved[2], &idx2 = saved[3];
  // a missing upstream gradient counts as zero
  Tensor g1 = grads[0].defined() ? grads[0].contiguous() : torch::zeros_like(idx1, xyz1.options());
  Tensor g2 = grads[1].defined() ? grads[1].contiguous() : torch::zeros_like(idx2, xyz2.options());
  TORCH_CHECK(g1.scalar_type() == torch::kFloat32 && g2.scalar_type() == torch::kFloat32,
              "graddist1 must be a float tensor");
  TORCH_CHECK(g1.device() == xyz1.device() && g2.device() == xyz1.device(),
              "graddist is on another device than xyz1 (", xyz1.device(), ")");
  Tensor out1 = torch::empty_like(xyz1), out2 = torch::empty_like(xyz2);  // fully overwritten by the kernel
  const c10::DeviceGuard guard(xyz1.device());
  const hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(xyz1.device().index()).stream();
  const int B = (int)xyz1.size(0), N = (int)xyz1.size(1), M = (int)xyz2.size(1), C = (int)xyz1.size(2);
  if (at::globalContext().deterministicAlgorithms()) {
    // the ordered form: ascending source order, no floating-point atomics (include/pp_hip.h)
    const int code = pp_nmdistance_backward_ordered_f32(xyz1.data_ptr<float>(), xyz2.data_ptr<float>(),
                                                        g1.data_ptr<float>(), g2.data_ptr<float>(), idx1.data_ptr<int>(),
                                                        idx2.data_ptr<int>(), out1.data_ptr<float>(),
                                                        out2.data_ptr<float>(), B, N, M, C, (void*)stream);
    if (code == 0) return {out1, out2};
    if (code != PP_ENOTSUP) check_code(code, "nmdistance_backward");
    const char* msg = "pytorch_points_amd: nmdistance_backward has no deterministic implementation for this shape "
                      "(torch.use_deterministic_algorithms(True) is set)";
    TORCH_CHECK(at::globalContext().deterministicAlgorithmsWarnOnly(), msg);
    TORCH_WARN(msg);
  }
  check_code(pp_nmdistance_backward_f32(xyz1.data_ptr<float>(), xyz2.data_ptr<float>(), g1.data_ptr<float>(),
                                        g2.data_ptr<float>(), idx1.data_ptr<int>(), idx2.data_ptr<int>(),
                                        out1.data_ptr<float>(), out2.data_ptr<float>(), B, N, M, C, (void*)stream),
             "nmdistance_backward");
  return {out1, out2};
}

bool g_force_brute = false;  // PP_NMDISTANCE_SEARCH=bruteforce, read once by the Python package at import

// nndistance(xyz1 (B,N,C), xyz2 (B,M,C)) -> (dist1 (B,N), dist2 (B,M), idx1, idx2); reference
// network/model_loss.py:401-439 + _ext/nmdistance.cpp:13-27
struct NmDistance : public torch::autograd::Function<NmDistance> {
  static variable_list forward(AutogradContext* ctx, const Tensor& a, const Tensor& b) {
    const Tensor xyz1 = a.contiguous(), xyz2 = b.contiguous();
    const Shapes s = check_inputs(xyz1, xyz2);
    const auto fopts = xyz1.options();
    const auto iopts = fopts.dtype(torch::kInt32);
    // uninitialised: the kernels write every element (and zero-fill when one cloud is empty)
    Tensor dist1 = torch::empty({s.b, s.n}, fopts), dist2 = torch::empty({s.b, s.m}, fopts);
    Tensor idx1 = torch::empty({s.b, s.n}, iopts), idx2 = torch::empty({s.b, s.m}, iopts);
    const c10::DeviceGuard guard(xyz1.device());
    const hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(xyz1.device().index()).stream();
    const size_t nbytes = g_force_brute ? 0 : pp_nmdistance_forward_workspace_bytes(s.b, s.n, s.m, s.c);
    const Tensor ws = workspace(xyz1.device(), stream, nbytes, false);
    check_code(pp_nmdistance_forward_ws_f32(xyz1.data_ptr<float>(), xyz2.data_ptr<float>(), dist1.data_ptr<float>(),
                                            idx1.data_ptr<int>(), dist2.data_ptr<float>(), idx2.data_ptr<int>(), s.b,
                                            s.n, s.m, s.c, ws.defined() ? ws.data_ptr() : nullptr, nbytes,
                                            (void*)stream),
               "nmdistance_forward");
    ctx->save_for_backward({xyz1, xyz2, idx1, idx2});
    ctx->mark_non_differentiable({idx1, idx2});
    ctx->set_materialize_grads(false);
    return {dist1, dist2, idx1, idx2};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) { return chamfer_backward(ctx, grads); }
};

// labeled_nndistance(xyz1, xyz2, label1 (B,N), label2 (B,M)); reference network/model_loss.py:445-481.  Labels
// are compared in the coordinates' dtype, as the reference's kernel does (_ext/nmdistance_cuda.cu:153).
struct LabeledNmDistance : public torch::autograd::Function<LabeledNmDistance> {
  static variable_list forward(AutogradContext* ctx, const Tensor& a, const Tensor& b, const Tensor& la,
                               const Tensor& lb) {
    const Tensor xyz1 = a.contiguous(), xyz2 = b.contiguous();
    const Shapes s = check_inputs(xyz1, xyz2);
    const Tensor label1 = la.to(xyz1.scalar_type()).contiguous(), label2 = lb.to(xyz1.scalar_type()).contiguous();
    TORCH_CHECK(label1.is_cuda() && label1.device() == xyz1.device(), "label1 must be a CUDA tensor on ", xyz1.device());
    TORCH_CHECK(label2.is_cuda() && label2.device() == xyz1.device(), "label2 must be a CUDA tensor on ", xyz1.device());
    TORCH_CHECK(label1.numel() == (int64_t)s.b * s.n && label2.numel() == (int64_t)s.b * s.m,
                "labels must be (B, N) and (B, M)");
    const auto fopts = xyz1.options();
    const auto iopts = fopts.dtype(torch::kInt32);
    Tensor dist1 = torch::empty({s.b, s.n}, fopts), dist2 = torch::empty({s.b, s.m}, fopts);
    Tensor idx1 = torch::empty({s.b, s.n}, iopts), idx2 = torch::empty({s.b, s.m}, iopts);
    const c10::DeviceGuard guard(xyz1.device());
    const hipStream_t stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(xyz1.device().index()).stream();
    const size_t nbytes = g_force_brute ? 0 : pp_labeled_nmdistance_forward_workspace_bytes(s.b, s.n, s.m, s.c);
    const Tensor ws = workspace(xyz1.device(), stream, nbytes, true);
    check_code(pp_labeled_nmdistance_forward_ws_f32(
                   xyz1.data_ptr<float>(), xyz2.data_ptr<float>(), label1.data_ptr<float>(), label2.data_ptr<float>(),
                   dist1.data_ptr<float>(), idx1.data_ptr<int>(), dist2.data_ptr<float>(), idx2.data_ptr<int>(), s.b, s.n,
                   s.m, s.c, ws.defined() ? ws.data_ptr() : nullptr, nbytes, (void*)stream),
               "labeled_nmdistance_forward");
    ctx->save_for_backward({xyz1, xyz2, idx1, idx2});
    ctx->mark_non_differentiable({idx1, idx2});
    ctx->set_materialize_grads(false);
    return {dist1, dist2, idx1, idx2};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    variable_list g = chamfer_backward(ctx, grads);
    g.push_back(Tensor());
    g.push_back(Tensor());
    return g;
  }
};

std::tuple<Tensor, Tensor, Tensor, Tensor> nndistance(const Tensor& xyz1, const Tensor& xyz2) {
  auto r = NmDistance::apply(xyz1, xyz2);
  return std::make_tuple(r[0], r[1], r[2], r[3]);
}

std::tuple<Tensor, Tensor, Tensor, Tensor> labeled_nndistance(const Tensor& xyz1, const Tensor& xyz2, const Tensor& l1,
                                                              const Tensor& l2) {
  auto r = LabeledNmDistance::apply(xyz1, xyz2, l1, l2);
  return std::make_tuple(r[0], r[1], r[2], r[3]);
}

// ---------------------------------------------------------------------------------------------------------------
// The batch-sharded operator's one exchange per step (pytorch_points_amd/sharded.py: PackedShardGather), issued from
// C++: pack the shard's (dist1 | dist2 | idx1 | idx2) into one buffer (pp_shard_pack_f32, on the
// current stream), ONE all-gather of it over RCCL (c10d: ProcessGroup::_allgather_base), and -- on a side stream that
// waits for the collective -- unpack the gathered bytes into the global-batch tensors (pp_shard_unpack_f32), all in one
// call that never touches Python: issued from Python the same three steps cost the thread ~50 us per step, more than
// the step's kernels leave idle.  Slots are double-buffered by the caller's `depth`; wait(slot) makes the CURRENT
// stream (not the host) wait for the slot's unpack and returns its tensors, valid until the slot is launched again.
struct PackedExchange {
  c10::intrusive_ptr<c10d::ProcessGroup> pg;
  int world, b, n, m, depth, compact;
  int64_t nbytes_padded;
  c10::Device dev;
  std::vector<Tensor> send, recv;
  std::vector<std::vector<Tensor>> out;
  std::vector<hipEvent_t> packed, done;
  c10::hip::HIPStreamMasqueradingAsCUDA side;
  int turn = 0;
  // The collective and the unpack are ISSUED BY A WORKER THREAD of this object (no Python in it, so no GIL to fight
  // over): the calling thread only launches the pack, records an event and queues the slot -- issuing the collective
  // itself (c10d + RCCL enqueue, ~30 us of host time) made the step host-bound (0.105 against 0.078 ms at config 2).
  // launched[k] / issued[k]: how often slot k has been handed to the worker / completed by it (its done event recorded).
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::deque<int> queue;
  std::vector<long long> launched, issued;
  std::string failure;  // the worker's first exception, re-raised in the calling thread
  bool stop = false;
  // Direct mode (init_direct): the all-gather is ONE ncclAllGather on a communicator of this object's own instead of a
  // c10d call -- c10d wraps the same RCCL enqueue in a Work object, two events and a stream wait, ~10 HIP runtime calls
  // per exchange issued from a second thread while the first launches the step's kernels; the runtime's locks made
  // the pair host-bound (0.105 ms per step against 0.078 of kernels at config 2).  nullptr: c10d.
  ncclComm_t comm = nullptr;
  double worker_ns = 0.0;  // host time the worker has spent issuing (collective + unpack + event), and how many slots
  long long worker_slots = 0;

  PackedExchange(const c10::intrusive_ptr<c10d::ProcessGroup>& group, int b_local, int n_, int m_, const c10::Device& device,
                 int depth_)
      : pg(group), world(group->getSize()), b(b_local), n(n_), m(m_), depth(depth_), dev(device),
        side(c10::hip::getStreamFromPoolMasqueradingAsCUDA(false, device.index())) {
    TORCH_CHECK(device.is_cuda() && depth >= 1, "PackedExchange needs a GPU device and depth >= 1");
    compact = std::max(n, m) <= 65535 ? 1 : 0;
    const int64_t isz = compact ? 2 : 4;
    const int64_t nbytes = (int64_t)4 * b * (n + m) + isz * b * (n + m);
    nbytes_padded = (nbytes + 15) / 16 * 16;
    const auto u8 = torch::TensorOptions().dtype(torch::kUInt8).device(device);
    const auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(device);
    const auto i32 = torch::TensorOptions().dtype(torch::kInt32).device(device);
    const c10::DeviceGuard guard(device);
    for (int k = 0; k < depth; ++k) {
      send.push_back(torch::empty({nbytes_padded}, u8));
      recv.push_back(torch::empty({(int64_t)world, nbytes_padded}, u8));
      out.push_back({torch::empty({(int64_t)world * b, n}, f32), torch::empty({(int64_t)world * b, m}, f32),
                     torch::empty({(int64_t)world * b, n}, i32), torch::empty({(int64_t)world * b, m}, i32)});
      hipEvent_t e0, e1;
      TORCH_CHECK(hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess,
                  "hipEventCreate failed");
      packed.push_back(e0);
      done.push_back(e1);
      launched.push_back(0);
      issued.push_back(0);
    }
    worker = std::thread([this] { run(); });
  }
  ~PackedExchange() {
    {
      std::lock_guard<std::mutex> lock(mu);
      stop = true;
    }
    cv_work.notify_all();
    if (worker.joinable()) worker.join();
    if (comm) (void)ncclCommDestroy(comm);
    for (hipEvent_t ev : packed) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : done) (void)hipEventDestroy(ev);
  }

  void run() {  // the worker: collective + unpack of every queued slot, on the side stream
    for (;;) {
      int slot;
      {
        std::unique_lock<std::mutex> lock(mu);
        cv_work.wait(lock, [this] { return stop || !queue.empty(); });
        if (queue.empty()) return;  // (stop)
        slot = queue.front();
        queue.pop_front();
      }
      const auto t_begin = std::chrono::steady_clock::now();
      try {
        const c10::DeviceGuard guard(dev);
        const c10::hip::HIPStreamGuardMasqueradingAsCUDA on_side(side);
        TORCH_CHECK(hipStreamWaitEvent(side.stream(), packed[slot], 0) == hipSuccess, "hipStreamWaitEvent failed");
        // (c10d orders the collective behind the CURRENT stream of the calling thread: here the side stream, which waits
        //  for the pack)
        if (comm) {
          const ncclResult_t rc = ncclAllGather(send[slot].data_ptr(), recv[slot].data_ptr(), (size_t)nbytes_padded, ncclChar, comm,
                                                side.stream());
          TORCH_CHECK(rc == ncclSuccess, "ncclAllGather failed: ", ncclGetErrorString(rc));
        } else {
          c10::intrusive_ptr<c10d::Work> work = pg->_allgather_base(recv[slot], send[slot]);
          work->wait();  // RCCL: the side stream waits for the collective's end, no host block
        }
        check_code(pp_shard_unpack_f32(recv[slot].data_ptr(), world, (long long)nbytes_padded, (long long)b * n,
                                       (long long)b * m, compact, out[slot][0].data_ptr<float>(),
                                       out[slot][1].data_ptr<float>(), out[slot][2].data_ptr<int>(),
                                       out[slot][3].data_ptr<int>(), (void*)side.stream()),
                   "shard_unpack");
        TORCH_CHECK(hipEventRecord(done[slot], side.stream()) == hipSuccess, "hipEventRecord failed");
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> lock(mu);
        if (failure.empty()) failure = e.what();
      }
      {
        std::lock_guard<std::mutex> lock(mu);
        ++issued[slot];
        worker_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t_begin).count();
        ++worker_slots;
      }
      cv_done.notify_all();
    }
  }

  // rank 0 makes the id (128 bytes), every rank receives it (the caller broadcasts it over the process group) and
  // joins: a collective call -- all ranks, before the first launch
  static pybind11::bytes unique_id() {
    ncclUniqueId id;
    const ncclResult_t rc = ncclGetUniqueId(&id);
    TORCH_CHECK(rc == ncclSuccess, "ncclGetUniqueId failed: ", ncclGetErrorString(rc));
    return pybind11::bytes(id.internal, NCCL_UNIQUE_ID_BYTES);
  }
  void init_direct(const std::string& id_bytes, int rank) {
    TORCH_CHECK(id_bytes.size() == NCCL_UNIQUE_ID_BYTES, "init_direct: the id must be ", NCCL_UNIQUE_ID_BYTES, " bytes");
    TORCH_CHECK(comm == nullptr, "init_direct: already initialised");
    ncclUniqueId id;
    std::memcpy(id.internal, id_bytes.data(), NCCL_UNIQUE_ID_BYTES);
    const c10::DeviceGuard guard(dev);
    ncclComm_t c = nullptr;
    const ncclResult_t rc = ncclCommInitRank(&c, world, id, rank);
    TORCH_CHECK(rc == ncclSuccess, "ncclCommInitRank failed: ", ncclGetErrorString(rc));
    std::lock_guard<std::mutex> lock(mu);
    comm = c;
  }

  void disable_direct() {  // back to c10d (a rank failed to join: every rank must then take the same path)
    drain();
    std::lock_guard<std::mutex> lock(mu);
    if (comm) (void)ncclCommDestroy(comm);
    comm = nullptr;
  }

  void finish(int slot) {  // the current stream waits for the slot's unpack (the host only until the worker has issued it)
    {
      std::unique_lock<std::mutex> lock(mu);
      cv_done.wait(lock, [this, slot] { return issued[slot] == launched[slot]; });
      TORCH_CHECK(failure.empty(), "PackedExchange: ", failure);
    }
    if (launched[slot] == 0) return;
    const hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
    TORCH_CHECK(hipStreamWaitEvent(cur, done[slot], 0) == hipSuccess, "hipStreamWaitEvent failed");
  }

  int launch(const Tensor& d1_, const Tensor& d2_, const Tensor& i1_, const Tensor& i2_) {
    const int slot = turn;
    turn = (turn + 1) % depth;
    finish(slot);  // the slot's buffers are about to be overwritten
    const Tensor d1 = d1_.detach().contiguous(), d2 = d2_.detach().contiguous(), i1 = i1_.contiguous(), i2 = i2_.contiguous();
    TORCH_CHECK(d1.numel() == (int64_t)b * n && d2.numel() == (int64_t)b * m && i1.numel() == d1.numel() &&
                    i2.numel() == d2.numel() && d1.scalar_type() == torch::kFloat32 && i1.scalar_type() == torch::kInt32 &&
                    d1.device() == dev,
                "PackedExchange.launch: shard outputs of another shape, dtype or device");
    const c10::DeviceGuard guard(dev);
    const hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream();
    check_code(pp_shard_pack_f32(d1.data_ptr<float>(), d2.data_ptr<float>(), i1.data_ptr<int>(), i2.data_ptr<int>(),
                                 send[slot].data_ptr(), (long long)b * n, (long long)b * m, compact, (void*)cur),
               "shard_pack");
    TORCH_CHECK(hipEventRecord(packed[slot], cur) == hipSuccess, "hipEventRecord failed");
    {
      std::lock_guard<std::mutex> lock(mu);
      ++launched[slot];
      queue.push_back(slot);
    }
    cv_work.notify_one();
    return slot;
  }

  std::vector<Tensor> wait(int slot) {
    TORCH_CHECK(slot >= 0 && slot < depth, "PackedExchange.wait: no such slot");
    finish(slot);
    return out[slot];
  }

  void drain() {
    for (int k = 0; k < depth; ++k) finish(k);
  }
};

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "C++ autograd nodes of pytorch_points_amd over the C ABI of libpp_hip.so";
  m.def("nndistance", &nndistance, "nndistance(xyz1, xyz2) -> (dist1, dist2, idx1, idx2)");
  m.def("labeled_nndistance", &labeled_nndistance,
        "labeled_nndistance(xyz1, xyz2, label1, label2) -> (dist1, dist2, idx1, idx2)");
  m.def("set_force_bruteforce", [](bool on) { g_force_brute = on; });
  m.def("library_version", []() { return std::string(pp_version()); });
  pybind11::class_<PackedExchange>(m, "PackedExchange")
      .def(pybind11::init<const c10::intrusive_ptr<c10d::ProcessGroup>&, int, int, int, const c10::Device&, int>())
      .def("launch", &PackedExchange::launch, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("wait", &PackedExchange::wait, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("drain", &PackedExchange::drain, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def_static("unique_id", &PackedExchange::unique_id)
      .def("init_direct", &PackedExchange::init_direct, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("disable_direct", &PackedExchange::disable_direct, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("worker_us_per_slot", [](PackedExchange& e) { std::lock_guard<std::mutex> l(e.mu); return e.worker_slots ? e.worker_ns / 1e3 / (double)e.worker_slots : 0.0; })
      .def_readonly("compact", &PackedExchange::compact)
      .def_readonly("nbytes_padded", &PackedExchange::nbytes_padded);
}
