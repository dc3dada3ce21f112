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

#include <map>
#include <string>
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
  // out1 / out2: where the distances are to be written (contiguous float32 (B, N) / (B, M) on the inputs' device --
  // e.g. a slot of the batch-sharded exchange, so that nothing has to be packed afterwards); undefined = allocate
  static variable_list forward(AutogradContext* ctx, const Tensor& a, const Tensor& b, const c10::optional<Tensor>& out1,
                               const c10::optional<Tensor>& out2) {
    const Tensor xyz1 = a.contiguous(), xyz2 = b.contiguous();
    const Shapes s = check_inputs(xyz1, xyz2);
    const auto fopts = xyz1.options();
    const auto iopts = fopts.dtype(torch::kInt32);
    // uninitialised: the kernels write every element (and zero-fill when one cloud is empty)
    Tensor dist1 = out1.has_value() ? *out1 : torch::empty({s.b, s.n}, fopts);
    Tensor dist2 = out2.has_value() ? *out2 : torch::empty({s.b, s.m}, fopts);
    TORCH_CHECK(dist1.is_contiguous() && dist2.is_contiguous() && dist1.scalar_type() == torch::kFloat32 &&
                    dist2.scalar_type() == torch::kFloat32 && dist1.device() == xyz1.device() &&
                    dist2.device() == xyz1.device() && dist1.numel() == (int64_t)s.b * s.n &&
                    dist2.numel() == (int64_t)s.b * s.m,
                "nndistance: out tensors must be contiguous float32 (B, N) and (B, M) on the inputs' device");
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
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    variable_list g = chamfer_backward(ctx, grads);
    g.push_back(Tensor());
    g.push_back(Tensor());
    return g;
  }
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
  auto r = NmDistance::apply(xyz1, xyz2, c10::optional<Tensor>(), c10::optional<Tensor>());
  return std::make_tuple(r[0], r[1], r[2], r[3]);
}

// the same operator with the distances written into tensors of the caller's (see NmDistance::forward)
std::tuple<Tensor, Tensor, Tensor, Tensor> nndistance_out(const Tensor& xyz1, const Tensor& xyz2, const Tensor& dist1,
                                                          const Tensor& dist2) {
  auto r = NmDistance::apply(xyz1, xyz2, c10::optional<Tensor>(dist1.detach()), c10::optional<Tensor>(dist2.detach()));
  return std::make_tuple(r[0], r[1], r[2], r[3]);
}

std::tuple<Tensor, Tensor, Tensor, Tensor> labeled_nndistance(const Tensor& xyz1, const Tensor& xyz2, const Tensor& l1,
                                                              const Tensor& l2) {
  auto r = LabeledNmDistance::apply(xyz1, xyz2, l1, l2);
  return std::make_tuple(r[0], r[1], r[2], r[3]);
}

// ---------------------------------------------------------------------------------------------------------------
// The batch-sharded operator's one exchange per step (pytorch_points_amd/sharded.py: PackedShardGather), issued from
// C++: ONE all-gather over RCCL of the shard's packed outputs (dist1 | dist2 | idx1 | idx2, indices as
// 16-bit words when they fit), on a side stream, in one call that never touches Python.
// Round 4:
//   * no pack of the distances: begin() hands out views of the slot's own first floats, which the caller gives to the
//     search as its dist outputs (nndistance_out / losses.nmdistance_forward); launch_in_place() narrows the indices in
//     behind them (one small kernel, 4 MB read / 2 MB written at config 2 instead of 8 / 6) and issues the gather;
//   * no unpack: wait() returns strided VIEWS of the gathered buffer -- distances as float32 (world, B, N), indices as
//     the 16-bit (or 32-bit) words they travelled as, (world, B, N); widen() makes int32 (world * B, N) tensors of the
//     indices for a consumer that wants them (the only kernel left on the receiving side, and only on demand);
//   * the collective is issued BY THE CALLING THREAD, in program order with the caller's other collectives (round 3
//     had a worker thread issue it on the caller's process group: enqueue order relative to the caller's own
//     collectives then differed from rank to rank -- ADVICE r3, medium).  The host cost of issuing it (~20-30 us
//     through c10d, ~10 with the direct communicator) is hidden behind a step's kernels at config 2 either way:
//     measured round 3, worker thread 0.105 against calling thread 0.107 ms per step.
struct PackedExchange {
  c10::intrusive_ptr<c10d::ProcessGroup> pg;
  int world, b, n, m, depth, compact;
  int64_t nbytes_padded;
  c10::Device dev;
  std::vector<Tensor> send, recv;
  std::vector<std::vector<Tensor>> wide;  // int32 indices of the global batch, made on demand (widen)
  // views made once, not per step (an as_strided per field and step was a third of the exchange's host time):
  std::vector<std::vector<Tensor>> own;       // [slot] -> the slot's own dist1 (b, n), dist2 (b, m)
  std::vector<std::vector<Tensor>> gathered;  // [slot] -> dist1, dist2, idx1, idx2 of every rank (wait)
  std::vector<hipEvent_t> packed, done;
  std::vector<long long> launched;
  c10::hip::HIPStreamMasqueradingAsCUDA side;
  int turn = 0;
  // Direct mode (init_direct): the all-gather is ONE ncclAllGather on a communicator of this object's own instead of a
  // c10d call (which wraps the same RCCL enqueue in a Work object, two events and a stream wait).  nullptr: c10d.
  ncclComm_t comm = nullptr;
  // p2p mode (set_p2p): the all-gather as ONE grouped set of world - 1 sends and world - 1 receives, in place between
  // the rows of the gathered buffers -- every part crosses the direct xGMI link between its two GPUs instead of
  // whatever ring or tree the library's all-gather picks (one link per hop, world - 1 hops back to back)
  bool p2p = false;
  int rank = 0;
  double issue_ns = 0.0;  // host time spent issuing exchanges (collective + events), and how many
  long long issue_slots = 0;

  static c10::Device with_index(const c10::Device& d) {  // "cuda" -> the current device
    if (d.has_index()) return d;
    int cur = 0;
    TORCH_CHECK(hipGetDevice(&cur) == hipSuccess, "hipGetDevice failed");
    return c10::Device(d.type(), (c10::DeviceIndex)cur);
  }

  PackedExchange(const c10::intrusive_ptr<c10d::ProcessGroup>& group, int b_local, int n_, int m_, const c10::Device& device,
                 int depth_)
      : pg(group), world(group->getSize()), b(b_local), n(n_), m(m_), depth(depth_), dev(with_index(device)),
        side(c10::hip::getStreamFromPoolMasqueradingAsCUDA(false, with_index(device).index())) {
    TORCH_CHECK(dev.is_cuda() && depth >= 1, "PackedExchange needs a GPU device and depth >= 1");
    rank = group->getRank();
    compact = std::max(n, m) <= 65535 ? 1 : 0;
    const int64_t isz = compact ? 2 : 4;
    const int64_t nbytes = (int64_t)4 * b * (n + m) + isz * b * (n + m);
    nbytes_padded = (nbytes + 15) / 16 * 16;
    const auto u8 = torch::TensorOptions().dtype(torch::kUInt8).device(dev);
    const c10::DeviceGuard guard(dev);
    for (int k = 0; k < depth; ++k) {
      send.push_back(torch::empty({nbytes_padded}, u8));
      recv.push_back(torch::empty({(int64_t)world, nbytes_padded}, u8));
      wide.push_back({});
      hipEvent_t e0, e1;
      TORCH_CHECK(hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess,
                  "hipEventCreate failed");
      packed.push_back(e0);
      done.push_back(e1);
      launched.push_back(0);
    }
    make_views();
  }
  void make_views() {  // (again whenever `send` changes: init_direct, disable_direct)
    own.clear();
    gathered.clear();
    for (int k = 0; k < depth; ++k) {
      const Tensor f = send[k].view(torch::kFloat32);
      own.push_back({f.narrow(0, 0, (int64_t)b * n).view({(int64_t)b, (int64_t)n}),
                     f.narrow(0, (int64_t)b * n, (int64_t)b * m).view({(int64_t)b, (int64_t)m})});
      gathered.push_back({dist_view(recv[k], world, 0), dist_view(recv[k], world, 1), idx_view(recv[k], world, 0),
                          idx_view(recv[k], world, 1)});
    }
  }
  ~PackedExchange() {
    // the side stream may still be writing the buffers this object is about to free
    (void)hipStreamSynchronize(side.stream());
    if (comm) (void)ncclCommDestroy(comm);
    for (hipEvent_t ev : packed) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : done) (void)hipEventDestroy(ev);
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
    comm = c;
    // IN PLACE from here on: a slot's own part is row `rank` of its gathered buffer (the search writes its distances
    // there, the indices are narrowed in behind them), which is what ncclAllGather takes as "sendbuff == recvbuff +
    // rank * count": RCCL then moves the seven foreign parts and nothing else -- and on one rank nothing at all.
    drain();
    for (int k = 0; k < depth; ++k) send[k] = recv[k].select(0, rank);
    make_views();
  }
  // The grouped send / receive form, in place like the direct form: a slot's own part is row `rank` of its gathered
  // buffer.  Over c10d's communicator (coalesced send / recv), or over the object's own after init_direct.  A
  // collective decision: every rank, before the first launch.
  void set_p2p() {
    drain();
    const c10::DeviceGuard guard(dev);
    for (int k = 0; k < depth; ++k) send[k] = recv[k].select(0, rank);
    make_views();
    p2p = true;
  }
  // back to c10d's all-gather (a rank failed to join, or the first exchange's self-check failed: every rank must then
  // take the same path).  EVERY slot's own part -- launched or only begun: the search may have written its distances
  // into the in-place row already -- moves into the send buffer the c10d call is given again.
  void disable_direct() {
    drain();
    (void)hipStreamSynchronize(side.stream());
    if (comm) (void)ncclCommDestroy(comm);
    comm = nullptr;
    const bool in_place = depth > 0 && send[0].data_ptr() == recv[0].select(0, rank).data_ptr();
    p2p = false;
    const c10::DeviceGuard guard(dev);
    for (int k = 0; k < depth; ++k) {
      send[k] = torch::empty({nbytes_padded}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
      if (in_place) send[k].copy_(recv[k].select(0, rank));
    }
    make_views();
  }

  hipStream_t current() const { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream(); }

  void finish(int slot) {  // the current stream waits for the slot's gather (no host block)
    if (launched[slot] == 0) return;
    TORCH_CHECK(hipStreamWaitEvent(current(), done[slot], 0) == hipSuccess, "hipStreamWaitEvent failed");
  }

  Tensor dist_view(const Tensor& buf, int64_t rows, int field) const {  // field 0: dist1 (rows, b, n), 1: dist2
    const int64_t stride = nbytes_padded / 4, off = field == 0 ? 0 : (int64_t)b * n, len = field == 0 ? n : m;
    const Tensor f = buf.view(torch::kFloat32);
    return f.as_strided({rows, (int64_t)b, len}, {stride, len, 1}, off);
  }
  Tensor idx_view(const Tensor& buf, int64_t rows, int field) const {  // the words the indices travelled as
    const int64_t isz = compact ? 2 : 4;
    const int64_t stride = nbytes_padded / isz, len = field == 0 ? n : m;
    const int64_t off = ((int64_t)4 * b * (n + m)) / isz + (field == 0 ? 0 : (int64_t)b * n);
    const Tensor w = buf.view(compact ? torch::kUInt16 : torch::kInt32);
    return w.as_strided({rows, (int64_t)b, len}, {stride, len, 1}, off);
  }

  // The next slot and views of its own distance fields, (b, n) and (b, m): the caller has the search write them
  // (nndistance_out), then calls launch_in_place(slot, idx1, idx2).  The current stream first waits for the slot's
  // previous exchange, whose bytes the search is about to overwrite.
  std::tuple<int, Tensor, Tensor> begin() {
    const int slot = turn;
    turn = (turn + 1) % depth;
    finish(slot);
    return std::make_tuple(slot, own[slot][0], own[slot][1]);
  }

  // begin + the search (the autograd node of nndistance, its distances written into the slot) + launch_in_place, in
  // ONE call from Python: -> (dist1, dist2, idx1, idx2, slot)
  std::tuple<Tensor, Tensor, Tensor, Tensor, int> forward(const Tensor& xyz1, const Tensor& xyz2) {
    const int slot = turn;
    turn = (turn + 1) % depth;
    finish(slot);
    auto r = NmDistance::apply(xyz1, xyz2, c10::optional<Tensor>(own[slot][0].detach()),
                               c10::optional<Tensor>(own[slot][1].detach()));
    launch_in_place(slot, r[2], r[3]);
    return std::make_tuple(r[0], r[1], r[2], r[3], slot);
  }

  void issue(int slot) {  // the gather of send[slot] on the side stream, behind everything the current stream holds
    const auto t_begin = std::chrono::steady_clock::now();
    const c10::DeviceGuard guard(dev);
    const hipStream_t cur = current();
    TORCH_CHECK(hipEventRecord(packed[slot], cur) == hipSuccess, "hipEventRecord failed");
    TORCH_CHECK(hipStreamWaitEvent(side.stream(), packed[slot], 0) == hipSuccess, "hipStreamWaitEvent failed");
    if (p2p && world == 1) {
      // (nothing to move: the own part is in place)
    } else if (p2p && comm) {
      ncclResult_t rc = ncclGroupStart();
      for (int d = 1; d < world && rc == ncclSuccess; ++d) {
        const int to = (rank + d) % world, from = (rank - d + world) % world;
        rc = ncclSend(send[slot].data_ptr(), (size_t)nbytes_padded, ncclChar, to, comm, side.stream());
        if (rc == ncclSuccess)
          rc = ncclRecv(recv[slot].select(0, from).data_ptr(), (size_t)nbytes_padded, ncclChar, from, comm, side.stream());
      }
      const ncclResult_t rc_end = ncclGroupEnd();
      TORCH_CHECK(rc == ncclSuccess && rc_end == ncclSuccess, "grouped ncclSend/ncclRecv failed: ",
                  ncclGetErrorString(rc != ncclSuccess ? rc : rc_end));
    } else if (p2p) {
      // c10d: the sends and receives coalesced into ONE group on the process group's communicator, ordered behind
      // the side stream like the all-gather below
      const c10::hip::HIPStreamGuardMasqueradingAsCUDA on_side(side);
      pg->startCoalescing(c10::DeviceType::CUDA);
      for (int d = 1; d < world; ++d) {
        const int to = (rank + d) % world, from = (rank - d + world) % world;
        std::vector<Tensor> out{send[slot]};
        std::vector<Tensor> in{recv[slot].select(0, from)};
        pg->send(out, to, 0);
        pg->recv(in, from, 0);
      }
      c10::intrusive_ptr<c10d::Work> work = pg->endCoalescing(c10::DeviceType::CUDA);
      if (work) work->wait();  // (the side stream waits for the group's end, no host block)
    } else if (comm) {
      const ncclResult_t rc = ncclAllGather(send[slot].data_ptr(), recv[slot].data_ptr(), (size_t)nbytes_padded, ncclChar, comm,
                                            side.stream());
      TORCH_CHECK(rc == ncclSuccess, "ncclAllGather failed: ", ncclGetErrorString(rc));
    } else {
      // (c10d orders the collective behind the CURRENT stream of the calling thread: the side stream, for this call)
      const c10::hip::HIPStreamGuardMasqueradingAsCUDA on_side(side);
      c10::intrusive_ptr<c10d::Work> work = pg->_allgather_base(recv[slot], send[slot]);
      work->wait();  // RCCL: the side stream waits for the collective's end, no host block
    }
    TORCH_CHECK(hipEventRecord(done[slot], side.stream()) == hipSuccess, "hipEventRecord failed");
    ++launched[slot];
    issue_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t_begin).count();
    ++issue_slots;
  }

  void check_idx(const Tensor& i1, const Tensor& i2) const {
    TORCH_CHECK(i1.numel() == (int64_t)b * n && i2.numel() == (int64_t)b * m && i1.scalar_type() == torch::kInt32 &&
                    i2.scalar_type() == torch::kInt32 && i1.device() == dev && i2.device() == dev,
                "PackedExchange: shard indices of another shape, dtype or device");
  }

  int launch_in_place(int slot, const Tensor& i1_, const Tensor& i2_) {
    TORCH_CHECK(slot >= 0 && slot < depth, "PackedExchange.launch_in_place: no such slot");
    const Tensor i1 = i1_.contiguous(), i2 = i2_.contiguous();
    check_idx(i1, i2);
    const c10::DeviceGuard guard(dev);
    check_code(pp_shard_pack_f32(nullptr, nullptr, i1.data_ptr<int>(), i2.data_ptr<int>(), send[slot].data_ptr(),
                                 (long long)b * n, (long long)b * m, compact, (void*)current()),
               "shard_pack");
    issue(slot);
    return slot;
  }

  // the general form: distances the caller holds elsewhere are copied in as well (one pack kernel)
  int launch(const Tensor& d1_, const Tensor& d2_, const Tensor& i1_, const Tensor& i2_) {
    const int slot = turn;
    turn = (turn + 1) % depth;
    finish(slot);  // the slot's buffers are about to be overwritten
    const Tensor d1 = d1_.detach().contiguous(), d2 = d2_.detach().contiguous(), i1 = i1_.contiguous(), i2 = i2_.contiguous();
    check_idx(i1, i2);
    TORCH_CHECK(d1.numel() == (int64_t)b * n && d2.numel() == (int64_t)b * m && d1.scalar_type() == torch::kFloat32 &&
                    d2.scalar_type() == torch::kFloat32 && d1.device() == dev && d2.device() == dev,
                "PackedExchange.launch: shard outputs of another shape, dtype or device");
    const c10::DeviceGuard guard(dev);
    check_code(pp_shard_pack_f32(d1.data_ptr<float>(), d2.data_ptr<float>(), i1.data_ptr<int>(), i2.data_ptr<int>(),
                                 send[slot].data_ptr(), (long long)b * n, (long long)b * m, compact, (void*)current()),
               "shard_pack");
    issue(slot);
    return slot;
  }

  // (dist1 (world, b, n), dist2 (world, b, m), idx1, idx2 (world, b, n / m) as they travelled: uint16 or int32) --
  // views of the slot's gathered buffer, valid until the slot is launched again; the current stream waits for the gather
  std::vector<Tensor> wait(int slot) {
    TORCH_CHECK(slot >= 0 && slot < depth, "PackedExchange.wait: no such slot");
    finish(slot);
    return gathered[slot];
  }

  // int32 indices of the global batch, (world * b, n) and (world * b, m), 0xFFFF -> -1: one kernel, on demand
  std::vector<Tensor> widen(int slot) {
    TORCH_CHECK(slot >= 0 && slot < depth, "PackedExchange.widen: no such slot");
    finish(slot);
    const c10::DeviceGuard guard(dev);
    if (wide[slot].empty()) {
      const auto i32 = torch::TensorOptions().dtype(torch::kInt32).device(dev);
      wide[slot] = {torch::empty({(int64_t)world * b, n}, i32), torch::empty({(int64_t)world * b, m}, i32)};
    }
    check_code(pp_shard_unpack_f32(recv[slot].data_ptr(), world, (long long)nbytes_padded, (long long)b * n,
                                   (long long)b * m, compact, nullptr, nullptr, wide[slot][0].data_ptr<int>(),
                                   wide[slot][1].data_ptr<int>(), (void*)current()),
               "shard_unpack");
    return wide[slot];
  }

  void drain() {
    for (int k = 0; k < depth; ++k) finish(k);
  }

  // the slot's gathered bytes, (world, nbytes_padded) uint8 (PackedShardGather's first-exchange self-check)
  Tensor raw(int slot) {
    TORCH_CHECK(slot >= 0 && slot < depth, "PackedExchange.raw: no such slot");
    finish(slot);
    return recv[slot];
  }

  // The self-check of the direct path failed somewhere: every rank is back on c10d (disable_direct) and gathers every
  // slot it has launched again, oldest first -- a slot's own part was row `rank` of its gathered buffer, which the
  // in-place form had the search write.
  void reissue(int rank) {
    TORCH_CHECK(comm == nullptr && rank >= 0 && rank < world, "PackedExchange.reissue: after disable_direct only");
    const c10::DeviceGuard guard(dev);
    for (int i = 0; i < depth; ++i) {
      const int slot = (turn + i) % depth;  // `turn` is the slot launched longest ago
      if (launched[slot] == 0) continue;
      send[slot].copy_(recv[slot].select(0, rank));
      issue(slot);
    }
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
  m.def("nndistance_out", &nndistance_out,
        "nndistance_out(xyz1, xyz2, dist1, dist2) -> (dist1, dist2, idx1, idx2): the distances written into the given tensors");
  pybind11::class_<PackedExchange>(m, "PackedExchange")
      .def(pybind11::init<const c10::intrusive_ptr<c10d::ProcessGroup>&, int, int, int, const c10::Device&, int>())
      .def("begin", &PackedExchange::begin, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("launch_in_place", &PackedExchange::launch_in_place, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("forward", &PackedExchange::forward, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("launch", &PackedExchange::launch, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("wait", &PackedExchange::wait, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("widen", &PackedExchange::widen, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("drain", &PackedExchange::drain, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("raw", &PackedExchange::raw, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("reissue", &PackedExchange::reissue, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def_static("unique_id", &PackedExchange::unique_id)
      .def("init_direct", &PackedExchange::init_direct, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("disable_direct", &PackedExchange::disable_direct, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def("set_p2p", &PackedExchange::set_p2p, pybind11::call_guard<pybind11::gil_scoped_release>())
      .def_readonly("p2p", &PackedExchange::p2p)
      .def("issue_us_per_slot", [](PackedExchange& e) { return e.issue_slots ? e.issue_ns / 1e3 / (double)e.issue_slots : 0.0; })
      .def_readonly("compact", &PackedExchange::compact)
      .def_readonly("nbytes_padded", &PackedExchange::nbytes_padded);
}
