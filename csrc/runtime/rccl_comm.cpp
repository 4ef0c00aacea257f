// RcclComm implementation (see rccl_comm.h): RCCL C API + HIP streams / events, no c10d in between.
#include "rccl_comm.h"

#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../kernels/api.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace py = pybind11;

namespace pdt {

namespace {

#define PDT_HIP(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    TORCH_CHECK(e_ == hipSuccess, "HIP error in RcclComm: ", hipGetErrorString(e_), " (" #expr ")");   \
  } while (0)
#define PDT_NCCL(expr)                                                                                 \
  do {                                                                                                 \
    ncclResult_t r_ = (expr);                                                                          \
    TORCH_CHECK(r_ == ncclSuccess, "RCCL error: ", ncclGetErrorString(r_), " (" #expr ")");            \
  } while (0)

ncclDataType_t nccl_dtype(at::ScalarType t) {
  switch (t) {
    case at::kFloat: return ncclFloat;
    case at::kHalf: return ncclHalf;
    case at::kBFloat16: return ncclBfloat16;
    case at::kDouble: return ncclDouble;
    case at::kLong: return ncclInt64;
    case at::kInt: return ncclInt32;
    case at::kByte: return ncclUint8;
    case at::kChar: return ncclInt8;
    case at::kBool: return ncclUint8;
    default: TORCH_CHECK(false, "RcclComm: unsupported dtype ", t);
  }
}

hipStream_t current_stream() { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

}  // namespace

std::string RcclComm::unique_id() {
  ncclUniqueId id;
  PDT_NCCL(ncclGetUniqueId(&id));
  return std::string(reinterpret_cast<const char*>(&id), sizeof(id));
}

int64_t RcclComm::rccl_version() {
  int v = 0;
  PDT_NCCL(ncclGetVersion(&v));
  return v;
}

RcclComm::RcclComm(const std::string& unique_id, int64_t rank, int64_t world, int64_t device)
    : rank_(rank), world_(world), device_(device) {
  TORCH_CHECK(unique_id.size() == sizeof(ncclUniqueId), "unique id has ", unique_id.size(), " bytes, expected ",
              sizeof(ncclUniqueId));
  TORCH_CHECK(world >= 1 && rank >= 0 && rank < world, "rank ", rank, " outside a group of ", world);
  ncclUniqueId id;
  std::memcpy(&id, unique_id.data(), sizeof(id));
  PDT_HIP(hipSetDevice((int)device));
  // The communication stream has NORMAL priority, like the compute stream.  Measured with bench/dev/rccl_stall.cpp
  // on ROCm 7.0 and 7.2 (one process per variant): once an RCCL communicator exists in the process, a chain of
  // record / wait pairs between the null stream and a stream created with the greatest priority never drains
  // (64 rounds enqueue in 2 ms and are still pending 15 s later; with or without an ncclAllReduce in between),
  // while the same chain on a normal-priority stream - or against an explicitly created compute stream - drains in
  // 2 ms.  PyTorch computes on the null stream, so the elevated priority is opt-in (PDT_COMM_STREAM_PRIORITY=high).
  const char* prio = std::getenv("PDT_COMM_STREAM_PRIORITY");
  if (prio && std::strcmp(prio, "high") == 0) {
    int least = 0, greatest = 0;
    PDT_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    PDT_HIP(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, greatest));
  } else {
    PDT_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  }
  for (int i = 0; i < kEvents; ++i) {
    PDT_HIP(hipEventCreateWithFlags(&ev_fork_[i], hipEventDisableTiming));
    PDT_HIP(hipEventCreateWithFlags(&ev_join_[i], hipEventDisableTiming));
  }
  PDT_NCCL(ncclCommInitRank(&comm_, (int)world, id, (int)rank));
  barrier_word_ = at::zeros({1}, at::TensorOptions().dtype(at::kFloat).device(at::kCUDA, (c10::DeviceIndex)device));
}

RcclComm::~RcclComm() {
  // No ncclCommDestroy here: a destructor may run while the interpreter (and the HIP runtime under it) is being
  // torn down, and a collective teardown from there can hang the exit of a rank whose peers are already gone.
  // close() is the orderly path.
}

void RcclComm::close() {
  if (!comm_) return;
  PDT_HIP(hipStreamSynchronize(stream_));
  PDT_NCCL(ncclCommDestroy(comm_));
  comm_ = nullptr;
  for (int i = 0; i < kEvents; ++i) {
    (void)hipEventDestroy(ev_fork_[i]);
    (void)hipEventDestroy(ev_join_[i]);
  }
  (void)hipStreamDestroy(stream_);
  stream_ = nullptr;
}

void RcclComm::check_tensor(const at::Tensor& t) const {
  TORCH_CHECK(comm_, "communicator is closed");
  TORCH_CHECK(t.is_cuda() && t.get_device() == device_, "tensor must live on GPU ", device_, " (the communicator's device)");
  TORCH_CHECK(t.is_contiguous(), "tensor must be contiguous");
}

void RcclComm::fork() {
  TORCH_CHECK(comm_, "communicator is closed");
  hipEvent_t ev = ev_fork_[n_fork_++ % kEvents];
  PDT_HIP(hipEventRecord(ev, current_stream()));
  PDT_HIP(hipStreamWaitEvent(stream_, ev, 0));
}

void RcclComm::join() {
  TORCH_CHECK(comm_, "communicator is closed");
  hipEvent_t ev = ev_join_[n_join_++ % kEvents];
  PDT_HIP(hipEventRecord(ev, stream_));
  PDT_HIP(hipStreamWaitEvent(current_stream(), ev, 0));
}

void RcclComm::all_reduce_async(const at::Tensor& t, bool average) {
  check_tensor(t);
  PDT_NCCL(ncclAllReduce(t.data_ptr(), t.data_ptr(), (size_t)t.numel(), nccl_dtype(t.scalar_type()),
                         average ? ncclAvg : ncclSum, comm_, stream_));
  ++launches_;
  bytes_ += t.numel() * t.element_size();
}

void RcclComm::broadcast_async(const at::Tensor& t, int64_t root) {
  check_tensor(t);
  TORCH_CHECK(root >= 0 && root < world_, "broadcast root outside the group");
  PDT_NCCL(ncclBroadcast(t.data_ptr(), t.data_ptr(), (size_t)t.numel(), nccl_dtype(t.scalar_type()), (int)root, comm_, stream_));
  ++launches_;
  bytes_ += t.numel() * t.element_size();
}

void RcclComm::all_gather_async(const at::Tensor& out, const at::Tensor& in) {
  check_tensor(out);
  check_tensor(in);
  TORCH_CHECK(out.scalar_type() == in.scalar_type() && out.numel() == in.numel() * world_,
              "all_gather: output must hold world_size copies of the input");
  PDT_NCCL(ncclAllGather(in.data_ptr(), out.data_ptr(), (size_t)in.numel(), nccl_dtype(in.scalar_type()), comm_, stream_));
  ++launches_;
  bytes_ += out.numel() * out.element_size();
}

void RcclComm::reduce_scatter_async(const at::Tensor& out, const at::Tensor& in, bool average) {
  check_tensor(out);
  check_tensor(in);
  TORCH_CHECK(out.scalar_type() == in.scalar_type() && in.numel() == out.numel() * world_,
              "reduce_scatter: input must hold world_size shards of the output's size");
  PDT_NCCL(ncclReduceScatter(in.data_ptr(), out.data_ptr(), (size_t)out.numel(), nccl_dtype(in.scalar_type()),
                             average ? ncclAvg : ncclSum, comm_, stream_));
  ++launches_;
  bytes_ += in.numel() * in.element_size();
}

void RcclComm::check_finite_async(const at::Tensor& t, const at::Tensor& scale_state) {
  check_tensor(t);
  TORCH_CHECK(t.scalar_type() == at::kFloat, "check_finite: fp32 gradients expected");
  TORCH_CHECK(scale_state.is_cuda() && scale_state.scalar_type() == at::kFloat && scale_state.numel() >= 4,
              "check_finite: scale_state is float[4] on the GPU");
  const int e = pdt_grad_check(t.data_ptr<float>(), t.numel(), scale_state.data_ptr<float>(), stream_);
  TORCH_CHECK(e == 0, "grad_check launch failed: ", e);
}

void RcclComm::adamw_async(int dtype, const PdtAdamW& args) {
  const int e = pdt_adamw_step(dtype, &args, stream_);
  TORCH_CHECK(e == 0, "adamw launch on the communication stream failed: ", e);
}

void RcclComm::p2p_async(const std::vector<int64_t>& kinds, const std::vector<at::Tensor>& tensors,
                         const std::vector<int64_t>& peers) {
  TORCH_CHECK(kinds.size() == tensors.size() && kinds.size() == peers.size(), "p2p: list lengths differ");
  for (size_t i = 0; i < kinds.size(); ++i) {
    check_tensor(tensors[i]);
    TORCH_CHECK(peers[i] >= 0 && peers[i] < world_, "p2p peer outside the group");
    TORCH_CHECK(kinds[i] == 0 || kinds[i] == 1, "p2p kind must be 0 (send) or 1 (recv)");
  }
  PDT_NCCL(ncclGroupStart());
  ncclResult_t first_error = ncclSuccess;
  for (size_t i = 0; i < kinds.size(); ++i) {
    const at::Tensor& t = tensors[i];
    ncclResult_t r = kinds[i] == 0
        ? ncclSend(t.data_ptr(), (size_t)t.numel(), nccl_dtype(t.scalar_type()), (int)peers[i], comm_, stream_)
        : ncclRecv(t.data_ptr(), (size_t)t.numel(), nccl_dtype(t.scalar_type()), (int)peers[i], comm_, stream_);
    if (r != ncclSuccess && first_error == ncclSuccess) first_error = r;
    bytes_ += t.numel() * t.element_size();
  }
  PDT_NCCL(ncclGroupEnd());  // the group is always closed, also after a failed call inside it
  PDT_NCCL(first_error);
  ++launches_;
}

void RcclComm::barrier() {
  fork();
  all_reduce_async(barrier_word_, false);
  join();
  synchronize();
}

void RcclComm::synchronize() {
  TORCH_CHECK(comm_, "communicator is closed");
  PDT_HIP(hipStreamSynchronize(stream_));
}

bool RcclComm::wait_idle(double seconds) {
  TORCH_CHECK(comm_, "communicator is closed");
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
  for (;;) {
    const hipError_t e = hipStreamQuery(stream_);
    if (e == hipSuccess) return true;
    TORCH_CHECK(e == hipErrorNotReady, "HIP error in RcclComm: ", hipGetErrorString(e), " (hipStreamQuery)");
    ncclResult_t async = ncclSuccess;
    if (ncclCommGetAsyncError(comm_, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) return false;
    if (std::chrono::steady_clock::now() >= deadline) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}

void RcclComm::abort() {
  if (!comm_) return;
  (void)ncclCommAbort(comm_);
  comm_ = nullptr;  // (stream and events are left to the process exit: work that was aborted may still refer to them)
}

int64_t RcclComm::comm_count() const {
  TORCH_CHECK(comm_, "communicator is closed");
  int n = 0;
  PDT_NCCL(ncclCommCount(comm_, &n));
  return n;
}

void RcclComm::check_async_error() {
  TORCH_CHECK(comm_, "communicator is closed");
  ncclResult_t async = ncclSuccess;
  PDT_NCCL(ncclCommGetAsyncError(comm_, &async));
  TORCH_CHECK(async == ncclSuccess || async == ncclInProgress, "RCCL asynchronous error on rank ", rank_, ": ",
              ncclGetErrorString(async));
}

}  // namespace pdt

void bind_rccl_comm(py::module_& m) {
  using pdt::RcclComm;
  py::class_<RcclComm, std::shared_ptr<RcclComm>>(m, "RcclComm")
      // (the GIL is released: ncclCommInitRank returns only when every rank of the group has called it)
      .def(py::init<const std::string&, int64_t, int64_t, int64_t>(), py::arg("unique_id"), py::arg("rank"),
           py::arg("world_size"), py::arg("device"), py::call_guard<py::gil_scoped_release>())
      .def_static("unique_id", []() { return py::bytes(RcclComm::unique_id()); })
      .def_static("rccl_version", &RcclComm::rccl_version)
      .def_property_readonly("rank", &RcclComm::rank)
      .def_property_readonly("world_size", &RcclComm::size)
      .def_property_readonly("device", &RcclComm::device)
      .def("fork", &RcclComm::fork)
      .def("join", &RcclComm::join)
      .def("all_reduce", &RcclComm::all_reduce, py::arg("tensor"), py::arg("average") = false)
      .def("broadcast", &RcclComm::broadcast, py::arg("tensor"), py::arg("root") = 0)
      .def("all_gather", &RcclComm::all_gather, py::arg("out"), py::arg("inp"))
      .def("reduce_scatter", &RcclComm::reduce_scatter, py::arg("out"), py::arg("inp"), py::arg("average") = false)
      .def("p2p", &RcclComm::p2p, py::arg("kinds"), py::arg("tensors"), py::arg("peers"))
      .def("all_reduce_async", &RcclComm::all_reduce_async, py::arg("tensor"), py::arg("average") = false)
      .def("broadcast_async", &RcclComm::broadcast_async, py::arg("tensor"), py::arg("root") = 0)
      .def("all_gather_async", &RcclComm::all_gather_async, py::arg("out"), py::arg("inp"))
      .def("reduce_scatter_async", &RcclComm::reduce_scatter_async, py::arg("out"), py::arg("inp"), py::arg("average") = false)
      .def("p2p_async", &RcclComm::p2p_async, py::arg("kinds"), py::arg("tensors"), py::arg("peers"))
      .def("barrier", &RcclComm::barrier, py::call_guard<py::gil_scoped_release>())
      .def("synchronize", &RcclComm::synchronize, py::call_guard<py::gil_scoped_release>())
      .def("wait_idle", &RcclComm::wait_idle, py::arg("seconds"), py::call_guard<py::gil_scoped_release>(),
           "poll the communication stream until it has drained: False after `seconds`, or on an asynchronous RCCL error")
      .def("abort", &RcclComm::abort, py::call_guard<py::gil_scoped_release>(), "ncclCommAbort: give up on a communicator whose work does not finish")
      .def("check_async_error", &RcclComm::check_async_error)
      .def("comm_count", &RcclComm::comm_count, "ranks of the communicator as RCCL itself reports them (ncclCommCount)")
      .def("close", &RcclComm::close, py::call_guard<py::gil_scoped_release>())
      .def("collectives", &RcclComm::collectives)
      .def("bytes", &RcclComm::bytes);
}
