// GradBucketReducer: the data-parallel gradient path.
//
// What the reference gets from torch's DDP reducer through `accelerator.prepare(model)`
// (/root/reference/test_data_parallelism.py:125-127) and the explicit
// `DistributedDataParallel(...)` + `no_sync()` of /root/reference/test_model_parallelism.py:248-253,292-296
// is rebuilt around one design decision: gradients are born inside a flat fp32 arena, and a bucket is a
// contiguous slice of that arena (one encoder layer ~ 50 MB for BERT-large, which keeps every one of the 7
// xGMI links of a GPU bandwidth-bound rather than latency-bound).  So there is no gather-into-bucket copy,
// no copy-back, and the all-reduce of a bucket is issued the moment the layer's backward has been enqueued:
// RCCL runs it on its own HIP stream (ordered after the producing kernels by an event) while the compute
// stream continues with the next layer's backward.
//
// The collective backend is either a c10d::ProcessGroup - "nccl" (= RCCL over xGMI) on GPUs, "gloo" for the
// CPU plumbing configuration; the identical C++ code path is exercised by the CPU multi-process tests - or
// this framework's own RcclComm (rccl_comm.h): RCCL's C API on a communication stream the reducer forks
// from the compute stream per bucket and joins once at the end of backward.
#include <torch/extension.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>
#include <torch/csrc/distributed/c10d/Types.hpp>
#include <torch/csrc/distributed/c10d/Work.hpp>

#include "rccl_comm.h"

#include <chrono>
#include <cstdlib>
#include <string>
#include <vector>

namespace py = pybind11;

namespace {

using at::Tensor;
using PG = c10::intrusive_ptr<c10d::ProcessGroup>;

class GradBucketReducer {
 public:
  GradBucketReducer(PG pg, Tensor arena, std::vector<int64_t> bounds, std::vector<int64_t> expected, bool average,
                    int64_t comm_dtype_code)
      : GradBucketReducer(std::move(pg), nullptr, std::move(arena), std::move(bounds), std::move(expected), average,
                          comm_dtype_code) {}
  GradBucketReducer(std::shared_ptr<pdt::RcclComm> comm, Tensor arena, std::vector<int64_t> bounds,
                    std::vector<int64_t> expected, bool average, int64_t comm_dtype_code)
      : GradBucketReducer(PG(), std::move(comm), std::move(arena), std::move(bounds), std::move(expected), average,
                          comm_dtype_code) {}

 private:
  GradBucketReducer(PG pg, std::shared_ptr<pdt::RcclComm> comm, Tensor arena, std::vector<int64_t> bounds,
                    std::vector<int64_t> expected, bool average, int64_t comm_dtype_code)
      : pg_(std::move(pg)), rccl_(std::move(comm)), arena_(std::move(arena)), bounds_(std::move(bounds)),
        expected_(std::move(expected)), average_(average) {
    TORCH_CHECK(bool(pg_) != bool(rccl_), "exactly one collective backend (process group or RcclComm) is needed");
    // wire dtype of the all-reduce: 0 = the arena's own (fp32), 1 = float16, 2 = bfloat16
    if (comm_dtype_code == 1) comm_dtype_ = at::kHalf;
    if (comm_dtype_code == 2) comm_dtype_ = at::kBFloat16;
    TORCH_CHECK(arena_.dim() == 1 && arena_.is_contiguous(), "gradient arena must be flat and contiguous");
    TORCH_CHECK(bounds_.size() >= 2 && bounds_.size() == expected_.size() + 1, "bounds/expected size mismatch");
    TORCH_CHECK(bounds_.front() == 0 && bounds_.back() <= arena_.numel(), "bucket bounds outside the arena");
    nb_ = (int)expected_.size();
    for (int i = 0; i < nb_; ++i) {
      TORCH_CHECK(bounds_[i] < bounds_[i + 1], "empty bucket ", i);
      views_.push_back(arena_.narrow(0, bounds_[i], bounds_[i + 1] - bounds_[i]));
    }
    world_ = rccl_ ? (int)rccl_->size() : pg_->getSize();
    is_nccl_ = rccl_ ? true : pg_->getBackendName() == "nccl";
    // (only when the backend positively is gloo: any other group gets the GPU tensor as it is)
    host_staged_ = !rccl_ && arena_.is_cuda() && pg_->getBackendName() == "gloo";
    if (rccl_) TORCH_CHECK(arena_.is_cuda() && arena_.get_device() == rccl_->device(), "arena is not on the communicator's GPU");
    // PDT_FORCE_COLLECTIVES=1 issues the collectives even in a 1-rank group (exercises the RCCL call path on
    // a single-GPU test box)
    const char* force = std::getenv("PDT_FORCE_COLLECTIVES");
    force_ = force && force[0] == '1';
    // PDT_COMM_ALGO=rs_ag (own communicator only): a bucket is reduced as reduce-scatter + all-gather, both in place -
    // the two halves of an all-reduce as separate RCCL calls, each rank owning the 1/world shard at its rank offset.
    // On a fully connected xGMI node every rank then exchanges one shard with each of its 7 peers per phase; which of
    // the two forms RCCL runs faster on a given node is a measurement (bench.py --comm-algo), not a given.
    const char* algo = std::getenv("PDT_COMM_ALGO");
    rs_ag_ = rccl_ && algo && std::string(algo) == "rs_ag";
    reset();
  }

 public:
  // Start a backward pass.  sync=false is the no_sync() mode: gradients only accumulate locally.
  void begin(bool sync) {
    reset();
    sync_ = sync;
    active_ = true;
  }

  // `count` producers of bucket `b` have enqueued their gradient kernels.
  void mark_ready(int64_t b, int64_t count) {
    if (!active_ || !sync_) return;
    TORCH_CHECK(b >= 0 && b < nb_, "bucket index out of range");
    ready_[b] += count;
    TORCH_CHECK(ready_[b] <= expected_[b], "bucket ", b, " marked ready more often than it has producers");
    launch_ready_prefix();
  }

  // Fused AdamW per bucket, on the communication stream right behind the bucket's collective (own communicator, fp32
  // wire).  For precisions without a global skip decision only (bf16 / fp32: the caller decides); a bucket that is
  // reduced as a left-over at the end of backward (nobody completed it: its gradient slice may hold unwritten
  // segments the optimizer has yet to zero) is left to the optimizer - optimizer_applied() says which.
  void set_optimizer(Tensor p, Tensor m, Tensor v, c10::optional<Tensor> p16, Tensor lr, Tensor step, Tensor scale_state,
                     double beta1, double beta2, double eps, double weight_decay, bool correct_bias) {
    TORCH_CHECK(p.numel() == arena_.numel() && m.numel() == arena_.numel() && v.numel() == arena_.numel(),
                "optimizer state must cover the gradient arena");
    TORCH_CHECK(p.scalar_type() == at::kFloat && m.scalar_type() == at::kFloat && v.scalar_type() == at::kFloat, "fp32 state expected");
    TORCH_CHECK(step.scalar_type() == at::kInt && scale_state.scalar_type() == at::kFloat && scale_state.numel() >= 4 &&
                lr.scalar_type() == at::kFloat, "lr / step / scale_state: device scalars of the fused optimizer");
    opt_p_ = std::move(p); opt_m_ = std::move(m); opt_v_ = std::move(v); opt_lr_ = std::move(lr); opt_step_ = std::move(step);
    opt_state_ = std::move(scale_state);
    opt_dtype_ = PDT_F16;
    if (p16.has_value()) {
      TORCH_CHECK(p16->numel() == arena_.numel(), "16-bit copy must cover the arena");
      opt_p16_ = *p16;
      opt_dtype_ = p16->scalar_type() == at::kBFloat16 ? PDT_BF16 : PDT_F16;
    }
    opt_args_ = PdtAdamW{};
    opt_args_.beta1 = (float)beta1; opt_args_.beta2 = (float)beta2; opt_args_.eps = (float)eps;
    opt_args_.weight_decay = (float)weight_decay; opt_args_.correct_bias = correct_bias ? 1 : 0;
    opt_set_ = true;
  }
  void clear_optimizer() { opt_set_ = false; }
  bool applies_optimizer() const { return rccl_ && opt_set_ && !comm_dtype_.has_value() && !host_staged_ && (world_ > 1 || force_); }
  // per bucket: did the last finished backward pass update it on the communication stream?
  std::vector<bool> optimizer_applied() const { return applied_; }

  // End of backward: buckets nobody marked (parameters that took no part in this graph - their arena slice
  // is still zero) are reduced too, in index order, so every rank issues the same sequence of collectives;
  // then the compute stream is made to wait for all of them.
  void finish() {
    if (!active_) return;
    active_ = false;
    if (!sync_) return;
    for (int b = next_; b < nb_; ++b) launch(b, /*leftover=*/true);
    next_ = nb_;
    const auto t0 = std::chrono::steady_clock::now();
    if (rccl_ && launched_any_) rccl_->join();  // the compute stream waits for every bucket; the host does not
    for (int b = 0; b < nb_; ++b) {
      if (!rccl_) {
        if (!works_[b]) continue;
        works_[b]->wait();
        works_[b].reset();
      }
      if (staged_[b].defined()) {
        views_[b].copy_(staged_[b]);
        staged_[b] = Tensor();
      }
      if (average_ && !native_avg()) views_[b].div_((double)world_);
    }
    wait_us_total_ += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    ++steps_;
  }

  // rank `root`'s values of `t` (any flat tensor, e.g. the parameter arena) to everyone, in chunks
  void broadcast(Tensor t, int64_t root, int64_t chunk_elems) {
    TORCH_CHECK(t.dim() == 1 && t.is_contiguous(), "broadcast expects a flat contiguous tensor");
    if (world_ == 1 && !force_) return;
    if (rccl_) {
      rccl_->fork();
      for (int64_t o = 0; o < t.numel(); o += chunk_elems)
        rccl_->broadcast_async(t.narrow(0, o, std::min(chunk_elems, t.numel() - o)), root);
      rccl_->join();
      return;
    }
    if (host_staged_) {
      Tensor host = t.to(at::kCPU);
      std::vector<Tensor> v{host};
      c10d::BroadcastOptions opts;
      opts.rootRank = root;
      pg_->broadcast(v, opts)->wait();
      t.copy_(host);
      return;
    }
    std::vector<c10::intrusive_ptr<c10d::Work>> ws;
    for (int64_t o = 0; o < t.numel(); o += chunk_elems) {
      std::vector<Tensor> v{t.narrow(0, o, std::min(chunk_elems, t.numel() - o))};
      c10d::BroadcastOptions opts;
      opts.rootRank = root;
      ws.push_back(pg_->broadcast(v, opts));
    }
    for (auto& w : ws) w->wait();
  }

  // Loss-scaler state {scale, tracker, found_inf, ...} of an fp16 run: with the framework's own communicator every
  // bucket is checked for inf / NaN on the communication stream right behind its all-reduce - on the REDUCED
  // values, so every rank takes the same decision - instead of in one pass over the whole arena after backward.
  void set_overflow_state(Tensor state) { overflow_state_ = std::move(state); }
  bool checks_overflow() const {
    return rccl_ && overflow_state_.defined() && !comm_dtype_.has_value() && (world_ > 1 || force_);
  }

  int64_t num_buckets() const { return nb_; }
  int64_t world_size() const { return world_; }
  std::vector<int64_t> launch_order() const { return order_; }
  py::dict stats() const {
    py::dict d;
    d["steps"] = steps_;
    d["collectives"] = launches_;
    d["bytes"] = bytes_;
    d["host_wait_us"] = wait_us_total_;
    return d;
  }

 private:
  bool native_avg() const { return is_nccl_; }

  void reset() {
    ready_.assign(nb_, 0);
    works_.assign(nb_, nullptr);
    staged_.assign(nb_, Tensor());
    order_.clear();
    applied_.assign(nb_, false);
    next_ = 0;
    launched_any_ = false;
  }
  void launch_ready_prefix() {
    while (next_ < nb_ && ready_[next_] >= expected_[next_]) launch(next_++);
  }
  void launch(int b, bool leftover = false) {
    order_.push_back(b);
    if (world_ == 1 && !force_) return;  // nothing to exchange; gradients are already the global ones
    Tensor buf = views_[b];
    if (comm_dtype_.has_value() && *comm_dtype_ != buf.scalar_type()) {
      staged_[b] = buf.to(*comm_dtype_);  // halve the bytes on the wire; accumulation stays fp32 in the arena
      buf = staged_[b];
    }
    if (host_staged_) {
      // several ranks rehearsing on one GPU over gloo: the bucket crosses gloo through host memory (this copy waits
      // for the kernels that produced the bucket); finish() copies the reduced values back
      staged_[b] = buf.to(at::kCPU);
      buf = staged_[b];
    }
    if (rccl_) {
      // ordered after the kernels that produced the bucket (everything enqueued on the compute stream so far)
      rccl_->fork();
      if (rs_ag_ && buf.numel() % world_ == 0) {
        Tensor shard = buf.narrow(0, (buf.numel() / world_) * rccl_->rank(), buf.numel() / world_);
        rccl_->reduce_scatter_async(shard, buf, average_);
        rccl_->all_gather_async(buf, shard);
      } else {
        rccl_->all_reduce_async(buf, average_);
      }
      if (checks_overflow()) rccl_->check_finite_async(views_[b], overflow_state_);
      if (applies_optimizer() && !leftover) {
        PdtAdamW a = opt_args_;
        const int64_t o = bounds_[b];
        a.p = opt_p_.data_ptr<float>() + o; a.m = opt_m_.data_ptr<float>() + o; a.v = opt_v_.data_ptr<float>() + o;
        a.g = arena_.data_ptr<float>() + o;
        a.p16 = opt_p16_.defined() ? (void*)((char*)opt_p16_.data_ptr() + 2 * o) : nullptr;
        a.n = bounds_[b + 1] - o;
        a.lr = opt_lr_.data_ptr<float>(); a.step = opt_step_.data_ptr<int32_t>(); a.scale_state = opt_state_.data_ptr<float>();
        rccl_->adamw_async(opt_dtype_, a);
        applied_[b] = true;
      }
      launched_any_ = true;
    } else {
      c10d::AllreduceOptions opts;
      opts.reduceOp = (average_ && native_avg()) ? c10d::ReduceOp::AVG : c10d::ReduceOp::SUM;
      std::vector<Tensor> v{buf};
      works_[b] = pg_->allreduce(v, opts);
    }
    ++launches_;
    bytes_ += buf.numel() * buf.element_size();
  }

  PG pg_;
  std::shared_ptr<pdt::RcclComm> rccl_;
  Tensor arena_, overflow_state_;
  Tensor opt_p_, opt_m_, opt_v_, opt_p16_, opt_lr_, opt_step_, opt_state_;
  PdtAdamW opt_args_{};
  int opt_dtype_ = PDT_F16;
  bool opt_set_ = false;
  std::vector<bool> applied_;
  std::vector<int64_t> bounds_, expected_;
  bool average_;
  c10::optional<at::ScalarType> comm_dtype_;
  int nb_ = 0, world_ = 1, next_ = 0;
  bool is_nccl_ = false, host_staged_ = false, sync_ = true, active_ = false, force_ = false, launched_any_ = false;
  bool rs_ag_ = false;
  std::vector<Tensor> views_, staged_;
  std::vector<int64_t> ready_, order_;
  std::vector<c10::intrusive_ptr<c10d::Work>> works_;
  int64_t steps_ = 0, launches_ = 0, bytes_ = 0, wait_us_total_ = 0;
};

}  // namespace

void bind_reducer(py::module_& m) {
  py::class_<GradBucketReducer>(m, "GradBucketReducer")
      .def(py::init<PG, at::Tensor, std::vector<int64_t>, std::vector<int64_t>, bool, int64_t>(),
           py::arg("process_group"), py::arg("arena"), py::arg("bounds"), py::arg("expected"), py::arg("average") = true,
           py::arg("comm_dtype") = 0)
      .def(py::init<std::shared_ptr<pdt::RcclComm>, at::Tensor, std::vector<int64_t>, std::vector<int64_t>, bool, int64_t>(),
           py::arg("rccl_comm"), py::arg("arena"), py::arg("bounds"), py::arg("expected"), py::arg("average") = true,
           py::arg("comm_dtype") = 0)
      .def("begin", &GradBucketReducer::begin, py::arg("sync") = true)
      .def("mark_ready", &GradBucketReducer::mark_ready, py::arg("bucket"), py::arg("count") = 1)
      .def("finish", &GradBucketReducer::finish, py::call_guard<py::gil_scoped_release>())
      .def("broadcast", &GradBucketReducer::broadcast, py::arg("tensor"), py::arg("root") = 0,
           py::arg("chunk_elems") = (int64_t)16 << 20, py::call_guard<py::gil_scoped_release>())
      .def("set_optimizer", &GradBucketReducer::set_optimizer)
      .def("clear_optimizer", &GradBucketReducer::clear_optimizer)
      .def("applies_optimizer", &GradBucketReducer::applies_optimizer)
      .def("optimizer_applied", &GradBucketReducer::optimizer_applied)
      .def("set_overflow_state", &GradBucketReducer::set_overflow_state)
      .def("checks_overflow", &GradBucketReducer::checks_overflow)
      .def("num_buckets", &GradBucketReducer::num_buckets)
      .def("world_size", &GradBucketReducer::world_size)
      .def("launch_order", &GradBucketReducer::launch_order)
      .def("stats", &GradBucketReducer::stats);
}
