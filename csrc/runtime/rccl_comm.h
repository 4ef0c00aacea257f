// RcclComm: a communicator of this framework's own, straight on the RCCL C API.
//
// One RcclComm = one ncclComm_t of one process-per-GPU rank + one HIP stream that belongs to it (non-blocking,
// normal priority - see the constructor) + the events that order that stream against the compute stream.  Collectives
// never block the host: `fork()` makes the communication stream wait for everything enqueued so far on the
// current compute stream, the collective is enqueued on the communication stream, `join()` makes the compute
// stream wait for everything enqueued on the communication stream.  The bucket reducer forks per bucket and
// joins once at the end of backward; the simple Python-facing calls fork and join around each operation.
//
// The unique id is created by group rank 0 (`unique_id()`), travels to the other ranks through whatever the
// launcher already has (the torch TCPStore behind the env-var rendezvous), and every rank then calls the
// constructor - the same bootstrap the reference gets implicitly from `Accelerator()` /
// `init_process_group` (/root/reference/test_data_parallelism.py:55, /root/reference/test_model_parallelism.py:166-171).
#pragma once
#include <torch/extension.h>
#include "../kernels/api.h"

#include <memory>
#include <string>
#include <vector>

struct ncclComm;
struct ihipStream_t;
struct ihipEvent_t;

namespace pdt {

class RcclComm {
 public:
  RcclComm(const std::string& unique_id, int64_t rank, int64_t world, int64_t device);
  ~RcclComm();
  RcclComm(const RcclComm&) = delete;
  RcclComm& operator=(const RcclComm&) = delete;

  static std::string unique_id();
  static int64_t rccl_version();

  int64_t rank() const { return rank_; }
  int64_t size() const { return world_; }
  int64_t device() const { return device_; }

  // stream ordering (see the header comment)
  void fork();
  void join();

  // enqueue on the communication stream; the caller brackets them with fork() / join()
  void all_reduce_async(const at::Tensor& t, bool average);
  void broadcast_async(const at::Tensor& t, int64_t root);
  void all_gather_async(const at::Tensor& out, const at::Tensor& in);
  void reduce_scatter_async(const at::Tensor& out, const at::Tensor& in, bool average);
  // enqueue the non-finite check of fp32 `t` on the communication stream: scale_state[2] = 1 if any element is inf / NaN
  // (the loss scaler's overflow flag; run right behind a bucket's all-reduce it overlaps backward like the all-reduce)
  void check_finite_async(const at::Tensor& t, const at::Tensor& scale_state);
  // enqueue the fused AdamW update of one bucket on the communication stream, right behind its all-reduce: runs of
  // precisions without a global skip decision (bf16, fp32) update a bucket while backward is still producing the next
  void adamw_async(int dtype, const PdtAdamW& args);
  // a group of sends / receives that must progress together: kinds[i] 0 = send, 1 = recv
  void p2p_async(const std::vector<int64_t>& kinds, const std::vector<at::Tensor>& tensors, const std::vector<int64_t>& peers);

  // fork + operation + join
  void all_reduce(const at::Tensor& t, bool average) { fork(); all_reduce_async(t, average); join(); }
  void broadcast(const at::Tensor& t, int64_t root) { fork(); broadcast_async(t, root); join(); }
  void all_gather(const at::Tensor& out, const at::Tensor& in) { fork(); all_gather_async(out, in); join(); }
  void reduce_scatter(const at::Tensor& out, const at::Tensor& in, bool average) { fork(); reduce_scatter_async(out, in, average); join(); }
  void p2p(const std::vector<int64_t>& kinds, const std::vector<at::Tensor>& tensors, const std::vector<int64_t>& peers) {
    fork(); p2p_async(kinds, tensors, peers); join();
  }
  void barrier();  // 1-element all-reduce + host wait for the communication stream

  int64_t comm_count() const;  // ncclCommCount
  // raises if RCCL has recorded an asynchronous error on this communicator (the watchdog's poll)
  void check_async_error();
  void synchronize();  // host wait for the communication stream
  // bounded host wait: true once the communication stream has drained, false after `seconds` or on an asynchronous
  // RCCL error (the start-up self test uses it so that a communicator that cannot make progress is abandoned, not waited for)
  bool wait_idle(double seconds);
  void abort();        // ncclCommAbort
  void close();        // orderly ncclCommDestroy; a communicator that was never closed is left to the process exit
  int64_t collectives() const { return launches_; }
  int64_t bytes() const { return bytes_; }

 private:
  void check_tensor(const at::Tensor& t) const;
  ncclComm* comm_ = nullptr;
  ihipStream_t* stream_ = nullptr;
  // rings of events: a fork / join never re-records an event that an earlier, possibly still pending, stream wait
  // refers to (one backward pass forks once per bucket)
  static constexpr int kEvents = 64;
  ihipEvent_t* ev_fork_[kEvents] = {};
  ihipEvent_t* ev_join_[kEvents] = {};
  uint64_t n_fork_ = 0, n_join_ = 0;
  at::Tensor barrier_word_;
  int64_t rank_ = 0, world_ = 1, device_ = 0, launches_ = 0, bytes_ = 0;
};

}  // namespace pdt
