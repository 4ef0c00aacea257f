"""Data parallelism: replicated parameters + bucketed gradient all-reduce overlapped with backward.

This is the role torch ``DistributedDataParallel`` plays in the reference
(/root/reference/test_data_parallelism.py:125-127 through Accelerate; explicitly with
``device_ids=None, output_device=None, find_unused_parameters=True`` and ``no_sync()`` at
/root/reference/test_model_parallelism.py:248-253,292-296).  Differences by design:

* buckets are contiguous slices of the model's flat gradient arena (one encoder layer each), reduced
  in place by the C++ ``GradBucketReducer`` (csrc/runtime/reducer.cpp) through RCCL ("nccl" backend)
  or gloo (CPU plumbing);
* readiness is signalled per arena *segment* by the native backward nodes (or per parameter by autograd
  hooks on the PyTorch path) - there is no per-iteration autograd-graph walk for unused parameters: a
  bucket nobody completed is still reduced at the end of backward, so ``find_unused_parameters`` is
  always "on" at zero cost;
* no float buffers exist in BERT, so there is no per-forward buffer broadcast.
"""
from __future__ import annotations

import contextlib
import os
import time
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from ..ops.arena import ParamArena


class _PyReducer:
    """Pure-Python mirror of the C++ reducer (used only if the extension cannot be loaded on a CPU box)."""

    def __init__(self, pg, arena_grad, bounds, expected, average=True, comm_dtype=0):
        self.pg, self.bounds, self.expected, self.average = pg, bounds, expected, average
        self.views = [arena_grad[bounds[i]:bounds[i + 1]] for i in range(len(expected))]
        self.world = dist.get_world_size(pg)
        self._reset()
        self.sync, self.active = True, False
        self._stats = {"steps": 0, "collectives": 0, "bytes": 0, "host_wait_us": 0}

    def _reset(self):
        self.ready = [0] * len(self.expected)
        self.works = [None] * len(self.expected)
        self.order: List[int] = []
        self.next = 0

    def begin(self, sync=True):
        self._reset()
        self.sync, self.active = sync, True

    def _launch(self, b):
        self.order.append(b)
        if self.world == 1:
            return
        self.works[b] = dist.all_reduce(self.views[b], group=self.pg, async_op=True)
        self._stats["collectives"] += 1
        self._stats["bytes"] += self.views[b].numel() * 4

    def mark_ready(self, bucket, count=1):
        if not (self.active and self.sync):
            return
        self.ready[bucket] += count
        while self.next < len(self.expected) and self.ready[self.next] >= self.expected[self.next]:
            self._launch(self.next)
            self.next += 1

    def finish(self):
        if not self.active:
            return
        self.active = False
        if not self.sync:
            return
        for b in range(self.next, len(self.expected)):
            self._launch(b)
        for b, w in enumerate(self.works):
            if w is not None:
                w.wait()
                if self.average:
                    self.views[b].div_(self.world)
        self._stats["steps"] += 1

    def broadcast(self, tensor, root=0, chunk_elems=16 << 20):
        if self.world > 1:
            dist.broadcast(tensor, src=dist.get_global_rank(self.pg, root) if self.pg is not None else root, group=self.pg)

    def launch_order(self):
        return list(self.order)

    def stats(self):
        return dict(self._stats)

    def num_buckets(self):
        return len(self.expected)


def make_reducer(pg, arena: ParamArena, expected: List[int], average: bool = True, comm_dtype: Optional[torch.dtype] = None,
                 comm_backend: Optional[str] = None):
    """The C++ bucket reducer over this framework's own RCCL communicator spanning the ranks of `pg` (GPU runs with
    one rank per device; the default) or over `pg` itself (``comm_backend="c10d"`` / ``PDT_COMM=c10d``: torch's
    "nccl" = RCCL group; always for gloo groups).  The reducer carries ``backend_label`` ("rccl", "c10d:nccl",
    "c10d:gloo") and ``rccl_comm`` (the RcclComm or None)."""
    code = {None: 0, torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[comm_dtype]
    from ..ops import available, native
    from .comm import backend_choice, create_rccl_comm

    label = "c10d:" + (dist.get_backend(pg) if dist.is_initialized() else "none")
    if available():
        comm = None
        # (PDT_COMM_FORCE_RCCL=1: also try on a gloo group - several ranks sharing one GPU - where RCCL refuses the
        # duplicate device; the tests use it to run the agreed fall-back)
        if backend_choice(comm_backend) == "rccl" and arena.device.type == "cuda" and (
                not dist.is_initialized() or dist.get_backend(pg) == "nccl" or os.environ.get("PDT_COMM_FORCE_RCCL") == "1"):
            comm = create_rccl_comm(None if pg is dist.group.WORLD else pg, arena.device)  # None: failed its self test
        if comm is not None:
            red = native().GradBucketReducer(comm, arena.grad, list(arena.bucket_bounds), list(expected), average, code)
            label = "rccl"
        else:
            red = native().GradBucketReducer(pg, arena.grad, list(arena.bucket_bounds), list(expected), average, code)
        return red, label, comm
    if arena.device.type == "cuda":
        native()  # raises with the build hint: never fall back silently on a GPU box
    return _PyReducer(pg, arena.grad, list(arena.bucket_bounds), list(expected), average, code), label, None


class DataParallel(nn.Module):
    """Wrap an arena-backed model for data-parallel training over `process_group`."""

    def __init__(self, module: nn.Module, arena: ParamArena, process_group=None, comm_dtype: Optional[torch.dtype] = None,
                 broadcast_from: Optional[int] = 0, comm_backend: Optional[str] = None):
        super().__init__()
        self.module = module
        self.arena = arena
        self.pg = process_group if process_group is not None else dist.group.WORLD
        self.world_size = dist.get_world_size(self.pg)
        self.require_sync = True
        self.auto_arm = True  # forward() arms the reducer; a pipeline driver arms it itself before the last backward
        # True: the window is closed by an explicit finish_backward() instead of autograd's end-of-backward callback
        # (a driver whose armed window spans several backward() calls, e.g. the tower-parallel layout)
        self.manual_finish = False
        self._armed = False
        self._callback_queued = False
        self.measure_exposed = os.environ.get("PDT_COMM_TIMING", "0") not in ("", "0")
        self._exposed: list = []
        runtime = getattr(module, "runtime", None)
        self.native_mode = runtime is not None
        nb = len(arena.bucket_bounds) - 1
        if self.native_mode:
            expected = list(arena.bucket_segments)
            seg_buckets = [s.buckets for s in arena.segments]
            runtime.segment_done = lambda seg: [self._mark(b) for b in seg_buckets[seg]]
        else:
            expected = [0] * nb
            for seg in arena.segments:
                for _, p in seg.params:
                    if p.requires_grad:
                        for b in seg.buckets:  # (a large segment is cut into several buckets: all of them wait for it)
                            expected[b] += 1
                        p.register_post_accumulate_grad_hook(self._make_hook(seg.buckets))
            expected = [max(1, e) for e in expected]
        self.reducer, self.backend_label, self.rccl_comm = make_reducer(self.pg, arena, expected, True, comm_dtype,
                                                                        comm_backend)
        self.comm_dtype = comm_dtype
        if dist.get_rank(self.pg) == 0 and (self.world_size > 1 or os.environ.get("PDT_COMM_VERBOSE")):
            print(self.describe(), flush=True)  # one line per data-parallel group: what a first-run log needs
        if self.world_size > 1:
            self._verify_replicas()
        if broadcast_from is not None and self.world_size > 1:
            # rank-0 parameters to everyone (what DDP's constructor does), then refresh the 16-bit copy
            self.reducer.broadcast(arena.master, broadcast_from)
            arena.sync_compute_copy()

    def _verify_replicas(self) -> None:
        """Every rank must hold the same parameters in the same arena layout (torch DDP's constructor verifies
        parameter shapes across ranks the same way); a mismatch would otherwise surface as a hang or as silently
        mixed-up gradients in the first all-reduce."""
        import zlib

        layout = ";".join(f"{n}:{o}:{k}" for n, (o, k) in sorted(self.arena.offsets.items()))
        mine = torch.tensor([self.arena.numel, len(self.arena.offsets), len(self.arena.bucket_bounds) - 1,
                             zlib.crc32(layout.encode())], dtype=torch.int64,
                            device=self.arena.device if dist.get_backend(self.pg) == "nccl" else "cpu")
        every = [torch.empty_like(mine) for _ in range(self.world_size)]
        dist.all_gather(every, mine, group=self.pg)
        bad = [r for r, t in enumerate(every) if not torch.equal(t, every[0])]
        if bad:
            raise RuntimeError(f"data-parallel replicas differ: ranks {bad} do not match rank 0 "
                               f"(arena elements / parameters / buckets / layout hash: "
                               f"{[t.tolist() for t in every]})")

    def describe(self) -> str:
        """The gradient path of this replica group in one line: transport, RCCL version and communicator size as RCCL
        reports them, wire dtype, collective per bucket, and the buckets' sizes in backward-completion order."""
        b = self.arena.bucket_bounds
        mb = [round((b[i + 1] - b[i]) * 4 / 2 ** 20, 1) for i in range(len(b) - 1)]
        version = ""
        try:
            from ..ops import native

            v = int(native().RcclComm.rccl_version())
            version = f" RCCL {v // 10000}.{v // 100 % 100}.{v % 100}"
        except Exception:  # noqa: BLE001 - CPU-only build / no extension: nothing to report
            pass
        wire = {None: "fp32", torch.float32: "fp32", torch.float16: "fp16", torch.bfloat16: "bf16"}[self.comm_dtype]
        ranks = dist.get_process_group_ranks(self.pg) if dist.is_initialized() else [0]
        return (f"[dp] ranks {list(ranks)} transport {self.backend_label}{version} ncclCommCount "
                f"{int(self.rccl_comm.comm_count()) if self.rccl_comm is not None else 'n/a'} wire {wire} "
                f"per-bucket {os.environ.get('PDT_COMM_ALGO', 'allreduce')} | {len(mb)} buckets, {sum(mb):.0f} MB fp32 "
                f"gradients, MB in arena order: {mb}")

    def attach_scaler(self, scaler) -> bool:
        """Let the reducer check every reduced bucket for inf / NaN on the communication stream (fp16 runs on the
        framework's own communicator).  True if it will, i.e. the optimizer can skip its own pass over the arena."""
        if scaler is None or not scaler.enabled or not hasattr(self.reducer, "set_overflow_state"):
            return False
        if os.environ.get("PDT_BUCKET_OVERFLOW_CHECK", "1") == "0":  # one pass over the arena in the optimizer instead
            return False
        self.reducer.set_overflow_state(scaler.state)
        return bool(self.reducer.checks_overflow())

    def rccl_ranks(self) -> int:
        """Ranks that exchange this model's gradients over RCCL: what RCCL itself reports for the framework's own
        communicator (ncclCommCount), the group size for torch's "nccl" group, 0 for gloo."""
        if self.rccl_comm is not None:
            return int(self.rccl_comm.comm_count())
        return self.world_size if self.backend_label == "c10d:nccl" else 0

    def check_replicas(self) -> None:
        """Raise unless every rank of the group holds bit-identical master parameters (host sync; collective).
        Data-parallel replicas start from one broadcast and apply the same all-reduced gradients with the same
        deterministic optimizer kernel, so any difference means a gradient that missed its reduction."""
        if self.world_size == 1:
            return
        m = self.arena.master
        bits = torch.sum(m.view(torch.int32), dtype=torch.int64)  # wrap-around sum of the raw words
        mine = torch.stack([bits, torch.sum(m, dtype=torch.float64).view(torch.int64)])
        if dist.get_backend(self.pg) != "nccl":
            mine = mine.cpu()
        every = [torch.empty_like(mine) for _ in range(self.world_size)]
        dist.all_gather(every, mine, group=self.pg)
        bad = [r for r, t in enumerate(every) if not torch.equal(t, every[0])]
        if bad:
            sums = [float(t[1:].view(torch.float64)) for t in every]
            raise RuntimeError(f"data-parallel replicas diverged: parameters of group ranks {bad} differ from rank 0 "
                               f"(parameter sums per rank: {sums})")

    # -- reducer plumbing ----------------------------------------------------------------------------
    def _make_hook(self, buckets):
        def hook(_p):
            for b in buckets:
                self._mark(b)
        return hook

    def _mark(self, bucket: int) -> None:
        if not self._armed:
            return
        if not self._callback_queued and not self.manual_finish:
            # end-of-backward callback, queued from inside the running backward pass
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
            self._callback_queued = True
        self.reducer.mark_ready(bucket, 1)

    def _finish(self) -> None:
        self._callback_queued = False
        if self._armed:
            self._armed = False
            if not (self.measure_exposed and self.require_sync):
                self.reducer.finish()
            elif self.arena.device.type == "cuda":
                # events on the compute stream around the waits for the collectives: the time the stream
                # stands still (plus the averaging kernels) is communication that backward did not hide
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.reducer.finish()
                e1.record()
                self._exposed.append((e0, e1))
            else:
                t0 = time.perf_counter()
                self.reducer.finish()
                self._exposed.append((time.perf_counter() - t0) * 1e3)

    @property
    def reduction_pending(self) -> bool:
        """True between arming (forward) and the end-of-backward callback."""
        return self._armed

    def comm_exposed_ms(self) -> Optional[float]:
        """Mean per-backward milliseconds of gradient communication NOT hidden behind backward since the last
        call (needs ``measure_exposed``; call it after a host sync, e.g. next to ``loss.item()``)."""
        if not self._exposed:
            return None
        vals = [v if isinstance(v, float) else (v[0].elapsed_time(v[1]) if v[1].query() else None) for v in self._exposed]
        self._exposed = [v for v, ms in zip(self._exposed, vals) if ms is None]
        vals = [v for v in vals if v is not None]
        return sum(vals) / len(vals) if vals else None

    def prepare_backward(self) -> None:
        """Arm the reducer for the next backward pass (called from forward)."""
        self.reducer.begin(self.require_sync)
        self._armed = True
        self._callback_queued = False

    def finish_backward(self) -> None:
        """Explicit end of backward for callers that do not go through autograd's callback (pipeline)."""
        self._finish()

    @contextlib.contextmanager
    def no_sync(self):
        """Gradients of backward passes inside the context only accumulate locally."""
        old, self.require_sync = self.require_sync, False
        try:
            yield
        finally:
            self.require_sync = old

    def forward(self, *args, **kwargs):
        if self.auto_arm and self.training and torch.is_grad_enabled():
            self.prepare_backward()
        return self.module(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)
