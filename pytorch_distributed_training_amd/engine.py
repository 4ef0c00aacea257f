"""Engine: the one object a training script talks to.

It offers the surface the reference's data-parallel script uses on ``accelerate.Accelerator``
(/root/reference/test_data_parallelism.py:55-60,97,125-127,146,160-161,166): ``device``,
``local_process_index``, ``process_index``, ``distributed_type``, ``prepare``, ``backward``, ``gather``,
``print`` - and what the model-parallel script does by hand (/root/reference/test_model_parallelism.py:
166-171 process group, 248-253 DDP wrap, 254-269 samplers, 292-296 ``no_sync``, 24-37 ``gather``).

``prepare(model)`` moves the parameters into a flat arena, turns on the native gfx950 path for
fp16/bf16 training on a GPU, and wraps the model for data parallelism when world_size > 1.
``prepare(optimizer)`` swaps a reference-style ``AdamW`` for the fused arena optimizer.
``prepare(dataloader)`` shards batches across ranks with Accelerate's index semantics and moves them to
the device asynchronously.
"""
from __future__ import annotations

import contextlib
import os
from typing import Any, Optional

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.utils.data import DataLoader

from . import launch
from .data.loader import DeviceLoader, shard_dataloader
from .ops.arena import ParamArena
from .optim import FusedAdamW, HFAdamW, LossScaler
from .parallel.dp import DataParallel
from .utils.debug import debug_enabled


class DistributedType:
    NO = "NO"
    MULTI_GPU = "MULTI_GPU"
    MULTI_CPU = "MULTI_CPU"
    TPU = "TPU"  # never produced; kept so that reference-style comparisons evaluate


class EngineOptimizer:
    """Handle returned by ``prepare(optimizer)``; mirrors the torch optimizer surface the scripts use."""

    def __init__(self, optimizer, engine: "Engine"):
        self.optimizer, self.engine = optimizer, engine

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    def step(self, **kw):
        debug = debug_enabled()
        if debug:
            for m in self.engine._models:
                if isinstance(m, DataParallel) and m.reduction_pending:
                    raise RuntimeError("optimizer.step() while a gradient reduction is still armed: the last "
                                       "backward pass did not reach its end-of-backward callback")
        self.optimizer.step(**kw)
        if debug and self.engine._arena is not None:
            self.engine._arena.check_guards()

    def zero_grad(self, set_to_none: bool = False):
        self.optimizer.zero_grad()

    def state_dict(self):
        return self.optimizer.state_dict()

    def load_state_dict(self, sd):
        self.optimizer.load_state_dict(sd)

    @property
    def step_was_skipped(self) -> bool:  # host sync
        sc = getattr(self.optimizer, "scaler", None)
        return bool(sc is not None and sc.last_step_skipped())


class Engine:
    def __init__(self, fp16: bool = False, cpu: bool = False, bf16: bool = False, seed: int = 42,
                 comm_dtype: Optional[torch.dtype] = None, min_bucket_mb: float = 4.0, process_group=None,
                 init_scale: float = 2.0 ** 16, native: Optional[bool] = None, comm_backend: Optional[str] = None,
                 update_in_backward: bool = True):
        if fp16 and bf16:
            raise ValueError("choose one of fp16 / bf16")
        self.env, self.device = launch.init_distributed(cpu=cpu)
        self.pg = process_group
        self.num_processes = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.process_index = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.local_process_index = self.env.local_rank
        self.is_main_process = self.process_index == 0
        self.is_local_main_process = self.local_process_index == 0
        self.use_fp16, self.use_bf16 = fp16, bf16
        self.mixed_precision = "fp16" if fp16 else ("bf16" if bf16 else "no")
        on_gpu = self.device.type == "cuda"
        self.compute_dtype = (torch.float16 if fp16 else torch.bfloat16) if (fp16 or bf16) else None
        # the native kernels are the 16-bit GPU path; fp32 (and CPU) runs use the PyTorch path
        self.native = (on_gpu and self.compute_dtype is not None) if native is None else native
        if self.native and not on_gpu:
            raise ValueError("the native path needs a GPU")
        if self.num_processes == 1:
            self.distributed_type = DistributedType.NO
        else:
            self.distributed_type = DistributedType.MULTI_GPU if on_gpu else DistributedType.MULTI_CPU
        # Persistent launches of the large GEMMs (256 workgroups drawing output tiles from ticket counters) stay on in
        # multi-rank runs: a workgroup never waits for anything but its counter, so communication kernels of other streams
        # (bucket all-reduces, pipeline send/recv) at worst start when the next GEMM workgroup retires, and GEMM workgroups
        # that find their CU taken simply leave their tiles to the others.  Measured in the 1-rank RCCL rehearsal (every
        # bucket collective really issued): 262.7 ms per step with persistent launches, 266.7 ms with one workgroup per
        # tile (PDT_GEMM_PERSIST=0 / bench.py --gemm-persist 0, a row of bench/scale_sweep.sh).
        self.seed = seed
        self.comm_dtype = comm_dtype
        # gradient all-reduce through c10d's "nccl" backend (default) or the framework's own RcclComm ("rccl")
        self.comm_backend = comm_backend
        # bf16 / fp32 data-parallel runs on the framework's own communicator: AdamW of a bucket right behind its
        # all-reduce, inside backward (FusedAdamW.attach_reducer).  False for drivers that touch the reduced gradients
        # between backward and the optimizer step (gradient clipping) or re-use weights after a bucket completed.
        self.update_in_backward = update_in_backward
        self.min_bucket_elems = int(min_bucket_mb * (1 << 20) / 4)
        self.scaler = LossScaler(self.device, enabled=fp16 and on_gpu, init_scale=init_scale)
        self._models = []
        self._fused = []
        self._arena: Optional[ParamArena] = None

    # ------------------------------------------------------------------------------------------
    @property
    def state(self):
        return self

    def print(self, *args, **kw):
        if self.is_main_process:
            print(*args, **kw)

    def wait_for_everyone(self):
        if self.num_processes > 1:
            dist.barrier(group=self.pg)

    # ------------------------------------------------------------------------------------------
    def prepare(self, *objs):
        out = []
        # models first: optimizers need the arena
        prepared = {}
        for o in objs:
            if isinstance(o, nn.Module):
                prepared[id(o)] = self.prepare_model(o)
        for o in objs:
            if id(o) in prepared:
                out.append(prepared[id(o)])
            elif isinstance(o, DataLoader):
                out.append(self.prepare_data_loader(o))
            elif isinstance(o, (torch.optim.Optimizer, FusedAdamW)):
                out.append(self.prepare_optimizer(o))
            else:
                out.append(o)
        return out[0] if len(out) == 1 else tuple(out)

    def prepare_model(self, model: nn.Module, arena_segments=None) -> nn.Module:
        model.to(self.device)
        use16 = self.compute_dtype if (self.native or self.device.type == "cuda") else None
        if not self.native:
            use16 = None
        arena = ParamArena(model, self.device, use16, self.min_bucket_elems, arena_segments)
        self._arena = arena
        model.arena = arena
        if self.native:
            if not hasattr(model, "enable_native"):
                raise TypeError("native 16-bit training needs a model with enable_native() (see models/bert.py)")
            # dropout streams: batches that carry global sample ids are keyed by (seed, step, sample) - the same masks
            # whatever the parallel layout; other batches by (seed, rank, micro-batch): distinct per rank, reproducible
            model.enable_native(arena, seed=self.seed * 100003 + self.process_index, global_seed=self.seed * 100003)
        if self.num_processes > 1:
            model = DataParallel(model, arena, self.pg, self.comm_dtype, comm_backend=self.comm_backend)
        self._models.append(model)
        return model

    def prepare_optimizer(self, optimizer):
        if isinstance(optimizer, FusedAdamW):
            optimizer.scaler = self.scaler
            return EngineOptimizer(optimizer, self)
        if self._arena is not None and isinstance(optimizer, (HFAdamW, torch.optim.AdamW)):
            g = optimizer.param_groups[0]
            fused = FusedAdamW(self._arena, lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"],
                               weight_decay=g["weight_decay"], correct_bias=g.get("correct_bias", True),
                               scaler=self.scaler)
            dps = [m for m in self._models if isinstance(m, DataParallel) and m.arena is self._arena]
            # (the overflow check rides behind the bucket all-reduces when the reducer can do it)
            fused.grads_checked_by_reducer = bool(dps) and all(m.attach_scaler(self.scaler) for m in dps)
            # (native path only: there a bucket is marked after the whole layer's backward - data gradients included - has
            # been enqueued; a per-parameter autograd hook may fire before the data gradient that still reads the weight)
            if (self.update_in_backward and len(dps) == 1 and dps[0].native_mode and dps[0].auto_arm
                    and not dps[0].manual_finish):
                fused.attach_reducer(dps[0].reducer)
            self._fused.append(fused)
            return EngineOptimizer(fused, self)
        return EngineOptimizer(optimizer, self)

    def prepare_data_loader(self, loader: DataLoader):
        sharded = shard_dataloader(loader, self.num_processes, self.process_index, seed=self.seed)
        return DeviceLoader(sharded, self.device)

    # ------------------------------------------------------------------------------------------
    def backward(self, loss: torch.Tensor, **kw):
        """Loss-scaled backward.  The bucket all-reduces run inside this call, overlapped with it."""
        if self.scaler.enabled:
            loss = loss * self.scaler.scale_tensor().to(loss.dtype)
        loss.backward(**kw)

    def clip_grad_norm_(self, model: nn.Module, max_norm: float, group=None) -> torch.Tensor:
        """Scale the gradients of `model` so that their global L2 norm is at most `max_norm` (Accelerate's
        ``clip_grad_norm_``; the reference never clips, so the recipes only call this with ``--max-grad-norm``).
        Returns the norm before clipping, un-scaled, as a 0-d tensor; nothing is read back to the host.

        Models adopted into an arena are clipped on the flat gradient: one norm and one multiply over the arena, with
        the loss scale divided out of the norm on the device.  Call it after the last backward of a step - in a
        data-parallel run the gradients are reduced by then and every replica computes the same factor.  `group`: ranks
        that hold disjoint parts of one model (the stages of a pipeline replica) - their squared norms are summed.
        A non-finite gradient gives a non-finite norm and leaves non-finite values in the arena, which the optimizer's
        overflow check skips as before."""
        inner = self.unwrap_model(model)
        arena = getattr(inner, "arena", None)
        if arena is None:
            params = [p for p in inner.parameters() if p.grad is not None]
            return torch.nn.utils.clip_grad_norm_(params, max_norm)
        if any(f.arena is arena and f._bucket_reducer is not None for f in self._fused):
            raise RuntimeError("clip_grad_norm_ after backward has already applied the optimizer bucket by bucket: "
                               "build the Engine with update_in_backward=False (or PDT_ADAMW_PER_BUCKET=0) to clip")
        sq = torch.linalg.vector_norm(arena.grad).pow(2)
        if group is not None and dist.is_initialized() and dist.get_world_size(group) > 1:
            if sq.is_cuda and dist.get_backend(group) == "gloo":  # single-GPU rehearsal: cross gloo through the host
                host = sq.detach().cpu()
                dist.all_reduce(host, group=group)
                sq = host.to(sq.device)
            else:
                dist.all_reduce(sq, group=group)
        norm = sq.sqrt() / self.scaler.scale_tensor()
        coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        arena.grad.mul_(coef)
        return norm

    @contextlib.contextmanager
    def no_sync(self, model: Optional[nn.Module] = None):
        models = [model] if model is not None else self._models
        with contextlib.ExitStack() as st:
            for m in models:
                if isinstance(m, DataParallel):
                    st.enter_context(m.no_sync())
            yield

    def accumulate(self, model: nn.Module, batches, overlap: Optional[bool] = None, threaded: Optional[bool] = None):
        """Forward + backward of one accumulation window (`batches` = its micro-batches, loss / len(batches) each, the
        gradient all-reduce armed for the last one only); returns the detached per-micro-batch losses.

        With ``overlap`` (default: on for the native GPU path) the window is software-pipelined over three HIP
        streams (micro-batch i on stream i % 3): the forward passes of the next micro-batches are enqueued before the
        backward of micro-batch i and run next to it.  The two are independent - the weights only change at the optimizer step - and with the reference's
        8-sample micro-batches (/root/reference/test_data_parallelism.py:90-93) neither fills the GPU on its own:
        a 1024-token GEMM has 64-256 output tiles for 256 CUs.  Every micro-batch lives on ONE stream (forward
        and backward, which is also where autograd runs a node's backward); even and odd micro-batches accumulate
        into two gradient buffers that the last backward merges (``PDT_DUAL_BACKWARD=0``: into one buffer, with the
        backward passes ordered by events - bit-identical to the plain loop), and the caller's stream waits for
        the other streams before the optimizer step."""
        n = len(batches)
        dp = model if isinstance(model, DataParallel) else None
        if overlap is None:
            overlap = self.native and self.device.type == "cuda" and os.environ.get("PDT_OVERLAP_MICRO", "1") != "0"
        if dp is not None:
            auto_arm, dp.auto_arm = dp.auto_arm, False
        losses = []
        split_k = None
        queued, rt, streams = {}, None, []  # (state of the overlapped window that the clean-up below has to undo)
        try:
            if not overlap or n < 2:
                for i, b in enumerate(batches):
                    out = model(**b)
                    if dp is not None:
                        dp.require_sync = i == n - 1
                        if i == n - 1:
                            dp.prepare_backward()
                    self.backward(out.loss / n)
                    losses.append(out.loss.detach())
                return losses
            main = torch.cuda.current_stream(self.device)
            # micro-batch i lives on stream i % ns (PDT_MICRO_STREAMS, default 3: the caller's stream and two more).
            # Measured at 12 x 8 samples of BERT-large: 2 streams 1455 samples/s, 3 streams 1571, 4 streams 1357 (1057
            # when the process gets 8 hardware queues instead of the default 4, GPU_MAX_HW_QUEUES=8, i.e. when the
            # fourth chain really runs concurrently): the GEMMs of these micro-batches hold 96-128 KiB of LDS, one
            # workgroup per CU, and a fourth chain of them only queues for CUs and evicts the others' operands.
            ns = max(2, min(n, int(os.environ.get("PDT_MICRO_STREAMS", "3"))))
            if getattr(self, "_alt_streams", None) is None:
                self._alt_streams = []
            while len(self._alt_streams) < ns - 1:
                self._alt_streams.append(torch.cuda.Stream(device=self.device))
            streams = [main] + self._alt_streams[:ns - 1]
            # The window's events come from a pool that lives as long as the engine and are re-recorded every window:
            # nothing is created or destroyed per micro-batch, and - what a hipGraph capture of the window depends on -
            # no event that a capturing stream has recorded is destroyed before the capture ends (utils/graph.py).
            if getattr(self, "_window_events", None) is None:
                self._window_events = []
            while len(self._window_events) < n + ns:
                self._window_events.append(torch.cuda.Event())
            start, done_ev, tail_ev = self._window_events[0], self._window_events[1:n + 1], self._window_events[n + 1:n + ns]
            start.record(main)
            for st in streams[1:]:
                st.wait_event(start)  # parameters / batches produced on the caller's stream so far
            for i, b in enumerate(batches):
                if i % ns:
                    for t in b.values():
                        if isinstance(t, torch.Tensor) and t.is_cuda:
                            t.record_stream(streams[i % ns])
            # Two gradient lanes (default; PDT_DUAL_BACKWARD=0 keeps one): micro-batch i accumulates into lane i % 2, a gradient buffer of
            # its own, so the backward passes of the two streams need no ordering between them and each stream runs its
            # micro-batches (forward, backward, forward, ...) independently of the other.  The last backward of the
            # window waits for the other stream and folds the second lane into the first segment by segment, right
            # before each segment is handed to the data-parallel reducer.  The sum is the same set of terms added in a
            # different (fixed) order than the one-lane loop.
            rt = getattr(self.unwrap_model(model), "runtime", None)
            dual = rt is not None and os.environ.get("PDT_DUAL_BACKWARD", "1") != "0"
            if dual:
                rt.arena.enable_alt()

            def forward(i):
                with torch.cuda.stream(streams[i % ns]):
                    if dual:
                        rt.lane = i % 2  # (captured by the nodes of this forward pass; one forward runs at a time)
                    out = model(**batches[i])
                    loss = out.loss / n
                    if self.scaler.enabled:
                        loss = loss * self.scaler.scale_tensor().to(loss.dtype)
                    return out, loss

            # The forward of the next micro-batch is issued by a second host thread while this one issues the
            # backward: the native ops release the GIL around their launches, and with 8-sample micro-batches the
            # step is bound by how fast ONE thread can launch ~1000 kernels per micro-batch, not by the GPU.
            if getattr(self, "_fwd_pool", None) is None:
                from concurrent.futures import ThreadPoolExecutor

                self._fwd_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="pdt-forward")
            # (threaded=False: everything is issued from the calling thread - what a hipGraph capture of the window
            # needs, see utils/graph.py; the streams still overlap on the GPU)
            if threaded is None:
                threaded = os.environ.get("PDT_OVERLAP_THREAD", "1") != "0"
            from .ops import native as _native

            defer_ok = os.environ.get("PDT_WGRAD_DEFER", "1") != "0"
            split_k = _native().gemm_split_k()
            # (K-split forward / dgrad GEMMs stay on inside the window: 1453 vs 1397 samples/s at 12 x 8 samples;
            # PDT_OVERLAP_SPLITK=0 switches them off for the window, which was the faster setting while a layer's
            # backward still cost 25 runtime calls on the host)
            if os.environ.get("PDT_OVERLAP_SPLITK", "1") == "0":
                _native().set_gemm_split_k(False)
            # How far the forward thread runs ahead of the backward passes (PDT_FWD_AHEAD, default 2): forward(i+2) is
            # queued while the host is still issuing backward(i), which it does not depend on.  With a distance of 1 a
            # rocprofv3 trace (where every launch costs the host more) shows all streams running dry for ~0.45 ms at the
            # end of every backward pass - the host then has to start the next forward AND the next backward from empty
            # queues; unprofiled, the 12 x 8 step measures the same with 1 and 2 (the GPU is the bound there).
            # (at most ns - 1: forward(i + ns) shares its stream with backward(i), and two host threads must never
            # enqueue into one stream at the same time)
            ahead = min(ns - 1, max(1, int(os.environ.get("PDT_FWD_AHEAD", "2")))) if threaded else 1
            cur, done = forward(0), [None] * n
            for j in range(1, min(n, ahead)):  # (forward(i + ahead) is submitted at the top of iteration i)
                queued[j] = self._fwd_pool.submit(forward, j)
            for i in range(n):
                nxt = None
                if threaded:
                    if i + ahead < n:
                        queued[i + ahead] = self._fwd_pool.submit(forward, i + ahead)
                    nxt = queued.pop(i + 1, None)
                elif i + 1 < n:  # started BEFORE backward(i): runs concurrently with it
                    nxt = forward(i + 1)
                s = streams[i % ns]
                with torch.cuda.stream(s):
                    if i >= 1 and (not dual or i == n - 1):
                        # one lane: backward passes accumulate into one buffer, strictly one after the other;
                        # two lanes: only the merging (last) backward has to see the other lane's gradients
                        s.wait_event(done[i - 1])
                    if dual and ns > 2 and i >= 2:
                        s.wait_event(done[i - 2])  # (the previous backward of this lane ran on another stream)
                    if dp is not None:
                        dp.require_sync = i == n - 1
                        if i == n - 1:
                            dp.prepare_backward()
                    if dual and i == n - 1:
                        rt.merge_on_done = True
                    # every backward of the window but the last: nothing reads the weight gradients before the
                    # window ends, so the layers queue their weight-gradient GEMMs behind one fork each and the
                    # stream joins once, here, instead of once per layer (fewer runtime calls per layer)
                    defer = dual and i < n - 1 and defer_ok
                    if defer:
                        _native().set_wgrad_defer(True)
                    try:
                        cur[1].backward()
                    finally:
                        if dual:
                            rt.merge_on_done = False
                        if defer:
                            _native().set_wgrad_defer(False)
                            _native().wgrad_join()
                    if dual and i == n - 1:
                        rt.arena.merge_alt()  # (segments that no node of the last backward reported)
                    done[i] = done_ev[i]
                    done[i].record(s)
                losses.append(cur[0].loss.detach())
                cur = nxt.result() if (threaded and nxt is not None) else nxt
            if dual:
                rt.lane = 0
            for st, tail in zip(streams[1:], tail_ev):  # the caller's stream continues after everything on the other streams
                tail.record(st)
                main.wait_event(tail)
            return losses
        finally:
            # A backward that raised leaves forward passes queued on (or running in) the forward thread: nothing may be
            # restored while they still run - they would read the switches reset below and set the gradient lane after
            # this function has returned.  Not-yet-started ones are cancelled, a running one is waited for.
            for fut in queued.values():
                if not fut.cancel():
                    try:
                        fut.result()
                    except Exception:  # noqa: BLE001 - the window is being abandoned anyway
                        pass
            if rt is not None:
                rt.lane = 0
                rt.merge_on_done = False
                from .ops import native as _native

                _native().set_wgrad_defer(False)
                for st in streams:  # deferred weight-gradient work of every stream the window used
                    with torch.cuda.stream(st):
                        _native().wgrad_join()
            if split_k is not None:
                _native().set_gemm_split_k(split_k)
            if dp is not None:
                dp.auto_arm = auto_arm
                dp.require_sync = True

    def unwrap_model(self, model: nn.Module) -> nn.Module:
        return model.module if isinstance(model, DataParallel) else model

    # ------------------------------------------------------------------------------------------
    def gather(self, obj: Any):
        """All-gather along dim 0 of a tensor or of every tensor in a nested list / tuple / dict."""
        return gather(obj, self.pg)

    def reduce_mean(self, t: torch.Tensor) -> torch.Tensor:
        if self.num_processes == 1:
            return t
        if t.is_cuda and dist.get_backend(self.pg) == "gloo":  # single-GPU rehearsal: cross gloo through the host
            host = t.detach().cpu()
            dist.all_reduce(host, group=self.pg)
            return host.to(t.device) / self.num_processes
        t = t.clone()
        dist.all_reduce(t, group=self.pg)
        return t / self.num_processes


def gather(obj: Any, group=None):
    """Fixed version of the reference's hand-copied ``gather`` (its list / dict branches call undefined
    names, /root/reference/test_model_parallelism.py:24-37): tensors are all-gathered along dim 0, 0-d
    tensors become 1-d first, containers are rebuilt with their own type."""
    if isinstance(obj, (list, tuple)):
        vals = [gather(o, group) for o in obj]
        if isinstance(obj, tuple) and hasattr(obj, "_fields"):  # namedtuple
            return type(obj)(*vals)
        return type(obj)(vals)
    if isinstance(obj, dict):
        return type(obj)({k: gather(v, group) for k, v in obj.items()})
    if not isinstance(obj, torch.Tensor):
        raise TypeError(f"Can't gather the values of type {type(obj)}, only of nested list/tuple/dicts of tensors.")
    t = obj
    if t.ndim == 0:
        t = t.clone()[None]
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    t = t.contiguous()
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # several ranks rehearsing on one GPU (gloo transport): GPU tensors cross gloo through the host
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, t.cpu(), group=group)
        return torch.cat(parts, dim=0).to(t.device)
    out = torch.empty((dist.get_world_size(group) * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out
