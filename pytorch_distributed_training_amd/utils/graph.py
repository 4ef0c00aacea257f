"""Whole-step HIP graph capture.

A training step of this framework is a fixed sequence of a few hundred to a few thousand kernel launches
(BERT-large: ~670 per micro-batch).  With large micro-batches the GPU is the bottleneck and the launches hide
behind it; with the reference's 8-sample micro-batches (/root/reference/test_data_parallelism.py:49,90-93) or with
small models the host becomes visible.  :class:`GraphedStep` captures forward + backward + optimizer update of a
fixed-shape batch into ONE hipGraph and replays it, so a step costs one graph launch.

What makes the step capturable by construction:

* every piece of per-step state the kernels read lives in device memory - loss scale and overflow flag, AdamW
  step counter, learning rate (``FusedAdamW.lr``), the dropout counter (``NativeRuntime.rng_state``) - and is
  advanced by kernels, never by values baked into launch arguments (the micro-batch index IS a launch argument:
  it is the position inside the captured accumulation window, identical in every replay, while the in-graph
  ``rng_state[1] += 1`` gives every replay fresh masks);
* gradients are born in the arena (no per-step allocation of parameter gradients) and the first micro-batch of a
  window overwrites, so there is no ``zero_grad`` dependence on host flags across replays;
* the weight-gradient side stream forks from and joins the capturing stream inside every layer.

The learning-rate schedule stays outside the graph: ``scheduler.step()`` after the call writes the new value into
the device scalar the captured optimizer kernel reads.  Data-parallel runs are not captured (the bucket
all-reduces are issued through c10d, which owns its own stream and work objects).
"""
from __future__ import annotations

import os
from typing import Callable, Optional

import torch

# Why the multi-stream accumulation window (Engine.accumulate: micro-batch i on stream i % 3, two gradient lanes, the
# weight-gradient lane) is not captured by default.  Measured on ROCm 7.2 / PyTorch 2.10 (bench/dev/graph_window_capture.py,
# BERT-large, 12 x 8 samples): the capture itself goes through - every launch issued from ONE host thread, so a second
# thread adding nodes to the graph is ruled out - and the process then dies with SIGSEGV inside
# torch.cuda.CUDAGraph.capture_end() (hipStreamEndCapture / instantiation) on the capturing thread.  The one-stream window
# with the same kernels, the same weight-gradient fork / join per layer and as many nodes captures and replays, so the
# crash belongs to the cross-stream part of the graph.  One difference of that part was found by reading the code and is
# removed: the window used to create its fork / join events per call and drop them when accumulate() returned, i.e.
# BEFORE the capture ended, while the capturing streams still list them; the events now live in a pool owned by the
# engine.  Whether that was the cause has not been re-measured (bench/dev/hip_capture_event_destroy.cpp is the
# stand-alone experiment for it), so the overlapped capture stays behind an explicit opt-in.
OVERLAP_REFUSAL = ("capturing the multi-stream accumulation window crashed inside hipStreamEndCapture on ROCm 7.2 "
                   "(see utils/graph.py); GraphedStep captures the one-stream window (window='plain'). "
                   "PDT_GRAPH_OVERLAP_WINDOW=1 tries the overlapped capture anyway.")


class GraphedStep:
    """``loss = step(batch)``: eager for the first ``warmup`` calls, then captured once and replayed.

    ``batch`` is a dict of tensors whose shapes must stay fixed (pad to a fixed length); it is copied into static
    device buffers before every replay.  ``accum`` micro-batches can be captured as one step by passing a list of
    ``accum`` dicts.  The returned loss is the static output tensor of the last micro-batch (clone it to keep it).
    """

    def __init__(self, engine, model, optimizer, warmup: int = 3, step_optimizer: bool = True,
                 loss_fn: Optional[Callable] = None, window: str = "plain"):
        self.engine, self.model, self.optimizer = engine, model, optimizer
        self.warmup, self.step_optimizer = max(1, warmup), step_optimizer
        self.loss_fn = loss_fn
        if window not in ("plain", "overlap"):
            raise ValueError("window is 'plain' (micro-batches one after the other on one stream) or 'overlap'")
        if window == "overlap" and loss_fn is not None:
            raise ValueError("the overlapped window computes the model's own loss")
        if window == "overlap" and os.environ.get("PDT_GRAPH_OVERLAP_WINDOW") != "1":
            raise RuntimeError(OVERLAP_REFUSAL)
        self.window = window
        if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) > 4:
            import warnings

            # measured on ROCm 7.2, 12 x 8 samples of BERT-large, one-stream capture: 96 ms per replayed step with HIP's
            # default of 4 hardware queues, 235 ms with the 16 this package asks for at import (the graph executor
            # spreads the weight-gradient branches over more queues and pays for every cross-queue dependency)
            warnings.warn("hipGraph replays are ~2.4x slower with more than 4 hardware queues: start the process with "
                          "PDT_HW_QUEUES=0 (before the package is imported) when steps are replayed from graphs")
        if engine.num_processes != 1:
            raise RuntimeError("GraphedStep captures single-process steps only")
        if engine.device.type != "cuda":
            raise RuntimeError("GraphedStep needs a GPU")
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static: Optional[list] = None
        self.static_loss: Optional[torch.Tensor] = None
        self.calls = 0
        self._side = torch.cuda.Stream(device=engine.device)

    # ------------------------------------------------------------------------------------------
    def _window(self) -> torch.Tensor:
        """One accumulation window on the static batches: the body that is captured."""
        rt = getattr(self.engine.unwrap_model(self.model), "runtime", None)
        if rt is not None:
            rt.rng_state[1:2].add_(1)  # in-graph: every replay draws new dropout masks
            rt.rng_state_global[1:2].add_(1)  # (the stream of batches that carry sample ids)
        n = len(self.static)
        loss = None
        if self.window == "overlap" and n > 1:
            # Engine.accumulate's multi-stream window (the streams fork from and join the capturing stream), with every
            # launch issued from THIS thread: no second host thread may add nodes to the graph while it is captured
            loss = self.engine.accumulate(self.model, self.static, overlap=True, threaded=False)[-1]
            if self.step_optimizer:
                self.optimizer.step()
                self.optimizer.zero_grad()
            return loss.detach()
        for mb in self.static:
            out = self.model(**mb)
            loss = out.loss if self.loss_fn is None else self.loss_fn(out, mb)
            self.engine.backward(loss / n if n > 1 else loss)
        if self.step_optimizer:
            self.optimizer.step()
            self.optimizer.zero_grad()
        return loss.detach()

    def _load(self, batches) -> None:
        if self.static is None:
            self.static = [{k: v.to(self.engine.device).clone() for k, v in b.items() if isinstance(v, torch.Tensor)}
                           for b in batches]
            return
        if len(batches) != len(self.static):
            raise ValueError("number of micro-batches changed after capture")
        for dst, src in zip(self.static, batches):
            for k, v in src.items():
                if not isinstance(v, torch.Tensor):
                    continue
                if dst[k].shape != v.shape:
                    raise ValueError(f"shape of batch[{k!r}] changed: {tuple(v.shape)} vs captured {tuple(dst[k].shape)}")
                dst[k].copy_(v, non_blocking=True)

    def __call__(self, batch) -> torch.Tensor:
        batches = batch if isinstance(batch, (list, tuple)) else [batch]
        self._load(batches)
        self.calls += 1
        if self.graph is not None:
            self.graph.replay()
            return self.static_loss
        if self.calls <= self.warmup:
            # eager warm-up on a side stream (lazy initialisation, allocator warm-up), as graph capture requires
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                loss = self._window()
            torch.cuda.current_stream().wait_stream(self._side)
            return loss
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self._side):
            self.static_loss = self._window()
        self.graph = g
        g.replay()  # capture does not execute: this replay IS the step of this call
        return self.static_loss
