"""Rank-0 structured logging.  The reference prints only rank banners and an end-of-epoch metric line
(/root/reference/test_data_parallelism.py:58-60,166); loss, learning rate and throughput are never shown.
:class:`StepLogger` keeps the same human-readable lines and adds a JSONL sink for the benchmark harness.
Loss values are fetched from the device only every ``every`` steps, so logging adds no per-step host sync."""
from __future__ import annotations

import json
import os
import time
from typing import Optional

import torch


class StepLogger:
    def __init__(self, is_main: bool, every: int = 10, path: Optional[str] = None, samples_per_step: int = 0,
                 tokens_per_step: int = 0):
        self.is_main, self.every = is_main, max(1, every)
        if path and is_main and os.path.dirname(path):
            os.makedirs(os.path.dirname(path), exist_ok=True)
        self.file = open(path, "a") if (path and is_main) else None
        self.samples_per_step, self.tokens_per_step = samples_per_step, tokens_per_step
        self.t_last, self.step_last = time.perf_counter(), 0
        self._tokens = 0

    def add_tokens(self, n: int) -> None:
        """Count `n` (padded) tokens processed by the whole job since the last log line; with dynamic padding
        the per-step token count varies, so the training loop reports it per micro-batch."""
        self._tokens += int(n)

    def log_step(self, step: int, loss: torch.Tensor, lr: float, scaler=None, extra=None) -> None:
        if not self.is_main:
            return
        if step % self.every:
            return
        now = time.perf_counter()
        dt, n = now - self.t_last, step - self.step_last
        self.t_last, self.step_last = now, step
        rec = {"step": step, "loss": float(loss.detach().float().item()), "lr": lr}
        if n > 0 and dt > 0:
            rec["steps_per_s"] = n / dt
            if self.samples_per_step:
                rec["samples_per_s"] = self.samples_per_step * n / dt
            tokens = self._tokens or self.tokens_per_step * n
            if tokens:
                rec["tokens_per_s"] = tokens / dt
        self._tokens = 0
        if scaler is not None and scaler.enabled:
            rec["loss_scale"] = scaler.get_scale()
        if callable(extra):  # evaluated after the loss fetch above, i.e. after the step's host sync
            extra = extra()
        if extra:
            rec.update({k: v for k, v in extra.items() if v is not None})
        print("step {step}: loss {loss:.4f} lr {lr:.3e}".format(**rec)
              + (f" {rec['samples_per_s']:.1f} samples/s" if "samples_per_s" in rec else "")
              + (f" comm-exposed {rec['comm_exposed_ms']:.2f} ms" if "comm_exposed_ms" in rec else ""), flush=True)
        self._write(rec)

    def log_epoch(self, epoch: int, metric: dict) -> None:
        if not self.is_main:
            return
        print(f"epoch {epoch}:", metric, flush=True)
        self._write({"epoch": epoch, **metric})

    def _write(self, rec: dict) -> None:
        if self.file:
            self.file.write(json.dumps(rec) + "\n")
            self.file.flush()

    def close(self) -> None:
        if self.file:
            self.file.close()
