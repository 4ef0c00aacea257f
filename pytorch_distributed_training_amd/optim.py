"""Optimizer, loss scaling and LR schedule.

* :class:`FusedAdamW` - the HF flavour of AdamW the reference instantiates
  (``AdamW(params=model.parameters(), lr=lr, correct_bias=correct_bias)``,
  /root/reference/test_data_parallelism.py:120, /root/reference/test_model_parallelism.py:244):
  ``eps=1e-6``, ``weight_decay=0``, bias correction optional, decoupled decay applied AFTER the Adam
  update.  On a GPU the whole model is updated by ONE kernel over the flat arena which also un-scales
  the gradients, skips the step on overflow and refreshes the 16-bit compute weights; the overflow flag,
  the loss scale, the step count and the learning rate are device scalars, so a step never syncs the host
  (torch's ``GradScaler.step`` reads ``found_inf`` back every step).
* :class:`HFAdamW` - literal per-parameter transcription, the oracle of the tests and the optimizer of
  models that were not adopted into an arena.
* :class:`LossScaler` - dynamic loss scale with torch ``GradScaler`` constants (init 2**16, x2 every 2000
  clean steps, x0.5 on overflow).
* :func:`get_linear_schedule_with_warmup` - ``lr * t/warmup`` then linear decay to 0, as used at
  /root/reference/test_data_parallelism.py:131-135.
"""
from __future__ import annotations

import math
from typing import Iterable, Optional

import torch

from .ops.arena import ParamArena


class LossScaler:
    """Device-resident loss-scale state ``[scale, growth_tracker, found_inf, last_step_overflowed]``."""

    def __init__(self, device, enabled: bool = True, init_scale: float = 2.0 ** 16, growth_factor: float = 2.0,
                 backoff_factor: float = 0.5, growth_interval: int = 2000, dynamic: bool = True):
        self.enabled = enabled
        self.dynamic = dynamic and enabled
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.state = torch.tensor([init_scale if enabled else 1.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)

    def scale_tensor(self) -> torch.Tensor:
        return self.state[0]

    def get_scale(self) -> float:  # host sync; for logging / tests only
        return float(self.state[0].item())

    def last_step_skipped(self) -> bool:  # host sync; for logging / tests only
        return bool(self.state[3].item() != 0.0)

    def state_dict(self):
        return {"state": self.state.detach().cpu()}

    def load_state_dict(self, sd):
        self.state.copy_(sd["state"].to(self.state.device))


class FusedAdamW:
    """HF-semantics AdamW over a :class:`ParamArena`."""

    def __init__(self, arena: ParamArena, lr: float = 2e-5, betas=(0.9, 0.999), eps: float = 1e-6,
                 weight_decay: float = 0.0, correct_bias: bool = True, scaler: Optional[LossScaler] = None):
        self.arena = arena
        dev = arena.device
        self.m = torch.zeros_like(arena.master)
        self.v = torch.zeros_like(arena.master)
        self.lr = torch.tensor([lr], dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)  # completed (non-skipped) updates
        self.betas, self.eps, self.weight_decay, self.correct_bias = betas, eps, weight_decay, correct_bias
        self.scaler = scaler or LossScaler(dev, enabled=False)
        self.grad_divisor = 1.0  # extra division folded into the un-scaling (not used by default)
        # True: the data-parallel reducer already flags non-finite values bucket by bucket, on the reduced gradients
        # (parallel/dp.py attach_scaler) - no separate pass over the arena here
        self.grads_checked_by_reducer = False
        # the data-parallel reducer that updates buckets on the communication stream as they land (attach_reducer)
        self._bucket_reducer = None
        self.param_groups = [{"lr": lr, "initial_lr": lr}]  # scheduler-facing, mirrors torch optimizers
        self._host_lr = lr

    # -- scheduler interface ----------------------------------------------------------------------
    def set_lr(self, lr: float) -> None:
        if lr != self._host_lr:
            self._host_lr = lr
            self.param_groups[0]["lr"] = lr
            self.lr.fill_(lr)

    def get_lr(self) -> float:
        return self._host_lr

    def attach_reducer(self, reducer) -> bool:
        """Let the data-parallel reducer run this optimizer's update bucket by bucket on the communication stream, right
        behind each bucket's all-reduce, so that the update of the first buckets overlaps the rest of backward instead
        of starting after the last bucket has landed (2.3 ms for BERT-large).  Only for precisions that never skip a
        step: with fp16's dynamic loss scale one overflowing bucket cancels the update of ALL of them, which an early
        update could not undo.  ``PDT_ADAMW_PER_BUCKET=0`` turns it off.  True if the reducer took the job."""
        import os

        if (self.scaler.enabled or not hasattr(reducer, "set_optimizer") or not self.arena.master.is_cuda
                or os.environ.get("PDT_ADAMW_PER_BUCKET", "1") == "0"):
            return False
        a = self.arena
        reducer.set_optimizer(a.master, self.m, self.v, a.p16, self.lr, self.step_count, self.scaler.state, self.betas[0],
                              self.betas[1], self.eps, self.weight_decay, self.correct_bias)
        if not reducer.applies_optimizer():  # (process-group transport, wire dtype conversion, single rank)
            reducer.clear_optimizer()
            return False
        self._bucket_reducer = reducer
        return True

    # -- optimizer interface -----------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False) -> None:
        self.arena.zero_grad()

    @torch.no_grad()
    def step(self, overflow_group=None) -> None:
        """`overflow_group`: ranks that must agree on skipping the step (the stages of one pipeline replica:
        an inf on one stage's gradients has to skip the update on all of them)."""
        a = self.arena
        a.zero_unwritten()
        if a.master.is_cuda:
            from .ops import native

            C = native()
            check = self.scaler.enabled and not self.grads_checked_by_reducer
            if self.scaler.enabled and overflow_group is not None:
                import torch.distributed as dist

                if check:
                    C.grad_check(a.grad, self.scaler.state)
                flag = self.scaler.state[2:3]
                if dist.get_backend(overflow_group) == "gloo":
                    # single-GPU rehearsal (ranks share the card, gloo transport): stage the flag through the host
                    # ourselves, like every other GPU tensor that crosses gloo in this framework
                    host = flag.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.MAX, group=overflow_group)
                    flag.copy_(host)
                else:
                    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=overflow_group)
                check = False
            applied = list(self._bucket_reducer.optimizer_applied()) if self._bucket_reducer is not None else []
            if any(applied):
                # buckets the reducer has updated behind their all-reduce are done; the rest (left-overs that nobody
                # completed, reduced at the end of backward) are updated here, as contiguous arena ranges
                b, i = a.bucket_bounds, 0
                while i < len(applied):
                    if applied[i]:
                        i += 1
                        continue
                    j = i
                    while j < len(applied) and not applied[j]:
                        j += 1
                    lo, n = b[i], b[j] - b[i]
                    C.adamw_step(a.master.narrow(0, lo, n), self.m.narrow(0, lo, n), self.v.narrow(0, lo, n),
                                 a.grad.narrow(0, lo, n), a.p16.narrow(0, lo, n) if a.p16 is not None else None, self.lr,
                                 self.step_count, self.scaler.state, self.betas[0], self.betas[1], self.eps,
                                 self.weight_decay, self.correct_bias, False)
                    i = j
            else:
                C.adamw_step(a.master, self.m, self.v, a.grad, a.p16, self.lr, self.step_count, self.scaler.state,
                             self.betas[0], self.betas[1], self.eps, self.weight_decay, self.correct_bias, check)
            C.scaler_update(self.scaler.state, self.step_count, self.scaler.growth_factor,
                            self.scaler.backoff_factor, self.scaler.growth_interval, self.scaler.dynamic)
        else:
            self._step_cpu()

    def _step_cpu(self) -> None:
        a, st = self.arena, self.scaler.state
        g = a.grad / st[0]
        overflow = self.scaler.enabled and not bool(torch.isfinite(g).all())
        if not overflow:
            b1, b2 = self.betas
            t = int(self.step_count.item()) + 1
            self.m.mul_(b1).add_(g, alpha=1 - b1)
            self.v.mul_(b2).addcmul_(g, g, value=1 - b2)
            step_size = self._host_lr
            if self.correct_bias:
                step_size *= math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            a.master.addcdiv_(self.m, self.v.sqrt().add_(self.eps), value=-step_size)
            if self.weight_decay:
                a.master.mul_(1 - self._host_lr * self.weight_decay)
            self.step_count += 1
            if self.scaler.dynamic:
                st[1] += 1
                if st[1] >= self.scaler.growth_interval:
                    st[0] *= self.scaler.growth_factor
                    st[1] = 0
            if a.p16 is not None:
                a.p16.copy_(a.master)
        elif self.scaler.dynamic:
            st[0] *= self.scaler.backoff_factor
            st[1] = 0
        st[3] = 1.0 if overflow else 0.0
        st[2] = 0.0

    def state_dict(self):
        return {"m": self.m, "v": self.v, "step": self.step_count, "lr": self._host_lr,
                "scaler": self.scaler.state_dict()}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"].to(self.m.device))
        self.v.copy_(sd["v"].to(self.v.device))
        self.step_count.copy_(sd["step"].to(self.step_count.device))
        self.set_lr(float(sd["lr"]))
        self.scaler.load_state_dict(sd["scaler"])


class HFAdamW(torch.optim.Optimizer):
    """Per-parameter AdamW with the exact update order of the (removed) ``transformers.AdamW``."""

    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-6,
                 weight_decay: float = 0.0, correct_bias: bool = True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"], st["exp_avg"], st["exp_avg_sq"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                st["step"] += 1
                st["exp_avg"].mul_(b1).add_(p.grad, alpha=1 - b1)
                st["exp_avg_sq"].mul_(b2).addcmul_(p.grad, p.grad, value=1 - b2)
                denom = st["exp_avg_sq"].sqrt().add_(group["eps"])
                step_size = group["lr"]
                if group["correct_bias"]:
                    step_size *= math.sqrt(1 - b2 ** st["step"]) / (1 - b1 ** st["step"])
                p.addcdiv_(st["exp_avg"], denom, value=-step_size)
                if group["weight_decay"] > 0:
                    p.add_(p, alpha=-group["lr"] * group["weight_decay"])


def AdamW(params=None, lr: float = 1e-3, correct_bias: bool = True, **kw):
    """Reference-spelled constructor (``AdamW(params=..., lr=..., correct_bias=...)``).  The engine swaps it
    for :class:`FusedAdamW` when the model is adopted into an arena (see ``Engine.prepare``)."""
    return HFAdamW(params, lr=lr, correct_bias=correct_bias, **kw)


class LinearWarmupSchedule:
    """``lambda(t) = t/warmup`` for ``t < warmup`` else ``max(0, (T-t)/(T-warmup))``.

    ``step()`` is expected once per optimizer step.  The reference sizes ``T`` in micro-batches
    (``len(train_dataloader) * num_epochs``) while stepping once per accumulation window, so its LR never
    leaves warm-up on MRPC; pass the same ``num_training_steps`` to reproduce that, or the number of
    optimizer steps to get the intended schedule.
    """

    def __init__(self, optimizer, num_warmup_steps: int, num_training_steps: int):
        self.optimizer = optimizer
        self.warmup, self.total = num_warmup_steps, num_training_steps
        self.base_lrs = [g.get("initial_lr", g["lr"]) for g in optimizer.param_groups]
        self.last_step = 0
        self._apply()

    def factor(self, t: int) -> float:
        if t < self.warmup:
            return t / max(1, self.warmup)
        return max(0.0, (self.total - t) / max(1, self.total - self.warmup))

    def _apply(self) -> None:
        f = self.factor(self.last_step)
        if hasattr(self.optimizer, "set_lr"):
            self.optimizer.set_lr(self.base_lrs[0] * f)
        else:
            for g, b in zip(self.optimizer.param_groups, self.base_lrs):
                g["lr"] = b * f

    def step(self) -> None:
        self.last_step += 1
        self._apply()

    def get_last_lr(self):
        return [b * self.factor(self.last_step) for b in self.base_lrs]

    def state_dict(self):
        return {"last_step": self.last_step}

    def load_state_dict(self, sd):
        self.last_step = sd["last_step"]
        self._apply()


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int):
    opt = getattr(optimizer, "optimizer", optimizer)  # unwrap Engine's handle
    return LinearWarmupSchedule(opt, num_warmup_steps, num_training_steps)
