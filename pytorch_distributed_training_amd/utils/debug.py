"""Race / ordering diagnostics.  The reference has none (its only concurrency is what torch DDP and
Accelerate do internally, /root/reference/test_data_parallelism.py:125-127); this framework owns its
streams, arenas and collectives, so it carries its own checks.  All of them are off in normal runs.

``PDT_DEBUG=1``
    * the parameter / gradient / 16-bit arenas are allocated with guard words on both sides and
      :meth:`ParamArena.check_guards` runs after every optimizer step (a kernel or collective that
      writes outside its slice trips it; alignment padding between parameters must stay zero too);
    * ``optimizer.step()`` raises if a gradient reduction of a prepared model is still armed
      (a backward pass that never reached its end-of-backward callback).
``PDT_DEBUG_SYNC=1`` (or :func:`set_serialized`)
    every native launch is followed by a device synchronise, so an asynchronous fault is raised at the
    op that caused it and kernels of different streams never overlap - a failure that disappears in this
    mode is a stream-ordering bug.
:func:`serialize_hip_runtime`
    sets ``AMD_SERIALIZE_KERNEL=3`` / ``AMD_SERIALIZE_COPY=3`` (the HIP runtime's own serialisation of every
    kernel and copy, library kernels and RCCL included); only possible before the process touches the GPU.
"""
from __future__ import annotations

import os

import torch


class ArenaGuardError(RuntimeError):
    pass


def debug_enabled() -> bool:
    return os.environ.get("PDT_DEBUG", "0") not in ("", "0")


def guard_elems() -> int:
    """Guard words per arena side (0 = arenas are allocated without guards)."""
    return 1024 if debug_enabled() else 0


def serialize_hip_runtime() -> None:
    if torch.cuda.is_initialized():
        raise RuntimeError("serialize_hip_runtime() must run before the first GPU call of the process")
    os.environ["AMD_SERIALIZE_KERNEL"] = "3"
    os.environ["AMD_SERIALIZE_COPY"] = "3"


def set_serialized(on: bool = True) -> None:
    """Drain the device after every native launch (see module docstring)."""
    from ..ops import native

    native().set_debug_sync(bool(on))
