"""Failure detection for multi-process runs.

The reference configures no timeouts (``init_process_group("gloo", ...)`` at
/root/reference/test_model_parallelism.py:171 blocks forever if a peer dies) and has no checkpoints, so a
torchrun restart starts from scratch.  :class:`StepWatchdog` is a host thread that aborts the process when
no training step completes within ``timeout_s`` (a hung collective, a dead peer): the elastic launcher then
restarts the worker group and the recipe resumes from the newest checkpoint.  On every tick it also polls the
asynchronous error state of the framework's own RCCL communicators (``ncclCommGetAsyncError``: a peer that
died, a link error) and aborts on the first error instead of waiting for the step timeout.  ``inject_fault`` is the test
hook: ``PDT_INJECT_FAULT=rank:step`` makes that rank exit abruptly at that optimizer step.
"""
from __future__ import annotations

import os
import sys
import threading
import time
from typing import Optional


class StepWatchdog:
    def __init__(self, timeout_s: float = 0.0, rank: int = 0):
        self.timeout_s, self.rank = timeout_s, rank
        self._last = time.monotonic()
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        if timeout_s > 0:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def heartbeat(self) -> None:
        self._last = time.monotonic()

    def _run(self) -> None:
        while not self._stop.wait(min(5.0, self.timeout_s / 4)):
            err = poll_comm_errors()
            if err:
                print(f"[watchdog] rank {self.rank}: {err} - aborting so the launcher can restart the group",
                      file=sys.stderr, flush=True)
                os._exit(76)
            if time.monotonic() - self._last > self.timeout_s:
                print(f"[watchdog] rank {self.rank}: no step for {self.timeout_s:.0f}s - aborting so the launcher "
                      "can restart the group", file=sys.stderr, flush=True)
                os._exit(75)

    def close(self) -> None:
        """Stop the thread and wait for it: after this returns nothing polls the communicators any more, so they
        can be closed."""
        self._stop.set()
        if self._thread is not None and self._thread is not threading.current_thread():
            self._thread.join(timeout=10.0)
            self._thread = None


def poll_comm_errors() -> Optional[str]:
    """First asynchronous RCCL error recorded on any communicator this process opened, or None."""
    try:
        from ..parallel.comm import _COMMS, _LOCK
    except Exception:  # pragma: no cover - partial installs
        return None
    with _LOCK:  # close_all() takes the same lock: a communicator is never destroyed under this poll
        for comm in [c for c in _COMMS.values() if c is not None]:
            try:
                comm.check_async_error()
            except RuntimeError as e:
                msg = str(e).splitlines()[0]
                if "communicator is closed" in msg:
                    continue  # closed in an orderly way, not an RCCL error
                return msg
    return None


def inject_fault(rank: int, step: int) -> None:
    spec = os.environ.get("PDT_INJECT_FAULT")
    if not spec:
        return
    r, s = (int(x) for x in spec.split(":"))
    marker = os.environ.get("PDT_INJECT_FAULT_ONCE")
    if r == rank and s == step:
        if marker:
            if os.path.exists(marker):
                return  # already crashed once; the restarted run passes through
            open(marker, "w").close()
        print(f"[fault-injection] rank {rank} exits at step {step}", file=sys.stderr, flush=True)
        os._exit(13)
