"""Tracing hooks: roctx ranges (visible in rocprofv3 --marker-trace), HIP-event timers, and the rocprofv3
command lines used for the numbers under profiles/.  The reference has no profiling at all."""
from __future__ import annotations

import contextlib
import ctypes
import os
from typing import Dict, List

import torch

_roctx = None


def _load_roctx():
    global _roctx
    if _roctx is None:
        _roctx = False
        # rocprofv3 (--marker-trace) records the rocprofiler-sdk flavour of roctx; the roctracer one is the fallback
        for name in ("librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so",
                     "/opt/rocm/lib/libroctx64.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                _roctx = lib
                break
            except OSError:
                continue
    return _roctx


@contextlib.contextmanager
def trace_range(name: str, enabled: bool = True):
    """roctx push/pop around a region (no-op when roctx is unavailable or PDT_TRACE is unset)."""
    lib = _load_roctx() if (enabled and os.environ.get("PDT_TRACE")) else None
    if lib:
        lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        if lib:
            lib.roctxRangePop()


class GpuTimer:
    """Named HIP-event timers; ``summary()`` syncs once and returns mean milliseconds per name."""

    def __init__(self, enabled: bool = True):
        self.enabled = enabled and torch.cuda.is_available()
        self.events: Dict[str, List] = {}

    @contextlib.contextmanager
    def time(self, name: str):
        if not self.enabled:
            yield
            return
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        try:
            yield
        finally:
            e.record()
            self.events.setdefault(name, []).append((s, e))

    def summary(self) -> Dict[str, float]:
        if not self.enabled:
            return {}
        torch.cuda.synchronize()
        out = {k: sum(s.elapsed_time(e) for s, e in v) / len(v) for k, v in self.events.items()}
        self.events.clear()
        return out


ROCPROF_RECIPES = {
    "kernel_time": "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 3 --warmup 2",
    "mfma_lds": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc -- python bench/gemm_bench.py --rounds 1 --iters 2",
}
