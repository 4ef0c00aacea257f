"""Multi-process test harness: run a function on N ranks over gloo on loopback (the way the reference
self-launches with ``mp.spawn`` on 127.0.0.1, /root/reference/test_model_parallelism.py:166-171,333-335) and
collect each rank's return value."""
from __future__ import annotations

import gc
import os
import pickle
import sys
import tempfile
import traceback
from typing import Any, Callable, List

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from .launch import DistEnv, find_free_port


def _entry(rank: int, world: int, port: int, fn: Callable, args: tuple, outdir: str, backend: str) -> None:
    DistEnv(rank, rank, world, "127.0.0.1", port).export()
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // max(2, world)))
    code = 0
    try:
        dist.init_process_group(backend, rank=rank, world_size=world)
        res = fn(rank, world, *args)
        with open(os.path.join(outdir, f"r{rank}.pkl"), "wb") as f:
            pickle.dump(("ok", res), f)
    except BaseException:  # noqa: BLE001 - the traceback goes to the parent through the result file
        with open(os.path.join(outdir, f"r{rank}.pkl"), "wb") as f:
            pickle.dump(("err", traceback.format_exc()), f)
        traceback.print_exc()
        code = 1
    finally:
        # Whatever still references the process group from C++ (a bucket reducer inside a reference cycle, or frames
        # kept alive by a traceback) would keep the group alive past destroy_process_group() until interpreter exit,
        # where its worker threads are torn down in arbitrary order ("terminate called without an active exception").
        # The result is on disk: collect, destroy the group, and leave without exit-time destructors.
        res = None
        gc.collect()
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(code)


def run_distributed(fn: Callable, world: int = 2, args: tuple = (), backend: str = "gloo") -> List[Any]:
    """Spawn `world` processes, run ``fn(rank, world, *args)`` in each, return the per-rank results."""
    # the rendezvous port is probed, released and then bound by rank 0: another process can take it in between
    # (or it can still be in TIME_WAIT), so a rendezvous failure is retried on a fresh port
    transient = ("address already in use", "eaddrinuse", "connection refused", "connection reset", "failed to bind")
    for attempt in range(3):
        with tempfile.TemporaryDirectory() as d:
            port = find_free_port()
            try:
                mp.spawn(_entry, args=(world, port, fn, args, d, backend), nprocs=world, join=True)
            except Exception as exc:
                msgs = [str(exc)]
                for r in range(world):
                    p = os.path.join(d, f"r{r}.pkl")
                    if os.path.exists(p):
                        kind, val = pickle.load(open(p, "rb"))
                        if kind == "err":
                            msgs.append(f"--- rank {r} ---\n{val}")
                text = "\n".join(msgs)
                if attempt < 2 and any(t in text.lower() for t in transient):
                    continue
                raise RuntimeError("distributed test failed:\n" + text)
            out = []
            for r in range(world):
                kind, val = pickle.load(open(os.path.join(d, f"r{r}.pkl"), "rb"))
                assert kind == "ok"
                out.append(val)
            return out
    raise RuntimeError("unreachable")

