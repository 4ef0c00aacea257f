"""Process bootstrap: the launcher contract of the reference, kept.

* ``python -m torch.distributed.run --nproc_per_node N --use_env script.py --fp16=True``
  (/root/reference/README.md:12-14): torchrun exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
  MASTER_PORT and :func:`init_distributed` consumes them.
* ``python script.py`` self-spawning one process per GPU on ``127.0.0.1:29500``
  (/root/reference/test_model_parallelism.py:166-171,331-335): :func:`spawn`.

One process drives one GPU.  The backend is "nccl" (RCCL over xGMI on ROCm) for GPU runs and "gloo"
for the CPU plumbing configuration.
"""
from __future__ import annotations

import datetime
import os
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


@dataclass
class DistEnv:
    rank: int = 0
    local_rank: int = 0
    world_size: int = 1
    master_addr: str = "127.0.0.1"
    master_port: int = 29500

    @classmethod
    def from_environ(cls) -> "DistEnv":
        e = os.environ
        return cls(int(e.get("RANK", 0)), int(e.get("LOCAL_RANK", e.get("RANK", 0))), int(e.get("WORLD_SIZE", 1)),
                   e.get("MASTER_ADDR", "127.0.0.1"), int(e.get("MASTER_PORT", 29500)))

    def export(self) -> None:
        os.environ.update(RANK=str(self.rank), LOCAL_RANK=str(self.local_rank), WORLD_SIZE=str(self.world_size),
                          MASTER_ADDR=self.master_addr, MASTER_PORT=str(self.master_port))


def export_gpu_ipc_env() -> None:
    """``HSA_ENABLE_IPC_MODE_LEGACY=0`` unless the user set it: dmabuf IPC handles are the only kind the host drivers of
    MI355X nodes export, and without it RCCL's intra-node transport (and any tensor shared between processes) fails with
    ``hipIpcGetMemHandle: invalid argument``.  The HIP runtime reads the variable when it initialises, so this runs at
    package import - before the first HIP call of the process - and again here for scripts that import late."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def agree_min(value: int, device: Optional[torch.device] = None, group=None) -> int:
    """The smallest `value` over the ranks of `group` (the value itself without a process group): how ranks settle on
    one number that each of them derived from its own device (free memory, ...).  Collective over the group."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return int(value)
    on_gpu = dist.get_backend(group) == "nccl" and device is not None and device.type == "cuda"
    t = torch.tensor([int(value)], dtype=torch.int64, device=device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def pick_device(cpu: bool, local_rank: int) -> torch.device:
    if cpu or not torch.cuda.is_available():
        return torch.device("cpu")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    return torch.device("cuda", local_rank % torch.cuda.device_count())


def default_backend(device: torch.device, env: DistEnv) -> str:
    """"nccl" (RCCL) when every local rank has a GPU of its own, otherwise "gloo".

    RCCL refuses two ranks on one device, so a rehearsal with more processes than GPUs (e.g. the 2-stage
    pipeline on a single-GPU box) runs its collectives over gloo: GPU tensors are staged through the host by
    the callers that support it (``parallel/pp.py``, the bucket reducer)."""
    if device.type != "cuda":
        return "gloo"
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", env.world_size))
    if local_world > torch.cuda.device_count():
        if env.rank == 0:
            print(f"[launch] {local_world} local ranks share {torch.cuda.device_count()} GPU(s): using the gloo backend "
                  "(rehearsal mode; RCCL needs one GPU per rank)", flush=True)
        return "gloo"
    return "nccl"


def init_distributed(cpu: bool = False, backend: Optional[str] = None, timeout_s: int = 1800,
                     env: Optional[DistEnv] = None) -> Tuple[DistEnv, torch.device]:
    """Join (or create) the default process group described by the environment; idempotent."""
    export_gpu_ipc_env()
    env = env or DistEnv.from_environ()
    device = pick_device(cpu, env.local_rank)
    if env.world_size > 1 and not dist.is_initialized():
        backend = backend or os.environ.get("PDT_BACKEND") or default_backend(device, env)
        env.export()
        kw = {}
        if backend == "nccl":
            kw["device_id"] = device  # eager communicator creation, binds the rank to its GPU
        restart = int(os.environ.get("TORCHELASTIC_RESTART_COUNT", "0") or 0)
        if restart > 0 and os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True":
            # A worker group restarted by torch.distributed.run (--max-restarts) on a fixed --master-port talks to the
            # SAME agent-hosted store as the group that died: the addresses its ranks published are still there, and
            # a new rank would try to connect to a dead peer.  Every attempt gets a key space of its own.
            store = dist.TCPStore(env.master_addr, env.master_port, env.world_size, is_master=False,
                                  timeout=datetime.timedelta(seconds=timeout_s))
            kw["store"] = dist.PrefixStore(f"/pdt/attempt_{restart}", store)
        dist.init_process_group(backend, rank=env.rank, world_size=env.world_size,
                                timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return env, device


def shutdown() -> None:
    """Destroy the default process group.  Objects that hold the group from C++ (bucket reducers) are collected
    first: a group that outlives this call is torn down at interpreter exit, with its worker threads still running."""
    import gc
    import sys

    gc.collect()
    if sys.exc_info()[0] is None:
        # orderly end of a run: close the framework's own RCCL communicators (a collective teardown - skipped when the
        # run is ending with an exception, where a peer may already be gone)
        from .parallel.comm import close_all

        close_all()
    if dist.is_initialized():
        dist.destroy_process_group()


def _spawn_entry(local_rank: int, fn: Callable, world_size: int, addr: str, port: int, args: tuple) -> None:
    DistEnv(local_rank, local_rank, world_size, addr, port).export()
    fn(local_rank, world_size, *args)


def spawn(fn: Callable, args: tuple = (), nprocs: Optional[int] = None, master_addr: str = "127.0.0.1",
          master_port: int = 29500, join: bool = True):
    """``mp.spawn``-style self launch: ``fn(rank, world_size, *args)`` in ``nprocs`` children
    (default: one per visible GPU, at least one)."""
    import torch.multiprocessing as mp

    if nprocs is None:
        nprocs = max(1, torch.cuda.device_count())
    return mp.spawn(_spawn_entry, args=(fn, nprocs, master_addr, master_port, args), nprocs=nprocs, join=join)


def find_free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
