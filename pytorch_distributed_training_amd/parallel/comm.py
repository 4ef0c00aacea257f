"""The framework's own RCCL communicator (csrc/runtime/rccl_comm.cpp) and its bootstrap.

The reference never creates a communicator itself: ``Accelerator()`` does it for the data-parallel script
(/root/reference/test_data_parallelism.py:55) and ``init_process_group("gloo")`` for the model-parallel one
(/root/reference/test_model_parallelism.py:166-171).  Here the launcher's env-var rendezvous (torch's TCPStore
behind ``init_process_group``) is only the *bootstrap channel*: group rank 0 creates an RCCL unique id, the id
travels over that channel, and every rank opens a communicator on the RCCL C API with a HIP stream of its own.
The bucket reducer (``GradBucketReducer(rccl_comm=...)``) and the pipeline's send/recv can then run on it
without going through c10d.

Selection: the framework's communicator is the default on GPUs with one rank per device; ``PDT_COMM=c10d`` (or
``Engine(comm_backend="c10d")`` / ``--comm-backend c10d``) routes the same traffic through torch's "nccl" process
group instead, which drives the same RCCL library.  CPU runs and rehearsals with several ranks on one GPU always
use the gloo process group.
"""
from __future__ import annotations

import os
import threading
from typing import Dict, Optional

import torch
import torch.distributed as dist

# communicators of this process, keyed by the global ranks they span (a group object's id() can be reused after the
# group is garbage collected; its rank tuple cannot mean anything else)
_COMMS: Dict[tuple, object] = {}
_LOCK = threading.Lock()  # held by the watchdog's poll and by close_all()


def backend_choice(explicit: Optional[str] = None) -> str:
    """"rccl" (default) or "c10d" (from the argument, else the PDT_COMM environment variable)."""
    choice = (explicit or os.environ.get("PDT_COMM") or "rccl").lower()
    if choice not in ("c10d", "rccl"):
        raise ValueError(f"unknown communication backend {choice!r} (c10d | rccl)")
    return choice


_AGREE_ROUND: Dict[tuple, int] = {}


def _agree(ok: bool, group, device) -> bool:
    """True iff `ok` on every rank of `group`.

    Decided over the rendezvous store (host sockets only), not with a collective on the "nccl" group: a communicator
    that just failed its self test may have left kernels of the same RCCL library stuck on the GPU, and a GPU
    collective queued behind them would turn the fall-back decision into the hang it is there to avoid.  Every rank
    publishes its verdict under a key of its own and reads the others' (``PDT_COMM_AGREE_S``, default 120 s); a rank that
    cannot be heard counts as failed.  Without a reachable store the process group decides, as before."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ok
    ranks = tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:  # noqa: BLE001 - private accessor: fall back to the collective
        store = None
    if store is None:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return bool(flag.item())
    import datetime

    rnd = _AGREE_ROUND.get(ranks, 0)
    _AGREE_ROUND[ranks] = rnd + 1
    prefix = f"pdt/comm_agree/{'-'.join(map(str, ranks))}/{rnd}/"
    store.set(prefix + str(dist.get_rank()), b"1" if ok else b"0")
    keys = [prefix + str(r) for r in ranks]
    try:
        store.wait(keys, datetime.timedelta(seconds=float(os.environ.get("PDT_COMM_AGREE_S", "120"))))
    except Exception:  # noqa: BLE001 - a silent rank: nobody may use the communicator
        return False
    return all(store.get(k) == b"1" for k in keys)


def self_test(comm) -> str:
    """Exercise the communicator the way a training step does before anybody relies on it, every wait under a
    deadline (``PDT_COMM_SELFTEST_S``, default 60 s), every result checked:

    1. one small all-reduce on the communicator's own stream (the low-latency protocol);
    2. three windows of ``fork -> two bucket-sized all-reduces (32 MB, averaged) -> join`` back to back with a
       consumer kernel on the compute stream after every join and no host synchronisation in between - the pattern of
       an overlapped backward; then a reduce-scatter + all-gather round over the same buffer.

    Returns "" if everything passed, else what went wrong."""
    world = comm.world_size
    device = torch.device("cuda", comm.device)
    deadline = float(os.environ.get("PDT_COMM_SELFTEST_S", "60"))
    probe = torch.ones(1024, dtype=torch.float32, device=device)
    comm.fork()
    comm.all_reduce_async(probe, False)
    if not comm.wait_idle(deadline):
        return "the self-test all-reduce did not finish"
    comm.join()
    if not bool((probe == float(world)).all().item()):
        return f"the self-test all-reduce returned {probe[0].item()} instead of {world}"
    n = (8 << 20) // world * world
    bucket = torch.full((n,), float(comm.rank + 1), dtype=torch.float32, device=device)
    mean = (world + 1) / 2.0  # average of rank+1 over the ranks; averaging it again leaves it unchanged
    for _ in range(3):
        comm.fork()
        comm.all_reduce_async(bucket, True)
        comm.all_reduce_async(bucket, True)
        comm.join()
        bucket.mul_(1.0)  # a consumer on the compute stream, ordered behind the join
    shard = torch.empty(n // world, dtype=torch.float32, device=device)
    comm.fork()
    comm.reduce_scatter_async(shard, bucket, True)
    comm.all_gather_async(bucket, shard)
    if not comm.wait_idle(deadline):
        return "the chained bucket collectives of the self test did not finish"
    comm.join()
    if not bool(torch.allclose(bucket, torch.full_like(bucket, mean), rtol=1e-6, atol=1e-6)):
        return f"the chained bucket collectives returned {bucket[0].item()} instead of {mean}"
    return ""


def create_rccl_comm(group=None, device: Optional[torch.device] = None):
    """Open (once per group) an RcclComm spanning the ranks of `group` (default: all ranks; a process that has
    no process group gets a 1-rank communicator).  Collective over the group: every member must call it.

    A communicator of more than one rank is self-tested before anybody relies on it (`self_test`: a small all-reduce,
    then bucket-sized collectives chained the way an overlapped backward issues them, every wait under the deadline
    ``PDT_COMM_SELFTEST_S``, default 60 s, every result checked).  The ranks
    agree on the outcome over the process group; if any rank failed to construct the communicator, timed out or read
    a wrong sum, every rank aborts its communicator and ``None`` is returned - the callers then route the traffic
    through the process group itself (torch's "nccl" backend) instead of hanging the job."""
    from ..ops import native

    if dist.is_initialized():
        key = tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))
    else:
        key = (0,)
    if key in _COMMS:
        return _COMMS[key]
    C = native()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if device.type != "cuda":
        raise ValueError("an RCCL communicator needs a GPU; CPU runs use the gloo process group")
    index = device.index if device.index is not None else torch.cuda.current_device()
    if dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    if world == 1:
        uid = C.RcclComm.unique_id()
    else:
        # the id goes from group rank 0 to the rest over the already initialised process group
        box = [C.RcclComm.unique_id() if rank == 0 else None]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group, device=device if dist.get_backend(group) == "nccl" else None)
        uid = box[0]
    comm, ok, why = None, True, ""
    try:
        with torch.cuda.device(index):
            comm = C.RcclComm(uid, rank, world, index)
            if world > 1:
                why = self_test(comm)
                ok = not why
    except RuntimeError as e:  # construction or launch failed on this rank
        ok, why = False, str(e).splitlines()[0]
    if world > 1 and os.environ.get("PDT_COMM_SELFTEST_FAIL") == str(rank):  # (tests: pretend this rank failed)
        ok, why = False, "PDT_COMM_SELFTEST_FAIL"
    if not _agree(ok, group, device):
        if comm is not None:
            comm.abort()
        if why or rank == 0:
            print(f"[comm] rank {rank}: RcclComm over {world} ranks is not usable"
                  f"{' (' + why + ')' if why else ' (another rank failed)'}; falling back to the process group", flush=True)
        comm = None
    _COMMS[key] = comm
    return comm


def close_all() -> None:
    """Orderly teardown of every communicator this process opened (collective: call it on every rank)."""
    with _LOCK:
        for comm in _COMMS.values():
            if comm is not None:
                comm.close()
        _COMMS.clear()
