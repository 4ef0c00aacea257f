"""Dataloader preparation: rank sharding + asynchronous host-to-device placement.

``Engine.prepare(dataloader)`` returns a :class:`DeviceLoader` around a re-built ``DataLoader`` whose batch
sampler is a :class:`ShardedBatchSampler`.  Shuffling is derived from ``(seed, epoch)`` on every rank, so no
RNG-state broadcast is needed at the start of an epoch (Accelerate broadcasts the generator state there).
Batches are pinned and copied on a side HIP stream one step ahead of the consumer.
"""
from __future__ import annotations

from typing import Iterator

import torch
from torch.utils.data import BatchSampler, DataLoader, RandomSampler, SequentialSampler

from .collate import PadCollator
from .samplers import DistributedShardSampler, ShardedBatchSampler


class _EpochSeededRandomSampler(torch.utils.data.Sampler):
    """Random permutation that is a pure function of (seed, epoch): identical on every rank."""

    def __init__(self, n: int, seed: int):
        self.n, self.seed, self.epoch = n, seed, 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return self.n

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed * 1000003 + self.epoch)
        return iter(torch.randperm(self.n, generator=g).tolist())


class _WithSampleIds(torch.utils.data.Dataset):
    """View of a dict-style dataset whose items also carry their dataset index as ``sample_ids``.  The native
    kernels key dropout masks by it, which makes a mask a property of the sample and not of the rank / micro-batch
    that happens to process it (ops.fused.NativeRuntime)."""

    def __init__(self, dataset):
        self.dataset = dataset

    def __len__(self) -> int:
        return len(self.dataset)

    def __getitem__(self, i):
        item = self.dataset[i]
        if isinstance(item, dict) and "sample_ids" not in item:
            item = dict(item)
            item["sample_ids"] = int(i)
        return item


def shard_dataloader(loader: DataLoader, num_processes: int, process_index: int, seed: int = 0,
                     even_batches: bool = True) -> DataLoader:
    """Re-build `loader` so that this rank iterates only its share of the batches."""
    bs = loader.batch_sampler
    if isinstance(bs, ShardedBatchSampler):
        return loader
    sampler = loader.sampler
    if isinstance(sampler, (DistributedShardSampler, torch.utils.data.distributed.DistributedSampler)):
        return loader  # already sharded per sample by the caller
    if isinstance(sampler, RandomSampler):
        sampler = _EpochSeededRandomSampler(len(loader.dataset), seed)
        inner = BatchSampler(sampler, loader.batch_size, loader.drop_last)
    elif isinstance(sampler, SequentialSampler):
        inner = BatchSampler(sampler, loader.batch_size, loader.drop_last)
    else:
        inner = bs
    sharded = ShardedBatchSampler(inner, num_processes, process_index, even_batches)
    dataset = loader.dataset
    if isinstance(loader.collate_fn, PadCollator) and not isinstance(dataset, _WithSampleIds):
        dataset = _WithSampleIds(dataset)  # (only with the framework's own collator, which passes the ids through)
    return DataLoader(dataset, batch_sampler=sharded, collate_fn=loader.collate_fn,
                      num_workers=loader.num_workers, pin_memory=loader.pin_memory,
                      persistent_workers=loader.persistent_workers if loader.num_workers > 0 else False)


def _samplers_of(loader: DataLoader):
    out = []
    bs = loader.batch_sampler
    while bs is not None:
        if hasattr(bs, "set_epoch"):
            out.append(bs)
        inner = getattr(bs, "sampler", None)
        if inner is not None and hasattr(inner, "set_epoch"):
            out.append(inner)
        bs = getattr(bs, "batch_sampler", None)
    if hasattr(loader.sampler, "set_epoch"):
        out.append(loader.sampler)
    return out


class DeviceLoader:
    """Iterate a CPU ``DataLoader`` with batches already on `device`, copied one step ahead on a side stream."""

    def __init__(self, loader: DataLoader, device: torch.device, prefetch: int = 2):
        self.loader, self.device, self.prefetch = loader, torch.device(device), max(1, prefetch)
        self.epoch = 0
        self._stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def __len__(self) -> int:
        return len(self.loader)

    @property
    def dataset(self):
        d = self.loader.dataset
        return d.dataset if isinstance(d, _WithSampleIds) else d

    @property
    def batch_size(self):
        return self.loader.batch_size

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def skip_batches(self, n: int) -> None:
        """Leave the first `n` batches of this rank out of the next iteration (resume inside an epoch).  With a
        :class:`ShardedBatchSampler` they are skipped as index lists, before any collation or copy."""
        self._skip = max(0, int(n))

    def _to_device(self, batch):
        if self._stream is None:
            return batch.to(self.device) if hasattr(batch, "to") else batch
        with torch.cuda.stream(self._stream):
            if hasattr(batch, "pin_memory"):
                batch = batch.pin_memory()
            moved = batch.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return moved, ev

    def __iter__(self) -> Iterator:
        for s in _samplers_of(self.loader):
            s.set_epoch(self.epoch)
        self.epoch += 1
        skip, self._skip = getattr(self, "_skip", 0), 0
        if skip and isinstance(self.loader.batch_sampler, ShardedBatchSampler):
            self.loader.batch_sampler.skip_next, skip = skip, 0
        # (a DataLoader iterator draws a base seed from the global torch generator when it is created; put the
        # generator back so that iterating the data never shifts the dropout stream of a CPU run - a run resumed in
        # the middle of an epoch creates one iterator more than the uninterrupted run it has to reproduce)
        rng = torch.get_rng_state()
        it = iter(self.loader)
        torch.set_rng_state(rng)
        for _ in range(skip):  # (a loader that is not batch-sharded: the skipped batches are built and dropped)
            next(it, None)
        if self._stream is None:
            for b in it:
                yield self._to_device(b)
            return
        queue = []
        for b in it:
            queue.append(self._to_device(b))
            if len(queue) > self.prefetch:
                yield self._hand_over(*queue.pop(0))
        for moved, ev in queue:
            yield self._hand_over(moved, ev)

    def _hand_over(self, moved, ev):
        """Order the consumer stream after the copy, and tell the caching allocator that the batch (allocated on the
        copy stream) is read on the consumer stream: without that, its memory could be handed to a later prefetch copy
        as soon as Python drops the batch, while kernels that read it are still queued."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        tensors = moved.values() if isinstance(moved, dict) else (moved if isinstance(moved, (list, tuple)) else [moved])
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
        return moved
