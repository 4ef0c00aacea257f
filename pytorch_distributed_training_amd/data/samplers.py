"""Sharding of a dataset across data-parallel ranks, in the two flavours the reference relies on.

* :class:`ShardedBatchSampler` - Accelerate's flavour, applied by ``accelerator.prepare(dataloader)``
  (/root/reference/test_data_parallelism.py:125-127): whole *batches* are dealt round-robin (rank r takes
  batches r, r+W, ...); the tail wraps around to the start of the epoch so that every rank sees the same
  number of full batches; ``len()`` shrinks to ceil(n_batches / W) - the reference sizes its LR schedule
  from that length (/root/reference/test_data_parallelism.py:131-135).
* :class:`DistributedShardSampler` - ``torch.utils.data.DistributedSampler``'s flavour
  (/root/reference/test_model_parallelism.py:254,262): a (seed+epoch)-shuffled permutation padded to a
  multiple of W, rank r takes ``indices[r::W]``.  The reference never calls ``set_epoch`` (same order every
  epoch); :class:`~pytorch_distributed_training_amd.data.loader.DeviceLoader` does it for you.
"""
from __future__ import annotations

import math
from typing import Iterator, List

import torch
from torch.utils.data import Sampler


class ShardedBatchSampler(Sampler):
    def __init__(self, batch_sampler, num_processes: int = 1, process_index: int = 0, even_batches: bool = True):
        self.batch_sampler = batch_sampler
        self.num_processes, self.process_index, self.even_batches = num_processes, process_index, even_batches
        self.batch_size = getattr(batch_sampler, "batch_size", None)
        self.drop_last = getattr(batch_sampler, "drop_last", False)
        self.skip_next = 0  # batches of this rank to leave out of the NEXT iteration (mid-epoch resume)

    def __iter__(self) -> Iterator[List[int]]:
        skip, self.skip_next = self.skip_next, 0
        for i, b in enumerate(self._batches()):
            if i >= skip:
                yield b

    def __len__(self) -> int:
        n, w = len(self.batch_sampler), self.num_processes
        if n % w == 0:
            return n // w
        if self.drop_last:
            return n // w
        if self.even_batches:
            return n // w + 1
        return n // w + (1 if self.process_index < n % w else 0)

    def _batches(self) -> Iterator[List[int]]:
        w, r = self.num_processes, self.process_index
        batches = [list(b) for b in self.batch_sampler]
        if not batches:
            return
        bs = self.batch_size or len(batches[0])
        if self.drop_last:
            full = [b for b in batches if len(b) == bs]
            for i in range(0, len(full) - len(full) % w, w):
                yield full[i + r]
            return
        if not self.even_batches:
            for i, b in enumerate(batches):
                if i % w == r:
                    yield b
            return
        # wrap-around pool: the indices of the first W batches, repeated if the epoch is tiny
        pool: List[int] = [i for b in batches[:w] for i in b]
        while len(pool) < w * bs:
            pool += pool
        cursor = 0
        if len(batches[-1]) < bs:
            need = bs - len(batches[-1])
            batches[-1] = batches[-1] + pool[cursor:cursor + need]
            cursor += need
        while len(batches) % w:
            batches.append(pool[cursor:cursor + bs])
            cursor += bs
        for i in range(r, len(batches), w):
            yield batches[i]


class DistributedShardSampler(Sampler):
    def __init__(self, dataset, num_replicas: int = 1, rank: int = 0, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = False):
        self.n = len(dataset)
        self.num_replicas, self.rank, self.shuffle, self.seed, self.drop_last = num_replicas, rank, shuffle, seed, drop_last
        self.epoch = 0
        self.frozen = False  # True: set_epoch() is ignored - a script that never calls it sees the same order every epoch
        if drop_last and self.n % num_replicas:
            self.num_samples = math.ceil((self.n - num_replicas) / num_replicas)
        else:
            self.num_samples = math.ceil(self.n / num_replicas)
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch: int) -> None:
        if not self.frozen:
            self.epoch = epoch

    def __len__(self) -> int:
        return self.num_samples

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        if self.drop_last:
            order = order[:self.total_size]
        else:
            pad = self.total_size - len(order)
            if pad > 0:
                order += (order * math.ceil(pad / len(order)))[:pad]
        return iter(order[self.rank:self.total_size:self.num_replicas])
