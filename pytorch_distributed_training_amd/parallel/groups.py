"""Rank mesh for data parallel x pipeline (layer-split model) parallel.

World rank ``r`` = ``replica * pp + stage``: the stages of one replica are adjacent ranks (GPU 2d and 2d+1
of a node - every GPU pair of an MI355X node shares a direct xGMI link), data-parallel groups connect equal
stages across replicas: 8 GPUs = 4 stage pairs, DP groups {0,2,4,6} (stage 0) and {1,3,5,7} (stage 1).
The reference composes DP and MP by giving every DDP rank a two-device module
(/root/reference/test_model_parallelism.py:190-191,248-253) - and lets both ranks alias the same two GPUs;
here every rank owns exactly one GPU and each group has its own communicator.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch.distributed as dist


@dataclass
class ParallelMesh:
    world_size: int
    rank: int
    pp: int = 1

    def __post_init__(self):
        if self.world_size % self.pp:
            raise ValueError(f"world size {self.world_size} is not a multiple of pipeline stages {self.pp}")
        self.dp = self.world_size // self.pp
        self.stage = self.rank % self.pp
        self.replica = self.rank // self.pp
        self.pp_group = None
        self.dp_group = None

    def pp_ranks(self, replica: Optional[int] = None) -> List[int]:
        d = self.replica if replica is None else replica
        return [d * self.pp + s for s in range(self.pp)]

    def dp_ranks(self, stage: Optional[int] = None) -> List[int]:
        s = self.stage if stage is None else stage
        return [d * self.pp + s for d in range(self.dp)]

    def build_groups(self) -> "ParallelMesh":
        """Create the communicators.  Collective: every rank creates every group, in the same order."""
        if not dist.is_initialized() or self.world_size == 1:
            return self
        for d in range(self.dp):
            g = dist.new_group(self.pp_ranks(d))
            if d == self.replica:
                self.pp_group = g
        for s in range(self.pp):
            g = dist.new_group(self.dp_ranks(s))
            if s == self.stage:
                self.dp_group = g
        return self

    @property
    def is_first_stage(self) -> bool:
        return self.stage == 0

    @property
    def is_last_stage(self) -> bool:
        return self.stage == self.pp - 1

    def peer(self, stage: int) -> int:
        """Global rank of `stage` inside this replica."""
        return self.replica * self.pp + stage
