"""``set_seed`` as called at /root/reference/test_data_parallelism.py:109: python, numpy and torch (CPU +
every GPU) generators.  Dropout masks of the native kernels do not use these generators: they are
counter-based functions of the Engine seed (see ops/fused.py)."""
import random

import numpy as np
import torch


def set_seed(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def seed_step(seed: int, rank: int, step: int) -> None:
    """Re-key the torch generators for the optimizer step that follows `step` completed ones.

    The PyTorch-path dropout draws from the global generator, and how much a rank draws depends on the (dynamically
    padded) shapes of its batches - so the generator state of a rank is not a function of the step count, and a run
    resumed from a checkpoint could not reproduce it without saving one state per rank.  Re-keying at every step
    boundary makes the stream a pure function of (seed, rank, step), like the counter-based masks of the native
    kernels: a restarted worker group continues with exactly the masks of the uninterrupted run."""
    torch.manual_seed((seed * 1000003 + rank * 7919 + step * 104729 + 12345) % (2 ** 63))
