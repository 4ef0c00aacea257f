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
