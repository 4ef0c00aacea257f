#!/usr/bin/env python3
"""Data-parallel fine-tuning of a BERT-class sequence classifier (the counterpart of the reference's
test_data_parallelism.py, same launch line):

    python -m torch.distributed.run --nproc_per_node 2 --use_env train_dp.py --fp16=True
    python train_dp.py --cpu=True --model bert-tiny            # single process, CPU plumbing

Defaults are the reference's: bert-large-cased, lr 2e-5, 3 epochs, batch 96 as 12 micro-batches of 8,
100 warm-up steps, seed 42, eval batch 32.  Data: synthetic MRPC-shaped pairs unless --data-dir/--vocab-file
point at a local GLUE/MRPC copy.
"""
import os

# dmabuf IPC (the only mode the host driver of these nodes supports) for RCCL / cross-process GPU memory:
# must be in the environment before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from pytorch_distributed_training_amd import launch  # noqa: E402
from pytorch_distributed_training_amd.config import parse_config  # noqa: E402
from pytorch_distributed_training_amd.recipes import run_data_parallel  # noqa: E402


def main():
    cfg = parse_config(description="Data-parallel sequence-classification fine-tuning on MI355X")
    try:
        run_data_parallel(cfg)
    finally:
        launch.shutdown()


if __name__ == "__main__":
    main()
