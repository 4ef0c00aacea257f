#!/usr/bin/env python3
"""Data parallelism combined with model parallelism, self-launching (the counterpart of the reference's
test_model_parallelism.py):

    python train_mp.py                       # one process per visible GPU: GPUs (2d, 2d+1) form a 2-stage replica
    python train_mp.py --cpu=True --nprocs 4 --model bert-tiny      # CPU/gloo plumbing, 2 replicas x 2 stages
    python -m torch.distributed.run --nproc_per_node 8 train_mp.py --fp16=True   # also works under torchrun

``--mp-layout pipeline`` (default): 2-stage layer split, micro-batched 1F1B over send/recv, data parallel
across replicas.  ``--mp-layout towers``: the reference's TriBert ensemble spread over the replica's ranks.
Unlike the reference (gloo, fp32 only, both ranks on the same two GPUs) mixed precision is supported and
every rank owns one GPU.
"""
import os

# dmabuf IPC (the only mode the host driver of these nodes supports) for RCCL / cross-process GPU memory:
# must be in the environment before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

from pytorch_distributed_training_amd import launch  # noqa: E402
from pytorch_distributed_training_amd.config import build_parser, TrainConfig  # noqa: E402
from pytorch_distributed_training_amd.recipes import run_model_parallel  # noqa: E402


def _worker(rank: int, world_size: int, cfg: TrainConfig):
    try:
        out = run_model_parallel(rank, world_size, cfg)
        if rank == 0:
            print({k: v for k, v in out.items() if k != "losses"})
    finally:
        launch.shutdown()


def main():
    p = build_parser("DP x MP sequence-classification fine-tuning on MI355X")
    p.add_argument("--nprocs", type=int, default=0, help="processes to self-spawn (default: one per GPU, min 2)")
    p.add_argument("--master-port", type=int, default=29500)
    ns = p.parse_args()
    nprocs, port = ns.nprocs, ns.master_port
    d = vars(ns)
    d.pop("nprocs"); d.pop("master_port")
    cfg = TrainConfig(**d)
    if cfg.pp_stages == 1:
        cfg.pp_stages = 2
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:  # launched by torch.distributed.run
        _worker(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), cfg)
        return
    world = nprocs or max(2, torch.cuda.device_count())
    launch.spawn(_worker, args=(cfg,), nprocs=world, master_port=port)


if __name__ == "__main__":
    main()
