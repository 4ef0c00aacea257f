"""Training configuration and command line.

The reference has two ``type=bool`` flags (``--fp16`` / ``--cpu``, for which ANY non-empty string - even
"False" - parses as True; /root/reference/test_data_parallelism.py:171-172) and otherwise hard-codes its
hyper-parameters in a dict (``{"lr": 2e-5, "num_epochs": 3, "correct_bias": True, "seed": 42, "batch_size": 96}``,
:174), module constants (``MAX_GPU_BATCH_SIZE = 8``, ``EVAL_BATCH_SIZE = 32``, :49-50) and literals (model
name :69,112; 100 warm-up steps :133).  Here every one of those is a dataclass field with the reference's
value as default and a CLI override; ``--fp16=True`` keeps working, ``--fp16=False`` now means False.
"""
from __future__ import annotations

import argparse
from dataclasses import asdict, dataclass, fields
from typing import Optional


def str2bool(v) -> bool:
    if isinstance(v, bool):
        return v
    s = str(v).strip().lower()
    if s in ("1", "true", "t", "yes", "y", "on", ""):
        return True
    if s in ("0", "false", "f", "no", "n", "off"):
        return False
    raise argparse.ArgumentTypeError(f"boolean expected, got {v!r}")


@dataclass
class TrainConfig:
    # reference hyper-parameter dict
    lr: float = 2e-5
    num_epochs: int = 3
    correct_bias: bool = True
    seed: int = 42
    batch_size: int = 96
    # reference module constants
    max_gpu_batch_size: int = 0  # 0: sized for the device (the reference's 8 on CPU / with --compat), see micro_batch_cap
    eval_batch_size: int = 32
    # reference literals
    model: str = "bert-large-cased"
    num_warmup_steps: int = 100
    # precision / device flags
    fp16: bool = False
    bf16: bool = False
    cpu: bool = False
    init_scale: float = 65536.0  # initial dynamic loss scale of fp16 runs (torch GradScaler's default)
    max_grad_norm: float = 0.0   # > 0: clip the global gradient norm before every optimizer step (the reference does not)
    # data (no hub access: synthetic MRPC-shaped pairs unless a local TSV directory is given)
    data_dir: Optional[str] = None
    vocab_file: Optional[str] = None
    train_examples: int = 3668
    eval_examples: int = 408
    data_rule: str = "copy"  # synthetic data: what makes a pair label 1 ("copy" | "topic", see data/synthetic.py)
    data_topics: int = 8     # synthetic data, rule "topic": number of vocabulary slices
    data_vocab: Optional[int] = None  # synthetic data: draw word tokens from this many ids only (None: whole vocabulary)
    max_length: Optional[int] = None  # None: pad to the longest sample of each batch
    pad_to_multiple_of: Optional[int] = None
    # behaviour switches
    varlen: bool = True         # native path (all layouts): run the encoder on the real tokens of a padded batch only
    compat: bool = False        # reproduce the reference's quirks (see train_dp.py)
    max_steps: int = 0          # stop after this many optimizer steps (0 = run all epochs)
    log_every: int = 10
    log_file: Optional[str] = None
    # parallelism
    pp_stages: int = 1          # 2 = layer-split model parallel composed with data parallel
    micro_batches: int = 4      # pipeline micro-batches per accumulation window
    mp_layout: str = "pipeline"  # "pipeline" (2-stage layer split) | "towers" (TriBert branch ensemble)
    comm_dtype: Optional[str] = None  # None|fp16|bf16: wire dtype of the gradient all-reduce
    min_bucket_mb: float = 4.0
    comm_backend: Optional[str] = None  # None|c10d|rccl: torch's RCCL process group, or the framework's own RcclComm
    # checkpointing
    output_dir: Optional[str] = None
    resume: Optional[str] = None
    save_every: int = 0
    init_from: Optional[str] = None  # directory with HF-layout weights (config.json + safetensors / bin)

    def micro_batch_cap(self, device=None, model_cfg=None, free_bytes: Optional[int] = None) -> int:
        """Largest micro-batch one GPU runs.  An explicit ``--max-gpu-batch-size N`` wins.  0 (default) sizes it for the
        GPU the process drives: the reference's constant 8 (/root/reference/test_data_parallelism.py:49, chosen for
        16-GB cards) on the CPU and with ``--compat``; on a GPU as many samples as fit a budget of half the free HBM
        and at most 131072 tokens (the shape the kernels are tuned for), counting a sample at the longest length the
        collator can produce and ~34 bytes of saved activations per token, layer and hidden unit (measured: 106 GB
        for 131072 tokens of BERT-large).

        Free memory is a per-rank reading (a co-tenant, several ranks rehearsing on one card), but the micro-batch
        decides the loaders' batch size, the number of optimizer steps per epoch and the pipeline's message shapes, so
        the ranks settle on the smallest cap among them (``launch.agree_min``: collective over the world - every rank of
        a run calls this at the same point of its recipe).  `free_bytes` replaces the device query (tests)."""
        if self.max_gpu_batch_size > 0:
            return self.max_gpu_batch_size
        if self.compat or device is None or getattr(device, "type", "cpu") != "cuda" or model_cfg is None:
            return 8
        from .launch import agree_min

        if free_bytes is None:
            import torch

            free_bytes, _total = torch.cuda.mem_get_info(device)
        per_token = 34 * model_cfg.num_hidden_layers * model_cfg.hidden_size
        seq = self.max_length or model_cfg.max_position_embeddings
        tokens = min(131072, int(0.5 * free_bytes) // per_token)
        cap = max(8, tokens // max(1, seq))
        agreed = agree_min(cap, device if hasattr(device, "index") else None)
        if agreed != cap:
            print(f"[config] micro-batch cap {cap} on this rank, {agreed} agreed across ranks", flush=True)
        return agreed

    def accumulation(self, device=None, model_cfg=None, free_bytes: Optional[int] = None):
        """The reference's accumulation policy (/root/reference/test_data_parallelism.py:90-93) on `micro_batch_cap`."""
        cap = self.micro_batch_cap(device, model_cfg, free_bytes)
        if self.batch_size > cap:
            return self.batch_size // cap, cap
        return 1, self.batch_size


def build_parser(description: str = "MI355X-native distributed fine-tuning") -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=description)
    for f in fields(TrainConfig):
        name = "--" + f.name.replace("_", "-")
        alias = ["--" + f.name] if "_" in f.name else []
        if f.type in (bool, "bool"):
            p.add_argument(name, *alias, dest=f.name, type=str2bool, nargs="?", const=True, default=f.default)
        else:
            base = {"int": int, "float": float, "str": str, "Optional[int]": int, "Optional[str]": str,
                    "Optional[float]": float}.get(str(f.type), str)
            p.add_argument(name, *alias, dest=f.name, type=base, default=f.default)
    return p


def parse_config(argv=None, description: str = "MI355X-native distributed fine-tuning") -> TrainConfig:
    ns = build_parser(description).parse_args(argv)
    return TrainConfig(**vars(ns))


def config_dict(cfg: TrainConfig):
    return asdict(cfg)
