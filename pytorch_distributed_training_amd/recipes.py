"""Training recipes: the two training functions of the reference, on this framework's engine.

:func:`run_data_parallel` is ``training_function`` of /root/reference/test_data_parallelism.py:53-166
(tokenise -> dataloaders -> seed -> model -> AdamW -> prepare -> linear schedule -> epochs of
train-with-accumulation + eval-with-gather).  :func:`run_model_parallel` is the one of
/root/reference/test_model_parallelism.py:174-315 (explicit process group, multi-device model wrapped for
data parallelism, ``DistributedSampler`` loaders, ``no_sync`` accumulation).

Reference quirks are reproduced only with ``--compat``:

====================================================  ===========================================
reference behaviour                                    default here
====================================================  ===========================================
optimizer fires when ``step % accum == 0`` (so after   fires after every ``accum`` micro-batches; a
the very first micro-batch, and tails leak across      partial window at the end of an epoch is
epochs)                                                flushed
gradients all-reduced on EVERY micro-batch (DP script)  all-reduced once per optimizer step
LR horizon counted in micro-batches -> never leaves    horizon counted in optimizer steps
warm-up on MRPC
eval sampler shuffles + pads, duplicates counted        sequential eval, wrap-around duplicates dropped
``DistributedSampler`` without ``set_epoch``            epoch-dependent shuffle
====================================================  ===========================================
"""
from __future__ import annotations

import os
import time
from typing import Optional

import torch
from torch.utils.data import DataLoader

from . import checkpoint as ckpt
from .config import TrainConfig
from .data import PadCollator, SyntheticPairDataset
from .data.samplers import DistributedShardSampler
from .engine import Engine
from .metrics import load_metric
from .models.bert import build_model
from .models.config import preset
from .optim import AdamW, get_linear_schedule_with_warmup
from .utils.logging import StepLogger
from .utils.profile import trace_range
from .utils.seed import seed_step, set_seed
from .utils.watchdog import StepWatchdog, inject_fault


def build_datasets(cfg: TrainConfig, model_cfg):
    """GLUE/MRPC from a local directory (``train.tsv`` / ``dev.tsv`` + a WordPiece ``vocab.txt``) when given,
    otherwise synthetic sentence pairs of MRPC size and shape."""
    if cfg.data_dir:
        from .data.glue import load_mrpc_tsv

        return load_mrpc_tsv(cfg.data_dir, cfg.vocab_file, max_length=model_cfg.max_position_embeddings)
    roberta = model_cfg.model_type == "roberta"
    # truncation to the model's maximum length, as the reference's tokenizer call does (truncation=True)
    room = model_cfg.max_position_embeddings - (model_cfg.pad_token_id + 1 if roberta else 0)
    limit = min(room, cfg.max_length) if cfg.max_length else room
    max_len = max(4, min(64, (limit - 3) // 2))
    train = SyntheticPairDataset(cfg.train_examples, model_cfg.vocab_size, max_len=max_len, seed=cfg.seed, roberta=roberta,
                                 data_vocab=cfg.data_vocab, rule=cfg.data_rule, topics=cfg.data_topics)
    val = SyntheticPairDataset(cfg.eval_examples, model_cfg.vocab_size, max_len=max_len, seed=cfg.seed + 1, roberta=roberta,
                               data_vocab=cfg.data_vocab, rule=cfg.data_rule, topics=cfg.data_topics)
    return train, val


def make_model(cfg: TrainConfig):
    if cfg.init_from:
        return ckpt.load_pretrained(cfg.init_from)
    return build_model(cfg.model)


def _collator(cfg: TrainConfig, model_cfg) -> PadCollator:
    varlen = bool(cfg.varlen)  # (the model-parallel layouts hand packed rows across the link as well)
    if cfg.max_length:
        return PadCollator(model_cfg.pad_token_id, "max_length", cfg.max_length, varlen=varlen)
    mult = cfg.pad_to_multiple_of if (cfg.pad_to_multiple_of or 0) > 0 else None
    return PadCollator(model_cfg.pad_token_id, "longest", pad_to_multiple_of=mult, varlen=varlen)


def model_parallel_loaders(cfg: TrainConfig, train_ds, val_ds, collate, mesh, train_batch_size: int):
    """The (CPU) loaders of the model-parallel recipes, before ``engine.prepare``.

    Default: plain shuffled / sequential loaders, which ``prepare`` shards per batch over the data-parallel replicas
    (the same flavour as the data-parallel recipe).  ``--compat``: the reference's own loaders
    (/root/reference/test_model_parallelism.py:254-269) - a ``DistributedSampler`` per split, i.e. per-sample
    round-robin sharding with the tail padded by repetition, its default ``shuffle=True, seed=0`` for the training
    AND the evaluation split, and no ``set_epoch`` call, so every epoch sees the same order."""
    if not cfg.compat:
        return (DataLoader(train_ds, shuffle=True, collate_fn=collate, batch_size=train_batch_size),
                DataLoader(val_ds, shuffle=False, collate_fn=collate, batch_size=cfg.eval_batch_size))
    from .data.loader import _WithSampleIds

    def loader(ds, batch_size):
        sampler = DistributedShardSampler(ds, num_replicas=mesh.dp, rank=mesh.replica)
        sampler.frozen = True
        data = _WithSampleIds(ds) if isinstance(collate, PadCollator) else ds
        return DataLoader(data, sampler=sampler, collate_fn=collate, batch_size=batch_size)

    return loader(train_ds, train_batch_size), loader(val_ds, cfg.eval_batch_size)


def evaluate(engine: Engine, model, eval_dl, metric, dataset_len: int, dedupe: bool) -> dict:
    model.eval()
    seen = 0
    for batch in eval_dl:
        with torch.no_grad():
            outputs = model(**batch)
        predictions = outputs.logits.argmax(dim=-1)
        preds, refs = engine.gather(predictions), engine.gather(batch["labels"])
        valid = None
        if dedupe:
            valid = max(0, min(preds.shape[0], dataset_len - seen))
            seen += preds.shape[0]
        metric.add_batch(predictions=preds, references=refs, valid=valid)
    return metric.compute()


def run_data_parallel(cfg: TrainConfig, engine: Optional[Engine] = None) -> dict:
    engine = engine or Engine(fp16=cfg.fp16, bf16=cfg.bf16, cpu=cfg.cpu, seed=cfg.seed,
                              comm_dtype={None: None, "fp16": torch.float16, "bf16": torch.bfloat16}[cfg.comm_dtype],
                              min_bucket_mb=cfg.min_bucket_mb, init_scale=cfg.init_scale,
                    comm_backend=cfg.comm_backend, update_in_backward=not cfg.max_grad_norm > 0)
    # rank banner, as the reference prints it on every process
    print("LOCAL_RANK: ", engine.local_process_index)
    print("DEVICE: ", engine.device)
    print("process_index: ", engine.process_index)

    model_cfg = preset(cfg.model) if not cfg.init_from else ckpt.EncoderConfig.load(cfg.init_from)
    train_ds, val_ds = build_datasets(cfg, model_cfg)
    metric = load_metric("glue", "mrpc")
    accum, micro_bs = cfg.accumulation(engine.device, model_cfg)
    engine.print(f"batch {cfg.batch_size} per process = {accum} micro-batch(es) of {micro_bs}")
    collate = _collator(cfg, model_cfg)
    train_dl = DataLoader(train_ds, shuffle=True, collate_fn=collate, batch_size=micro_bs)
    eval_dl = DataLoader(val_ds, shuffle=False, collate_fn=collate, batch_size=cfg.eval_batch_size)

    set_seed(cfg.seed)
    model = make_model(cfg)
    optimizer = AdamW(params=model.parameters(), lr=cfg.lr, correct_bias=cfg.correct_bias)
    model, optimizer, train_dl, eval_dl = engine.prepare(model, optimizer, train_dl, eval_dl)

    micro_per_epoch = len(train_dl)
    steps_per_epoch = micro_per_epoch if cfg.compat else -(-micro_per_epoch // accum)
    horizon = micro_per_epoch * cfg.num_epochs if cfg.compat else steps_per_epoch * cfg.num_epochs
    lr_scheduler = get_linear_schedule_with_warmup(optimizer, cfg.num_warmup_steps, horizon)

    # resume (an explicit --resume, or the newest checkpoint of --output-dir after an elastic restart): weights,
    # optimizer, schedule, RNG counters, and the position inside the epoch - the batches already consumed are
    # skipped as index lists, so the run continues with exactly the batch an uninterrupted run would see next
    start_epoch, global_step, start_micro = 0, 0, 0
    resume_dir = cfg.resume or (cfg.output_dir if ckpt.has_checkpoint(cfg.output_dir) else None)
    if resume_dir and ckpt.has_checkpoint(resume_dir):
        extra = ckpt.load_training_state(resume_dir, model, optimizer, lr_scheduler)
        start_epoch, global_step = int(extra.get("epoch", 0)), int(extra.get("global_step", 0))
        start_micro = int(extra.get("micro_offset", 0))
        engine.print(f"resumed from {resume_dir}: epoch {start_epoch}, batch {start_micro}, optimizer step {global_step}")

    seed_step(cfg.seed, engine.process_index, global_step)
    logger = StepLogger(engine.is_main_process, cfg.log_every, cfg.log_file,
                        samples_per_step=micro_bs * accum * engine.num_processes)
    dog = StepWatchdog(float(os.environ.get("PDT_WATCHDOG_S", "0")), engine.process_index)
    # PDT_COMM_TIMING=1: the log line carries the gradient-communication time that backward did not hide
    comm_extra = (lambda: {"comm_exposed_ms": model.comm_exposed_ms()}) if getattr(model, "measure_exposed", False) else None
    last_metric: dict = {}
    done = False
    t_start = time.perf_counter()
    for epoch in range(start_epoch, cfg.num_epochs):
        model.train()
        if hasattr(train_dl, "set_epoch"):
            train_dl.set_epoch(epoch)
        pending = []
        first = start_micro if epoch == start_epoch else 0
        if first and hasattr(train_dl, "skip_batches"):
            train_dl.skip_batches(first)
        for step, batch in enumerate(train_dl, start=first):
            window = accum
            logger.add_tokens(batch["input_ids"].numel() * engine.num_processes)
            if cfg.compat:
                # the reference's DP script: every micro-batch is reduced, the optimizer fires when step % accum == 0
                fire = step % accum == 0
                with trace_range("forward"):
                    outputs = model(**batch)
                    loss = outputs.loss / window
                with trace_range("backward+reduce"):
                    engine.backward(loss)
            else:
                # one accumulation window = `accum` micro-batches (the partial window that ends an epoch is a mean
                # over its own micro-batches): forward / backward software-pipelined by Engine.accumulate, gradients
                # reduced once, with the last backward
                pending.append(batch)
                fire = len(pending) == accum or step == micro_per_epoch - 1
                if not fire:
                    continue
                window = len(pending)
                with trace_range("window forward+backward+reduce"):
                    loss = torch.stack(engine.accumulate(model, pending)).mean() / window
            if fire:
                with trace_range("optimizer"):
                    if cfg.max_grad_norm > 0:
                        engine.clip_grad_norm_(model, cfg.max_grad_norm)
                    optimizer.step()
                    lr_scheduler.step()
                    optimizer.zero_grad()
                pending = []
                global_step += 1
                seed_step(cfg.seed, engine.process_index, global_step)
                dog.heartbeat()
                logger.log_step(global_step, loss * window, lr_scheduler.get_last_lr()[0], engine.scaler, comm_extra)
                if cfg.save_every and cfg.output_dir and global_step % cfg.save_every == 0 and engine.is_main_process:
                    ckpt.save_training_state(cfg.output_dir, model, optimizer, lr_scheduler,
                                             {"epoch": epoch, "global_step": global_step, "micro_offset": step + 1})
                inject_fault(engine.process_index, global_step)
                if cfg.max_steps and global_step >= cfg.max_steps:
                    done = True
                    break
        last_metric = evaluate(engine, model, eval_dl, metric, len(val_ds), dedupe=not cfg.compat)
        logger.log_epoch(epoch, last_metric)
        if cfg.output_dir and engine.is_main_process:
            ckpt.save_training_state(cfg.output_dir, model, optimizer, lr_scheduler,
                                     {"epoch": epoch + 1, "global_step": global_step})
        engine.wait_for_everyone()
        if done:
            break
    dog.close()
    logger.close()
    if cfg.output_dir and engine.is_main_process:
        ckpt.save_pretrained(model, cfg.output_dir)  # the trained model as an HF directory (config.json + safetensors)
    if hasattr(model, "check_replicas"):
        model.check_replicas()  # raises if a gradient ever missed its reduction
        last_metric["replicas_bit_equal"] = True
    last_metric["optimizer_steps"] = global_step
    last_metric["wall_s"] = time.perf_counter() - t_start
    return last_metric


# ---------------------------------------------------------------------------------------------------------
def run_model_parallel(rank: int, world_size: int, cfg: TrainConfig) -> dict:
    """Data parallel x model parallel.  ``mp_layout="pipeline"``: each data-parallel replica is a pair of
    ranks holding a 2-stage layer split, micro-batched 1F1B over RCCL send/recv (world_size must be even).
    ``mp_layout="towers"``: the reference's TriBert ensemble (three encoders, mean of last hidden states,
    shared embedding / pooler / classifier) with the towers spread over the replica's stage ranks."""
    from .parallel.pp import run_pipeline_training

    return run_pipeline_training(rank, world_size, cfg)
