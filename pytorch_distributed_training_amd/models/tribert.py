"""The multi-encoder model-parallel modules of the reference, on this framework.

* :class:`TriBert` (/root/reference/test_model_parallelism.py:92-163): three BERT classifiers; the word
  embedding, pooler and classifier of tower 1 are shared; the three encoders run on the same
  ``inputs_embeds``, their last hidden states are averaged, then pooler -> dropout -> classifier -> CE.
  Towers 2/3 keep (unused, gradient-free) embeddings / poolers / classifiers exactly like the reference,
  which is why it needs ``find_unused_parameters=True`` - here unused arena slices simply stay zero.
  Only token 0 of the averaged hidden states reaches the pooler, so the mean is taken on ``[B, H]``.
* :class:`ConcatBert` (:40-89, unused there): stage 1 = all of ``bert_1``, stage 2 = ``bert_2``'s encoder
  stack + pooler + dropout + classifier + CE, with the additive mask built by
  ``get_extended_attention_mask``.  The pipeline of ``parallel/pp.py`` is the multi-process form of it.
* :func:`run_tower_training`: TriBert with the towers spread over the two ranks of a replica (tower 1 +
  shared parts on the first rank, towers 2 and 3 on the second - the reference's 1:2 placement), composed
  with data parallelism across replicas.  The embedding output travels once (the reference copies it twice),
  and only the token-0 rows come back.

Inside one process every tower lives on that process's GPU; the reference's ``first_device`` /
``second_device`` arguments are accepted and honoured on the PyTorch path (fp32), where a plain
``Tensor.to`` moves activations as in the reference.
"""
from __future__ import annotations

import os
import time
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from .bert import SequenceClassifierOutput, build_model


class TriBert(nn.Module):
    def __init__(self, bert_1, bert_2, bert_3, first_device=None, second_device=None):
        super().__init__()
        self.first_device = torch.device(first_device) if first_device is not None else None
        self.second_device = torch.device(second_device) if second_device is not None else None
        self.bert_1, self.bert_2, self.bert_3 = bert_1, bert_2, bert_3
        if self.first_device is not None:
            self.bert_1.to(self.first_device)
        if self.second_device is not None:
            self.bert_2.to(self.second_device)
            self.bert_3.to(self.second_device)
        self.config = bert_1.config
        self.num_labels = bert_1.num_labels
        self.arena = None
        self.runtime = None
        self.verbose = False

    # -- arena / native plumbing -----------------------------------------------------------------------
    def arena_segments(self):
        segs, heads = [], []
        for k, b in enumerate((self.bert_1, self.bert_2, self.bert_3), 1):
            for s in b.arena_segments(prefix=f"bert_{k}."):
                (heads if s.name.endswith("head") else segs).append(s)
        return segs + list(reversed(heads))  # tower 1's (the used) head completes first in backward

    def enable_native(self, arena, seed: int = 0, global_seed=None):
        from ..ops import fused

        rt = fused.NativeRuntime(arena, seed, global_seed)
        for k, b in enumerate((self.bert_1, self.bert_2, self.bert_3), 1):
            b.enable_native(arena, prefix=f"bert_{k}.", runtime=rt, site_offset=1024 * k)
        self.arena, self.runtime = arena, rt
        return rt

    # -- forward ---------------------------------------------------------------------------------------
    def _tower(self, bert, inputs_embeds, attention_mask, token_type_ids, device):
        if device is not None and inputs_embeds.device != device:
            inputs_embeds, attention_mask, token_type_ids = (t.to(device) for t in (inputs_embeds, attention_mask, token_type_ids))
        return bert.base_model(inputs_embeds=inputs_embeds, attention_mask=attention_mask,
                               token_type_ids=token_type_ids).last_hidden_state

    def forward(self, **kwargs):
        input_ids = kwargs.pop("input_ids")
        attention_mask = kwargs.pop("attention_mask")
        token_type_ids = kwargs.pop("token_type_ids")
        labels = kwargs.pop("labels", None)
        sample_ids = kwargs.pop("sample_ids", None)
        cu, tok = kwargs.pop("cu_seqlens", None), kwargs.pop("token_index", None)
        if self.verbose:
            print("Input shape ", input_ids.size())
        if self.runtime is not None:
            self.runtime.begin_forward(sample_ids, train=self.training and torch.is_grad_enabled())
        # the same embedding, pooler and classifier for the three towers
        word_embed = self.bert_1.get_input_embeddings()
        if cu is not None and tok is not None and self.runtime is not None and input_ids.is_cuda:
            # unpadded execution: one lookup of the real tokens, every tower on the packed [T, H] rows
            firsts = _packed_towers((self.bert_1, self.bert_2, self.bert_3), word_embed(input_ids.reshape(-1)[tok]),
                                    token_type_ids, sample_ids, cu, tok, input_ids.shape)
            first = (firsts[0].float() + firsts[1].float() + firsts[2].float()) / 3.0
            out = self.bert_1.head(first.to(firsts[0].dtype), labels)
            return SequenceClassifierOutput(loss=out.loss, logits=out.logits, hidden_states=None, attentions=None)
        inputs_embeds = word_embed(input_ids)
        h1 = self._tower(self.bert_1, inputs_embeds, attention_mask, token_type_ids, self.first_device)
        h2 = self._tower(self.bert_2, inputs_embeds, attention_mask, token_type_ids, self.second_device)
        h3 = self._tower(self.bert_3, inputs_embeds, attention_mask, token_type_ids, self.second_device)
        dev = h1.device
        # mean of the last hidden states; the pooler reads token 0 only, so only those rows are combined
        first = (h1[:, :1].float() + h2[:, :1].to(dev).float() + h3[:, :1].to(dev).float()) / 3.0
        out = self.bert_1.head(first.to(h1.dtype), labels.to(dev) if labels is not None else None)
        return SequenceClassifierOutput(loss=out.loss, logits=out.logits, hidden_states=None, attentions=None)


def _packed_towers(towers, inputs_embeds, token_type_ids, sample_ids, cu, tok, padded_shape):
    """Every tower on the packed [T, H] word embeddings of an unpadded batch; returns the towers' token-0 rows
    ([B, 1, H] each), the only rows the ensemble combines."""
    rows = cu[:-1].to(torch.int64)
    out = []
    for t in towers:
        h = t._forward_unpadded(None, token_type_ids, None, sample_ids, cu, tok, padded_shape=tuple(padded_shape),
                                inputs_embeds=inputs_embeds, run_head=False).last_hidden_state
        out.append(h.index_select(0, rows).unsqueeze(1))
    return out


class ConcatBert(nn.Module):
    """Two-stage layer split inside one process (see the module docstring)."""

    def __init__(self, bert_1, bert_2, device=None, second_device=None):
        super().__init__()
        self.bert_1, self.bert_2 = bert_1, bert_2
        self.device_1 = torch.device(device) if device is not None else None
        self.second_device = torch.device(second_device) if second_device is not None else self.device_1
        if self.device_1 is not None:
            self.bert_1.to(self.device_1)
        if self.second_device is not None:
            self.bert_2.to(self.second_device)
        self.config = bert_2.config

    def forward(self, *args, **kwargs):
        labels = kwargs.pop("labels", None)
        kwargs.pop("sample_ids", None)
        input_shape = kwargs["input_ids"].size()
        hidden_states = self.bert_1.base_model(*args, **kwargs).last_hidden_state
        dev = self.second_device or hidden_states.device
        ext = self.bert_2.get_extended_attention_mask(kwargs["attention_mask"], input_shape, dev)
        hidden_states = hidden_states.to(dev)
        sequence_output = self.bert_2.base_model.encoder(hidden_states=hidden_states, attention_mask=ext)[0]
        out = self.bert_2.head(sequence_output, labels.to(dev) if labels is not None else None)
        return SequenceClassifierOutput(loss=out.loss, logits=out.logits, hidden_states=None, attentions=None)


def build_tribert(name_or_cfg="bert-base-cased", first_device=None, second_device=None, seed: Optional[int] = None) -> TriBert:
    """Three independently initialised classifiers, like three ``from_pretrained`` calls give three modules."""
    if seed is not None:
        torch.manual_seed(seed)
    return TriBert(build_model(name_or_cfg), build_model(name_or_cfg), build_model(name_or_cfg), first_device, second_device)


# ---------------------------------------------------------------------------------------------------------
class _TowerRank(nn.Module):
    """What one rank of a tower-parallel replica owns: rank 0 -> tower 1 (+ shared embedding / pooler /
    classifier), rank 1 -> towers 2 and 3."""

    def __init__(self, towers: List[nn.Module], indices: List[int]):
        super().__init__()
        self.indices = indices
        for k, t in zip(indices, towers):
            setattr(self, f"bert_{k}", t)
        self.config = towers[0].config
        self.arena = None
        self.runtime = None

    def towers(self):
        return [getattr(self, f"bert_{k}") for k in self.indices]

    def arena_segments(self):
        segs, heads = [], []
        for k in self.indices:
            for s in getattr(self, f"bert_{k}").arena_segments(prefix=f"bert_{k}."):
                (heads if s.name.endswith("head") else segs).append(s)
        return segs + list(reversed(heads))

    def enable_native(self, arena, seed: int = 0, global_seed=None):
        from ..ops import fused

        rt = fused.NativeRuntime(arena, seed, global_seed)
        for k in self.indices:
            getattr(self, f"bert_{k}").enable_native(arena, prefix=f"bert_{k}.", runtime=rt, site_offset=1024 * k)
        self.arena, self.runtime = arena, rt
        return rt


def _p2p(ops, mesh, engine):
    """Activations and their gradients between the two ranks of a replica: device to device over the replica's own
    RCCL communicator, like the pipeline layout (the reference moves them with ``.to(device)``,
    /root/reference/test_model_parallelism.py:127-143); through the process group where there is none."""
    from ..parallel.pp import _p2p as run, replica_comm

    run(ops, replica_comm(mesh, engine), mesh.pp)


def tower_step(module, mesh, engine, batch, train: bool = True):
    """One forward(+backward) of a tower-parallel replica.  Returns (loss, logits) on the first rank."""
    inner = module.module if hasattr(module, "module") and isinstance(module.module, _TowerRank) else module
    dp = module if inner is not module else None
    rt = inner.runtime
    if dp is not None:
        # the armed window of the first rank spans two backward() calls (tower 1 + head, then the embedding lookup),
        # so it is closed explicitly after the second one instead of by autograd's end-of-backward callback
        dp.auto_arm = False
        dp.manual_finish = True
    if rt is not None:
        rt.begin_forward(batch.get("sample_ids"), train=train)
    dev = engine.device
    ids, mask, types = batch["input_ids"], batch["attention_mask"], batch["token_type_ids"]
    B, S = ids.shape
    H = inner.config.hidden_size
    arena = inner.arena
    adt = arena.p16.dtype if (arena is not None and arena.p16 is not None) else torch.float32
    peer0, peer1 = mesh.peer(0), mesh.peer(1)
    # unpadded execution (native kernels, collator with packing entries): the word embeddings and their gradient
    # cross the link as packed [T, H] rows of the real tokens, the towers run on those rows
    packed = rt is not None and dev.type == "cuda" and all(k in batch for k in ("cu_seqlens", "token_index", "lengths"))
    cu, tok = (batch["cu_seqlens"], batch["token_index"]) if packed else (None, None)
    emb_shape = (sum(int(v) for v in batch["lengths"]), H) if packed else (B, S, H)
    sids = batch.get("sample_ids")

    def run_towers(towers, emb_in):
        if packed:
            return _packed_towers(towers, emb_in, types, sids, cu, tok, (B, S))
        return [t.base_model(inputs_embeds=emb_in, attention_mask=mask, token_type_ids=types).last_hidden_state[:, :1]
                for t in towers]

    with torch.set_grad_enabled(train):
        if mesh.stage == 0:
            b1 = inner.bert_1
            emb = b1.get_input_embeddings()(ids.reshape(-1)[tok] if packed else ids)
            _p2p([dist.P2POp(dist.isend, emb.detach().to(adt).contiguous(), peer1)], mesh, engine)
            # tower 1 reads a detached copy, so the embedding lookup is differentiated ONCE, at the very end,
            # with the sum of this rank's and the peer's gradient
            emb_in = emb.detach().requires_grad_() if train else emb
            h1 = run_towers([b1], emb_in)[0]
            s23 = torch.empty(B, 1, H, dtype=torch.float32, device=dev)
            _p2p([dist.P2POp(dist.irecv, s23, peer1)], mesh, engine)
            if train:
                s23.requires_grad_()
            first = (h1.float() + s23) / 3.0
            out = b1.head(first.to(h1.dtype), batch.get("labels"))
            if not train:
                return out.loss, out.logits
            scale = engine.scaler.scale_tensor() if engine.scaler.enabled else None
            loss = out.loss * scale.to(out.loss.dtype) if scale is not None else out.loss
            emb_seg = arena.seg_index.get("bert_1.embeddings") if arena is not None else None
            if rt is not None and emb_seg is not None:
                rt.hold.add(emb_seg)  # its word-embedding gradient only arrives with the second backward below
            if dp is not None:
                dp.prepare_backward()  # layer buckets are reduced while the rest of this backward still runs
            # tower-1 / head gradients; d(s23) goes to the second rank, which answers with d(embedding)
            loss.backward()
            d_emb = torch.empty(*emb_shape, dtype=adt, device=dev)
            _p2p([dist.P2POp(dist.isend, s23.grad.contiguous(), peer1), dist.P2POp(dist.irecv, d_emb, peer1)], mesh, engine)
            emb.backward(emb_in.grad.to(emb.dtype) + d_emb.to(emb.dtype))
            if rt is not None:
                rt.release()
            if dp is not None:
                dp.finish_backward()  # the embedding bucket (and anything unused) is reduced here; then the join
            return out.loss.detach(), out.logits
        emb = torch.empty(*emb_shape, dtype=adt, device=dev)
        _p2p([dist.P2POp(dist.irecv, emb, peer0)], mesh, engine)
        if train:
            emb.requires_grad_()
        hs = run_towers(inner.towers(), emb)
        s = hs[0].float() + hs[1].float()
        _p2p([dist.P2POp(dist.isend, s.detach().contiguous(), peer0)], mesh, engine)
        if not train:
            return None, None
        ds = torch.empty_like(s)
        _p2p([dist.P2POp(dist.irecv, ds, peer0)], mesh, engine)
        if dp is not None:
            dp.prepare_backward()
        s.backward(ds)
        if dp is not None:
            dp.finish_backward()
        _p2p([dist.P2POp(dist.isend, emb.grad.contiguous(), peer0)], mesh, engine)
        return None, None


def run_tower_training(rank: int, world_size: int, cfg) -> dict:
    """TriBert training with the towers spread over rank pairs, data parallel across pairs."""

    from .. import checkpoint as ckpt
    from .. import launch
    from ..engine import Engine
    from ..metrics import load_metric
    from ..models.config import preset
    from ..optim import AdamW, get_linear_schedule_with_warmup
    from ..parallel.groups import ParallelMesh
    from ..recipes import _collator, build_datasets, model_parallel_loaders
    from ..utils.logging import StepLogger
    from ..utils.seed import seed_step, set_seed
    from ..utils.watchdog import StepWatchdog, inject_fault

    launch.init_distributed(cpu=cfg.cpu)
    mesh = ParallelMesh(world_size, rank, pp=2).build_groups()
    engine = Engine(fp16=cfg.fp16, bf16=cfg.bf16, cpu=cfg.cpu, seed=cfg.seed, process_group=mesh.dp_group,
                    min_bucket_mb=cfg.min_bucket_mb, init_scale=cfg.init_scale,
                    comm_backend=cfg.comm_backend, update_in_backward=False)
    model_cfg = preset(cfg.model)
    train_ds, val_ds = build_datasets(cfg, model_cfg)
    accum, micro_bs = cfg.accumulation(engine.device, model_cfg)
    collate = _collator(cfg, model_cfg)
    train_dl, eval_dl = model_parallel_loaders(cfg, train_ds, val_ds, collate, mesh, micro_bs)
    set_seed(cfg.seed)
    torch.manual_seed(cfg.seed)
    towers = [build_model(model_cfg) for _ in range(3)]  # same three initialisations on every rank
    part = _TowerRank([towers[0]], [1]) if mesh.stage == 0 else _TowerRank(towers[1:], [2, 3])
    optimizer = AdamW(params=part.parameters(), lr=cfg.lr, correct_bias=cfg.correct_bias)
    part, optimizer, train_dl, eval_dl = engine.prepare(part, optimizer, train_dl, eval_dl)
    steps_per_epoch = -(-len(train_dl) // accum)
    sched = get_linear_schedule_with_warmup(optimizer, cfg.num_warmup_steps, steps_per_epoch * cfg.num_epochs)
    metric = load_metric("glue", "mrpc")
    log_rank = mesh.stage == 0 and engine.is_main_process
    logger = StepLogger(log_rank, cfg.log_every, cfg.log_file, samples_per_step=micro_bs * accum * mesh.dp)
    losses, result, global_step, done = [], {}, 0, False
    t0 = time.perf_counter()
    inner_dp = part if hasattr(part, "no_sync") else None
    # checkpoints: every tower rank saves its own towers; resume from the newest checkpoint both ranks completed
    suffix = f"tower{mesh.stage}"
    start_epoch, start_micro = 0, 0
    resume_dir = cfg.resume or cfg.output_dir
    tag = ckpt.common_tag(resume_dir, suffix) if resume_dir else None
    if tag is not None:
        extra = ckpt.load_training_state(resume_dir, part, optimizer, sched, suffix, tag=tag)
        start_epoch, global_step = int(extra.get("epoch", 0)), int(extra.get("global_step", 0))
        start_micro = int(extra.get("micro_offset", 0))
        engine.print(f"resumed from {resume_dir}/checkpoint-{tag}: epoch {start_epoch}, batch {start_micro}")
    dog = StepWatchdog(float(os.environ.get("PDT_WATCHDOG_S", "0")), rank)
    seed_step(cfg.seed, rank, global_step)

    def save(epoch: int, offset: int) -> None:
        if cfg.output_dir and engine.is_main_process:
            ckpt.save_training_state(cfg.output_dir, part, optimizer, sched,
                                     {"epoch": epoch, "global_step": global_step, "micro_offset": offset}, suffix,
                                     write_config=mesh.stage == 0)

    for epoch in range(start_epoch, cfg.num_epochs):
        part.train()
        train_dl.set_epoch(epoch)
        pending = 0
        n_micro = len(train_dl)
        first = start_micro if epoch == start_epoch else 0
        if first:
            train_dl.skip_batches(first)
        for step, batch in enumerate(train_dl, start=first):
            pending += 1
            fire = pending == accum or step == n_micro - 1
            if inner_dp is not None:
                inner_dp.require_sync = fire
            loss, _ = tower_step(part, mesh, engine, batch, train=True)
            if fire:
                if cfg.max_grad_norm > 0:  # (the two ranks of a replica hold disjoint parameters)
                    engine.clip_grad_norm_(part, cfg.max_grad_norm, group=mesh.pp_group)
                optimizer.step(overflow_group=mesh.pp_group if engine.scaler.enabled else None)
                sched.step()
                optimizer.zero_grad()
                pending = 0
                global_step += 1
                seed_step(cfg.seed, rank, global_step)
                dog.heartbeat()
                if cfg.save_every and global_step % cfg.save_every == 0:
                    save(epoch, step + 1)
                inject_fault(rank, global_step)
                if loss is not None:
                    record = global_step % cfg.log_every == 0 or bool(cfg.max_steps)
                    if record and mesh.dp > 1:
                        loss = engine.reduce_mean(loss)  # global-batch loss: mean over the replicas' shards
                    if log_rank and record:
                        losses.append(float(loss.item()))
                    logger.log_step(global_step, loss, sched.get_last_lr()[0], engine.scaler)
                if cfg.max_steps and global_step >= cfg.max_steps:
                    done = True
                    break
        part.eval()
        seen = 0
        for batch in eval_dl:
            _, logits = tower_step(part, mesh, engine, batch, train=False)
            if logits is not None:
                preds, refs = engine.gather(logits.argmax(dim=-1)), engine.gather(batch["labels"])
                valid = max(0, min(preds.shape[0], len(val_ds) - seen))
                seen += preds.shape[0]
                metric.add_batch(predictions=preds, references=refs, valid=valid)
        result = metric.compute() if mesh.stage == 0 else {}
        logger.log_epoch(epoch, result)
        save(epoch + 1, 0)
        dist.barrier()
        if done:
            break
    dog.close()
    replicas_equal = None
    if inner_dp is not None:
        inner_dp.check_replicas()  # raises if a gradient ever missed its reduction
        replicas_equal = True
    box = [result if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    out = dict(box[0] or {})
    out.update({"optimizer_steps": global_step, "replicas_bit_equal": replicas_equal, "wall_s": time.perf_counter() - t0, "losses": losses,
                "stage": mesh.stage, "replica": mesh.replica})
    logger.close()
    return out
