"""2-stage layer-split model parallelism, micro-batched (1F1B), composed with data parallelism.

The reference's model-parallel idea - stage 1 = embeddings + a stack of layers on one device, the hidden
states and the mask hop to a second device, stage 2 = the rest of the stack + pooler + classifier + loss
(``ConcatBert``, /root/reference/test_model_parallelism.py:40-89) - is done here with one process per GPU:
activations travel forward and their gradients backward over send/recv of the pipeline group ("nccl" =
RCCL over the direct xGMI link of the GPU pair; gloo on CPU), a batch is cut into micro-batches so that
both stages work at the same time (bubble fraction 1/(m+1) for two stages), and each stage is additionally
data-parallel across replicas through its own bucket reducer.  Only hidden states are exchanged: both stage
ranks of a replica read the same batches, so the attention mask and the labels never cross the link.

Send/recv pairs that must progress together are issued as ONE batched P2P group
(``send next activation`` + ``recv previous gradient``), otherwise the two ranks would wait on each other's
matching call.
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from .comm import backend_choice, create_rccl_comm
from .dp import DataParallel
from .groups import ParallelMesh
from ..utils.profile import trace_range


def split_layers(num_layers: int, stages: int) -> List[Tuple[int, int]]:
    """Contiguous layer ranges, as even as possible (earlier stages get the remainder)."""
    base, rem = divmod(num_layers, stages)
    out, lo = [], 0
    for s in range(stages):
        hi = lo + base + (1 if s < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def build_stage(model_cfg, stage: int, stages: int, full_state: Optional[Dict[str, torch.Tensor]] = None,
                seed: Optional[int] = None) -> nn.Module:
    """Module of one pipeline stage.  Its initial weights are the corresponding slice of the FULL model
    built from `seed` (or of `full_state`), so a pipelined run starts from the same point as a
    single-process run - a precondition for matching loss curves."""
    from ..models.bert import build_model

    lo, hi = split_layers(model_cfg.num_hidden_layers, stages)[stage]
    if full_state is None:
        if seed is not None:
            torch.manual_seed(seed)
        full_state = build_model(model_cfg).state_dict()
    part = build_model(model_cfg, layer_range=(lo, hi), with_embeddings=stage == 0, with_head=stage == stages - 1)
    own = part.state_dict()
    missing = [k for k in own if k not in full_state]
    if missing:
        raise KeyError(f"stage {stage}: parameters missing from the full state dict: {missing[:4]}")
    part.load_state_dict({k: full_state[k] for k in own})
    return part


_PACKED_KEYS = ("cu_seqlens", "token_index", "lengths")


def _strip_packing(batch: Dict) -> Dict:
    """The batch without the unpadded-execution entries of the collator (padded execution)."""
    return {k: v for k, v in batch.items() if k not in _PACKED_KEYS}


def _split_batch(batch: Dict, m: int) -> List[Dict]:
    """Cut a batch into `m` micro-batches along the sample dimension.  A batch that carries the collator's packing
    entries (``cu_seqlens`` [B+1], ``token_index`` [T], and the host-side list ``lengths``) is cut so that every
    micro-batch is again a self-contained packed batch: its ``cu_seqlens`` start at 0, its ``token_index`` are
    flat positions inside its own [b, S] block, and ``tokens`` is its row count as a host integer (the size of the
    block that crosses the link) - all without reading anything back from the device."""
    n = batch["attention_mask"].shape[0] if "attention_mask" in batch else next(iter(batch.values())).shape[0]
    m = max(1, min(m, n))
    bounds = [round(i * n / m) for i in range(m + 1)]
    packed = all(k in batch for k in _PACKED_KEYS)
    if not packed:
        batch = _strip_packing(batch)
        return [{k: v[bounds[i]:bounds[i + 1]] for k, v in batch.items()} for i in range(m)]
    lengths = [int(v) for v in batch["lengths"]]
    seq = batch["attention_mask"].shape[1]
    starts = [0]
    for v in lengths:
        starts.append(starts[-1] + v)
    out = []
    for i in range(m):
        b0, b1 = bounds[i], bounds[i + 1]
        t0, t1 = starts[b0], starts[b1]
        c = {k: v[b0:b1] for k, v in batch.items() if k not in _PACKED_KEYS}
        c["cu_seqlens"] = batch["cu_seqlens"][b0:b1 + 1] - t0
        c["token_index"] = batch["token_index"][t0:t1] - b0 * seq
        c["tokens"] = t1 - t0
        out.append(c)
    return out


def _p2p(ops, comm=None, pp: int = 2) -> None:
    """Run a group of point-to-point ops that must progress together.  On "nccl" (RCCL) they are fused into
    one group call on the communicator's stream; gloo has no device transport for send/recv, so GPU tensors
    are staged through host memory there (CPU plumbing / single-GPU rehearsal only).  With `comm` (an RcclComm
    over the replica's pipeline group, ``PDT_COMM=rccl``) the group goes straight to ncclSend / ncclRecv on the
    framework's own communication stream; a peer's rank inside that group is its stage index."""
    if comm is not None:
        comm.p2p([0 if op.op is dist.isend else 1 for op in ops], [op.tensor for op in ops], [op.peer % pp for op in ops])
        return
    if dist.get_backend() == "gloo" and any(op.tensor.is_cuda for op in ops):
        staged, works = [], []
        for op in ops:
            host = op.tensor.detach().cpu() if op.op is dist.isend else torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
            staged.append((op, host))
            works.append(op.op(host, op.peer))
        for w in works:
            w.wait()
        for op, host in staged:
            if op.op is dist.irecv:
                op.tensor.copy_(host)
        return
    for w in dist.batch_isend_irecv(ops):
        w.wait()


def replica_comm(mesh: ParallelMesh, engine):
    """The framework's own RCCL communicator over the ranks of this replica (its model-parallel group), or None where
    the point-to-point traffic goes through the process group: CPU / gloo runs, ``--comm-backend c10d``, a communicator
    that failed its self test.  Collective over the replica the first time; remembered on the mesh afterwards."""
    if not hasattr(mesh, "_replica_comm"):
        comm = None
        if (engine.device.type == "cuda" and dist.is_initialized() and dist.get_backend() == "nccl"
                and backend_choice(getattr(engine, "comm_backend", None)) == "rccl"):
            comm = create_rccl_comm(mesh.pp_group, engine.device)
        mesh._replica_comm = comm
    return mesh._replica_comm


class PipelineStage:
    """Driver of one rank of a 2-stage pipeline replica."""

    def __init__(self, module: nn.Module, mesh: ParallelMesh, engine, micro_batches: int = 4):
        if mesh.pp != 2:
            raise ValueError("the pipeline schedule is written for exactly 2 stages")
        self.module, self.mesh, self.engine, self.m = module, mesh, engine, micro_batches
        self.dp = module if isinstance(module, DataParallel) else None
        if self.dp is not None:
            self.dp.auto_arm = False
        inner = module.module if self.dp is not None else module
        self.inner = inner
        arena = getattr(inner, "arena", None)
        self.act_dtype = arena.p16.dtype if (arena is not None and arena.p16 is not None) else torch.float32
        self.hidden = inner.config.hidden_size
        self.first, self.last = mesh.is_first_stage, mesh.is_last_stage
        self.prev, self.next = mesh.peer(0), mesh.peer(1)
        self.timers = {"p2p_wait_s": 0.0}
        self.link_bytes = 0  # bytes this rank has put on / taken off the link (activations and their gradients)
        # unpadded execution needs the native kernels on both stages; both ranks of a replica decide alike (same
        # build, same device type, same precision mode)
        self.packed = getattr(inner, "runtime", None) is not None and engine.device.type == "cuda"
        self.comm = replica_comm(mesh, engine)

    # -- communication -------------------------------------------------------------------------------
    def _buf(self, chunk: Dict[str, torch.Tensor]) -> torch.Tensor:
        if "tokens" in chunk:  # packed micro-batch: only the rows of real tokens cross the link
            return torch.empty(chunk["tokens"], self.hidden, dtype=self.act_dtype, device=self.engine.device)
        b, s = chunk["attention_mask"].shape
        return torch.empty(b, s, self.hidden, dtype=self.act_dtype, device=self.engine.device)

    def _chunks(self, batch: Dict, m: int) -> List[Dict]:
        return _split_batch(batch if self.packed else _strip_packing(batch), m)

    def _send(self, t: torch.Tensor, dst: int) -> None:
        self.link_bytes += t.numel() * t.element_size()
        with trace_range("pp.send"):
            _p2p([dist.P2POp(dist.isend, t.contiguous(), dst)], self.comm, self.mesh.pp)

    def _recv(self, buf: torch.Tensor, src: int) -> torch.Tensor:
        self.link_bytes += buf.numel() * buf.element_size()
        with trace_range("pp.recv"):
            _p2p([dist.P2POp(dist.irecv, buf, src)], self.comm, self.mesh.pp)
        return buf

    def _send_recv(self, t: torch.Tensor, buf: torch.Tensor, peer: int) -> torch.Tensor:
        self.link_bytes += t.numel() * t.element_size() + buf.numel() * buf.element_size()
        with trace_range("pp.send_recv"):
            _p2p([dist.P2POp(dist.isend, t.contiguous(), peer), dist.P2POp(dist.irecv, buf, peer)], self.comm, self.mesh.pp)
        return buf

    def _arm(self) -> None:
        if self.dp is not None:
            self.dp.prepare_backward()

    # -- one optimizer step worth of micro-batches -----------------------------------------------------
    def train_step(self, batch: Dict[str, torch.Tensor]) -> Optional[torch.Tensor]:
        """Forward + backward of `batch` cut into micro-batches.  Returns the batch-mean loss on the last
        stage (None on the first).  Gradients accumulate in the arena; the data-parallel all-reduce is armed
        only for the final micro-batch's backward."""
        chunks = self._chunks(batch, self.m)
        m, total = len(chunks), batch["attention_mask"].shape[0]
        self.module.train()
        if self.first:
            outs: List[Optional[torch.Tensor]] = []
            out = self._fwd_first(chunks[0])
            self._send(out.detach(), self.next)
            outs.append(out)
            for i in range(m - 1):
                nxt = self._fwd_first(chunks[i + 1])
                outs.append(nxt)
                g = self._send_recv(nxt.detach(), self._buf(chunks[i]), self.next)
                outs[i].backward(g)
                outs[i] = None
            g = self._recv(self._buf(chunks[m - 1]), self.next)
            self._arm()
            outs[m - 1].backward(g)
            return None
        loss_sum = None
        act = self._recv(self._buf(chunks[0]), self.prev)
        for i in range(m):
            act.requires_grad_()
            c = chunks[i]
            out = self.module(hidden_states=act, attention_mask=c["attention_mask"], labels=c["labels"],
                              sample_ids=c.get("sample_ids"), cu_seqlens=c.get("cu_seqlens"),
                              token_index=c.get("token_index"))
            loss = out.loss * (c["labels"].shape[0] / total)  # exact batch-mean over unequal micro-batches
            if i == m - 1:
                self._arm()
            self.engine.backward(loss)
            g = act.grad
            if i < m - 1:
                act = self._send_recv(g, self._buf(chunks[i + 1]), self.prev)
            else:
                self._send(g, self.prev)
            loss_sum = loss.detach() if loss_sum is None else loss_sum + loss.detach()
        return loss_sum

    def _fwd_first(self, c: Dict[str, torch.Tensor]) -> torch.Tensor:
        out = self.module(input_ids=c["input_ids"], attention_mask=c["attention_mask"],
                          token_type_ids=c.get("token_type_ids"), sample_ids=c.get("sample_ids"),
                          cu_seqlens=c.get("cu_seqlens"), token_index=c.get("token_index"))
        return out.last_hidden_state

    @torch.no_grad()
    def eval_step(self, batch: Dict[str, torch.Tensor]) -> Optional[torch.Tensor]:
        """Forward only; logits on the last stage."""
        self.module.eval()
        batch = self._chunks(batch, 1)[0]
        if self.first:
            self._send(self._fwd_first(batch), self.next)
            return None
        act = self._recv(self._buf(batch), self.prev)
        return self.module(hidden_states=act, attention_mask=batch["attention_mask"], cu_seqlens=batch.get("cu_seqlens"),
                           token_index=batch.get("token_index")).logits

    def optimizer_step(self, optimizer, scheduler=None, max_grad_norm: float = 0.0) -> None:
        if max_grad_norm > 0:  # the norm of the whole replica: squared norms of the two stages summed over the pair
            self.engine.clip_grad_norm_(self.module, max_grad_norm, group=self.mesh.pp_group)
        optimizer.step(overflow_group=self.mesh.pp_group if self.engine.scaler.enabled else None)
        if scheduler is not None:
            scheduler.step()
        optimizer.zero_grad()


# ---------------------------------------------------------------------------------------------------------
def run_pipeline_training(rank: int, world_size: int, cfg) -> dict:
    """DP x 2-stage-MP training recipe (see recipes.run_model_parallel)."""

    from .. import checkpoint as ckpt
    from ..engine import Engine
    from ..metrics import load_metric
    from ..models.config import preset
    from ..optim import AdamW, get_linear_schedule_with_warmup
    from ..recipes import _collator, build_datasets, model_parallel_loaders
    from ..utils.logging import StepLogger
    from ..utils.seed import seed_step, set_seed
    from ..utils.watchdog import StepWatchdog, inject_fault

    if cfg.mp_layout == "towers":
        from ..models.tribert import run_tower_training

        return run_tower_training(rank, world_size, cfg)
    from .. import launch

    env, device = launch.init_distributed(cpu=cfg.cpu)
    print("LOCAL_RANK: ", rank)
    print("world_size: ", world_size)
    print("MASTER_ADDR:", os.environ.get("MASTER_ADDR"))
    print("MASTER_PORT:", os.environ.get("MASTER_PORT"))
    mesh = ParallelMesh(world_size, rank, pp=2).build_groups()
    engine = Engine(fp16=cfg.fp16, bf16=cfg.bf16, cpu=cfg.cpu, seed=cfg.seed, process_group=mesh.dp_group,
                    min_bucket_mb=cfg.min_bucket_mb, init_scale=cfg.init_scale,
                    comm_backend=cfg.comm_backend, update_in_backward=False)
    model_cfg = preset(cfg.model) if not cfg.init_from else ckpt.EncoderConfig.load(cfg.init_from)
    train_ds, val_ds = build_datasets(cfg, model_cfg)
    accum, micro_bs = cfg.accumulation(engine.device, model_cfg)
    collate = _collator(cfg, model_cfg)
    # one loader batch = one optimizer step = `accum` pipeline micro-batches
    train_dl, eval_dl = model_parallel_loaders(cfg, train_ds, val_ds, collate, mesh, micro_bs * accum)
    set_seed(cfg.seed)
    full_state = ckpt._read_weights(cfg.init_from) if cfg.init_from else None
    stage_model = build_stage(model_cfg, mesh.stage, 2, full_state, seed=cfg.seed)
    print(f"[RANK: {rank}] Initialized model (stage {mesh.stage}, replica {mesh.replica}).")
    optimizer = AdamW(params=stage_model.parameters(), lr=cfg.lr, correct_bias=cfg.correct_bias)
    print(f"[RANK: {rank}] Initialized optimizer.")
    stage_model, optimizer, train_dl, eval_dl = engine.prepare(stage_model, optimizer, train_dl, eval_dl)
    print(f"[RANK: {rank}] All prepared.")
    steps_per_epoch = len(train_dl)
    sched = get_linear_schedule_with_warmup(optimizer, cfg.num_warmup_steps, steps_per_epoch * cfg.num_epochs)
    pipe = PipelineStage(stage_model, mesh, engine, micro_batches=accum if accum > 1 else cfg.micro_batches)
    metric = load_metric("glue", "mrpc")
    log_rank = mesh.is_last_stage and engine.is_main_process
    logger = StepLogger(log_rank, cfg.log_every, cfg.log_file, samples_per_step=micro_bs * accum * mesh.dp)
    suffix = f"stage{mesh.stage}"
    start_epoch, global_step, start_batch = 0, 0, 0
    resume_dir = cfg.resume or cfg.output_dir
    # the two stages publish their checkpoints one after the other: resume from the newest one BOTH completed
    tag = ckpt.common_tag(resume_dir, suffix) if resume_dir else None
    if tag is not None:
        extra = ckpt.load_training_state(resume_dir, stage_model, optimizer, sched, suffix, tag=tag)
        start_epoch, global_step = int(extra.get("epoch", 0)), int(extra.get("global_step", 0))
        start_batch = int(extra.get("micro_offset", 0))
        engine.print(f"resumed from {resume_dir}/checkpoint-{tag}: epoch {start_epoch}, batch {start_batch}")
    dog = StepWatchdog(float(os.environ.get("PDT_WATCHDOG_S", "0")), rank)
    seed_step(cfg.seed, rank, global_step)

    def save(epoch: int, offset: int) -> None:
        if cfg.output_dir and engine.is_main_process:  # data-parallel rank 0 of each stage writes its part
            ckpt.save_training_state(cfg.output_dir, stage_model, optimizer, sched,
                                     {"epoch": epoch, "global_step": global_step, "micro_offset": offset}, suffix,
                                     write_config=mesh.stage == 0)

    losses: List[float] = []
    result: dict = {}
    t0 = time.perf_counter()
    done = False
    for epoch in range(start_epoch, cfg.num_epochs):
        train_dl.set_epoch(epoch)
        first = start_batch if epoch == start_epoch else 0
        if first:
            train_dl.skip_batches(first)
        for index, batch in enumerate(train_dl, start=first):
            loss = pipe.train_step(batch)
            pipe.optimizer_step(optimizer, sched, cfg.max_grad_norm)
            global_step += 1
            seed_step(cfg.seed, rank, global_step)
            dog.heartbeat()
            if cfg.save_every and global_step % cfg.save_every == 0:
                save(epoch, index + 1)
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
        seen = 0
        for batch in eval_dl:
            logits = pipe.eval_step(batch)
            if logits is not None:
                preds, refs = engine.gather(logits.argmax(dim=-1)), engine.gather(batch["labels"])
                valid = max(0, min(preds.shape[0], len(val_ds) - seen))
                seen += preds.shape[0]
                metric.add_batch(predictions=preds, references=refs, valid=valid)
        result = metric.compute() if mesh.is_last_stage else {}
        logger.log_epoch(epoch, result)
        save(epoch + 1, 0)
        dist.barrier()
        if done:
            break
    dog.close()
    replicas_equal = None
    if pipe.dp is not None:
        pipe.dp.check_replicas()  # raises if a gradient ever missed its reduction
        replicas_equal = True
    # every rank returns the metric of the last stage of replica 0
    box = [result if (mesh.is_last_stage and mesh.replica == 0) else None]
    dist.broadcast_object_list(box, src=ParallelMesh(world_size, 0, 2).peer(1))
    out = dict(box[0] or {})
    out.update({"optimizer_steps": global_step, "replicas_bit_equal": replicas_equal, "wall_s": time.perf_counter() - t0, "losses": losses,
                "stage": mesh.stage, "replica": mesh.replica, "packed": pipe.packed and bool(cfg.varlen),
                "link_mb": pipe.link_bytes / 1e6})
    logger.close()
    return out
