"""Rank functions of the multi-process CPU tests (top-level so that mp.spawn can import them)."""
import os

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from pytorch_distributed_training_amd.data import PadCollator, SyntheticPairDataset
from pytorch_distributed_training_amd.engine import Engine, gather
from pytorch_distributed_training_amd.models.bert import build_model
from pytorch_distributed_training_amd.optim import AdamW


def _model(seed=0):
    torch.manual_seed(seed)
    return build_model("bert-tiny", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


def _global_batch(n=8, s=12):
    g = torch.Generator().manual_seed(123)
    ids = torch.randint(1, 512, (n, s), generator=g)
    mask = torch.ones(n, s, dtype=torch.long)
    mask[1, 7:] = 0
    return dict(input_ids=ids, attention_mask=mask, token_type_ids=torch.zeros_like(ids),
                labels=torch.randint(0, 2, (n,), generator=g))


def dp_grads_match_big_batch(rank, world, split_mb=None):
    """DP gradients (average over ranks of per-shard mean loss) == gradients of the whole batch on one process.
    `split_mb`: cut every bucket into pieces of at most that many MB (half of it for the last bucket of backward)."""
    if split_mb is not None:
        import os

        os.environ["PDT_MAX_BUCKET_MB"], os.environ["PDT_TAIL_BUCKET_MB"] = str(split_mb), str(split_mb / 2)
    eng = Engine(cpu=True, min_bucket_mb=0.05)
    model = _model(seed=rank)  # different init per rank: prepare() must broadcast rank 0's weights
    model = eng.prepare(model)
    ref = _model(seed=0)
    inner = eng.unwrap_model(model)
    for (n, p), (_, q) in zip(inner.named_parameters(), ref.named_parameters()):
        assert torch.equal(p, q), f"broadcast failed for {n}"
    batch = _global_batch()
    shard = {k: v[rank::world] for k, v in batch.items()}
    model.train()
    eng.backward(model(**shard).loss)
    ref.train()
    ref(**batch).loss.backward()
    worst = 0.0
    for (n, p), (_, q) in zip(inner.named_parameters(), ref.named_parameters()):
        worst = max(worst, (p.grad - q.grad).abs().max().item())
    order = model.reducer.launch_order()
    nb = model.reducer.num_buckets()
    return {"worst": worst, "order": list(order), "nb": nb, "stats": dict(model.reducer.stats()),
            "pieces": [len(s.buckets) for s in inner.arena.segments]}


def dp_no_sync_accumulation(rank, world):
    """Two micro-batches, the first under no_sync: one reduction, equal to the 2-micro-batch big-batch gradient."""
    eng = Engine(cpu=True, min_bucket_mb=0.05)
    model = eng.prepare(_model())
    inner = eng.unwrap_model(model)
    batch = _global_batch(16)
    model.train()
    mb = [{k: v[i * 8:(i + 1) * 8][rank::world] for k, v in batch.items()} for i in range(2)]
    with eng.no_sync(model):
        eng.backward(model(**mb[0]).loss / 2)
    local_only = inner.arena.grad.clone()
    collectives_before = model.reducer.stats()["collectives"]
    eng.backward(model(**mb[1]).loss / 2)
    ref = _model()
    ref.train()
    (ref(**{k: v[:8] for k, v in batch.items()}).loss / 2).backward()
    (ref(**{k: v[8:] for k, v in batch.items()}).loss / 2).backward()
    worst = max((p.grad - q.grad).abs().max().item() for p, q in zip(inner.parameters(), ref.parameters()))
    gathered = [torch.zeros_like(local_only) for _ in range(world)]
    dist.all_gather(gathered, local_only)
    differs = not torch.equal(gathered[0], gathered[1])
    return {"worst": worst, "no_sync_kept_local": differs, "collectives_before": collectives_before,
            "collectives_after": model.reducer.stats()["collectives"], "nb": model.reducer.num_buckets()}


class _PartlyUsed(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 8)
        self.unused = torch.nn.Linear(8, 8)
        self.b = torch.nn.Linear(8, 2)

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


def dp_unused_parameters(rank, world):
    """Parameters outside the graph (the reference needs find_unused_parameters=True for TriBert): the step
    completes, their gradient stays zero, everything else is averaged."""
    eng = Engine(cpu=True, min_bucket_mb=1e-5)
    torch.manual_seed(0)
    model = eng.prepare(_PartlyUsed())
    x = torch.arange(32, dtype=torch.float32).reshape(4, 8) * (rank + 1)
    model.train()
    model(x).sum().backward()
    inner = eng.unwrap_model(model)
    ref = _PartlyUsed()
    ref.load_state_dict(inner.state_dict())
    tot = sum(ref(torch.arange(32, dtype=torch.float32).reshape(4, 8) * (r + 1)).sum() for r in range(world)) / world
    tot.backward()
    worst = max((p.grad - (q.grad if q.grad is not None else torch.zeros_like(q))).abs().max().item()
                for p, q in zip(inner.parameters(), ref.parameters()))
    return {"worst": worst, "unused_zero": bool(inner.unused.weight.grad.abs().sum() == 0),
            "order": list(model.reducer.launch_order()), "nb": model.reducer.num_buckets()}


def gather_semantics(rank, world):
    t = torch.arange(3) + 10 * rank
    out = {"vec": gather(t).tolist(), "scalar": gather(torch.tensor(rank)).tolist()}
    nested = gather({"a": torch.full((2, 2), rank), "b": [torch.tensor([rank]), (torch.tensor([rank + 5]),)]})
    out["nested_a"] = nested["a"].tolist()
    out["nested_b0"] = nested["b"][0].tolist()
    out["nested_b1_is_tuple"] = isinstance(nested["b"][1], tuple)
    try:
        gather("nope")
    except TypeError:
        out["type_error"] = True
    return out


def sharded_loader_and_training(rank, world):
    """End-to-end DP recipe pieces on 2 ranks: sharded loader lengths, identical parameters after training."""
    eng = Engine(cpu=True, seed=1)
    ds = SyntheticPairDataset(50, vocab_size=512, min_len=3, max_len=8, seed=0)
    dl = DataLoader(ds, shuffle=True, collate_fn=PadCollator(), batch_size=4)
    model = _model()
    opt = AdamW(params=model.parameters(), lr=1e-3)
    model, opt, dl = eng.prepare(model, opt, dl)
    n_batches, first_epoch_ids = len(dl), []
    model.train()
    for epoch in range(2):
        for batch in dl:
            if epoch == 0:
                first_epoch_ids.append(batch["input_ids"][:, 1].tolist())
            eng.backward(model(**batch).loss)
            opt.step()
            opt.zero_grad()
    params = eng.unwrap_model(model).arena.master.clone()
    both = [torch.zeros_like(params) for _ in range(world)]
    dist.all_gather(both, params)
    return {"len": n_batches, "same_params": bool(torch.equal(both[0], both[1])), "ids": first_epoch_ids}


def dp_recipe(rank, world, tmp):
    from pytorch_distributed_training_amd.config import TrainConfig
    from pytorch_distributed_training_amd.recipes import run_data_parallel

    cfg = TrainConfig(model="bert-tiny", cpu=True, train_examples=48, eval_examples=20, num_epochs=2, batch_size=16,
                      lr=1e-3, num_warmup_steps=2, log_every=100, output_dir=tmp if rank == 0 else tmp)
    return run_data_parallel(cfg)


def pipeline_recipe(rank, world, micro_bs, accum, compat=False, max_grad_norm=0.0):
    from pytorch_distributed_training_amd.config import TrainConfig
    from pytorch_distributed_training_amd.recipes import run_model_parallel

    cfg = TrainConfig(model="bert-tiny-nodrop", cpu=True, train_examples=64, eval_examples=16, num_epochs=2,
                      batch_size=micro_bs * accum, max_gpu_batch_size=micro_bs, lr=2e-3, num_warmup_steps=1,
                      log_every=1, max_steps=6, seed=5, compat=compat, max_grad_norm=max_grad_norm)
    return run_model_parallel(rank, world, cfg)


def single_process_reference(batch_size, steps=6, max_grad_norm=0.0):
    """Loss curve of ONE process on the whole global batch, same seed / data order / init as the pipeline runs."""
    from pytorch_distributed_training_amd.config import TrainConfig
    from pytorch_distributed_training_amd.models.config import preset
    from pytorch_distributed_training_amd.optim import get_linear_schedule_with_warmup
    from pytorch_distributed_training_amd.recipes import _collator, build_datasets
    from pytorch_distributed_training_amd.utils.seed import set_seed

    cfg = TrainConfig(model="bert-tiny-nodrop", cpu=True, train_examples=64, eval_examples=16, num_epochs=2,
                      batch_size=batch_size, lr=2e-3, num_warmup_steps=1, seed=5)
    eng = Engine(cpu=True, seed=5)
    mc = preset(cfg.model)
    train_ds, _ = build_datasets(cfg, mc)
    dl = DataLoader(train_ds, shuffle=True, collate_fn=_collator(cfg, mc), batch_size=batch_size)
    set_seed(5)
    torch.manual_seed(5)
    model = build_model(mc)
    opt = AdamW(params=model.parameters(), lr=2e-3)
    model, opt, dl = eng.prepare(model, opt, dl)
    sched = get_linear_schedule_with_warmup(opt, 1, len(dl) * 2)
    losses = []
    model.train()
    for epoch in range(2):
        dl.set_epoch(epoch)
        for batch in dl:
            loss = model(**batch).loss
            eng.backward(loss)
            if max_grad_norm > 0:
                eng.clip_grad_norm_(model, max_grad_norm)
            opt.step(); sched.step(); opt.zero_grad()
            losses.append(loss.item())
            if len(losses) >= steps:
                return losses
    return losses


def tower_recipe(rank, world, examples=32):
    from pytorch_distributed_training_amd.config import TrainConfig
    from pytorch_distributed_training_amd.recipes import run_model_parallel

    cfg = TrainConfig(model="bert-tiny-nodrop", cpu=True, train_examples=examples, eval_examples=8, num_epochs=1,
                      batch_size=8, max_gpu_batch_size=8, lr=2e-3, num_warmup_steps=1, log_every=1, max_steps=4,
                      seed=5, mp_layout="towers", min_bucket_mb=0.05)
    return run_model_parallel(rank, world, cfg)


def tribert_single_process(steps=4, batch_size=8, examples=32):
    """TriBert in one process on the same data / seeds as tower_recipe: the loss curve to match."""
    from pytorch_distributed_training_amd.config import TrainConfig
    from pytorch_distributed_training_amd.models.config import preset
    from pytorch_distributed_training_amd.models.tribert import TriBert
    from pytorch_distributed_training_amd.optim import get_linear_schedule_with_warmup
    from pytorch_distributed_training_amd.recipes import _collator, build_datasets
    from pytorch_distributed_training_amd.utils.seed import set_seed

    cfg = TrainConfig(model="bert-tiny-nodrop", cpu=True, train_examples=examples, eval_examples=8, num_epochs=1,
                      batch_size=batch_size, seed=5)
    eng = Engine(cpu=True, seed=5)
    mc = preset(cfg.model)
    train_ds, _ = build_datasets(cfg, mc)
    dl = DataLoader(train_ds, shuffle=True, collate_fn=_collator(cfg, mc), batch_size=batch_size)
    set_seed(5)
    torch.manual_seed(5)
    model = TriBert(build_model(mc), build_model(mc), build_model(mc))
    opt = AdamW(params=model.parameters(), lr=2e-3)
    model, opt, dl = eng.prepare(model, opt, dl)
    sched = get_linear_schedule_with_warmup(opt, 1, len(dl))
    losses = []
    model.train()
    dl.set_epoch(0)
    for batch in dl:
        loss = model(**batch).loss
        eng.backward(loss)
        opt.step(); sched.step(); opt.zero_grad()
        losses.append(loss.item())
        if len(losses) >= steps:
            break
    return losses


# ------------------------------------------------------------------------------------------ GPU rehearsals
# Two (or four) processes share the ONE GPU of the test box over gloo: RCCL refuses two ranks on one device, but
# the native kernels, the arena reducer, the pipeline schedule and the tower exchange run exactly as they do
# with one GPU per rank.
def gpu_dp_native(rank, world):
    eng = Engine(fp16=True, seed=7, min_bucket_mb=0.05, init_scale=256.0)
    assert eng.native and eng.device.type == "cuda"
    model = _model(seed=rank)
    opt = AdamW(params=model.parameters(), lr=1e-3)
    model, opt = eng.prepare(model, opt)
    inner = eng.unwrap_model(model)
    batch = {k: v.cuda() for k, v in _global_batch(16).items()}
    shard = {k: v[rank::world] for k, v in batch.items()}
    model.train()
    eng.backward(model(**shard).loss)
    torch.cuda.synchronize()
    scale = eng.scaler.get_scale()
    g_dp = inner.arena.grad.clone() / scale
    # single-process big batch on the same (broadcast) weights through the same native path
    ref = _model(seed=0).cuda()
    from pytorch_distributed_training_amd.ops.arena import ParamArena

    arena = ParamArena(ref, eng.device, torch.float16)
    ref.enable_native(arena, seed=1)
    ref.train()
    ref(**batch).loss.backward()
    err = ((g_dp - arena.grad).norm() / arena.grad.norm()).item()
    opt.step(); opt.zero_grad()
    both = [torch.zeros_like(inner.arena.master) for _ in range(world)]
    dist.all_gather(both, inner.arena.master)
    return {"err": err, "same": bool(torch.equal(both[0], both[1])), "order": list(model.reducer.launch_order()),
            "nb": model.reducer.num_buckets(), "label": model.backend_label}


def gpu_dp_accumulate_two_lanes(rank, world):
    """An accumulation window of 4 micro-batches per rank through Engine.accumulate (two streams, two gradient lanes,
    the all-reduce armed for the last micro-batch, whose backward merges the lanes segment by segment in front of the
    reducer): the reduced gradient equals the gradient of ONE process on the 32 samples of both ranks."""
    eng = Engine(fp16=True, seed=7, min_bucket_mb=0.05, init_scale=256.0)
    model = _model(seed=rank)
    opt = AdamW(params=model.parameters(), lr=1e-3)
    model, opt = eng.prepare(model, opt)
    inner = eng.unwrap_model(model)
    batch = {k: v.cuda() for k, v in _global_batch(32).items()}
    shard = {k: v[rank::world] for k, v in batch.items()}
    micro = [{k: v[4 * i:4 * i + 4] for k, v in shard.items()} for i in range(4)]
    model.train()
    eng.accumulate(model, micro)
    torch.cuda.synchronize()
    g_dp = inner.arena.grad.clone() / eng.scaler.get_scale()
    ref = _model(seed=0).cuda()
    from pytorch_distributed_training_amd.ops.arena import ParamArena

    arena = ParamArena(ref, eng.device, torch.float16)
    ref.enable_native(arena, seed=1)
    ref.train()
    ref(**batch).loss.backward()
    err = ((g_dp - arena.grad).norm() / arena.grad.norm()).item()
    opt.step(); opt.zero_grad()
    model.check_replicas()
    return {"err": err, "lanes": inner.arena.grad_alt is not None, "order": list(model.reducer.launch_order()),
            "nb": model.reducer.num_buckets()}


def gpu_dp_comm_fallback(rank, world):
    os.environ["PDT_COMM_FORCE_RCCL"] = "1"
    os.environ["PDT_COMM_SELFTEST_FAIL"] = "1"
    os.environ["PDT_COMM_SELFTEST_S"] = "20"
    try:
        out = gpu_dp_native(rank, world)
        from pytorch_distributed_training_amd.parallel import comm as C

        out["comm_is_none"] = len(C._COMMS) == 1 and all(c is None for c in C._COMMS.values())
    finally:
        for k in ("PDT_COMM_FORCE_RCCL", "PDT_COMM_SELFTEST_FAIL", "PDT_COMM_SELFTEST_S"):
            os.environ.pop(k, None)
    return out


def gpu_pipeline_recipe(rank, world, layout, debug=False, model="bert-tiny-nodrop", varlen=True, examples=64, lr=2e-3,
                        init_scale=256.0):
    from pytorch_distributed_training_amd.config import TrainConfig
    from pytorch_distributed_training_amd.recipes import run_model_parallel

    if debug:  # arena guard words + "optimizer step while a reduction is still armed" check
        import os

        os.environ["PDT_DEBUG"] = "1"
    cfg = TrainConfig(model=model, fp16=True, train_examples=examples, eval_examples=16, num_epochs=2,
                      batch_size=16, max_gpu_batch_size=4, lr=lr, num_warmup_steps=1, log_every=1, max_steps=6,
                      seed=5, mp_layout=layout, init_scale=init_scale, varlen=varlen)
    if layout == "towers":
        cfg.batch_size, cfg.max_gpu_batch_size = 8, 8
    return run_model_parallel(rank, world, cfg)


def gpu_dp_dropout(rank, world, steps=6, model="bert-tiny", examples=64, lr=2e-3):
    """Data-parallel native fp16 training WITH dropout; the logged loss is the mean over the ranks' shards."""
    from pytorch_distributed_training_amd.config import TrainConfig
    from pytorch_distributed_training_amd.models.config import preset
    from pytorch_distributed_training_amd.optim import get_linear_schedule_with_warmup
    from pytorch_distributed_training_amd.recipes import _collator, build_datasets
    from pytorch_distributed_training_amd.utils.seed import set_seed

    cfg = TrainConfig(model=model, fp16=True, train_examples=examples, eval_examples=16, num_epochs=2, batch_size=8, seed=5)
    eng = Engine(fp16=True, seed=5, init_scale=256.0, min_bucket_mb=0.05)
    mc = preset(cfg.model)
    train_ds, _ = build_datasets(cfg, mc)
    dl = DataLoader(train_ds, shuffle=True, collate_fn=_collator(cfg, mc), batch_size=8)
    set_seed(5)
    torch.manual_seed(5)
    net = build_model(mc)
    opt = AdamW(params=net.parameters(), lr=lr)
    model, opt, dl = eng.prepare(net, opt, dl)
    sched = get_linear_schedule_with_warmup(opt, 1, len(dl) * 2)
    losses = []
    model.train()
    for epoch in range(2):
        dl.set_epoch(epoch)
        for batch in dl:
            assert "sample_ids" in batch
            loss = model(**batch).loss
            eng.backward(loss)
            opt.step(); sched.step(); opt.zero_grad()
            losses.append(eng.reduce_mean(loss.detach()).item())
            if len(losses) >= steps:
                return losses
    return losses


def gpu_single_reference(layout, steps=6, replicas=1, model="bert-tiny-nodrop", examples=64, lr=2e-3, init_scale=256.0):
    from pytorch_distributed_training_amd.config import TrainConfig
    from pytorch_distributed_training_amd.models.config import preset
    from pytorch_distributed_training_amd.models.tribert import TriBert
    from pytorch_distributed_training_amd.optim import get_linear_schedule_with_warmup
    from pytorch_distributed_training_amd.recipes import _collator, build_datasets
    from pytorch_distributed_training_amd.utils.seed import set_seed

    bs = (8 if layout in ("towers", "dp") else 16) * replicas  # the global batch of `replicas` data-parallel replicas
    cfg = TrainConfig(model=model, fp16=True, train_examples=examples, eval_examples=16, num_epochs=2,
                      batch_size=bs, seed=5)
    eng = Engine(fp16=True, seed=5, init_scale=init_scale)
    mc = preset(cfg.model)
    train_ds, _ = build_datasets(cfg, mc)
    dl = DataLoader(train_ds, shuffle=True, collate_fn=_collator(cfg, mc), batch_size=bs)
    set_seed(5)
    torch.manual_seed(5)
    model = TriBert(build_model(mc), build_model(mc), build_model(mc)) if layout == "towers" else build_model(mc)
    opt = AdamW(params=model.parameters(), lr=lr)
    model, opt, dl = eng.prepare(model, opt, dl)
    sched = get_linear_schedule_with_warmup(opt, 1, len(dl) * 2)
    losses = []
    model.train()
    for epoch in range(2):
        dl.set_epoch(epoch)
        for batch in dl:
            loss = model(**batch).loss
            eng.backward(loss)
            opt.step(); sched.step(); opt.zero_grad()
            losses.append(loss.item())
            if len(losses) >= steps:
                return losses
    return losses


def dp_debug_mode(rank, world):
    """PDT_DEBUG arena guards + PDT_COMM_TIMING exposed-communication timer through a real DP step."""
    import os

    os.environ["PDT_DEBUG"] = "1"
    os.environ["PDT_COMM_TIMING"] = "1"
    from pytorch_distributed_training_amd.utils.debug import ArenaGuardError

    eng = Engine(cpu=True, min_bucket_mb=0.05)
    model = _model()
    opt = AdamW(model.parameters(), lr=1e-3)
    model, opt = eng.prepare(model, opt)
    arena = eng.unwrap_model(model).arena
    assert arena.guard_elems > 0 and model.measure_exposed
    batch = _global_batch()
    shard = {k: v[rank::world] for k, v in batch.items()}
    model.train()
    for _ in range(2):
        eng.backward(model(**shard).loss)
        opt.step()  # runs check_guards()
        opt.zero_grad()
    exposed = model.comm_exposed_ms()
    # armed but never finished: forward without backward, then step
    model(**shard)
    armed_error = False
    try:
        opt.step()
    except RuntimeError as e:
        armed_error = "still armed" in str(e)
    model.finish_backward()
    # a stray write below the gradient arena
    arena._backing["grad"][arena.guard_elems - 1] = 0.0
    tripped = False
    try:
        opt.step()
    except ArenaGuardError as e:
        tripped = "grad arena" in str(e)
    return {"exposed": exposed, "armed_error": armed_error, "tripped": tripped}


def dp_replica_mismatch(rank, world):
    """Rank 1 builds a 3-label head: the DataParallel constructor must refuse on every rank."""
    eng = Engine(cpu=True, min_bucket_mb=0.05)
    torch.manual_seed(0)
    model = build_model("bert-tiny", num_labels=2 if rank == 0 else 3)
    try:
        eng.prepare(model)
    except RuntimeError as e:
        return "replicas differ" in str(e) and "[1]" in str(e)
    return False


def cap_agreement(rank, world):
    """Ranks that read different amounts of free HBM must end up with the same (accumulation steps, micro-batch)."""
    import types

    from pytorch_distributed_training_amd.config import TrainConfig
    from pytorch_distributed_training_amd.launch import init_distributed
    from pytorch_distributed_training_amd.models.config import preset

    init_distributed(cpu=True)
    cfg = TrainConfig(batch_size=4096, max_length=128)
    fake_gpu = types.SimpleNamespace(type="cuda")  # (no index: the agreement runs over the gloo group's CPU tensors)
    free = (200 << 30) if rank == 0 else (20 << 30)  # rank 1 shares its card with somebody
    return cfg.accumulation(fake_gpu, preset("bert-large-cased"), free_bytes=free)
