"""CPU-side plumbing: config, samplers, collation, optimizer, schedule, metric, arena, checkpoints, HF parity."""
import math
import os

import pytest
import torch

import pytorch_distributed_training_amd as pdt  # noqa: F401
from pytorch_distributed_training_amd.config import TrainConfig, parse_config, str2bool
from pytorch_distributed_training_amd.data import PadCollator, ShardedBatchSampler, SyntheticPairDataset
from pytorch_distributed_training_amd.data.samplers import DistributedShardSampler
from pytorch_distributed_training_amd.models.bert import build_model
from pytorch_distributed_training_amd.ops.arena import ParamArena
from pytorch_distributed_training_amd.optim import FusedAdamW, HFAdamW, LinearWarmupSchedule, LossScaler


# ------------------------------------------------------------------------------------------- config
def test_bool_flags_parse_properly_and_keep_reference_spelling():
    assert parse_config(["--fp16=True"]).fp16 is True
    assert parse_config(["--fp16=False"]).fp16 is False  # the reference's type=bool would give True
    assert parse_config(["--fp16"]).fp16 is True
    assert parse_config([]).fp16 is False
    assert parse_config(["--cpu", "yes"]).cpu is True
    with pytest.raises(Exception):
        str2bool("maybe")


def test_reference_defaults_and_accumulation_policy():
    c = TrainConfig()
    assert (c.lr, c.num_epochs, c.correct_bias, c.seed, c.batch_size) == (2e-5, 3, True, 42, 96)
    assert c.accumulation() == (12, 8)
    assert TrainConfig(batch_size=8).accumulation() == (1, 8)
    assert TrainConfig(batch_size=20).accumulation() == (2, 8)
    # the micro-batch cap: an explicit value wins; unset, it is the reference's 8 on the CPU and with --compat
    import torch

    from pytorch_distributed_training_amd.models.config import preset

    large = preset("bert-large-cased")
    assert TrainConfig(max_gpu_batch_size=32).accumulation(torch.device("cpu"), large) == (3, 32)
    assert TrainConfig().micro_batch_cap(torch.device("cpu"), large) == 8
    assert TrainConfig(compat=True).micro_batch_cap(torch.device("cuda", 0), large) == 8


# ------------------------------------------------------------------------------------------- samplers
@pytest.mark.parametrize("n,bs,w", [(100, 8, 2), (3668, 8, 2), (408, 32, 2), (17, 4, 4), (64, 8, 8), (5, 4, 4), (33, 8, 3)])
def test_sharded_batch_sampler_matches_accelerate(n, bs, w):
    from accelerate.data_loader import BatchSamplerShard
    from torch.utils.data import BatchSampler, SequentialSampler

    for drop_last in (False, True):
        base = BatchSampler(SequentialSampler(range(n)), bs, drop_last)
        for r in range(w):
            ours = ShardedBatchSampler(base, w, r)
            ref = BatchSamplerShard(base, w, r)
            assert list(ours) == list(ref), (n, bs, w, r, drop_last)
            assert len(ours) == len(ref)


@pytest.mark.parametrize("n,w", [(408, 2), (3668, 2), (10, 4), (7, 8)])
def test_distributed_shard_sampler_matches_torch(n, w):
    from torch.utils.data.distributed import DistributedSampler

    data = list(range(n))
    for r in range(w):
        for shuffle in (True, False):
            ours = DistributedShardSampler(data, w, r, shuffle=shuffle, seed=3)
            ref = DistributedSampler(data, w, r, shuffle=shuffle, seed=3)
            for epoch in (0, 1, 5):
                ours.set_epoch(epoch); ref.set_epoch(epoch)
                assert list(ours) == list(ref)
            assert len(ours) == len(ref)


def test_sharded_epoch_covers_dataset_and_wraps_evenly():
    n, bs, w = 3668, 8, 2
    from torch.utils.data import BatchSampler, SequentialSampler

    base = BatchSampler(SequentialSampler(range(n)), bs, False)
    seen, lens = [], []
    for r in range(w):
        b = list(ShardedBatchSampler(base, w, r))
        lens.append(len(b))
        assert all(len(x) == bs for x in b)
        seen += [i for x in b for i in x]
    assert lens == [230, 230]  # 459 batches -> 230 per rank, the schedule length the reference relies on
    assert set(seen) == set(range(n))


def test_collator_pads_longest_and_fixed():
    ds = SyntheticPairDataset(8, vocab_size=2000, min_len=3, max_len=9, seed=0)
    ex = [ds[i] for i in range(8)]
    b = PadCollator()(ex)
    L = max(len(e["input_ids"]) for e in ex)
    assert b["input_ids"].shape == (8, L) and b["labels"].shape == (8,)
    for i, e in enumerate(ex):
        n = len(e["input_ids"])
        assert b["attention_mask"][i].sum() == n and (b["input_ids"][i, n:] == 0).all()
        assert b["input_ids"][i, 0] == 101 and b["token_type_ids"][i, n - 1] == 1
    b = PadCollator(padding="max_length", max_length=128)(ex)
    assert b["input_ids"].shape == (8, 128)
    b = PadCollator(pad_to_multiple_of=16)(ex)
    assert b["input_ids"].shape[1] % 16 == 0
    assert b.to("cpu") is b


# ------------------------------------------------------------------------------------------- optimizer
def _hf_formula(p, g, m, v, t, lr, b1, b2, eps, wd, correct):
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    ss = lr * (math.sqrt(1 - b2 ** t) / (1 - b1 ** t) if correct else 1.0)
    p = p - ss * m / (v.sqrt() + eps)
    if wd:
        p = p - lr * wd * p
    return p, m, v


@pytest.mark.parametrize("correct_bias", [True, False])
def test_hf_adamw_and_fused_cpu_follow_the_hf_update(correct_bias):
    torch.manual_seed(0)
    lin = torch.nn.Linear(7, 5)
    ref = [p.detach().clone().double() for p in lin.parameters()]
    ms = [torch.zeros_like(r) for r in ref]; vs = [torch.zeros_like(r) for r in ref]
    lin2 = torch.nn.Linear(7, 5); lin2.load_state_dict(lin.state_dict())
    opt = HFAdamW(lin.parameters(), lr=1e-2, weight_decay=0.1, correct_bias=correct_bias)
    arena = ParamArena(lin2, torch.device("cpu"))
    fused = FusedAdamW(arena, lr=1e-2, weight_decay=0.1, correct_bias=correct_bias)
    for t in range(1, 5):
        x = torch.randn(3, 7)
        for mod in (lin, lin2):
            mod.zero_grad() if mod is lin else arena.zero_grad()
            mod(x).pow(2).sum().backward()
        gs = [p.grad.detach().clone().double() for p in lin.parameters()]
        opt.step(); fused.step()
        for i in range(len(ref)):
            ref[i], ms[i], vs[i] = _hf_formula(ref[i], gs[i], ms[i], vs[i], t, 1e-2, 0.9, 0.999, 1e-6, 0.1, correct_bias)
        for p, p2, r in zip(lin.parameters(), lin2.parameters(), ref):
            torch.testing.assert_close(p.double(), r, rtol=1e-5, atol=1e-7)
            torch.testing.assert_close(p2.double(), r, rtol=1e-5, atol=1e-7)


def test_loss_scaler_skip_and_growth_on_cpu():
    lin = torch.nn.Linear(4, 4)
    arena = ParamArena(lin, torch.device("cpu"))
    sc = LossScaler("cpu", enabled=True, init_scale=8.0, growth_interval=2)
    opt = FusedAdamW(arena, lr=0.1, scaler=sc)
    before = arena.master.clone()
    arena.grad.fill_(float("inf"))
    opt.step()
    assert torch.equal(arena.master, before) and sc.get_scale() == 4.0 and sc.last_step_skipped()
    assert int(opt.step_count) == 0
    arena.grad.fill_(4.0)  # = 1.0 after un-scaling
    opt.step(); opt.step()
    assert sc.get_scale() == 8.0 and int(opt.step_count) == 2 and not sc.last_step_skipped()
    assert not torch.equal(arena.master, before)


def test_linear_schedule_matches_transformers():
    from transformers import get_linear_schedule_with_warmup as hf_sched

    p = torch.nn.Parameter(torch.zeros(1))
    o1 = torch.optim.SGD([p], lr=2e-5); o2 = HFAdamW([p], lr=2e-5)
    s1 = hf_sched(o1, 100, 690); s2 = LinearWarmupSchedule(o2, 100, 690)
    for _ in range(700):
        assert abs(s1.get_last_lr()[0] - s2.get_last_lr()[0]) < 1e-12
        assert abs(o1.param_groups[0]["lr"] - o2.param_groups[0]["lr"]) < 1e-12
        o1.step(); s1.step(); s2.step()


# ------------------------------------------------------------------------------------------- metric
def test_mrpc_metric_matches_sklearn():
    from sklearn.metrics import accuracy_score, f1_score

    from pytorch_distributed_training_amd.metrics import load_metric

    g = torch.Generator().manual_seed(0)
    m = load_metric("glue", "mrpc")
    P, R = [], []
    for _ in range(5):
        p, r = torch.randint(0, 2, (32,), generator=g), torch.randint(0, 2, (32,), generator=g)
        m.add_batch(predictions=p, references=r)
        P += p.tolist(); R += r.tolist()
    out = m.compute()
    assert abs(out["accuracy"] - accuracy_score(R, P)) < 1e-12 and abs(out["f1"] - f1_score(R, P)) < 1e-12
    m.add_batch(predictions=torch.tensor([1, 1, 0, 0]), references=torch.tensor([1, 0, 0, 1]), valid=2)
    assert m.compute()["accuracy"] == 0.5


# ------------------------------------------------------------------------------------------- arena
def test_arena_layout_views_and_buckets():
    model = build_model("bert-tiny")
    names = [n for n, _ in model.named_parameters()]
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    arena = ParamArena(model, torch.device("cpu"), min_bucket_elems=1 << 16)
    # parameters are views of the arena with unchanged values and HF names
    assert [n for n, _ in model.named_parameters()] == names
    for n, p in model.named_parameters():
        assert torch.equal(p, before[n])
        o, k = arena.offsets[n]
        assert p.data_ptr() == arena.master[o:].data_ptr() and p.grad.data_ptr() == arena.grad[o:].data_ptr()
    # reverse forward order: head first, embeddings last; Q/K/V adjacent
    assert arena.segments[0].name == "head" and arena.segments[-1].name == "embeddings"
    pre = "bert.encoder.layer.0.attention.self."
    assert arena.adjacent([pre + "query.weight", pre + "key.weight", pre + "value.weight"])
    assert arena.adjacent([pre + "query.bias", pre + "key.bias", pre + "value.bias"])
    assert arena.bucket_bounds[0] == 0 and arena.bucket_bounds[-1] == arena.numel
    assert sum(arena.bucket_segments) == len(arena.segments)
    assert all(b % 64 == 0 for b in arena.bucket_bounds)
    # autograd accumulates into the arena
    ids = torch.randint(0, 512, (2, 8))
    model(input_ids=ids, labels=torch.tensor([0, 1])).loss.backward()
    assert arena.grad.abs().sum() > 0
    arena.zero_grad()
    assert arena.grad.abs().sum() == 0


def test_arena_cuts_oversized_buckets_into_aligned_pieces():
    """Every bucket above the cap becomes several pieces (the last bucket of backward has the smaller cap): contiguous,
    covering the arena, boundaries aligned like parameters (64 elements), pieces of one bucket equal multiples of 512
    elements (divisible by any rank count up to 8) but for the last, none larger than its cap plus that unit, and every
    segment knows all the pieces that hold its gradients."""
    model = build_model("bert-tiny")
    whole = ParamArena(build_model("bert-tiny"), torch.device("cpu"), min_bucket_elems=1 << 14, max_bucket_elems=1 << 30,
                       tail_bucket_elems=1 << 30)
    arena = ParamArena(model, torch.device("cpu"), min_bucket_elems=1 << 14, max_bucket_elems=1 << 15, tail_bucket_elems=1 << 13)
    b = arena.bucket_bounds
    assert b[0] == 0 and b[-1] == arena.numel and all(x < y for x, y in zip(b, b[1:]))
    assert len(b) > len(whole.bucket_bounds) and set(whole.bucket_bounds) <= set(b)  # cuts only add boundaries
    assert all(x % 64 == 0 for x in b)
    sizes = [y - x for x, y in zip(b, b[1:])]
    assert max(sizes[:-1]) <= (1 << 15) + 512 and sizes[-1] <= (1 << 13) + 512
    assert len(arena.bucket_segments) == len(sizes)
    covered = set()
    for seg, ref in zip(arena.segments, whole.segments):
        assert seg.bucket == seg.buckets[0] and list(seg.buckets) == list(range(seg.buckets[0], seg.buckets[-1] + 1))
        lo, hi = b[seg.buckets[0]], b[seg.buckets[-1] + 1]
        assert (lo, hi) == (whole.bucket_bounds[ref.bucket], whole.bucket_bounds[ref.bucket + 1])  # its original bucket
        assert all(sizes[k] % 512 == 0 for k in seg.buckets[:-1])
        covered.update(seg.buckets)
    assert covered == set(range(len(sizes)))
    assert len(arena.segments[-1].buckets) >= 2  # the embedding tables: the tail cap applies


# ------------------------------------------------------------------------------------------- HF parity
def _hf_bert(cfg):
    from transformers import BertConfig, BertForSequenceClassification

    hc = BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                    num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                    max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size,
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, num_labels=cfg.num_labels)
    hc._attn_implementation = "eager"
    return BertForSequenceClassification(hc).eval()


def test_bert_matches_transformers_names_logits_and_grads():
    ours = build_model("bert-tiny", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0).eval()
    hf = _hf_bert(ours.config)
    assert set(dict(ours.named_parameters())) == set(dict(hf.named_parameters()))
    hf.load_state_dict(ours.state_dict(), strict=False)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(1, 512, (3, 20), generator=g)
    mask = torch.ones(3, 20, dtype=torch.long); mask[1, 12:] = 0; mask[2, 5:] = 0
    types = torch.zeros(3, 20, dtype=torch.long); types[:, 10:] = 1
    labels = torch.tensor([0, 1, 1])
    a = ours(input_ids=ids, attention_mask=mask, token_type_ids=types, labels=labels, output_hidden_states=True)
    b = hf(input_ids=ids, attention_mask=mask, token_type_ids=types, labels=labels, output_hidden_states=True)
    torch.testing.assert_close(a.logits, b.logits, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(a.loss, b.loss, rtol=1e-5, atol=1e-6)
    assert len(a.hidden_states) == len(b.hidden_states) == 3
    valid = mask.bool()
    torch.testing.assert_close(a.hidden_states[-1][valid], b.hidden_states[-1][valid], rtol=1e-4, atol=1e-5)
    a.loss.backward(); b.loss.backward()
    hfp = dict(hf.named_parameters())
    for n, p in ours.named_parameters():
        torch.testing.assert_close(p.grad, hfp[n].grad, rtol=1e-3, atol=1e-6, msg=n)


def test_roberta_matches_transformers():
    from transformers import RobertaConfig, RobertaForSequenceClassification

    ours = build_model("roberta-tiny", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0).eval()
    c = ours.config
    hc = RobertaConfig(vocab_size=c.vocab_size, hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers,
                       num_attention_heads=c.num_attention_heads, intermediate_size=c.intermediate_size,
                       max_position_embeddings=c.max_position_embeddings, type_vocab_size=c.type_vocab_size,
                       hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=c.layer_norm_eps,
                       pad_token_id=c.pad_token_id, num_labels=2)
    hc._attn_implementation = "eager"
    hf = RobertaForSequenceClassification(hc).eval()
    assert set(dict(ours.named_parameters())) == set(dict(hf.named_parameters()))
    hf.load_state_dict(ours.state_dict(), strict=False)
    ids = torch.randint(3, 512, (2, 16)); ids[1, 9:] = c.pad_token_id
    mask = ids.ne(c.pad_token_id).long()
    a = ours(input_ids=ids, attention_mask=mask, labels=torch.tensor([1, 0]))
    b = hf(input_ids=ids, attention_mask=mask, labels=torch.tensor([1, 0]))
    torch.testing.assert_close(a.logits, b.logits, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(a.loss, b.loss, rtol=1e-5, atol=1e-6)


def test_piecemeal_access_used_by_model_parallel_script():
    m = build_model("bert-tiny").eval()
    ids = torch.randint(1, 512, (2, 12)); mask = torch.ones(2, 12, dtype=torch.long); mask[1, 8:] = 0
    emb = m.get_input_embeddings()(ids)
    out = m(inputs_embeds=emb, attention_mask=mask, token_type_ids=torch.zeros_like(ids), output_hidden_states=True)
    ref = m(input_ids=ids, attention_mask=mask, token_type_ids=torch.zeros_like(ids), output_hidden_states=True)
    torch.testing.assert_close(out.hidden_states[-1], ref.hidden_states[-1])
    ext = m.get_extended_attention_mask(mask, ids.shape, ids.device)
    assert ext.shape == (2, 1, 1, 12) and ext[1, 0, 0, -1] == torch.finfo(torch.float32).min and ext[0].abs().sum() == 0
    seq = m.bert.encoder(hidden_states=ref.hidden_states[0], attention_mask=ext)[0]
    torch.testing.assert_close(seq, ref.hidden_states[-1])
    logits = m.classifier(m.dropout(m.bert.pooler(seq)))
    torch.testing.assert_close(logits, ref.logits)
    assert m.num_labels == 2


# ------------------------------------------------------------------------------------------- checkpoints
def test_checkpoint_roundtrip_and_loads_into_transformers(tmp_path):
    from transformers import BertForSequenceClassification

    from pytorch_distributed_training_amd import checkpoint as ck

    m = build_model("bert-tiny", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0).eval()
    d = str(tmp_path / "ckpt")
    ck.save_pretrained(m, d)
    assert os.path.exists(os.path.join(d, "config.json")) and os.path.exists(os.path.join(d, "model.safetensors"))
    m2 = ck.load_pretrained(d).eval()
    ids = torch.randint(1, 512, (2, 10))
    torch.testing.assert_close(m(input_ids=ids).logits, m2(input_ids=ids).logits)
    hf = BertForSequenceClassification.from_pretrained(d, attn_implementation="eager").eval()
    torch.testing.assert_close(m(input_ids=ids).logits, hf(input_ids=ids).logits, rtol=1e-4, atol=1e-5)
    # and the other direction: a directory written by transformers loads here
    d2 = str(tmp_path / "hf")
    hf.save_pretrained(d2)
    m3 = ck.load_pretrained(d2).eval()
    torch.testing.assert_close(m3(input_ids=ids).logits, hf(input_ids=ids).logits, rtol=1e-4, atol=1e-5)


def test_training_state_resume_is_exact(tmp_path):
    from pytorch_distributed_training_amd import checkpoint as ck
    from pytorch_distributed_training_amd.optim import get_linear_schedule_with_warmup

    def make():
        torch.manual_seed(0)
        m = build_model("bert-tiny", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        a = ParamArena(m, torch.device("cpu"))
        o = FusedAdamW(a, lr=1e-3)
        s = get_linear_schedule_with_warmup(o, 2, 20)
        return m, a, o, s

    def step(m, o, s, i):
        g = torch.Generator().manual_seed(i)
        ids = torch.randint(1, 512, (4, 10), generator=g)
        loss = m(input_ids=ids, labels=torch.randint(0, 2, (4,), generator=g)).loss
        loss.backward(); o.step(); s.step(); o.zero_grad()
        return loss.item()

    m, a, o, s = make()
    for i in range(3):
        step(m, o, s, i)
    ck.save_training_state(str(tmp_path), m, o, s, {"epoch": 1, "global_step": 3})
    cont = [step(m, o, s, i) for i in range(3, 6)]
    m2, a2, o2, s2 = make()
    extra = ck.load_training_state(str(tmp_path), m2, o2, s2)
    assert extra == {"epoch": 1, "global_step": 3} and ck.has_checkpoint(str(tmp_path))
    again = [step(m2, o2, s2, i) for i in range(3, 6)]
    assert cont == again
    assert torch.equal(a.master, a2.master)


def test_default_backend_choice(monkeypatch):
    """CPU runs use gloo; the GPU choice (nccl unless ranks outnumber GPUs) needs a device to be exercised."""
    import torch

    from pytorch_distributed_training_amd import launch

    env = launch.DistEnv(rank=1, local_rank=1, world_size=2)
    assert launch.default_backend(torch.device("cpu"), env) == "gloo"
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert launch.default_backend(torch.device("cuda", 0), env) == "gloo"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    assert launch.default_backend(torch.device("cuda", 0), env) == "nccl"


def test_arena_guards_and_padding_canaries():
    from pytorch_distributed_training_amd.utils.debug import ArenaGuardError

    torch.manual_seed(0)
    model = build_model("bert-tiny")
    plain = ParamArena(build_model("bert-tiny"), torch.device("cpu"))
    assert plain.guard_elems == 0 and not plain._backing
    arena = ParamArena(model, torch.device("cpu"), guard_elems=256)
    assert arena.master.numel() == plain.numel and arena.grad.data_ptr() % 16 == 0
    arena.check_guards()
    opt = FusedAdamW(arena, lr=1e-3, weight_decay=0.01)
    for p in model.parameters():
        p.grad.normal_()
    opt.step()
    arena.check_guards()  # optimizer and weight decay keep guards and padding intact
    pad = arena._padding_index()
    assert pad.numel() > 0  # the 2-element classifier bias is padded to the 64-element alignment
    arena.grad[pad[0]] = 1.0
    with pytest.raises(ArenaGuardError, match="alignment padding after 'classifier.bias'"):
        arena.check_guards()
    arena.grad[pad[0]] = 0.0
    arena._backing["master"][-1] = 3.0
    with pytest.raises(ArenaGuardError, match="master arena: 1 guard words above"):
        arena.check_guards()


def test_step_logger_evaluates_lazy_extra(tmp_path, capsys):
    import json

    from pytorch_distributed_training_amd.utils.logging import StepLogger

    path = tmp_path / "log.jsonl"
    lg = StepLogger(True, every=1, path=str(path), samples_per_step=8)
    lg.log_step(1, torch.tensor(0.5), 1e-3, extra=lambda: {"comm_exposed_ms": 1.25, "skipped": None})
    lg.close()
    rec = json.loads(path.read_text().strip())
    assert rec["comm_exposed_ms"] == 1.25 and "skipped" not in rec
    assert "comm-exposed 1.25 ms" in capsys.readouterr().out


def test_comm_backend_choice_and_flag(monkeypatch):
    """`--comm-backend rccl` / PDT_COMM select the framework's own RCCL communicator for the gradient path; CPU runs
    keep the process group (gloo) whatever the choice; the watchdog's communicator poll is a no-op without one."""
    from pytorch_distributed_training_amd.config import parse_config
    from pytorch_distributed_training_amd.parallel.comm import backend_choice, create_rccl_comm
    from pytorch_distributed_training_amd.utils.watchdog import poll_comm_errors

    monkeypatch.delenv("PDT_COMM", raising=False)
    assert backend_choice() == "rccl" and backend_choice("c10d") == "c10d"
    monkeypatch.setenv("PDT_COMM", "rccl")
    assert backend_choice() == "rccl" and backend_choice("c10d") == "c10d"
    with pytest.raises(ValueError):
        backend_choice("mpi")
    assert parse_config(["--comm-backend", "rccl"]).comm_backend == "rccl"
    assert parse_config([]).comm_backend is None
    with pytest.raises(ValueError):
        create_rccl_comm(device=torch.device("cpu"))
    assert poll_comm_errors() is None


def test_native_extension_exports_runtime_switches():
    """The in-tree extension carries the communicator class and the run-time switches the engine uses."""
    from pytorch_distributed_training_amd.ops import native

    C = native()
    assert hasattr(C, "RcclComm") and hasattr(C, "GradBucketReducer")
    before = C.gemm_persistent_mode()
    C.set_gemm_persistent(0)
    assert C.gemm_persistent_mode() == 0
    C.set_gemm_persistent(before)
    assert C.gemm_persistent_mode() == before


def test_clip_grad_norm_on_the_flat_gradient_arena():
    """Engine.clip_grad_norm_ scales the arena's gradient to the requested global norm (one norm, one multiply), returns
    the norm before clipping, leaves gradients below the bound untouched, and the recipe flag reaches it."""
    from pytorch_distributed_training_amd.data.synthetic import fixed_batch
    from pytorch_distributed_training_amd.engine import Engine
    from pytorch_distributed_training_amd.optim import AdamW

    torch.manual_seed(0)
    eng = Engine(cpu=True)
    model = build_model("bert-tiny")
    opt = AdamW(params=model.parameters(), lr=1e-3)
    model, opt = eng.prepare(model, opt)
    model.train()
    batch = fixed_batch(4, 16, model.config.vocab_size, eng.device, seed=1)
    eng.backward(model(**batch).loss)
    arena = eng.unwrap_model(model).arena
    before = arena.grad.clone()
    total = before.norm().item()
    assert total > 0
    got = eng.clip_grad_norm_(model, 10.0 * total)  # far above: nothing changes
    assert abs(got.item() - total) < 1e-4 * total and torch.equal(arena.grad, before)
    got = eng.clip_grad_norm_(model, 0.5 * total)
    assert abs(got.item() - total) < 1e-4 * total
    assert abs(arena.grad.norm().item() - 0.5 * total) < 1e-4 * total
    assert torch.allclose(arena.grad, before * 0.5, rtol=1e-4, atol=1e-8)
    assert parse_config(["--max-grad-norm", "1.0"]).max_grad_norm == 1.0 and TrainConfig().max_grad_norm == 0.0


def test_in_memory_round_trip_with_a_transformers_model():
    """checkpoint.from_transformers / to_transformers: a transformers model becomes this framework's model (same
    configuration, same weights, same logits) and back, without going through files."""
    from pytorch_distributed_training_amd import checkpoint as ckpt

    hf = _hf_bert(build_model("bert-tiny").config)
    ours = ckpt.from_transformers(hf).eval()
    assert ours.config.hidden_size == hf.config.hidden_size and ours.config.num_hidden_layers == hf.config.num_hidden_layers
    assert ours.config.hidden_dropout_prob == 0.0
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1, 512, (2, 12), generator=g)
    mask = torch.ones(2, 12, dtype=torch.long); mask[1, 7:] = 0
    with torch.no_grad():
        a = ours(input_ids=ids, attention_mask=mask).logits
        b = hf(input_ids=ids, attention_mask=mask).logits
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    back = ckpt.to_transformers(ours).eval()
    with torch.no_grad():
        c = back(input_ids=ids, attention_mask=mask).logits
    torch.testing.assert_close(c, b, rtol=1e-4, atol=1e-5)


def test_checkpoint_tags_order_numerically(tmp_path):
    """Only the step field of a tag is zero-padded: epoch 10 must sort after epoch 9 (and be the one that is kept)."""
    from pytorch_distributed_training_amd import checkpoint as ck

    assert sorted(["00000040-10-0", "00000040-9-57", "00000039-9-56"], key=ck._tag_key) == \
        ["00000039-9-56", "00000040-9-57", "00000040-10-0"]
    torch.manual_seed(0)
    model = build_model("bert-tiny")
    for epoch, off in ((9, 57), (10, 0)):
        ck.save_training_state(str(tmp_path), model, extra={"global_step": 40, "epoch": epoch, "micro_offset": off}, keep=1)
    assert ck.available_tags(str(tmp_path)) == ["00000040-10-0"]  # the older one was pruned, not the newer
    assert ck.common_tag(str(tmp_path), "") == "00000040-10-0"


def test_graphed_step_refuses_the_overlapped_window_without_the_opt_in(monkeypatch):
    """The multi-stream accumulation window crashed inside hipStreamEndCapture when captured (utils/graph.py has the
    findings): GraphedStep(window="overlap") says so instead of crashing unless PDT_GRAPH_OVERLAP_WINDOW=1 is set, and
    rejects unknown window names and a custom loss for the overlapped window."""
    import types

    from pytorch_distributed_training_amd.utils.graph import OVERLAP_REFUSAL, GraphedStep

    eng = types.SimpleNamespace(num_processes=1, device=torch.device("cpu"))
    monkeypatch.delenv("PDT_GRAPH_OVERLAP_WINDOW", raising=False)
    with pytest.raises(RuntimeError) as e:
        GraphedStep(eng, None, None, window="overlap")
    assert str(e.value) == OVERLAP_REFUSAL and "hipStreamEndCapture" in OVERLAP_REFUSAL
    with pytest.raises(ValueError):
        GraphedStep(eng, None, None, window="both")
    with pytest.raises(ValueError):
        GraphedStep(eng, None, None, window="overlap", loss_fn=lambda out, mb: out.loss)
    monkeypatch.setenv("PDT_GRAPH_OVERLAP_WINDOW", "1")
    with pytest.raises(RuntimeError, match="needs a GPU"):  # past the refusal; this fake engine has no GPU
        GraphedStep(eng, None, None, window="overlap")
