"""Native (gfx950 kernels) model path against the fp32 PyTorch path of the same module tree."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(name, dtype, **over):
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.ops.arena import ParamArena

    torch.manual_seed(0)
    ref = build_model(name, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **over).cuda()
    # larger-than-init weights make the comparison meaningful
    with torch.no_grad():
        for p in ref.parameters():
            if p.dim() > 1:
                p.mul_(3.0)
            else:
                p.add_(0.05 * torch.randn_like(p))
    nat = copy.deepcopy(ref)
    arena = ParamArena(nat, torch.device("cuda"), dtype)
    nat.enable_native(arena, seed=1)
    return ref, nat, arena


def _batch(cfg, B=4, S=48, roberta=False):
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(5, cfg.vocab_size, (B, S), generator=g)
    lens = torch.tensor([S, S - 7, S // 2, S - 1])[:B]
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    ids = ids * mask + cfg.pad_token_id * (1 - mask)
    types = (torch.arange(S)[None] >= (lens[:, None] // 2)).long() * mask
    if roberta:
        types = torch.zeros_like(types)
    labels = torch.randint(0, cfg.num_labels, (B,), generator=g)
    return {k: v.cuda() for k, v in dict(input_ids=ids, attention_mask=mask, token_type_ids=types, labels=labels).items()}


@pytest.mark.parametrize("name", ["bert-tiny", "roberta-tiny"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_native_matches_torch_path(name, dtype):
    ref, nat, arena = _pair(name, dtype)
    batch = _batch(ref.config, roberta=name.startswith("roberta"))
    ref.train(); nat.train()
    o_ref = ref(**batch)
    o_nat = nat(**batch)
    tol = 3e-2 if dtype == torch.bfloat16 else 5e-3
    torch.testing.assert_close(o_nat.logits, o_ref.logits, rtol=tol, atol=tol)
    torch.testing.assert_close(o_nat.loss, o_ref.loss, rtol=tol, atol=tol)
    o_ref.loss.backward()
    arena.zero_grad()
    o_nat.loss.backward()
    named = dict(nat.named_parameters())
    worst = 0.0
    for n, p in ref.named_parameters():
        g_ref, g_nat = p.grad.float(), named[n].grad.float()
        denom = g_ref.norm().item()
        if denom < 1e-8:
            assert g_nat.norm().item() < 1e-4, n
            continue
        err = (g_nat - g_ref).norm().item() / denom
        worst = max(worst, err)
        assert err < (8e-2 if dtype == torch.bfloat16 else 2e-2), (n, err)
    assert worst > 0.0


def test_gradient_accumulation_overwrite_then_add():
    ref, nat, arena = _pair("bert-tiny", torch.float16)
    batch = _batch(ref.config)
    nat.train()
    arena.zero_grad()
    nat(**batch).loss.backward()
    g1 = arena.grad.clone()
    nat(**batch).loss.backward()  # second micro-batch of the same window: adds
    torch.testing.assert_close(arena.grad, 2 * g1, rtol=1e-3, atol=1e-6)
    arena.zero_grad()
    nat(**batch).loss.backward()  # new window: overwrites without a memset of the layer slices
    torch.testing.assert_close(arena.grad, g1, rtol=1e-3, atol=1e-6)


def test_dropout_is_active_reproducible_and_backward_consistent():
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.ops.arena import ParamArena

    torch.manual_seed(0)
    m = build_model("bert-tiny").cuda()
    arena = ParamArena(m, torch.device("cuda"), torch.float16)
    rt = m.enable_native(arena, seed=7)
    batch = _batch(m.config)
    m.train()
    rt.micro = 10
    a = m(**batch).logits.clone()
    rt.micro = 10
    b = m(**batch).logits.clone()
    c = m(**batch).logits.clone()  # next micro-batch index: different masks
    assert torch.equal(a, b) and not torch.equal(a, c)
    m.eval()
    d = m(**batch).logits
    e = m(**batch).logits
    assert torch.equal(d, e)
    # finite-difference check of the loss along the gradient direction with a frozen mask
    m.train()
    rt.micro = 20
    arena.zero_grad()
    loss0 = m(**batch).loss
    loss0.backward()
    g = arena.grad.clone()
    eps = 1e-2 / (g.norm().item() + 1e-12)
    with torch.no_grad():
        arena.master.add_(g, alpha=eps)
    arena.sync_compute_copy()
    rt.micro = 20
    loss1 = m(**batch).loss
    predicted = eps * g.norm().item() ** 2
    assert abs((loss1 - loss0).item() - predicted) < 0.35 * abs(predicted) + 2e-3


def test_engine_training_reduces_loss_fp16_and_bf16():
    from torch.utils.data import DataLoader

    from pytorch_distributed_training_amd.data import PadCollator, SyntheticPairDataset
    from pytorch_distributed_training_amd.engine import Engine
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.optim import AdamW

    for kw in (dict(fp16=True), dict(bf16=True)):
        eng = Engine(seed=3, **kw)
        torch.manual_seed(0)
        model = build_model("bert-tiny")
        opt = AdamW(params=model.parameters(), lr=5e-4)
        ds = SyntheticPairDataset(256, vocab_size=512, min_len=4, max_len=20, seed=1)
        dl = DataLoader(ds, shuffle=True, collate_fn=PadCollator(), batch_size=32)
        model, opt, dl = eng.prepare(model, opt, dl)
        model.train()
        first, last = None, None
        for epoch in range(12):
            tot, n = 0.0, 0
            for batch in dl:
                out = model(**batch)
                eng.backward(out.loss)
                opt.step()
                opt.zero_grad()
                tot += out.loss.item(); n += 1
            if first is None:
                first = tot / n
            last = tot / n
        assert last < first - 0.05, (kw, first, last)


def test_debug_mode_guards_hold_and_serialized_mode_agrees(monkeypatch):
    """PDT_DEBUG: no native kernel (wgrad, reductions, optimizer, cast) writes outside its arena slice or into
    alignment padding.  Serialized mode (device drained after every launch) reproduces the normal run with
    dropout off up to the summation order of the fp32 atomics (bias-gradient column sums, embedding scatter) -
    a larger difference would mean the normal mode depends on a race between streams."""
    from pytorch_distributed_training_amd.data.synthetic import fixed_batch
    from pytorch_distributed_training_amd.engine import Engine
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.ops import native
    from pytorch_distributed_training_amd.optim import AdamW
    from pytorch_distributed_training_amd.utils.debug import set_serialized

    monkeypatch.setenv("PDT_DEBUG", "1")

    def run(serialized):
        set_serialized(serialized)
        try:
            eng = Engine(fp16=True, seed=5)
            torch.manual_seed(0)
            model = build_model("bert-tiny", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
            opt = AdamW(params=model.parameters(), lr=1e-3)
            model, opt = eng.prepare(model, opt)
            assert model.arena.guard_elems > 0
            batch = fixed_batch(8, 64, model.config.vocab_size, device="cuda", seed=2)
            model.train()
            losses = []
            for _ in range(4):
                out = model(**batch)
                eng.backward(out.loss)
                opt.step()  # check_guards() inside
                opt.zero_grad()
                losses.append(out.loss.detach().clone())
            return torch.stack(losses), model.arena.master.clone()
        finally:
            set_serialized(False)

    n0 = native().debug_sync_count()
    l_ser, w_ser = run(True)
    assert native().debug_sync_count() > n0 + 40
    n1 = native().debug_sync_count()
    l_par, w_par = run(False)
    assert native().debug_sync_count() == n1
    torch.testing.assert_close(l_ser, l_par, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(w_ser, w_par, rtol=0, atol=2e-4)  # 4 steps of lr 1e-3: a wrong gradient sign moves 2e-3


def test_training_steps_do_not_retain_activations():
    """No reference cycle through the autograd nodes: with the cycle collector off, device memory after every
    optimizer step returns to the same level (a cycle would pin one [tokens, hidden] tensor per layer per step)."""
    import gc

    from pytorch_distributed_training_amd.data.synthetic import fixed_batch
    from pytorch_distributed_training_amd.engine import Engine
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.optim import AdamW

    eng = Engine(fp16=True, seed=1)
    model = build_model("bert-tiny")
    opt = AdamW(params=model.parameters(), lr=1e-4)
    model, opt = eng.prepare(model, opt)
    batch = fixed_batch(64, 128, model.config.vocab_size, device="cuda", seed=2)
    model.train()
    gc.collect()
    gc.disable()
    try:
        levels = []
        for _ in range(5):
            out = model(**batch)
            eng.backward(out.loss)
            opt.step()
            opt.zero_grad()
            del out
            torch.cuda.synchronize()
            levels.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    assert max(levels[1:]) - levels[1] < 64 * 128 * model.config.hidden_size, levels


def test_graphed_step_matches_eager_and_redraws_dropout():
    """A step replayed from a captured hipGraph gives the eager loss curve (dropout off), and with dropout on
    every replay draws new masks although the launch arguments are frozen."""
    from pytorch_distributed_training_amd.data.synthetic import fixed_batch
    from pytorch_distributed_training_amd.engine import Engine
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.optim import AdamW
    from pytorch_distributed_training_amd.utils.graph import GraphedStep

    def run(graphed: bool, model_name: str, lr: float, steps: int):
        torch.manual_seed(3)
        eng = Engine(fp16=True, seed=11, init_scale=1024.0)
        model = build_model(model_name)
        opt = AdamW(params=model.parameters(), lr=lr)
        model, opt = eng.prepare(model, opt)
        model.train()
        vocab = eng.unwrap_model(model).config.vocab_size
        batches = [fixed_batch(8, 64, vocab, eng.device, seed=50 + i) for i in range(2)]
        losses = []
        if graphed:
            g = GraphedStep(eng, model, opt, warmup=2)
            for i in range(steps):
                losses.append(float(g(batches[i % 2]).item()))
            assert g.graph is not None
        else:
            for i in range(steps):
                out = model(**batches[i % 2])
                eng.backward(out.loss)
                opt.step()
                opt.zero_grad()
                losses.append(float(out.loss.item()))
        return losses

    eager = run(False, "bert-tiny-nodrop", 1e-3, 7)
    graphed = run(True, "bert-tiny-nodrop", 1e-3, 7)
    assert max(abs(a - b) for a, b in zip(eager, graphed)) < 2e-3, (eager, graphed)
    assert graphed[0] != graphed[-1]  # the captured optimizer really updates the weights
    same_batch = run(True, "bert-tiny", 0.0, 8)  # lr 0, two alternating batches: only the masks change
    replays = same_batch[3::2]  # replays of batch 1 (calls 4, 6, 8)
    assert len(set(replays)) == len(replays), same_batch


def test_bench_contract_json_line():
    """bench.py prints exactly one JSON line with the fields the benchmark driver reads."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1", "--model", "bert-tiny",
                        "--batch", "8", "--seq-len", "64"], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["unit"] == "samples/s" and d["value"] > 0
    assert abs(d["value"] - 8 * 3 / (d["ms_per_step"] * 3 / 1000)) / d["value"] < 1e-2
    for key in ("model", "global_batch", "seq_len", "parallelism"):
        assert key in d["config"], key
    assert d["model_tflops_per_gpu"] >= 0


def test_data_parallel_recipe_end_to_end_single_gpu(tmp_path):
    """The DP recipe on the GPU the way train_dp.py runs it: dynamically padded batches through the prefetching
    DeviceLoader, native fp16 training without a host sync per step, evaluation, checkpoint at the end of every
    epoch, and a second invocation that resumes from it (nothing left to train)."""
    from pytorch_distributed_training_amd.config import TrainConfig
    from pytorch_distributed_training_amd.recipes import run_data_parallel

    cfg = TrainConfig(model="bert-tiny", fp16=True, train_examples=256, eval_examples=64, num_epochs=2, batch_size=32,
                      max_gpu_batch_size=32, lr=1e-3, num_warmup_steps=2, log_every=1000, seed=11, init_scale=1024.0,
                      output_dir=str(tmp_path))
    out = run_data_parallel(cfg)
    assert out["optimizer_steps"] == 16
    assert 0.0 <= out["accuracy"] <= 1.0 and 0.0 <= out["f1"] <= 1.0
    again = run_data_parallel(cfg)
    assert again["optimizer_steps"] == 16  # resumed at the end of the last epoch: no further steps


def test_segment_outside_the_graph_contributes_zero_gradient():
    """zero_grad() does not memset slices whose native producer overwrites them; a layer that takes no part in a
    step (here: the forward stops after layer 0) must still count as zero gradient, not as last step's values."""
    from pytorch_distributed_training_amd.optim import FusedAdamW

    ref, nat, arena = _pair("bert-tiny", torch.float16)
    opt = FusedAdamW(arena, lr=1e-2)
    batch = _batch(ref.config)
    nat.train()
    nat(**batch).loss.backward()          # every segment written
    opt.step(); opt.zero_grad()
    last = nat.bert.encoder.layers()[-1]
    seg = arena.segments[last._native.segment]
    m_before = opt.m[seg.start:seg.end].clone()
    assert arena.grad[seg.start:seg.end].abs().sum() > 0   # stale values are still there (no memset)
    # a step whose graph skips the last layer: embeddings -> layer 0 -> head
    nat._begin_forward()
    x = nat.bert.embeddings(batch["input_ids"], batch["token_type_ids"])
    from pytorch_distributed_training_amd.models.bert import MaskInfo

    x = nat.bert.encoder.layers()[0](x, MaskInfo(batch["attention_mask"], *batch["input_ids"].shape, x.device))
    nat.head(x, batch["labels"]).loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert arena.grad[seg.start:seg.end].abs().sum() == 0
    # the optimizer saw a zero gradient there: its first moment only decayed
    torch.testing.assert_close(opt.m[seg.start:seg.end], 0.9 * m_before, rtol=1e-6, atol=0)


def test_sample_keyed_dropout_is_invariant_to_batch_split_and_padding():
    """With sample ids the dropout masks (embeddings, attention probabilities, hidden states, head) belong to the
    SAMPLE: a batch of 8, its two halves, and the same samples padded to a longer length give the same per-sample
    logits with dropout 0.3 on - which is what lets data-/model-parallel runs reproduce a single-GPU run."""
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.ops.arena import ParamArena

    torch.manual_seed(0)
    m = build_model("bert-tiny", hidden_dropout_prob=0.3, attention_probs_dropout_prob=0.3).cuda()
    arena = ParamArena(m, torch.device("cuda"), torch.float16)
    m.enable_native(arena, seed=7, global_seed=99)
    g = torch.Generator().manual_seed(1)
    B, S = 8, 40
    ids = torch.randint(5, 512, (B, S), generator=g)
    lens = torch.tensor([40, 33, 21, 40, 9, 17, 38, 25])
    mask = (torch.arange(S)[None] < lens[:, None]).long()
    batch = {k: v.cuda() for k, v in dict(input_ids=ids * mask, attention_mask=mask, token_type_ids=torch.zeros_like(ids),
                                           labels=torch.randint(0, 2, (B,), generator=g)).items()}
    sids = torch.arange(100, 100 + B, dtype=torch.int32).cuda()
    m.train()
    full = m(**batch, sample_ids=sids).logits.clone()
    again = m(**batch, sample_ids=sids).logits.clone()
    assert torch.equal(full, again)                      # same window, same samples: same masks
    halves = torch.cat([m(**{k: v[i:i + 4] for k, v in batch.items()}, sample_ids=sids[i:i + 4]).logits for i in (0, 4)])
    torch.testing.assert_close(halves, full, rtol=2e-2, atol=2e-2)
    # padded to 64 tokens instead of 40
    pad = lambda t: torch.nn.functional.pad(t, (0, 24))
    wide = {k: (pad(v) if v.dim() == 2 else v) for k, v in batch.items()}
    torch.testing.assert_close(m(**wide, sample_ids=sids).logits, full, rtol=2e-2, atol=2e-2)
    # other ids, or the next optimizer window: other masks
    other = m(**batch, sample_ids=sids + 1000).logits
    arena.zero_grad()
    later = m(**batch, sample_ids=sids).logits
    assert (other - full).abs().max() > 5e-2 and (later - full).abs().max() > 5e-2
    # backward replays the masks of its own forward although another forward ran in between
    arena.zero_grad()
    out1 = m(**batch, sample_ids=sids)
    _ = m(**{k: v[:4] for k, v in batch.items()}, sample_ids=sids[:4] + 7)
    out1.loss.backward()
    g1 = arena.grad.clone()
    arena.window -= 1
    arena.zero_grad()
    m(**batch, sample_ids=sids).loss.backward()
    assert torch.equal(arena.grad, g1)  # every reduction in backward is deterministic


def test_overlapped_accumulation_window_equals_the_plain_loop(monkeypatch):
    """Engine.accumulate software-pipelines a window over two streams (forward of micro-batch i+1 next to backward of
    micro-batch i).  Same micro-batches, same sample-keyed dropout masks: with one gradient lane the accumulated
    gradients and the losses are bit-identical to the sequential loop, and so are the weights after three optimizer
    steps."""
    monkeypatch.setenv("PDT_DUAL_BACKWARD", "0")
    from pytorch_distributed_training_amd.data.synthetic import fixed_batch
    from pytorch_distributed_training_amd.engine import Engine
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.optim import AdamW

    def run(overlap):
        torch.manual_seed(3)
        eng = Engine(fp16=True, seed=11, init_scale=1024.0)
        model = build_model("bert-tiny")
        opt = AdamW(params=model.parameters(), lr=1e-3)
        model, opt = eng.prepare(model, opt)
        model.train()
        batches = [fixed_batch(4, 32, 512, eng.device, seed=70 + i, first_id=4 * i) for i in range(5)]
        losses, grads = [], None
        for step in range(3):
            ls = eng.accumulate(model, batches, overlap=overlap)
            torch.cuda.synchronize()
            if step == 0:
                grads = model.arena.grad.clone()
            losses += [float(x) for x in ls]
            opt.step(); opt.zero_grad()
        torch.cuda.synchronize()
        return losses, grads, model.arena.master.clone()

    l0, g0, m0 = run(False)
    l1, g1, m1 = run(True)
    assert l0 == l1 and g0.abs().sum() > 0
    assert torch.equal(g0, g1) and torch.equal(m0, m1)


@pytest.mark.parametrize("n_micro", [5, 4])
def test_two_gradient_lanes_accumulate_the_same_window(monkeypatch, n_micro):
    """PDT_DUAL_BACKWARD=1: the micro-batches of a window alternate between two gradient buffers whose backward passes
    run concurrently; the last backward merges the second lane into the first segment by segment.  Same terms, another
    (fixed) order of additions: losses identical, gradients and weights equal to the one-lane window within rounding,
    and bit-identical between two runs."""
    from pytorch_distributed_training_amd.data.synthetic import fixed_batch
    from pytorch_distributed_training_amd.engine import Engine
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.optim import AdamW

    def run(dual):
        monkeypatch.setenv("PDT_DUAL_BACKWARD", "1" if dual else "0")
        torch.manual_seed(3)
        eng = Engine(fp16=True, seed=11, init_scale=1024.0)
        model = build_model("bert-tiny")
        opt = AdamW(params=model.parameters(), lr=1e-3)
        model, opt = eng.prepare(model, opt)
        model.train()
        batches = [fixed_batch(4, 32, 512, eng.device, seed=70 + i, first_id=4 * i) for i in range(n_micro)]
        losses, grads = [], None
        for step in range(3):
            ls = eng.accumulate(model, batches, overlap=True)
            torch.cuda.synchronize()
            if step == 0:
                grads = model.arena.grad.clone()
            losses += [float(x) for x in ls]
            opt.step(); opt.zero_grad()
        torch.cuda.synchronize()
        return losses, grads, model.arena.master.clone(), model.arena.grad_alt is not None

    l0, g0, m0, alt0 = run(False)
    l1, g1, m1, alt1 = run(True)
    l2, g2, m2, _ = run(True)
    assert not alt0 and alt1
    assert l0[:n_micro] == l1[:n_micro] and g0.abs().sum() > 0
    assert ((g1 - g0).norm() / g0.norm()).item() < 1e-5
    assert ((m1 - m0).norm() / m0.norm()).item() < 1e-5
    assert l1 == l2 and torch.equal(g1, g2) and torch.equal(m1, m2)


@pytest.mark.parametrize("name", ["bert-tiny", "roberta-tiny"])
def test_unpadded_execution_matches_the_padded_batch(name):
    """`cu_seqlens` / `token_index` (PadCollator(varlen=True)): the encoder runs on the real tokens only.  Logits,
    loss and every parameter gradient equal the padded run - with dropout ON, because the masks are keyed by
    (sample, position) in both layouts."""
    from pytorch_distributed_training_amd.data import PadCollator
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.ops.arena import ParamArena

    torch.manual_seed(0)
    m = build_model(name).cuda()
    arena = ParamArena(m, torch.device("cuda"), torch.float16)
    m.enable_native(arena, seed=7, global_seed=5)
    g = torch.Generator().manual_seed(2)
    lens = [37, 12, 64, 5, 29, 48]
    pad = m.config.pad_token_id
    examples = [{"input_ids": torch.randint(5, 500, (n,), generator=g).tolist(), "attention_mask": [1] * n,
                 "token_type_ids": ([0] * (n // 2) + [1] * (n - n // 2)) if name == "bert-tiny" else [0] * n,
                 "labels": int(i % 2), "sample_ids": 40 + i} for i, n in enumerate(lens)]
    batch = PadCollator(pad, varlen=True)(examples).to("cuda")
    assert batch["cu_seqlens"].tolist() == [0, 37, 49, 113, 118, 147, 195] and batch["token_index"].numel() == 195
    padded = {k: v for k, v in batch.items() if k not in ("cu_seqlens", "token_index")}
    m.train()
    arena.zero_grad()
    out_p = m(**padded)
    out_p.loss.backward()
    g_p = arena.grad.clone()
    arena.window -= 1
    arena.zero_grad()
    out_u = m(**batch)
    assert out_u.last_hidden_state.shape == (195, m.config.hidden_size)
    out_u.loss.backward()
    torch.testing.assert_close(out_u.logits, out_p.logits, rtol=5e-3, atol=5e-3)
    torch.testing.assert_close(out_u.loss, out_p.loss, rtol=2e-3, atol=2e-3)
    rel = ((arena.grad - g_p).norm() / g_p.norm()).item()
    assert rel < 2e-2, rel
    # rows of the packed output are the real-token rows of the padded output
    tok = batch["token_index"]
    torch.testing.assert_close(out_u.last_hidden_state, out_p.last_hidden_state.reshape(-1, m.config.hidden_size)[tok],
                               rtol=2e-2, atol=2e-2)
    m.eval()
    with torch.no_grad():
        torch.testing.assert_close(m(**batch).logits, m(**padded).logits, rtol=5e-3, atol=5e-3)


def _ragged_batch(pad, lens=(37, 12, 64, 5, 29, 48), seed=2):
    from pytorch_distributed_training_amd.data import PadCollator

    g = torch.Generator().manual_seed(seed)
    examples = [{"input_ids": torch.randint(5, 500, (n,), generator=g).tolist(), "attention_mask": [1] * n,
                 "token_type_ids": [0] * (n // 2) + [1] * (n - n // 2), "labels": int(i % 2), "sample_ids": 40 + i}
                for i, n in enumerate(lens)]
    return PadCollator(pad, varlen=True)(examples).to("cuda")


def test_unpadded_execution_of_a_two_stage_split():
    """The packed [T, H] block a first stage produces, fed to a second stage together with the packing entries of
    the batch, gives the loss and the parameter gradients of the whole model on that batch (dropout on)."""
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.models.config import preset
    from pytorch_distributed_training_amd.ops.arena import ParamArena
    from pytorch_distributed_training_amd.parallel.pp import build_stage

    cfg = preset("bert-tiny")
    torch.manual_seed(0)
    full = build_model(cfg).cuda()
    state = {k: v.detach().cpu().clone() for k, v in full.state_dict().items()}
    arena = ParamArena(full, torch.device("cuda"), torch.float16)
    full.enable_native(arena, seed=7, global_seed=5)
    stages, arenas = [], []
    for s in range(2):
        part = build_stage(cfg, s, 2, state).cuda()
        a = ParamArena(part, torch.device("cuda"), torch.float16)
        part.enable_native(a, seed=7, global_seed=5)
        stages.append(part.train())
        arenas.append(a)
    batch = _ragged_batch(cfg.pad_token_id)
    T = int(batch["cu_seqlens"][-1])
    full.train()
    arena.zero_grad()
    ref = full(**batch)
    ref.loss.backward()
    for a in arenas:
        a.zero_grad()
    packing = {k: batch[k] for k in ("cu_seqlens", "token_index", "sample_ids")}
    h = stages[0](input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                  token_type_ids=batch["token_type_ids"], **packing).last_hidden_state
    assert h.shape == (T, cfg.hidden_size)
    act = h.detach().clone().requires_grad_()
    out = stages[1](hidden_states=act, attention_mask=batch["attention_mask"], labels=batch["labels"], **packing)
    out.loss.backward()
    assert act.grad.shape == (T, cfg.hidden_size)
    h.backward(act.grad)
    torch.testing.assert_close(out.loss, ref.loss, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(out.logits, ref.logits, rtol=5e-3, atol=5e-3)
    for a in arenas:
        for name, (off, n) in a.offsets.items():
            o = arena.offsets[name][0]
            want, got = arena.grad[o:o + n], a.grad[off:off + n]
            assert ((got - want).norm() / (want.norm() + 1e-6)).item() < 2e-2, name


def test_unpadded_execution_of_the_tower_ensemble():
    """TriBert on the real tokens only (one packed word-embedding lookup shared by the three towers) equals the
    padded run: loss, logits, gradients - dropout on."""
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.models.tribert import TriBert
    from pytorch_distributed_training_amd.ops.arena import ParamArena

    torch.manual_seed(0)
    m = TriBert(build_model("bert-tiny"), build_model("bert-tiny"), build_model("bert-tiny")).cuda()
    arena = ParamArena(m, torch.device("cuda"), torch.float16)
    m.enable_native(arena, seed=7, global_seed=5)
    batch = _ragged_batch(m.config.pad_token_id)
    padded = {k: v for k, v in batch.items() if k not in ("cu_seqlens", "token_index", "lengths")}
    m.train()
    arena.zero_grad()
    out_p = m(**padded)
    out_p.loss.backward()
    g_p = arena.grad.clone()
    arena.window -= 1
    arena.zero_grad()
    out_u = m(**batch)
    out_u.loss.backward()
    torch.testing.assert_close(out_u.logits, out_p.logits, rtol=5e-3, atol=5e-3)
    torch.testing.assert_close(out_u.loss, out_p.loss, rtol=2e-3, atol=2e-3)
    assert g_p.abs().sum() > 0 and ((arena.grad - g_p).norm() / g_p.norm()).item() < 2e-2


def test_micro_batch_cap_is_sized_for_the_hbm_of_this_gpu():
    """Unset, the micro-batch cap of the recipes is sized for the device: the reference's whole batch of 96 runs as
    one micro-batch (its constant 8 was chosen for 16-GB cards), capped at 131072 tokens at the longest length."""
    from pytorch_distributed_training_amd.config import TrainConfig
    from pytorch_distributed_training_amd.models.config import preset

    dev = torch.device("cuda", 0)
    large = preset("bert-large-cased")
    cap = TrainConfig().micro_batch_cap(dev, large)
    assert 96 <= cap <= 131072 // large.max_position_embeddings
    assert TrainConfig().accumulation(dev, large) == (1, 96)
    assert TrainConfig(max_length=128).micro_batch_cap(dev, large) <= 1024
    assert TrainConfig(max_gpu_batch_size=8).accumulation(dev, large) == (12, 8)


@pytest.mark.parametrize("streams", ["2", "4"])
def test_encoder_stack_as_one_autograd_node_equals_one_node_per_layer(monkeypatch, streams):
    """The run of encoder layers is ONE autograd node by default (stack_forward / stack_backward issue every layer from
    C++ and report each layer's gradient segment from inside the call); PDT_LAYER_NODES=1 keeps a node per layer.  Same
    kernels in the same order: losses, accumulated gradients and weights after three optimizer steps are bit-identical,
    also with the accumulation window spread over 2 or 4 streams instead of the default 3."""
    from pytorch_distributed_training_amd.data.synthetic import fixed_batch
    from pytorch_distributed_training_amd.engine import Engine
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.optim import AdamW

    monkeypatch.setenv("PDT_DUAL_BACKWARD", "0")

    def run(per_layer, ns):
        monkeypatch.setenv("PDT_LAYER_NODES", "1" if per_layer else "0")
        monkeypatch.setenv("PDT_MICRO_STREAMS", ns)
        torch.manual_seed(3)
        eng = Engine(fp16=True, seed=11, init_scale=1024.0)
        model = build_model("bert-tiny")
        opt = AdamW(params=model.parameters(), lr=1e-3)
        model, opt = eng.prepare(model, opt)
        model.train()
        batches = [fixed_batch(4, 32, 512, eng.device, seed=70 + i, first_id=4 * i) for i in range(5)]
        losses, grads = [], None
        for step in range(3):
            ls = eng.accumulate(model, batches, overlap=True)
            torch.cuda.synchronize()
            if step == 0:
                grads = model.arena.grad.clone()
            losses += [float(x) for x in ls]
            opt.step(); opt.zero_grad()
        torch.cuda.synchronize()
        return losses, grads, model.arena.master.clone()

    l0, g0, m0 = run(True, "3")
    l1, g1, m1 = run(False, "3")
    l2, g2, m2 = run(False, streams)
    assert g0.abs().sum() > 0 and l0 == l1 == l2
    assert torch.equal(g0, g1) and torch.equal(m0, m1)
    assert torch.equal(g0, g2) and torch.equal(m0, m2)


def test_clip_grad_norm_divides_the_loss_scale_out_on_the_device():
    """fp16 with a loss scale: the returned norm is the un-scaled one, the clipped gradients have the requested
    un-scaled norm, and an optimizer step after the clip moves the weights (no overflow is flagged by it)."""
    from pytorch_distributed_training_amd.data.synthetic import fixed_batch
    from pytorch_distributed_training_amd.engine import Engine
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.optim import AdamW

    torch.manual_seed(0)
    eng = Engine(fp16=True, seed=3, init_scale=1024.0)
    model = build_model("bert-tiny")
    opt = AdamW(params=model.parameters(), lr=1e-3)
    model, opt = eng.prepare(model, opt)
    model.train()
    batch = fixed_batch(8, 32, model.config.vocab_size, eng.device, seed=1)
    eng.backward(model(**batch).loss)
    arena = model.arena
    unscaled = (arena.grad.norm() / 1024.0).item()
    got = eng.clip_grad_norm_(model, 0.25 * unscaled)
    assert abs(got.item() - unscaled) < 1e-4 * unscaled
    assert abs((arena.grad.norm() / 1024.0).item() - 0.25 * unscaled) < 1e-3 * unscaled
    w0 = arena.master.clone()
    opt.step(); opt.zero_grad()
    torch.cuda.synchronize()
    assert not eng.scaler.last_step_skipped() and not torch.equal(w0, arena.master)
