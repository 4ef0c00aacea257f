"""2-stage pipeline (+ data parallel) on CPU processes: loss curve identical to a single process."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dist_workers as W  # noqa: E402
from pytorch_distributed_training_amd.parallel.groups import ParallelMesh  # noqa: E402
from pytorch_distributed_training_amd.parallel.pp import _split_batch, build_stage, split_layers  # noqa: E402
from pytorch_distributed_training_amd.testing import run_distributed  # noqa: E402


def test_mesh_and_layer_split():
    assert split_layers(24, 2) == [(0, 12), (12, 24)] and split_layers(5, 2) == [(0, 3), (3, 5)]
    m = ParallelMesh(8, 5, pp=2)
    assert (m.stage, m.replica, m.dp) == (1, 2, 4)
    assert m.pp_ranks() == [4, 5] and m.dp_ranks() == [1, 3, 5, 7] and m.peer(0) == 4
    assert ParallelMesh(8, 0, pp=2).dp_ranks() == [0, 2, 4, 6]
    with pytest.raises(ValueError):
        ParallelMesh(3, 0, pp=2)


def test_stage_modules_partition_the_full_model():
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.models.config import preset

    cfg = preset("bert-tiny-nodrop")
    torch.manual_seed(3)
    full = build_model(cfg)
    s0 = build_stage(cfg, 0, 2, full.state_dict())
    s1 = build_stage(cfg, 1, 2, full.state_dict())
    k0, k1 = set(s0.state_dict()), set(s1.state_dict())
    assert not (k0 & k1) and (k0 | k1) == set(full.state_dict())
    assert any(k.startswith("bert.embeddings") for k in k0) and "classifier.weight" in k1
    assert "bert.encoder.layer.1.output.dense.weight" in k1  # global layer indices are kept
    ids = torch.randint(1, 512, (3, 9)); mask = torch.ones(3, 9, dtype=torch.long); mask[2, 5:] = 0
    labels = torch.tensor([0, 1, 0])
    full.eval(); s0.eval(); s1.eval()
    h = s0(input_ids=ids, attention_mask=mask, token_type_ids=torch.zeros_like(ids)).last_hidden_state
    out = s1(hidden_states=h, attention_mask=mask, labels=labels)
    ref = full(input_ids=ids, attention_mask=mask, token_type_ids=torch.zeros_like(ids), labels=labels)
    torch.testing.assert_close(out.logits, ref.logits)
    torch.testing.assert_close(out.loss, ref.loss)


def test_split_batch_covers_everything():
    b = {"x": torch.arange(10), "y": torch.arange(10) * 2}
    parts = _split_batch(b, 4)
    assert sum(p["x"].numel() for p in parts) == 10 and torch.equal(torch.cat([p["y"] for p in parts]), b["y"])
    assert len(_split_batch(b, 100)) == 10


def test_split_batch_keeps_packed_micro_batches_self_contained():
    """A batch with the collator's packing entries is cut into micro-batches that are packed batches themselves."""
    from pytorch_distributed_training_amd.data.collate import PadCollator

    lens = [5, 3, 7, 2, 6, 4, 1]
    ex = [{"input_ids": list(range(10, 10 + n)), "attention_mask": [1] * n, "token_type_ids": [0] * n, "label": i % 2}
          for i, n in enumerate(lens)]
    batch = PadCollator(0, "longest", pad_to_multiple_of=8, varlen=True)(ex)
    assert batch["lengths"] == lens and batch["attention_mask"].shape == (7, 8)
    parts = _split_batch(batch, 3)
    assert [p["labels"].shape[0] for p in parts] == [2, 3, 2]
    flat = batch["input_ids"].reshape(-1)
    seen = 0
    for p in parts:
        b = p["labels"].shape[0]
        own = lens[seen:seen + b]
        assert p["tokens"] == sum(own) and p["cu_seqlens"].tolist() == [sum(own[:i]) for i in range(b + 1)]
        # the micro-batch's token_index addresses its own [b, S] block and selects exactly its real tokens
        sel = p["input_ids"].reshape(-1)[p["token_index"]]
        assert sel.tolist() == [t for n in own for t in range(10, 10 + n)]
        assert "lengths" not in p
        seen += b
    assert seen == len(lens) and flat.numel() == 56
    # without the host-side lengths (or on a path that runs padded) the packing entries are dropped
    padded = _split_batch({k: v for k, v in batch.items() if k != "lengths"}, 2)
    assert all("cu_seqlens" not in p and "token_index" not in p for p in padded)


def test_pipeline_loss_curve_matches_single_process():
    ref = W.single_process_reference(batch_size=16)
    res = run_distributed(W.pipeline_recipe, 2, args=(4, 4))  # 1 replica x 2 stages, 4 micro-batches of 4
    last = [r for r in res if r["stage"] == 1][0]
    assert last["optimizer_steps"] == 6 and len(last["losses"]) == 6
    for a, b in zip(last["losses"], ref):
        assert abs(a - b) < 2e-5, (last["losses"], ref)
    assert res[0]["accuracy"] == res[1]["accuracy"]


def test_gradient_clipping_uses_the_norm_of_the_whole_replica():
    """--max-grad-norm in the pipeline recipe: each stage holds half of the parameters, the squared norms are summed
    over the pair, so the clipped run follows ONE process clipping the full model (a bound low enough to bind at
    every step; with a norm per stage the curves part after the first update)."""
    ref = W.single_process_reference(batch_size=16, max_grad_norm=0.05)
    free = W.single_process_reference(batch_size=16)
    assert max(abs(a - b) for a, b in zip(ref, free)) > 1e-3  # the bound binds
    res = run_distributed(W.pipeline_recipe, 2, args=(4, 4, False, 0.05))
    last = [r for r in res if r["stage"] == 1][0]
    for a, b in zip(last["losses"], ref):
        assert abs(a - b) < 2e-5, (last["losses"], ref)


def test_dp_times_pipeline_loss_curve_matches_single_process():
    ref = W.single_process_reference(batch_size=16)
    res = run_distributed(W.pipeline_recipe, 4, args=(4, 2))  # 2 replicas x 2 stages, per-replica batch 8
    # the logged loss is the mean over the replicas' shards = the loss of the whole global batch
    assert all(r["optimizer_steps"] == 6 for r in res)
    logged = [r for r in res if r["stage"] == 1 and r["replica"] == 0][0]["losses"]
    assert len(logged) == 6
    for a, b in zip(logged, ref):
        assert abs(a - b) < 2e-5, (logged, ref)


def test_compat_model_parallel_loaders_are_distributed_sampler_flavoured():
    """--compat: per-sample round-robin shards with torch's DistributedSampler order (shuffle seed 0), for the
    training and the evaluation split, and the same order in every epoch (the reference never calls set_epoch:
    /root/reference/test_model_parallelism.py:254-269)."""
    from types import SimpleNamespace

    from torch.utils.data.distributed import DistributedSampler

    from pytorch_distributed_training_amd.config import TrainConfig
    from pytorch_distributed_training_amd.data.loader import DeviceLoader
    from pytorch_distributed_training_amd.models.config import preset
    from pytorch_distributed_training_amd.recipes import _collator, build_datasets, model_parallel_loaders

    cfg = TrainConfig(model="bert-tiny-nodrop", cpu=True, train_examples=37, eval_examples=11, compat=True, eval_batch_size=4)
    mcfg = preset(cfg.model)
    train_ds, val_ds = build_datasets(cfg, mcfg)
    for replica in range(2):
        mesh = SimpleNamespace(dp=2, replica=replica)
        tl, el = model_parallel_loaders(cfg, train_ds, val_ds, _collator(cfg, mcfg), mesh, 8)
        for ds, loader in ((train_ds, tl), (val_ds, el)):
            want = list(DistributedSampler(ds, num_replicas=2, rank=replica))
            dev = DeviceLoader(loader, torch.device("cpu"))
            for _ in range(2):  # two epochs, same order
                got = [int(i) for b in dev for i in b["sample_ids"]]
                assert got == want


def test_dp_times_pipeline_compat_recipe_runs():
    res = run_distributed(W.pipeline_recipe, 4, args=(4, 2, True))
    last = [r for r in res if r["stage"] == 1 and r["replica"] == 0][0]
    assert all(l == l and l < 2.0 for l in last["losses"]) and len(last["losses"]) > 0
