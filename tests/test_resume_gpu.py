"""Kill -> resume on the native 16-bit GPU path: a run that dies in the middle of an epoch and is started again (a fresh
process) ends with exactly the weights of the uninterrupted run.  Dropout is on and the run is fp16, so the resumed
process has to restore everything the native path keeps outside the torch modules: the fused AdamW moments and step
counter on the parameter arena, the dynamic loss-scale state, the arena's dropout window counter and the runtime's
micro-batch counter.  (The reference has no checkpoints: a restart by torch.distributed.run,
/root/reference/README.md:12-14, recomputes from scratch.)"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--fp16=True", "--model", "bert-tiny", "--train-examples", "48", "--eval-examples", "16", "--num-epochs", "2",
          "--batch-size", "8", "--max-gpu-batch-size", "4", "--lr", "1e-3", "--num-warmup-steps", "2", "--log-every", "1",
          "--save-every", "1", "--seed", "3", "--init-scale", "1024"]


def _run(cmd, env=None, timeout=300):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def _losses(path):
    return {json.loads(l)["step"]: json.loads(l)["loss"] for l in open(path) if '"step"' in l}


def test_native_mid_epoch_resume_reproduces_the_uninterrupted_run(tmp_path):
    """6 optimizer steps per epoch (12 micro-batches of 4, accumulation 2).  The second run is killed by the fault hook
    right after step 4 - inside epoch 0 - and restarted: bit-equal final weights, identical losses for steps 5..12."""
    from safetensors.torch import load_file

    from pytorch_distributed_training_amd import checkpoint as ck

    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    base = [sys.executable, "train_dp.py"] + COMMON
    r = _run(base + ["--output-dir", a, "--log-file", os.path.join(a, "log.jsonl")])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    r = _run(base + ["--output-dir", b], env={"PDT_INJECT_FAULT": "0:4"})
    assert r.returncode == 13 and "fault-injection" in r.stderr, r.stdout[-2000:] + r.stderr[-3000:]
    assert ck.latest_tag(b) == "00000004-0-8" and not os.path.exists(os.path.join(b, "model.safetensors"))
    r = _run(base + ["--output-dir", b, "--log-file", os.path.join(b, "log.jsonl")])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "resumed from" in r.stdout and "epoch 0, batch 8, optimizer step 4" in r.stdout
    wa, wb = load_file(os.path.join(a, "model.safetensors")), load_file(os.path.join(b, "model.safetensors"))
    assert wa.keys() == wb.keys() and len(wa) > 0
    for k in wa:
        assert torch.equal(wa[k], wb[k]), k
    la, lb = _losses(os.path.join(a, "log.jsonl")), _losses(os.path.join(b, "log.jsonl"))
    assert sorted(lb) == list(range(5, 13)) and all(la[s] == lb[s] for s in lb), (la, lb)


def test_native_pipeline_restart_and_resume_two_ranks_one_gpu(tmp_path):
    """The 2-stage pipeline with both ranks on the one GPU (gloo rehearsal) under torch.distributed.run --max-restarts 1:
    the second rank dies after optimizer step 2 (once); both ranks of the restarted group resume from the newest
    checkpoint that BOTH had completed and end with the stage weights of the uninterrupted run."""
    from safetensors.torch import load_file

    from pytorch_distributed_training_amd import checkpoint as ck
    from pytorch_distributed_training_amd.launch import find_free_port

    args = ["--fp16=True", "--model", "bert-tiny", "--train-examples", "32", "--eval-examples", "8", "--num-epochs", "2",
            "--batch-size", "8", "--max-gpu-batch-size", "4", "--lr", "1e-3", "--num-warmup-steps", "1", "--log-every", "1",
            "--save-every", "1", "--seed", "3", "--init-scale", "1024", "--mp-layout", "pipeline"]

    def launch(out, extra_env=None, restarts=0):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(find_free_port()), f"--max-restarts={restarts}", "train_mp.py"] + args + ["--output-dir", out]
        return _run(cmd, env=extra_env, timeout=420)

    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    r = launch(a)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    r = launch(b, {"PDT_INJECT_FAULT": "1:2", "PDT_INJECT_FAULT_ONCE": str(tmp_path / "once")}, restarts=1)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "fault-injection" in r.stderr and "resumed from" in r.stdout
    for suffix in ("stage0", "stage1"):
        ta, tb = ck.latest_tag(a, suffix), ck.latest_tag(b, suffix)
        assert ta == tb and ta is not None
        wa = load_file(os.path.join(a, f"checkpoint-{ta}", f"model.{suffix}.safetensors"))
        wb = load_file(os.path.join(b, f"checkpoint-{tb}", f"model.{suffix}.safetensors"))
        assert wa.keys() == wb.keys() and len(wa) > 0
        for k in wa:
            assert torch.equal(wa[k], wb[k]), (suffix, k)
