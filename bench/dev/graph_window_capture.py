#!/usr/bin/env python3
"""Capture Engine.accumulate's multi-stream accumulation window in a hipGraph, one variant per process.

    python bench/dev/graph_window_capture.py            # eager, plain: each in a child process, one line each
    python bench/dev/graph_window_capture.py <variant>  # one variant in this process

Variants: eager (no graph: the window as Engine.accumulate issues it, forward passes from a second host thread), plain
(one-stream capture of the plain loop), overlap (the multi-stream window captured with every launch issued from one
host thread; needs PDT_GRAPH_OVERLAP_WINDOW=1 and is not in the default list: it ended in SIGSEGV inside
CUDAGraph.capture_end() on ROCm 7.2 - 61.6 ms eager, 232.7 ms plain replay at 12 x 8 samples of BERT-large).
"""
import faulthandler
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(variant: str, model_name: str, micro: int, accum: int, steps: int) -> None:
    faulthandler.enable()
    import torch

    from pytorch_distributed_training_amd.data.synthetic import fixed_batch
    from pytorch_distributed_training_amd.engine import Engine
    from pytorch_distributed_training_amd.models.bert import build_model
    from pytorch_distributed_training_amd.optim import AdamW
    from pytorch_distributed_training_amd.utils.graph import GraphedStep

    eng = Engine(fp16=True, seed=42)
    torch.manual_seed(42)
    model = build_model(model_name)
    opt = AdamW(params=model.parameters(), lr=2e-5)
    model, opt = eng.prepare(model, opt)
    model.train()
    vocab = eng.unwrap_model(model).config.vocab_size
    batches = [fixed_batch(micro, 128, vocab, eng.device, seed=i, first_id=i * micro) for i in range(accum)]
    if variant == "eager":
        def step():
            losses = eng.accumulate(model, batches)
            opt.step(); opt.zero_grad()
            return losses[-1]
    else:
        g = GraphedStep(eng, model, opt, warmup=2, window=variant)

        def step():
            return g(batches)
    for _ in range(4):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    print(json.dumps({"variant": variant, "model": model_name, "micro_batch": micro, "accum": accum,
                      "ms_per_step": round(ms, 3), "samples_per_s": round(micro * accum / ms * 1e3, 1),
                      "loss": float(loss)}), flush=True)


def main() -> None:
    model_name = os.environ.get("PDT_GW_MODEL", "bert-large-cased")
    micro, accum, steps = int(os.environ.get("PDT_GW_MICRO", "8")), int(os.environ.get("PDT_GW_ACCUM", "12")), 20
    if len(sys.argv) > 1:
        return child(sys.argv[1], model_name, micro, accum, steps)
    for variant in ("eager", "plain"):
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), variant], capture_output=True, text=True, timeout=240)
        except subprocess.TimeoutExpired:
            print(json.dumps({"variant": variant, "outcome": "no result after 240 s"}), flush=True)
            break  # (nothing more on the GPU after a hang)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            print(lines[-1], flush=True)
        else:
            print(json.dumps({"variant": variant, "outcome": f"exit status {r.returncode}",
                              "stderr_tail": r.stderr[-1800:]}), flush=True)
            if r.returncode not in (0, 1):
                break  # a crash: nothing more on the GPU in this call


if __name__ == "__main__":
    main()
