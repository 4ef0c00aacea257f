#!/usr/bin/env python3
"""The reference's software stack on this hardware, as a yardstick for bench.py.

The reference trains HF BERT with PyTorch AMP (autocast fp16 + GradScaler) and an eager AdamW
(/root/reference/test_data_parallelism.py:55,112-150).  Pretrained weights and `transformers.AdamW` are not
available offline, so this script runs the SAME architecture through this repository's pure-PyTorch module
path (ATen + hipBLASLt + SDPA-free eager attention exactly like HF's eager BertSelfAttention) under
``torch.autocast`` with ``torch.cuda.amp.GradScaler`` and ``torch.optim.AdamW`` - i.e. what
`accelerate launch ... --fp16` would execute on an MI355X - on the same synthetic batches as bench.py.
Optionally (--hf) it uses `transformers.BertForSequenceClassification` itself, random-init.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_training_amd.data.synthetic import fixed_batch  # noqa: E402
from pytorch_distributed_training_amd.models.bert import build_model  # noqa: E402
from pytorch_distributed_training_amd.models.config import preset  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="bert-large-cased")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--hf", action="store_true")
    ap.add_argument("--sdpa", action="store_true", help="HF model with attn_implementation=sdpa")
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    cfg = preset(args.model)
    if args.hf:
        from transformers import BertConfig, BertForSequenceClassification

        hc = BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                        num_labels=2)
        hc._attn_implementation = "sdpa" if args.sdpa else "eager"
        model = BertForSequenceClassification(hc).to(dev)
    else:
        model = build_model(cfg).to(dev)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-5, eps=1e-6, weight_decay=0.0)
    scaler = torch.amp.GradScaler("cuda")
    pool = [fixed_batch(args.batch, args.seq_len, cfg.vocab_size, dev, seed=i) for i in range(4)]

    def step(i):
        b = pool[i % 4]
        with torch.autocast("cuda", dtype=torch.float16):
            out = model(**b)
        scaler.scale(out.loss).backward()
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        return out.loss

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"stack": "transformers+torch AMP eager" if args.hf else "torch AMP eager (own module tree)",
                      "attn": "sdpa" if args.sdpa else "eager", "model": args.model, "batch": args.batch,
                      "seq_len": args.seq_len, "samples_per_s": round(args.batch * args.steps / dt, 2),
                      "ms_per_step": round(1000 * dt / args.steps, 2), "loss": float(loss.item()),
                      "max_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
