#!/usr/bin/env python3
"""Loss-curve parity at real size on one GPU: the native 16-bit path against the fp32 PyTorch path of the same module
tree, same initial weights, same batches (synthetic sentence pairs, dynamically padded), dropout off.

    python bench/parity_curve.py [--model bert-base-cased] [--steps 50] [--batch 32] [--accum 1] [--lr 2e-5] [--dtype fp16]

`--batch 96 --accum 12 --compat` is the reference's literal layout: 12 micro-batches of 8, the optimizer fires when
``step % 12 == 0`` (/root/reference/test_data_parallelism.py:90-93,145-150)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

from pytorch_distributed_training_amd.data import PadCollator, SyntheticPairDataset  # noqa: E402
from pytorch_distributed_training_amd.engine import Engine  # noqa: E402
from pytorch_distributed_training_amd.models.bert import build_model  # noqa: E402
from pytorch_distributed_training_amd.optim import AdamW, get_linear_schedule_with_warmup  # noqa: E402


def run(args, native: bool):
    eng = Engine(fp16=native and args.dtype == "fp16", bf16=native and args.dtype == "bf16", seed=1)
    torch.manual_seed(0)
    model = build_model(args.model, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ds = SyntheticPairDataset(args.steps * args.batch, model.config.vocab_size, seed=3, data_vocab=args.data_vocab)
    micro = args.batch // args.accum
    dl = DataLoader(ds, shuffle=True, collate_fn=PadCollator(), batch_size=micro)
    opt = AdamW(params=model.parameters(), lr=args.lr)
    model, opt, dl = eng.prepare(model, opt, dl)
    sched = get_linear_schedule_with_warmup(opt, args.warmup, args.steps)
    model.train()
    losses, window = [], []
    for step, batch in enumerate(dl):
        loss = model(**batch).loss
        eng.backward(loss / args.accum)
        window.append(loss.detach().float())
        fire = step % args.accum == 0 if args.compat else (step + 1) % args.accum == 0
        if fire:
            opt.step(); sched.step(); opt.zero_grad()
            losses.append(float(torch.stack(window).mean().item()))
            window = []
    return losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="bert-base-cased")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--accum", type=int, default=1)
    ap.add_argument("--lr", type=float, default=2e-5)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--data-vocab", type=int, default=None)
    ap.add_argument("--compat", action="store_true")
    args = ap.parse_args()
    ref = run(args, native=False)
    nat = run(args, native=True)
    n = min(len(ref), len(nat))
    diffs = [abs(a - b) for a, b in zip(ref[:n], nat[:n])]
    print(json.dumps({"model": args.model, "optimizer_steps": n, "batch": args.batch, "micro_batches": args.accum,
                      "compat": args.compat, "dtype": args.dtype, "lr": args.lr,
                      "max_abs_loss_diff": max(diffs), "mean_abs_loss_diff": sum(diffs) / n,
                      "loss_fp32_torch": [round(x, 4) for x in ref[:n]], "loss_native": [round(x, 4) for x in nat[:n]]}))


if __name__ == "__main__":
    main()
