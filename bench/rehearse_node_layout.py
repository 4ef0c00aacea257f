#!/usr/bin/env python3
"""The whole-node layout rehearsed on ONE GPU at real model size: 8 processes = 4 data-parallel replicas x 2 pipeline
stages (rank = replica * 2 + stage, gloo transport, fp16, dropout 0.1 on) against ONE process training the same
global batches.  Prints both loss curves and their largest difference as one JSON line.

    python bench/rehearse_node_layout.py [--model bert-base-cased] [--world 8]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import dist_workers as W  # noqa: E402
from pytorch_distributed_training_amd.testing import run_distributed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="bert-base-cased")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--lr", type=float, default=5e-5)
    ap.add_argument("--layout", default="pp", choices=["pp", "dp"])
    args = ap.parse_args()
    if args.layout == "dp":  # pure data parallel: every rank one shard of the global batch of 8 x world samples
        examples = 8 * args.world * 4
        ref = W.gpu_single_reference("dp", steps=6, replicas=args.world, model=args.model, examples=examples, lr=args.lr)
        res = run_distributed(W.gpu_dp_dropout, args.world, args=(6, args.model, examples, args.lr))
        got = res[0]
        print(json.dumps({
            "model": args.model, "ranks": args.world, "layout": f"dp{args.world}", "dtype": "fp16", "dropout": 0.1,
            "lr": args.lr, "global_batch": 8 * args.world, "transport": f"gloo ({args.world} processes on one GPU)",
            "loss_single_process": [round(x, 5) for x in ref], "loss_ranks": [round(x, 5) for x in got],
            "max_abs_loss_diff": max(abs(a - b) for a, b in zip(got, ref)),
        }))
        return
    replicas = args.world // 2
    examples = 64 * replicas
    ref = W.gpu_single_reference("pipeline", steps=6, replicas=replicas, model=args.model, examples=examples, lr=args.lr)
    res = run_distributed(W.gpu_pipeline_recipe, args.world, args=("pipeline", False, args.model, True, examples, args.lr))
    got = [r for r in res if r["stage"] == 1 and r["replica"] == 0][0]["losses"]
    print(json.dumps({
        "model": args.model, "ranks": args.world, "layout": f"dp{replicas} x pp2 (1F1B, 4 micro-batches)", "dtype": "fp16",
        "dropout": 0.1, "lr": args.lr, "global_batch": 16 * replicas, "transport": f"gloo ({args.world} processes on one GPU)",
        "loss_single_process": [round(x, 5) for x in ref], "loss_ranks": [round(x, 5) for x in got],
        "max_abs_loss_diff": max(abs(a - b) for a, b in zip(got, ref)),
        "replicas_bit_equal": all(r["replicas_bit_equal"] is True for r in res),
    }))


if __name__ == "__main__":
    main()
