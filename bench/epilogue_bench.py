#!/usr/bin/env python3
"""What the fused GEMM epilogues cost: the BERT-large FFN / projection GEMMs timed with a plain epilogue and
with the epilogue the training step uses (bias+GELU, bias+dropout+residual, GELU-grad multiply + column sums).
Prints microseconds and nominal TFLOP/s per variant (median of rounds, variants interleaved)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_training_amd.ops import native  # noqa: E402


def timeit(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=32768)
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    C = native()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    T, H = args.tokens, args.hidden
    F = 4 * H
    dev = "cuda"
    torch.manual_seed(0)
    x = (torch.randn(T, H, device=dev) * 0.5).to(dt)
    w1 = (torch.randn(F, H, device=dev) * 0.03).to(dt)
    b1 = (torch.randn(F, device=dev) * 0.1).to(dt)
    z1 = torch.empty(T, F, device=dev, dtype=dt)
    g = (torch.randn(T, F, device=dev) * 0.5).to(dt)
    w2 = (torch.randn(H, F, device=dev) * 0.03).to(dt)
    b2 = (torch.randn(H, device=dev) * 0.1).to(dt)
    dy = (torch.randn(T, H, device=dev) * 0.5).to(dt)
    cs = torch.zeros(F, device=dev)
    rng = torch.tensor([1234, 0], dtype=torch.int64, device=dev)
    wqkv = (torch.randn(3 * H, H, device=dev) * 0.03).to(dt)
    bqkv = (torch.randn(3 * H, device=dev) * 0.1).to(dt)

    groups = {
        "ffn1_fwd nt [T,4H]<-[T,H]": (2.0 * T * F * H, {
            "plain": lambda: C.op_gemm("nt", x, w1, 0, None, None, None, None, 0.0, 0, 0, -1),
            "bias": lambda: C.op_gemm("nt", x, w1, 1, b1, None, None, None, 0.0, 0, 0, -1),
            "bias_gelu": lambda: C.op_gemm("nt", x, w1, 2, b1, None, z1, None, 0.0, 0, 0, -1),
        }),
        "qkv_fwd nt [T,3H]<-[T,H]": (2.0 * T * 3 * H * H, {
            "plain": lambda: C.op_gemm("nt", x, wqkv, 0, None, None, None, None, 0.0, 0, 0, -1),
            "bias": lambda: C.op_gemm("nt", x, wqkv, 1, bqkv, None, None, None, 0.0, 0, 0, -1),
        }),
        "ffn2_fwd nt [T,H]<-[T,4H]": (2.0 * T * F * H, {
            "plain": lambda: C.op_gemm("nt", g, w2, 0, None, None, None, None, 0.0, 0, 0, -1),
            "bias_res": lambda: C.op_gemm("nt", g, w2, 3, b2, x, None, None, 0.0, 0, 0, -1),
            "bias_drop_res": lambda: C.op_gemm("nt", g, w2, 3, b2, x, None, rng, 0.1, 0, 3, -1),
        }),
        "ffn2_dgrad nn [T,4H]<-[T,H]": (2.0 * T * F * H, {
            "plain": lambda: C.op_gemm("nn", dy, w2, 0, None, None, None, None, 0.0, 0, 0, -1),
            "mul_dgelu": lambda: C.op_gemm("nn", dy, w2, 4, None, None, z1, None, 0.0, 0, 0, -1),
            "mul_dgelu_colsum": lambda: C.op_gemm("nn", dy, w2, 4, None, None, z1, None, 0.0, 0, 0, -1, cs),
        }),
        "ffn1_dgrad nn [T,H]<-[T,4H]": (2.0 * T * F * H, {
            "plain": lambda: C.op_gemm("nn", g, w1, 0, None, None, None, None, 0.0, 0, 0, -1),
            "add_res": lambda: C.op_gemm("nn", g, w1, 5, None, x, None, None, 0.0, 0, 0, -1),
        }),
    }
    C.op_gemm("nt", x, w1, 2, b1, None, z1, None, 0.0, 0, 0, -1)  # fill z1 with real GELU derivatives
    for name, (fl, cands) in groups.items():
        for f in cands.values():
            f()
        times = {k: [] for k in cands}
        for _ in range(args.rounds):
            for k, f in cands.items():
                times[k].append(timeit(f, args.iters))
        res = {"site": name}
        for k, v in times.items():
            v.sort()
            us = v[len(v) // 2] * 1e3
            res[k] = f"{us:.1f} us / {fl / us / 1e6:.0f} TF"
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
