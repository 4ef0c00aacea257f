#!/usr/bin/env python3
"""GEMM micro-benchmark on the BERT-large shapes: this library's kernels (per tile variant) against
torch.matmul (hipBLASLt) on the same random data, interleaved in one process.  TFLOP/s, median of rounds."""
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
    ap.add_argument("--tokens", type=int, default=16384)
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--tiles", default="1,2")
    args = ap.parse_args()
    C = native()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    T, H = args.tokens, args.hidden
    tiles = [int(t) for t in args.tiles.split(",")]
    shapes = [("nt", T, 3 * H, H), ("nt", T, H, H), ("nt", T, 4 * H, H), ("nt", T, H, 4 * H),
              ("nn", T, H, 3 * H), ("nn", T, 4 * H, H), ("nn", T, H, 4 * H),
              ("tn", 3 * H, H, T), ("tn", H, H, T), ("tn", 4 * H, H, T), ("tn", H, 4 * H, T)]
    out = []
    for layout, M, N, K in shapes:
        torch.manual_seed(0)
        if layout == "nt":
            A = torch.randn(M, K, device="cuda").to(dt); B = torch.randn(N, K, device="cuda").to(dt)
            ref = lambda: torch.matmul(A, B.t())
            mine = lambda tile: C.op_gemm("nt", A, B, 0, None, None, None, None, 0.0, 0, 0, tile)
        elif layout == "nn":
            A = torch.randn(M, K, device="cuda").to(dt); B = torch.randn(K, N, device="cuda").to(dt)
            ref = lambda: torch.matmul(A, B)
            mine = lambda tile: C.op_gemm("nn", A, B, 0, None, None, None, None, 0.0, 0, 0, tile)
        else:
            A = torch.randn(K, M, device="cuda").to(dt); B = torch.randn(K, N, device="cuda").to(dt)
            G = torch.zeros(M, N, device="cuda")
            ref = lambda: torch.matmul(A.t(), B)
            mine = lambda tile: C.op_gemm_tn(A, B, G, False, tile, 0)
        fl = 2.0 * M * N * K
        res = {"layout": layout, "M": M, "N": N, "K": K}
        cands = {"torch": ref}
        for t in tiles:
            cands[f"tile{t}"] = (lambda tt: (lambda: mine(tt)))(t)
        for f in cands.values():
            f()
        times = {k: [] for k in cands}
        for _ in range(args.rounds):
            for k, f in cands.items():
                times[k].append(timeit(f, args.iters))
        for k, v in times.items():
            v.sort()
            res[k + "_tflops"] = round(fl / (v[len(v) // 2] * 1e-3) / 1e12, 1)
        print(json.dumps(res), flush=True)
        out.append(res)
    return out


if __name__ == "__main__":
    main()
