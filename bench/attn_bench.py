#!/usr/bin/env python3
"""Attention forward / backward kernel times at the training shape (B x 16 heads x S, head dim 64), with and
without dropout, against the HBM-traffic floor of each pass."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_distributed_training_amd.ops import native  # noqa: E402


def timeit(fn, iters=20):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--dtype", default="fp16")
    args = ap.parse_args()
    C = native()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    B, S, A = args.batch, args.seq, args.heads
    H = A * 64
    qkv = (torch.randn(B * S, 3 * H, device="cuda") * 0.5).to(dt)
    mask = torch.ones(B, S, dtype=torch.int32, device="cuda")
    kv_len = torch.full((B,), S, dtype=torch.int32, device="cuda")
    st = torch.tensor([1234, 0], dtype=torch.int64, device="cuda")
    dctx = torch.randn(B * S, H, device="cuda").to(dt)
    db = torch.zeros(3 * H, device="cuda") if C.attn_bwd_fuses_dbias(S) else None
    mb = B * S * H * 2 / 1e6
    for p in (0.0, 0.1):
        rs = st if p > 0 else None
        ctx, lse = C.op_attn_fwd(qkv, mask, kv_len, B, S, A, rs, p, 0, 0)
        tf = timeit(lambda: C.op_attn_fwd(qkv, mask, kv_len, B, S, A, rs, p, 0, 0))
        tb = timeit(lambda: C.op_attn_bwd(qkv, mask, kv_len, ctx, dctx, lse, B, S, A, rs, p, 0, 0, db))
        tn = timeit(lambda: C.op_attn_bwd(qkv, mask, kv_len, ctx, dctx, lse, B, S, A, rs, p, 0, 0, None))
        print(f"p={p}: bwd without the bias-gradient reduction {tn:.1f} us")
        print(f"p={p}: fwd {tf:.1f} us ({4 * mb / tf:.2f} TB/s of its 4x{mb:.0f} MB floor)  "
              f"bwd {tb:.1f} us ({8 * mb / tb:.2f} TB/s of its 8x{mb:.0f} MB floor)", flush=True)


if __name__ == "__main__":
    main()
