"""Does the relative placement of AdamW's five arrays (p, m, v, g fp32 and the 16-bit copy) change the rate it
streams at?  One buffer, arrays at n*4 + k*delta bytes for a sweep of delta; prints us per call and TB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from pytorch_distributed_training_amd.ops import native

C = native()
dev = torch.device("cuda")
n = 335_143_938 // 64 * 64
nbytes = n * 4
buf = torch.empty(5 * nbytes + (64 << 20), dtype=torch.uint8, device=dev)
buf.zero_()
lr = torch.tensor([1e-4], dtype=torch.float32, device=dev)
step = torch.zeros(1, dtype=torch.int32, device=dev)
state = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
traffic = n * (4 * 4 + 3 * 4 + 2)


def view(off, count, dtype):
    width = 4 if dtype == torch.float32 else 2
    return buf[off:off + count * width].view(dtype)


for delta in (0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 3 << 20):
    offs = [k * (nbytes + delta) for k in range(5)]
    p, m, v, g = (view(o, n, torch.float32) for o in offs[:4])
    p16 = view(offs[4], n, torch.float16)
    g.fill_(1e-3)
    for _ in range(2):
        C.adamw_step(p, m, v, g, p16, lr, step, state, 0.9, 0.999, 1e-8, 0.01, True, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        C.adamw_step(p, m, v, g, p16, lr, step, state, 0.9, 0.999, 1e-8, 0.01, True, False)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    print(f"delta {delta:>8d} B: {us:8.1f} us  {traffic / us / 1e6:5.2f} TB/s", flush=True)
