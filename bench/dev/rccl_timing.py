"""Times every step of the 1-rank RcclComm exercise (where do the seconds go?)."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from pytorch_distributed_training_amd.parallel.comm import create_rccl_comm, close_all
t0 = time.time()
def lap(msg):
    global t0
    torch.cuda.synchronize(); print(f"{time.time()-t0:7.2f}s {msg}", flush=True); t0 = time.time()
torch.cuda.set_device(0); torch.zeros(1, device="cuda"); lap("cuda init")
comm = create_rccl_comm(); lap("comm init")
y = torch.randn(1 << 22, device="cuda"); comm.all_reduce(y, average=True); lap("all_reduce fp32 avg")
for dt in (torch.float16, torch.bfloat16, torch.float32, torch.int64, torch.int32):
    t = torch.arange(1000, device="cuda").to(dt)
    comm.all_reduce(t); lap(f"all_reduce {dt}")
    comm.broadcast(t, 0); lap(f"broadcast {dt}")
    out = torch.empty_like(t); comm.all_gather(out, t); lap(f"all_gather {dt}")
    rs = torch.empty_like(t); comm.reduce_scatter(rs, out); lap(f"reduce_scatter {dt}")
src = torch.randn(4096, 64, device="cuda", dtype=torch.float16); dst = torch.zeros_like(src)
comm.p2p([0, 1], [src, dst], [0, 0]); lap("self send/recv")
comm.barrier(); lap("barrier")
close_all(); lap("close")
