"""Host-side profile (torch.profiler, CPU activities) of one accumulation window of 12 x 8 samples, one host thread."""
import os
import sys

os.environ.setdefault("PDT_OVERLAP_THREAD", "0")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_distributed_training_amd.data.synthetic import fixed_batch  # noqa: E402
from pytorch_distributed_training_amd.engine import Engine  # noqa: E402
from pytorch_distributed_training_amd.models.bert import build_model  # noqa: E402
from pytorch_distributed_training_amd.optim import AdamW  # noqa: E402

seq = int(sys.argv[1]) if len(sys.argv) > 1 else 32
eng = Engine(fp16=True, seed=0)
model = build_model("bert-large-cased")
opt = AdamW(params=model.parameters(), lr=1e-5)
model, opt = eng.prepare(model, opt)
model.train()
pool = [fixed_batch(8, seq, model.config.vocab_size, eng.device, seed=i, first_id=8 * i) for i in range(12)]


def step():
    eng.accumulate(model, pool)
    opt.step()
    opt.zero_grad()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile

with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=35, max_name_column_width=60))
