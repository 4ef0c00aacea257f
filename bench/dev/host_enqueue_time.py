import os, sys, time, torch
sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
from pytorch_distributed_training_amd.engine import Engine
from pytorch_distributed_training_amd.models.bert import build_model
from pytorch_distributed_training_amd.optim import AdamW
from pytorch_distributed_training_amd.data.synthetic import fixed_batch
eng = Engine(fp16=True, seed=0)
model = build_model("bert-large-cased")
opt = AdamW(params=model.parameters(), lr=1e-5)
model, opt = eng.prepare(model, opt)
model.train()
b = fixed_batch(8, 128, eng.unwrap_model(model).config.vocab_size, eng.device, seed=1)
for _ in range(3):
    out = model(**b); eng.backward(out.loss); opt.step(); opt.zero_grad()
torch.cuda.synchronize()
tf = tb = 0.0; n = 20
g0 = time.perf_counter()
for _ in range(n):
    t0 = time.perf_counter(); out = model(**b); t1 = time.perf_counter()
    eng.backward(out.loss); t2 = time.perf_counter()
    tf += t1 - t0; tb += t2 - t1
host = time.perf_counter() - g0
torch.cuda.synchronize()
tot = time.perf_counter() - g0
print(f"host enqueue: forward {tf/n*1e3:.2f} ms, backward {tb/n*1e3:.2f} ms per 8-sample micro-batch; host total {host/n*1e3:.2f} ms, with GPU drain {tot/n*1e3:.2f} ms")
